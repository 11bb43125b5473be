"""The reference's own benchmark case — the dry convective boundary layer of benchmarking/src/convective_boundary_layer.jl:59-185
(BreezeBenchmarks; .github/workflows/Benchmarks.yml:34-45) — through the host mirror breeze.jl_amd/benchmarks.py:
AnelasticDynamics on a 309 K reference state, 5-cell halos, FPlane, geostrophic forcing (9, 0) m/s, a 0.35 K m/s surface heat flux and
the u* = 0.4 m/s drag conditions with the case's regulariser epsilon = 1e-10.  Compared with the oracle carrying the same physics list
(oracle/forcings.py; parity status as in tests/test_forcings.py: the Oceananigans side of the flux application is restated)."""
import numpy as np
import pytest

from helpers import PROG


def _oracle_cbl(oracle, size, order=5, topology=("Periodic", "Periodic", "Bounded")):
    from oracle.forcings import ColumnForcings
    from breeze_jl_amd import benchmarks as bm
    C = bm.CBL
    og = oracle.Grid(size, x=(0.0, C["Lx"]), y=(0.0, C["Ly"]), z=(0.0, C["Lz"]), halo=(5, 5, 5), topology=topology)
    f, rho0 = bm.cbl_coriolis_parameter(), bm.cbl_surface_density()
    ones = np.ones(size[2])
    F = ColumnForcings(Fu=-f * C["Vg"] * ones, Fv=f * C["Ug"] * ones, coriolis_f=f, flux_theta=rho0 * C["heat_flux"],
                       drag_rho0_ustar2=rho0 * C["ustar"] ** 2, drag_epsilon=C["drag_epsilon"])
    kw = {} if order == 5 else {"advection": f"WENO{order}"}
    om = oracle.OracleModel(og, surface_pressure=C["p0"], potential_temperature=C["theta0"], forcings=F, **kw)
    om.set(theta=bm.cbl_initial_theta(size, seed=0), u=C["Ug"], v=C["Vg"])
    return om


def test_cbl_case_constants():
    """f = 2 Omega sin(33.5 deg) ~ 8.0e-5 1/s and rho0 = p0 / (R^d theta0) (convective_boundary_layer.jl:114-127)"""
    from breeze_jl_amd import benchmarks as bm
    assert abs(bm.cbl_coriolis_parameter() - 8.05e-5) < 1e-6
    assert abs(bm.cbl_surface_density() - 101325.0 / (8.314462618 / 0.02897 * 309.0)) < 1e-12
    th = bm.cbl_initial_theta((8, 8, 30))
    zc = (np.arange(30) + 0.5) * 100.0
    assert np.all(np.abs(th[zc < 400.0] - 309.0) <= 0.25) and np.all(th[zc > 600.0, 0, 0] == 309.0 + (zc[zc > 600.0] - 600.0) * 0.004)
    assert np.all(th[(zc > 400.0) & (zc < 600.0)] == 309.0)


@pytest.mark.gpu
@pytest.mark.parametrize("order", [5, 9])
def test_cbl_float64_steps_match_oracle(oracle, bz, order):
    """three steps of the case's physics list in Float64 against the oracle: 1e-9 of the field scale (WENO9: 2e-8, the tolerance of
    tests/test_weno_orders.py for wide stencils on kinked data)"""
    size = (32, 24, 16)
    om = _oracle_cbl(oracle, size, order)
    hm = bz.benchmarks.convective_boundary_layer(size, float_type=np.float64, advection=bz.WENO(order=order))
    for _ in range(3):
        om.time_step(0.5)
        hm.time_step(0.5)
    hm.synchronize()
    g = om.grid
    mom = max(np.abs(g.interior(getattr(om, n), zface=(n == "rw"))).max() for n in ("ru", "rv", "rw"))
    tol = 1e-9 if order == 5 else 2e-8
    for n, k in PROG.items():
        want, got = g.interior(getattr(om, n), zface=(n == "rw")), hm.prognostic_fields()[k].interior_cpu()
        scale = mom if n in ("ru", "rv", "rw") else max(np.abs(want).max(), 1e-3)
        assert np.abs(got - want).max() / scale < tol, (n, np.abs(got - want).max() / scale)
    # the surface fluxes did something: the lowest level slowed down below the geostrophic wind it started from, and warmed
    assert g.interior(om.u)[0].mean() < 9.0 - 1e-4 and g.interior(om.u)[-1].mean() > 9.0 - 1e-6
    assert g.interior(om.theta)[0].mean() > 309.0


@pytest.mark.gpu
def test_cbl_with_walls_in_y_matches_oracle(oracle, bz):
    """the benchmark driver's PBB topology (benchmarking/run_benchmarks.jl:130: (Periodic, Bounded, Bounded)) with the case's full
    physics list, three Float64 steps against the oracle; the wall faces of rho v stay closed"""
    size = (32, 24, 16)
    om = _oracle_cbl(oracle, size, topology=("Periodic", "Bounded", "Bounded"))
    hm = bz.benchmarks.convective_boundary_layer(size, float_type=np.float64, topology=(bz.Periodic, bz.Bounded, bz.Bounded))
    for _ in range(3):
        om.time_step(0.5)
        hm.time_step(0.5)
    hm.synchronize()
    g = om.grid
    mom = max(np.abs(g.interior(getattr(om, n), zface=(n == "rw"))).max() for n in ("ru", "rv", "rw"))
    for n, k in PROG.items():
        want, got = g.interior(getattr(om, n), zface=(n == "rw")), hm.prognostic_fields()[k].interior_cpu()
        scale = mom if n in ("ru", "rv", "rw") else max(np.abs(want).max(), 1e-3)
        assert np.abs(got - want).max() / scale < 1e-9, (n, np.abs(got - want).max() / scale)
    assert float(hm.momentum["ρv"].interior[:, 0, :].abs().max()) == 0.0
    assert np.abs(g.interior(om.rv)[:, 1:]).max() > 1e-4          # Coriolis turned the wind: rho v is no longer zero inside


@pytest.mark.gpu
@pytest.mark.parametrize("walls", [False, True])
def test_cbl_float32_steps_match_the_float64_oracle(oracle, bz, walls):
    """the case in the precision the reference benchmarks it in (Float32), three steps against the Float64 oracle: App. C tolerances
    (1e-4 of the field scale after steps); walls: the PBB topology"""
    import torch
    size = (32, 24, 16)
    om = _oracle_cbl(oracle, size, topology=("Periodic", "Bounded" if walls else "Periodic", "Bounded"))
    hm = bz.benchmarks.convective_boundary_layer(size, float_type=np.float32, topology=(bz.Periodic, bz.Bounded if walls else bz.Periodic, bz.Bounded))
    assert hm.momentum["ρu"].parent.dtype == torch.float32
    for _ in range(3):
        om.time_step(0.5)
        hm.time_step(0.5)
    hm.synchronize()
    g = om.grid
    mom = max(np.abs(g.interior(getattr(om, n), zface=(n == "rw"))).max() for n in ("ru", "rv", "rw"))
    for n, k in PROG.items():
        if n == "rq":
            continue
        want, got = g.interior(getattr(om, n), zface=(n == "rw")), hm.prognostic_fields()[k].interior_cpu().astype(np.float64)
        scale = mom if n in ("ru", "rv", "rw") else np.abs(want).max()
        assert np.abs(got - want).max() / scale < 1e-4, (n, np.abs(got - want).max() / scale)


@pytest.mark.gpu
def test_cbl_lean_seam_equals_the_operator_sequence(bz):
    """The case's forcing stack (momentum profiles + Coriolis + bottom fluxes, no subsidence) rides the lean whole-step seam
    (csrc/bz_step.hip: bzi_lean_forcings_ok); three steps against the reference's call sequence through the per-operator entry
    points, whole parent arrays incl. halos: same arithmetic up to the rounding of alpha dt (G + F) against alpha dt G + alpha dt F."""
    models = []
    for whole in (True, False):
        m = bz.benchmarks.convective_boundary_layer((32, 24, 16), float_type=np.float64)
        for _ in range(3):
            bz.time_step_(m, 0.5, whole_step=whole)
        m.synchronize()
        models.append(m)
    a, b = models
    for k in a.prognostic_fields():
        x, y = a.prognostic_fields()[k].cpu(), b.prognostic_fields()[k].cpu()
        assert np.abs(x - y).max() / max(np.abs(y).max(), 1e-3) < 1e-12, k
    for fa, fb in ((a.temperature, b.temperature), (a.velocities["u"], b.velocities["u"]), (a.velocities["w"], b.velocities["w"])):
        assert np.abs(fa.cpu() - fb.cpu()).max() / np.abs(fb.cpu()).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("walls", [False, True])
def test_cbl_budgets_at_ci_size(bz, walls):
    """The case at the reference CI's smallest size (256 x 256 x 128) in Float64, ten steps on the tier the bench times (lean seam;
    walls: the operator-by-operator tier of the PBB topology) — size-independent properties instead of an oracle comparison:
    the only source of sum(rho theta dz) is the prescribed surface heat flux rho0 * 0.35 K m/s (advection is in flux form, the walls
    are closed), so the column content grows by exactly flux * dt per step; the momentum stays discretely divergence-free; the drag
    slows the lowest level."""
    import torch
    from breeze_jl_amd import benchmarks as bm
    size, dt, steps = (256, 256, 128), 0.05, 10
    m = bz.benchmarks.convective_boundary_layer(size, float_type=np.float64, topology=(bz.Periodic, bz.Bounded if walls else bz.Periodic, bz.Bounded))
    dz = bm.CBL["Lz"] / size[2]

    def content():
        return float(m.potential_temperature_density.interior.sum(dtype=torch.float64)) * dz / (size[0] * size[1])

    c0, u_low0 = content(), float(m.velocities["u"].interior[0].mean())
    for _ in range(steps):
        m.time_step(dt)
    m.synchronize()
    flux = bm.cbl_surface_density() * bm.CBL["heat_flux"]
    gained = content() - c0
    assert abs(gained - steps * dt * flux) < 2e-6 * steps * dt * flux, (gained, steps * dt * flux)
    assert m.max_abs_divergence() < 1e-10
    assert float(m.velocities["u"].interior[0].mean()) < u_low0 - 1e-4
    for f in (m.temperature, m.velocities["w"], m.momentum["ρv"]):
        assert bool(torch.isfinite(f.interior).all())
    if walls:
        assert float(m.momentum["ρv"].interior[:, 0, :].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("ftype", ["f64", "f32"])
def test_cbl_weno9_budgets_at_ci_size(bz, ftype):
    """WENO(order = 9) — the second scheme of the reference's CI matrix — at 256 x 256 x 128, ten steps through bz_time_steps_anelastic
    (what bench.py --workload cbl --cbl-order 9 times): scalars and x / y momentum run the single marching pass of
    csrc/bz_tendency_generic.hip (rows of a multiple of 64 cells), the z-momentum its two passes.  Same size-independent properties as
    above; Float32 is the precision the reference benchmarks in (sums accumulate in Float64 here, the fields are Float32)."""
    import torch
    from breeze_jl_amd import benchmarks as bm
    size, dt, steps = (256, 256, 128), 0.05, 10
    f32 = ftype == "f32"
    m = bz.benchmarks.convective_boundary_layer(size, float_type=np.float32 if f32 else np.float64, advection=bz.WENO(order=9))
    dz = bm.CBL["Lz"] / size[2]

    def content():
        return float(m.potential_temperature_density.interior.sum(dtype=torch.float64)) * dz / (size[0] * size[1])

    c0, u_low0 = content(), float(m.velocities["u"].interior[0].double().mean())
    m.time_steps(dt, steps)
    m.synchronize()
    flux = bm.cbl_surface_density() * bm.CBL["heat_flux"]
    gained = content() - c0
    # Float32: a stored rho theta carries 6e-8 of its ~350 kg K / m^3; the column content (~3e5) is known to ~1e-2 of the 1.9e-1 gained
    assert abs(gained - steps * dt * flux) < (5e-2 if f32 else 2e-6) * steps * dt * flux, (gained, steps * dt * flux)
    assert m.max_abs_divergence() < (1e-3 if f32 else 1e-10)
    assert float(m.velocities["u"].interior[0].double().mean()) < u_low0 - 1e-4
    for f in (m.temperature, m.velocities["w"], m.momentum["ρv"]):
        assert bool(torch.isfinite(f.interior).all())
