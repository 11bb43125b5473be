"""(Bounded, Flat, Bounded): walls in x of a 2-D x-z model — the grid of the reference's examples/cloudy_thermal_bubble.jl:20-24 (128 x 128,
halo 5, WENO(order = 9), dry and then with saturation adjustment) and examples/tropical_cyclone_with_rainband.jl:294.

x is the lane dimension of every kernel here, so the wall treatment is per lane: WENO and Centered buffers by column (divergent only in the
first and last wavefront of a row), rho u / u with wall faces i = 0 and Nx (face Nx in the first upper halo cell), a no-flux halo cell for
centre-in-x fields, the cosine transform along x around the real row transform of the Flat-y pressure solve, a projection that leaves the
wall faces alone.  Semantics recalled from Oceananigans (PARITY UNPINNED, as for the walls in z and y); the oracle is checked through its
defining properties, the HIP path against the oracle."""
import numpy as np
import pytest

from helpers import PROG, push_state, randomize, relerr

EXT = dict(x=(-10e3, 10e3), z=(0.0, 10e3))
TOPO = ("Bounded", "Flat", "Bounded")


def theta_bubble(x, z, x0=-6e3):
    r = np.sqrt(((x - x0) / 2e3) ** 2 + ((z - 2e3) / 2e3) ** 2)
    return 300.0 + 2.0 * np.cos(np.pi * np.minimum(1.0, r) / 2) ** 2


def test_oracle_projects_exactly_and_keeps_the_walls_closed(oracle):
    g = oracle.Grid((32, 16), topology=TOPO, **EXT)
    m = oracle.OracleModel(g, potential_temperature=300.0)
    lam = oracle.poisson_eigenvalues(g.Nx, g.dx, oracle.BOUNDED)
    assert lam[0] == 0.0 and np.all(np.diff(lam) > 0)
    rng = np.random.default_rng(1)
    g.interior(m.ru)[...] = rng.standard_normal(g.interior(m.ru).shape)
    g.interior(m.rw, True)[1:-1] = rng.standard_normal(g.interior(m.rw, True)[1:-1].shape)
    m.fill_momentum_halos()
    assert np.all(g.interior(m.ru)[:, :, 0] == 0.0) and np.all(m.ru[:, :, g.Hx + g.Nx] == 0.0)
    m.compute_pressure_correction(1.0)
    m.make_pressure_correction(1.0)
    m.fill_momentum_halos()
    assert np.abs(m.divergence()).max() < 1e-13
    m.set(theta=lambda x, y, z: theta_bubble(x, z), u=lambda x, y, z: np.sin(np.pi * (x + 10e3) / 20e3) + 0 * z)
    s0 = g.interior(m.rtheta).sum()
    for _ in range(4):
        m.time_step(2.0)
    assert np.all(g.interior(m.ru)[:, :, 0] == 0.0)
    assert abs(g.interior(m.rtheta).sum() - s0) < 1e-13 * s0
    assert np.abs(g.interior(m.w, True)).max() > 1e-2


def _pair(oracle, bz, size, order, halo, **kw):
    g = oracle.Grid(size, topology=TOPO, halo=halo, **EXT)
    okw = {} if order == 5 else {"advection": f"WENO{order}"}
    om = oracle.OracleModel(g, potential_temperature=300.0, **okw, **kw.get("okw", {}))
    grid = bz.RectilinearGrid(size, topology=(bz.Bounded, bz.Flat, bz.Bounded), halo=halo, **EXT)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, potential_temperature=300.0)),
                            advection=bz.WENO(order=order), **kw.get("hkw", {}))
    return g, om, hm


@pytest.mark.gpu
@pytest.mark.parametrize("order", [5, 7, 9])
def test_bounded_x_tendencies_match_oracle(oracle, bz, order):
    """Nx = 16: every buffer of the cascade occurs next to both x walls"""
    g, om, hm = _pair(oracle, bz, (16, 14), order, (5, 5))
    randomize(om, seed=8)
    om.update_state(compute_tendencies=True)
    push_state(om, hm, names=("ru", "rv", "rw", "rtheta", "rq", "u", "v", "w", "theta", "q", "T"))
    for f in hm.G.values():
        f.parent.zero_()
    bz.compute_tendencies_(hm)
    hm.synchronize()
    for n, k in PROG.items():
        zf = n == "rw"
        want, got = g.interior(om.G[n], zface=zf), hm.G[k].interior_cpu()
        if zf:
            want, got = want[1:-1], got[1:-1]
        if n == "ru":      # the wall face i = 0 is never written
            want, got = want[:, :, 1:], got[:, :, 1:]
        assert relerr(got, want) < (1e-12 if order == 5 else 1e-11), (n, relerr(got, want))


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(32, 16), (128, 24), (200, 12)])
def test_bounded_x_pressure_solve_matches_oracle(oracle, bz, size):
    g, om, hm = _pair(oracle, bz, size, 5, (3, 3))
    rng = np.random.default_rng(5)
    g.interior(om.ru)[...] = rng.standard_normal(g.interior(om.ru).shape)
    g.interior(om.rw, True)[1:-1] = rng.standard_normal(g.interior(om.rw, True)[1:-1].shape)
    om.fill_momentum_halos()
    import torch
    for n, k in (("ru", "ρu"), ("rv", "ρv"), ("rw", "ρw")):
        hm.momentum[k].parent.copy_(torch.from_numpy(getattr(om, n)))
    om.compute_pressure_correction(0.7)
    bz.compute_pressure_correction_(hm, 0.7)
    hm.synchronize()
    assert relerr(hm.dynamics.pressure_anomaly.interior_cpu(), g.interior(om.phi)) < 1e-11
    om.make_pressure_correction(0.7)
    bz.make_pressure_correction_(hm, 0.7)
    for n, k in (("ru", "ρu"), ("rw", "ρw")):
        assert relerr(hm.momentum[k].interior_cpu(), g.interior(getattr(om, n), n == "rw")) < 1e-11, n
    assert hm.max_abs_divergence() < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("order", [5, 9])
def test_bounded_x_steps_match_oracle(oracle, bz, order):
    """the dry half of examples/cloudy_thermal_bubble.jl at reduced size: a bubble next to the west wall, WENO(order = 9), three steps"""
    g, om, hm = _pair(oracle, bz, (64, 32), order, (5, 5))
    th = lambda x, y, z: theta_bubble(x, z)
    om.set(theta=th)
    hm.set(θ=lambda x, z: theta_bubble(x, z))
    for _ in range(3):
        om.time_step(2.0)
        hm.time_step(2.0)
    hm.synchronize()
    mom = max(np.abs(g.interior(getattr(om, n), n == "rw")).max() for n in ("ru", "rw"))
    for n, f in (("ru", hm.momentum["ρu"]), ("rw", hm.momentum["ρw"]), ("rtheta", hm.potential_temperature_density), ("T", hm.temperature)):
        want, got = g.interior(getattr(om, n), n == "rw"), f.interior_cpu()
        scale = mom if n in ("ru", "rw") else np.abs(want).max()
        assert np.abs(got - want).max() < (1e-9 if order == 5 else 2e-8) * scale, n
    assert float(hm.momentum["ρu"].interior[:, :, 0].abs().max()) == 0.0
    assert hm.max_abs_divergence() < 1e-11


@pytest.mark.gpu
def test_cloudy_bubble_between_x_walls_matches_oracle(oracle, bz):
    """the moist half of the example: saturation adjustment (warm phase) in the walled 2-D box"""
    g, om, hm = _pair(oracle, bz, (64, 32), 9, (5, 5), okw=dict(microphysics="SaturationAdjustment"),
                      hkw=dict(microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium())))
    qt = lambda x, z: 0.020 * np.exp(-z / 2500.0) + 0 * x
    th = lambda x, z: theta_bubble(x, z) - 4.0 + 0.001 * z
    om.set(qt=lambda x, y, z: qt(x, z), theta=lambda x, y, z: th(x, z))
    hm.set(qᵗ=qt, θ=th)
    for _ in range(3):
        om.time_step(2.0)
        hm.time_step(2.0)
    hm.synchronize()
    assert (g.interior(om.ql) > 0).any()
    mom = max(np.abs(g.interior(getattr(om, n), n == "rw")).max() for n in ("ru", "rw"))
    for n, f in (("ru", hm.momentum["ρu"]), ("rw", hm.momentum["ρw"]), ("rtheta", hm.potential_temperature_density),
                 ("rq", hm.moisture_density), ("T", hm.temperature), ("ql", hm.microphysical_fields["qˡ"])):
        want, got = g.interior(getattr(om, n), n == "rw"), f.interior_cpu()
        scale = mom if n in ("ru", "rw") else max(np.abs(want).max(), 1e-6)
        assert np.abs(got - want).max() < 2e-8 * scale, n


@pytest.mark.gpu
@pytest.mark.parametrize("option", ["tracers", "StaticEnergy", "Kessler"])
def test_cell_local_options_between_x_walls(oracle, bz, option):
    """tracers (the scalar kernel again), the StaticEnergy formulation and the Kessler column physics in the walled 2-D box, three steps"""
    tol = 1e-9
    if option == "tracers":
        g, om, hm = _pair(oracle, bz, (32, 16), 5, (3, 3), okw=dict(tracers=1), hkw=dict(tracers=("a",)))
        a = lambda x, z: 1.0 + 0.5 * np.cos(np.pi * (x + 10e3) / 20e3) * (z / 10e3)
        om.set(theta=lambda x, y, z: theta_bubble(x, z), rc0=lambda x, y, z: a(x, z))
        hm.tracers["a"].set_interior(a)
        hm.set(θ=lambda x, z: theta_bubble(x, z))
        extra = [("rc0", hm.tracers["a"])]
    elif option == "StaticEnergy":
        g, om, hm = _pair(oracle, bz, (32, 16), 5, (3, 3), okw=dict(formulation="StaticEnergy"), hkw=dict(formulation="StaticEnergy"))
        om.set(theta=lambda x, y, z: theta_bubble(x, z))
        hm.set(θ=lambda x, z: theta_bubble(x, z))
        extra, tol = [], 2e-8          # e ~ 3e5 J/kg: two more digits of cancellation in the smoothness indicators (tests/test_bounded_y.py)
    else:
        tc = bz.ThermodynamicConstants(saturation_vapor_pressure=bz.TetensFormula())
        g = oracle.Grid((32, 20), topology=TOPO, x=(0.0, 4e3), z=(0.0, 5e3))
        om = oracle.OracleModel(g, surface_pressure=1e5, potential_temperature=300.0, microphysics="Kessler")
        grid = bz.RectilinearGrid((32, 20), topology=(bz.Bounded, bz.Flat, bz.Bounded), x=(0.0, 4e3), z=(0.0, 5e3))
        hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, tc, surface_pressure=1e5, potential_temperature=300.0)),
                                advection=bz.WENO(order=5), thermodynamic_constants=tc, microphysics=bz.DCMIP2016KesslerMicrophysics())
        bub = lambda x, z: np.maximum(0.0, 1.0 - np.sqrt((x - 1e3) ** 2 + (z - 1500.0) ** 2) / 1200.0)
        ic = dict(qt=lambda x, z: 0.016 * np.exp(-z / 3000.0) + 0.004 * bub(x, z), theta=lambda x, z: 300.0 + 0.004 * z + 1.0 * bub(x, z),
                  qcl=lambda x, z: 0.003 * bub(x, z), qr=lambda x, z: 0.001 * bub(x, z))
        om.set(**{k: (lambda f: (lambda x, y, z: f(x, z)))(f) for k, f in ic.items()}, u=2.0)
        hm.set(qᵗ=ic["qt"], θ=ic["theta"], qcl=ic["qcl"], qr=ic["qr"], u=2.0)
        μ = hm.microphysical_fields
        extra, tol = [("rqcl", μ["ρqᶜˡ"]), ("rqr", μ["ρqʳ"])], 1e-8
    for _ in range(3):
        om.time_step(2.0)
        hm.time_step(2.0)
    hm.synchronize()
    mom = max(np.abs(g.interior(getattr(om, n), n == "rw")).max() for n in ("ru", "rw"))
    for n, f in [("ru", hm.momentum["ρu"]), ("rw", hm.momentum["ρw"]), ("rtheta", hm.potential_temperature_density), ("T", hm.temperature)] + extra:
        want, got = g.interior(getattr(om, n), n == "rw"), f.interior_cpu()
        scale = mom if n in ("ru", "rw") else max(np.abs(want).max(), 1e-6)
        assert np.abs(got - want).max() < tol * scale, (n, np.abs(got - want).max() / scale)
    assert float(hm.momentum["ρu"].interior[:, :, 0].abs().max()) == 0.0


def test_closure_between_x_walls_keeps_them_closed(oracle):
    from oracle.closure import SmagorinskyLilly
    g = oracle.Grid((32, 16), topology=TOPO, halo=(3, 3), **EXT)
    m = oracle.OracleModel(g, potential_temperature=300.0, closure=SmagorinskyLilly())
    m.set(theta=lambda x, y, z: theta_bubble(x, z))
    s0 = g.interior(m.rtheta).sum()
    for _ in range(3):
        m.time_step(2.0)
    assert m.nu_e.max() > 0
    assert np.all(g.interior(m.ru)[:, :, 0] == 0.0)
    assert abs(g.interior(m.rtheta).sum() - s0) < 1e-13 * s0


@pytest.mark.gpu
def test_closure_between_x_walls_matches_oracle(oracle, bz):
    """SmagorinskyLilly in the walled 2-D box: nu_e mirrors across the walls, the wall face of rho u is never updated"""
    from oracle.closure import SmagorinskyLilly
    g, om, hm = _pair(oracle, bz, (64, 32), 5, (3, 3), okw=dict(closure=SmagorinskyLilly()), hkw=dict(closure=bz.SmagorinskyLilly()))
    om.set(theta=lambda x, y, z: theta_bubble(x, z), u=lambda x, y, z: 2.0 * np.sin(np.pi * (x - EXT["x"][0]) / (EXT["x"][1] - EXT["x"][0])) * z / 1e4)
    hm.set(θ=lambda x, z: theta_bubble(x, z), u=lambda x, z: 2.0 * np.sin(np.pi * (x - EXT["x"][0]) / (EXT["x"][1] - EXT["x"][0])) * z / 1e4)
    for _ in range(3):
        om.time_step(2.0)
        hm.time_step(2.0)
    hm.synchronize()
    nu = hm.closure_fields["νₑ"].interior_cpu()
    assert om.nu_e.max() > 0.01
    # after three steps: sqrt(1 - Cb N^2 / Sigma^2) next to its stability cutoff amplifies the 1e-12 differences of the state
    assert np.abs(nu - om.nu_e).max() < 1e-5 * om.nu_e.max()
    mom = max(np.abs(g.interior(getattr(om, n), n == "rw")).max() for n in ("ru", "rw"))
    for n, f in (("ru", hm.momentum["ρu"]), ("rw", hm.momentum["ρw"]), ("rtheta", hm.potential_temperature_density), ("T", hm.temperature)):
        want, got = g.interior(getattr(om, n), n == "rw"), f.interior_cpu()
        scale = mom if n in ("ru", "rw") else np.abs(want).max()
        assert np.abs(got - want).max() < 2e-9 * scale, n
    assert float(hm.momentum["ρu"].interior[:, :, 0].abs().max()) == 0.0
