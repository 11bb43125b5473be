"""(Periodic, Flat, Bounded): BASELINE configs[0], the reference's 2-D x-z dry thermal bubble (README.md:67-75,
examples/dry_thermal_bubble.jl), on the device as a genuinely two-dimensional problem (Ny = 1, no y halos): the per-operator
WENO-5 kernels drop their y terms, the pressure solve transforms rows only.  Checked against the oracle's own Flat implementation
(parity status as for the 3-D path: WENO / halos recalled from Oceananigans, unpinned) and against the y-invariant 3-D run."""
import numpy as np
import pytest

from helpers import PROG, relerr

pytestmark = pytest.mark.gpu
EXT = dict(x=(-10e3, 10e3), z=(0.0, 10e3))


def theta2d(x, z):      # README.md:71: 300 + 2 cos^2(pi/2 min(1, r / 2000)), r from (0, 2000 m)
    r = np.sqrt(x ** 2 + (z - 2000.0) ** 2)
    return 300.0 + 2.0 * np.cos(np.pi / 2 * np.minimum(1.0, r / 2000.0)) ** 2


def _pair(oracle, bz, size):
    og = oracle.Grid(size, x=EXT["x"], z=EXT["z"], topology=("Periodic", "Flat", "Bounded"))
    om = oracle.OracleModel(og, surface_pressure=101325.0, potential_temperature=300.0)
    grid = bz.RectilinearGrid(size, x=EXT["x"], z=EXT["z"], topology=(bz.Periodic, bz.Flat, bz.Bounded))
    ref = bz.ReferenceState(grid, surface_pressure=101325.0, potential_temperature=300.0)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5))
    return om, hm


def test_two_dimensional_tendencies_match_oracle(oracle, bz):
    import torch
    om, hm = _pair(oracle, bz, (48, 40))
    g = om.grid
    rng = np.random.default_rng(3)
    rho = om.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
    sh = (g.Nz, 1, g.Nx)
    g.interior(om.ru)[...] = rho * 4.0 * rng.standard_normal(sh)
    g.interior(om.rv)[...] = rho * 3.0 * rng.standard_normal(sh)        # v is advected as a passive component in 2-D
    wi = np.zeros((g.Nz + 1, 1, g.Nx))
    wi[1:-1] = 2.0 * rng.standard_normal((g.Nz - 1, 1, g.Nx))
    g.interior(om.rw, True)[...] = wi
    g.interior(om.rtheta)[...] = rho * (300.0 + 2.0 * rng.standard_normal(sh))
    g.interior(om.rq)[...] = rho * np.abs(5e-3 * rng.standard_normal(sh))
    om.update_state(compute_tendencies=True)
    from helpers import ORACLE_TO_HIP
    for n in ("ru", "rv", "rw", "rtheta", "rq"):
        ORACLE_TO_HIP[n](hm).parent.copy_(torch.from_numpy(getattr(om, n)))
    bz.update_state_(hm, compute_tendencies=True)
    hm.synchronize()
    for n, f in (("u", hm.velocities["u"]), ("w", hm.velocities["w"]), ("T", hm.temperature)):
        assert relerr(f.cpu(), getattr(om, n)) < 1e-14, n
    for n, k in PROG.items():
        zf = n == "rw"
        want, got = g.interior(om.G[n], zface=zf), hm.G[k].interior_cpu()
        if zf:
            want, got = want[1:-1], got[1:-1]
        assert relerr(got, want) < 1e-12, (n, relerr(got, want))


def test_config0_two_dimensional_bubble_steps_match_oracle(oracle, bz):
    """README.md:67-75 at reduced resolution (64 x 64 instead of 256 x 256 so that the CPU side stays in seconds), dt = 2 s"""
    om, hm = _pair(oracle, bz, (64, 64))
    om.set(theta=lambda x, y, z: theta2d(x, z) + 0 * y)
    hm.set(θ=theta2d)                                  # f(x, z), as on the reference's Flat grids
    for _ in range(3):
        om.time_step(2.0)
        hm.time_step(2.0)
    hm.synchronize()
    g = om.grid
    for n, k in PROG.items():
        got = hm.prognostic_fields()[k].interior_cpu()
        want = g.interior(getattr(om, n), zface=(n == "rw"))
        scale = max(np.max(np.abs(want)), 1e-3)
        assert np.max(np.abs(got - want)) / scale < 1e-9, (n, np.max(np.abs(got - want)) / scale)
    assert np.abs(hm.momentum["ρv"].interior_cpu()).max() == 0.0
    scale = np.max(np.abs(g.interior(om.ru))) / g.dx
    assert hm.max_abs_divergence() < 1e-12 * max(scale, 1e-6)


def test_two_dimensional_run_equals_the_y_invariant_three_dimensional_run(bz):
    size2, Ny = (64, 48), 8
    g2 = bz.RectilinearGrid(size2, x=EXT["x"], z=EXT["z"], topology=(bz.Periodic, bz.Flat, bz.Bounded))
    g3 = bz.RectilinearGrid((size2[0], Ny, size2[1]), x=EXT["x"], y=(0.0, 1.0 * Ny), z=EXT["z"])
    out = []
    for g in (g2, g3):
        m = bz.AtmosphereModel(g, dynamics=bz.AnelasticDynamics(bz.ReferenceState(g, potential_temperature=300.0)), advection=bz.WENO(order=5))
        m.set(θ=(theta2d if g is g2 else (lambda x, y, z: theta2d(x, z) + 0 * y)), u=2.0)
        for _ in range(3):
            m.time_step(2.0)
        m.synchronize()
        out.append({k: f.interior_cpu()[:, 0, :] for k, f in m.prognostic_fields().items()})
    for k in out[0]:
        scale = max(np.abs(out[1][k]).max(), 1e-3)
        assert np.abs(out[0][k] - out[1][k]).max() / scale < 1e-10, k


# ---- compressible split-explicit model on (Periodic, Flat, Bounded): examples/acoustic_wave.jl:51, inertia_gravity_wave.jl:70 ----------

def _cpair(oracle, bz, size=(40, 24), zext=(0.0, 8e3), xext=(-4e3, 4e3), theta_ref=300.0, **td):
    from oracle import oracle_compressible as oc
    og = oracle.Grid(size, x=xext, z=zext, topology=("Periodic", "Flat", "Bounded"))
    otd = oc.SplitExplicit(**td)
    om = oc.CompressibleOracleModel(og, time_discretization=otd, reference_potential_temperature=theta_ref, reference_state=True)
    grid = bz.RectilinearGrid(size, x=xext, z=zext, topology=(bz.Periodic, bz.Flat, bz.Bounded))
    damping = (bz.NoDivergenceDamping() if otd.damping_coefficient is None
               else bz.DirectDivergenceDamping(coefficient=otd.damping_coefficient) if otd.direct_damping
               else bz.ThermalDivergenceDamping(coefficient=otd.damping_coefficient, damp_vertical=otd.damp_vertical))
    sponge = None
    if otd.sponge is not None:
        ramp = {"linear": bz.LinearRamp, "cubic": bz.CubicRamp, "sin2": bz.Sin2Ramp}[otd.sponge[2]]()
        sponge = bz.UpperSponge(damping_rate=otd.sponge[0], depth=otd.sponge[1], ramp=ramp)
    btd = bz.SplitExplicitTimeDiscretization(substeps=otd.substeps, acoustic_cfl=otd.acoustic_cfl, forward_weight=otd.forward_weight,
                                             damping=damping, sponge=sponge,
                                             apply_first_substep_pressure_gradient=otd.apply_first)
    dyn = bz.CompressibleDynamics(btd, reference_potential_temperature=theta_ref, reference_state="auto")
    hm = bz.CompressibleAtmosphereModel(grid, dyn, advection=bz.WENO(order=5))
    return oc, om, hm


def test_compressible_two_dimensional_update_state_and_slow_tendencies_match_oracle(oracle, bz):
    import test_gpu_compressible as tc
    oc, om, hm = _cpair(oracle, bz)
    tc.seeded_state(om, 11)
    om.compute_slow_tendencies()
    tc.push(om, hm, substepper=False)
    bz.compressible.update_state_(hm)
    tc.cmp_interior(om, hm, ("rho", "u", "v", "w", "theta", "q", "T", "p"), 1e-14)
    for k in hm.G:
        if k != "ρq":
            hm.G[k].parent.zero_()
    bz.compressible.compute_slow_tendencies_(hm)
    g = om.grid
    for n, k in tc.PROG.items():
        if n == "rq":
            continue
        a, b = hm.G[k].interior_cpu(), g.interior(om.G[n], n == "rw")
        if n == "rw":
            a, b = a[1:-1], b[1:-1]
        assert tc.rel(a, b) <= 1e-12, (n, tc.rel(a, b))


@pytest.mark.parametrize("td", [dict(substeps=6), dict(), dict(substeps=4, damping_coefficient=0.05, damp_vertical=True),
                                dict(substeps=6, direct_damping=True), dict(substeps=6, sponge=(0.2, 3000.0, "cubic"))])
def test_compressible_two_dimensional_acoustic_loop_matches_oracle(oracle, bz, td):
    """acoustic_rk3_substep_loop! on a Flat-y grid: the y neighbours of every column are the column itself, Flat spacings do not enter
    the acoustic CFL or the damping length (acoustic_substepping.jl:458-465, 1102-1110)."""
    import test_gpu_compressible as tc
    oc, om, hm = _cpair(oracle, bz, **td)
    tc.seeded_state(om, 4)
    for n in om.PROGNOSTIC:
        om.U0[n][...] = getattr(om, n)
    g = om.grid
    rng = np.random.default_rng(5)
    for n in ("rho_d", "rtheta", "ru", "rv"):
        g.interior(getattr(om, n))[...] *= 1 + 1e-3 * rng.standard_normal((g.Nz, g.Ny, g.Nx))
    g.interior(om.rw, True)[1:-1] *= 1 + 1e-3 * rng.standard_normal((g.Nz - 1, g.Ny, g.Nx))
    om.update_state(compute_tendencies=True)
    om.refresh_linearization()
    om.compute_slow_tendencies()
    tc.push(om, hm)
    bz.compressible.refresh_linearization_(hm)
    dt, beta = 2.0, 1.0
    om.acoustic_substep_loop(dt, beta)
    bz.compressible.acoustic_rk3_substep_loop_(hm, dt, beta)
    n_tau, _ = hm.stage_substeps(dt, beta)
    assert n_tau == om.last_substeps[-1]
    sub = hm.timestepper.substepper
    for n, k in tc.SUB.items():
        if n in ("Pi", "thL", "gR"):
            continue
        zf = n in ("rwp", "aw", "Gs")
        a, b = getattr(sub, k).interior_cpu(), g.interior(getattr(om, n), zf)
        scale = {"rp": np.abs(g.interior(om.rho_d)).max() * 1e-3, "rthp": np.abs(g.interior(om.rtheta)).max() * 1e-3}.get(n)
        err = np.abs(a - b).max() / (scale or max(np.abs(b).max(), 1e-300))
        assert err <= 2e-10, (n, err, n_tau)      # rounding level: rp / rthp are scaled by 1e-3 of the full density
    tc.cmp_interior(om, hm, ("rho_d", "rtheta", "ru", "rv", "rw", "u", "v", "w"), 5e-12)


def test_compressible_inertia_gravity_wave_steps_match_oracle(oracle, bz):
    """examples/inertia_gravity_wave.jl:63-95 (Skamarock & Klemp 1994) at reduced length: theta_bg(z) = theta_0 exp(N^2 z / g) with the
    0.01 K perturbation, 20 m/s mean wind, split-explicit compressible dynamics; three steps against the oracle."""
    Nx, Nz, Lx, Lz = 96, 10, 96e3, 10e3
    th0, N2, grav = 300.0, 1e-4, 9.80665
    thbg = lambda z: th0 * np.exp(N2 * z / grav)
    thi = lambda x, z: thbg(z) + 0.01 * np.sin(np.pi * z / Lz) / (1 + (x - Lx / 3) ** 2 / 5000.0 ** 2)
    oc, om, hm = _cpair(oracle, bz, size=(Nx, Nz), zext=(0.0, Lz), xext=(0.0, Lx), theta_ref=thbg)
    g = om.grid
    rho = om.ref.density[g.Hz:g.Hz + g.Nz][:, None, None] + np.zeros((Nz, 1, Nx))
    om.set(rho=rho, theta=lambda x, y, z: thi(x, z) + 0 * y, u=20.0, v=0.0, w=0.0)
    hm.set(ρ=rho, θ=thi, u=20.0, v=0.0, w=0.0)
    import test_gpu_compressible as tc
    tc.cmp_interior(om, hm, ("rho_d", "rho", "rtheta", "ru", "T", "p"), 1e-14)
    for _ in range(3):
        om.time_step(6.0)
        hm.time_step(6.0)
    assert np.abs(g.interior(om.rw, True)).max() > 1e-6
    tc.cmp_interior(om, hm, ("rho_d", "rtheta", "ru", "rw", "u", "w", "theta", "T", "p"), 5e-9)
    assert np.abs(hm.momentum["ρv"].interior_cpu()).max() == 0.0


def test_dry_thermal_bubble_example_configuration_matches_oracle(oracle, bz):
    """examples/dry_thermal_bubble.jl:15-25 as written, at reduced resolution: (Periodic, Flat, Bounded) with halo (5, 5),
    formulation = :StaticEnergy, advection = WENO(order = 9), the 10 K bubble on a weakly stratified background."""
    size, ext = (64, 48), dict(x=(-10e3, 10e3), z=(0.0, 10e3))
    og = oracle.Grid(size, topology=("Periodic", "Flat", "Bounded"), halo=(5, 5), **ext)
    om = oracle.OracleModel(og, surface_pressure=101325.0, potential_temperature=288.0, formulation="StaticEnergy", advection="WENO9")
    grid = bz.RectilinearGrid(size, topology=(bz.Periodic, bz.Flat, bz.Bounded), halo=(5, 5), **ext)
    ref = bz.ReferenceState(grid, surface_pressure=101325.0, potential_temperature=288.0)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), formulation=":StaticEnergy", advection=bz.WENO(order=9))

    def thi(x, z):      # :36-48: theta_0 exp(N^2 z / g) + 10 K cos^2 bubble of radius 2 km at 0.3 Lz
        r = np.sqrt(x ** 2 + (z - 3000.0) ** 2)
        return 288.0 * np.exp(1e-6 * z / 9.80665) + 10.0 * np.where(r < 2000.0, np.cos(np.pi / 2 * r / 2000.0) ** 2, 0.0)

    om.set(theta=lambda x, y, z: thi(x, z) + 0 * y)
    hm.set(θ=thi)
    for _ in range(3):
        om.time_step(1.0)
        hm.time_step(1.0)
    hm.synchronize()
    g = om.grid
    assert np.abs(g.interior(om.rw, True)).max() > 1e-2
    for n, k in PROG.items():
        got = hm.prognostic_fields()[k].interior_cpu()
        want = g.interior(getattr(om, n), zface=(n == "rw"))
        scale = max(np.max(np.abs(want)), 1e-3)
        assert np.max(np.abs(got - want)) / scale < 2e-8, (n, np.max(np.abs(got - want)) / scale)      # the WENO-9 tolerance of tests/test_weno_orders.py
    assert np.abs(hm.momentum["ρv"].interior_cpu()).max() == 0.0


def test_two_dimensional_kessler_and_tracer_model_matches_oracle(oracle, bz):
    """2-D moist convection ingredients: DCMIP2016 Kessler species and a user tracer on (Periodic, Flat, Bounded), two steps against the oracle."""
    size, ext = (48, 24), dict(x=(0.0, 12e3), z=(0.0, 6e3))
    og = oracle.Grid(size, topology=("Periodic", "Flat", "Bounded"), **ext)
    om = oracle.OracleModel(og, surface_pressure=1e5, potential_temperature=300.0, microphysics="Kessler", tracers=1)
    grid = bz.RectilinearGrid(size, topology=(bz.Periodic, bz.Flat, bz.Bounded), **ext)
    tc = bz.ThermodynamicConstants(saturation_vapor_pressure=bz.TetensFormula())
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, tc, surface_pressure=1e5, potential_temperature=300.0)),
                            advection=bz.WENO(order=5), thermodynamic_constants=tc, microphysics=bz.DCMIP2016KesslerMicrophysics(), tracers=("a",))
    bub = lambda x, z: np.maximum(0.0, 1.0 - np.hypot(x - 6e3, z - 1500.0) / 1200.0)
    ic = dict(qt=lambda x, z: 0.016 * np.exp(-z / 3000.0) + 0.004 * bub(x, z), theta=lambda x, z: 300.0 + 0.004 * z + 1.0 * bub(x, z),
              qcl=lambda x, z: 0.003 * bub(x, z), qr=lambda x, z: 0.001 * bub(x, z))
    a = lambda x, z: 1.0 + 0.5 * np.sin(2 * np.pi * x / 12e3) * (z / 6e3)
    om.set(u=2.0, rc0=lambda x, y, z: a(x, z) + 0 * y, **{k: (lambda f: (lambda x, y, z: f(x, z) + 0 * y))(v) for k, v in ic.items()})
    hm.tracers["a"].set_interior(a)
    hm.set(qᵗ=ic["qt"], θ=ic["theta"], qcl=ic["qcl"], qr=ic["qr"], u=2.0)
    for _ in range(2):
        om.time_step(5.0)
        hm.time_step(5.0)
    hm.synchronize()
    g, μ = om.grid, hm.microphysical_fields
    mom = max(np.abs(g.interior(getattr(om, n), n == "rw")).max() for n in ("ru", "rw"))
    for n, f in (("ru", hm.momentum["ρu"]), ("rw", hm.momentum["ρw"]), ("rtheta", hm.potential_temperature_density), ("rq", hm.moisture_density),
                 ("rqr", μ["ρqʳ"]), ("T", hm.temperature), ("rc0", hm.tracers["a"])):
        want = g.interior(getattr(om, n), n == "rw")
        scale = mom if n in ("ru", "rw") else max(np.abs(want).max(), 1e-6)
        assert np.abs(f.interior_cpu() - want).max() / scale < 1e-8, n


def test_two_dimensional_smagorinsky_model_matches_oracle(oracle, bz):
    """SmagorinskyLilly on (Periodic, Flat, Bounded): the y derivatives of the strain vanish, the corner averages in y collapse onto the row
    and the filter width uses the unit spacing of the Flat direction; sheared moist layer, three steps against the oracle's closure."""
    from oracle.closure import SmagorinskyLilly
    size, ext = (64, 32), dict(x=(0.0, 6.4e3), z=(0.0, 3e3))
    og = oracle.Grid(size, topology=("Periodic", "Flat", "Bounded"), **ext)
    om = oracle.OracleModel(og, surface_pressure=101500.0, potential_temperature=299.1, microphysics="SaturationAdjustment", closure=SmagorinskyLilly())
    grid = bz.RectilinearGrid(size, topology=(bz.Periodic, bz.Flat, bz.Bounded), **ext)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, surface_pressure=101500.0, potential_temperature=299.1)),
                            advection=bz.WENO(order=5), closure=bz.SmagorinskyLilly(),
                            microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium()))
    rng = np.random.default_rng(4)
    noise = rng.standard_normal((size[1], 1, size[0]))
    x, y, z = og.nodes("ccc")
    th = 298.7 + 0.004 * np.maximum(z - 520.0, 0.0) + 0.2 * noise
    qt = 0.0185 * np.exp(-z / 2200.0) * (1 + 0.02 * noise)
    u = -8.75 + 3e-3 * z + 0.6 * noise
    om.set(theta=th, qt=qt, u=u, v=0.3 * noise)
    hm.set(θ=th, qᵗ=qt, u=u, v=0.3 * noise)
    for _ in range(3):
        om.time_step(2.0)
        hm.time_step(2.0)
    hm.synchronize()
    assert om.nu_e.max() > 0.01
    mom = max(np.abs(og.interior(getattr(om, n), n == "rw")).max() for n in ("ru", "rv", "rw"))
    for n, k in PROG.items():
        want = og.interior(getattr(om, n), zface=(n == "rw"))
        got = hm.prognostic_fields()[k].interior_cpu()
        scale = mom if n in ("ru", "rv", "rw") else np.abs(want).max()
        assert np.abs(got - want).max() / scale < 2e-9, (n, np.abs(got - want).max() / scale)
    assert relerr(hm.closure_fields["νₑ"].interior_cpu(), om.nu_e) < 1e-9


def test_two_dimensional_bounded_moisture_tendency_matches_oracle(oracle, bz):
    """Bounds-preserving WENO on a (Periodic, Flat, Bounded) grid with v != 0 (the bounded kernel had no Flat branch
    and added a spurious y-flux divergence read from other z levels): device vs the oracle's Flat implementation, 1e-12."""
    import torch
    from helpers import ORACLE_TO_HIP
    size = (48, 40)
    og = oracle.Grid(size, x=EXT["x"], z=EXT["z"], topology=("Periodic", "Flat", "Bounded"))
    om = oracle.OracleModel(og, surface_pressure=101325.0, potential_temperature=300.0)
    om.bounded = {"rq": (0.0, 1.0)}
    grid = bz.RectilinearGrid(size, x=EXT["x"], z=EXT["z"], topology=(bz.Periodic, bz.Flat, bz.Bounded))
    ref = bz.ReferenceState(grid, surface_pressure=101325.0, potential_temperature=300.0)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref),
                            advection={"momentum": bz.WENO(), "ρθ": bz.WENO(), "ρqᵛ": bz.WENO(bounds=(0.0, 1.0))})
    g = om.grid
    rng = np.random.default_rng(9)
    rho = om.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
    sh = (g.Nz, 1, g.Nx)
    g.interior(om.ru)[...] = rho * 4.0 * rng.standard_normal(sh)
    g.interior(om.rv)[...] = rho * 3.0 * rng.standard_normal(sh)
    wi = np.zeros((g.Nz + 1, 1, g.Nx))
    wi[1:-1] = 2.0 * rng.standard_normal((g.Nz - 1, 1, g.Nx))
    g.interior(om.rw, True)[...] = wi
    g.interior(om.rtheta)[...] = rho * (300.0 + 2.0 * rng.standard_normal(sh))
    g.interior(om.rq)[...] = rho * np.abs(0.3 * rng.standard_normal(sh))      # the lower bound bites
    om.update_state(compute_tendencies=True)
    for n in ("ru", "rv", "rw", "rtheta", "rq"):
        ORACLE_TO_HIP[n](hm).parent.copy_(torch.from_numpy(getattr(om, n)))
    bz.update_state_(hm, compute_tendencies=True)
    hm.synchronize()
    want, got = g.interior(om.G["rq"]), hm.G["ρq"].interior_cpu()
    assert relerr(got, want) < 1e-12, relerr(got, want)


def test_two_dimensional_default_advection_is_rejected(bz):
    """advection = nothing resolves to Centered(order = 2) before the Flat guard; that combination is not built"""
    grid = bz.RectilinearGrid((16, 16), x=EXT["x"], z=EXT["z"], topology=(bz.Periodic, bz.Flat, bz.Bounded))
    with pytest.raises(NotImplementedError):
        bz.AtmosphereModel(grid)


@pytest.mark.gpu
def test_two_dimensional_forcing_stack_matches_oracle(oracle, bz):
    """the BOMEX forcing / boundary-condition stack on a (Periodic, Flat, Bounded) grid — what the 2-D
    examples/prescribed_sea_surface_temperature.jl:39-73 and radiative_convection.jl:57-62 attach to their x-z models: f-plane + geostrophic
    + subsidence + drying / cooling profiles + bottom fluxes + friction-velocity drag, saturation adjustment; three steps vs the oracle"""
    import test_forcings as tf
    size = (64, 24)
    ext = dict(x=tf.EXTENT[0], z=tf.EXTENT[2])
    og = oracle.Grid(size, topology=("Periodic", "Flat", "Bounded"), **ext)
    om = oracle.OracleModel(og, surface_pressure=101500.0, potential_temperature=299.1, microphysics="SaturationAdjustment",
                            forcings=tf._oracle_forcings(oracle, og, True))
    grid = bz.RectilinearGrid(size, topology=(bz.Periodic, bz.Flat, bz.Bounded), **ext)
    ref = bz.ReferenceState(grid, surface_pressure=101500.0, potential_temperature=299.1)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5),
                            microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium()), **tf._hip_forcing_kwargs(bz, True))
    ic = tf._ic()
    om.set(theta=ic["theta"], qt=lambda x, y, z: 0.0185 * np.exp(-z / 2200.0) + 0 * x, u=lambda x, y, z: -8.75 + 1.5e-3 * z + 0 * x, v=ic["v"])
    hm.set(θ=lambda x, z: ic["theta"](x, 0.0, z), qᵗ=lambda x, z: 0.0185 * np.exp(-z / 2200.0) + 0 * x, u=lambda x, z: -8.75 + 1.5e-3 * z + 0 * x,
           v=lambda x, z: ic["v"](x, 0.0, z))
    for _ in range(3):
        om.time_step(3.0)
        hm.time_step(3.0)
    hm.synchronize()
    mom = max(np.abs(og.interior(getattr(om, n), n == "rw")).max() for n in ("ru", "rv", "rw"))
    for n, f in (("ru", hm.momentum["ρu"]), ("rv", hm.momentum["ρv"]), ("rw", hm.momentum["ρw"]), ("rtheta", hm.potential_temperature_density),
                 ("rq", hm.moisture_density), ("T", hm.temperature)):
        want, got = og.interior(getattr(om, n), n == "rw"), f.interior_cpu()
        scale = mom if n in ("ru", "rv", "rw") else max(np.abs(want).max(), 1e-6)
        assert np.abs(got - want).max() < 2e-9 * scale, (n, np.abs(got - want).max() / scale)


@pytest.mark.gpu
def test_two_dimensional_bulk_surface_fluxes_match_oracle(oracle, bz):
    """BulkDrag / BulkSensibleHeatFlux / BulkVaporFlux (constant coefficients) on the 2-D grid of
    examples/prescribed_sea_surface_temperature.jl:39-42 (its polynomial coefficients are outside this build), with a w sponge; three steps"""
    import test_forcings as tf
    from oracle.forcings import BulkFluxes, ColumnForcings
    size = (64, 24)
    ext = dict(x=tf.EXTENT[0], z=tf.EXTENT[2])
    og = oracle.Grid(size, topology=("Periodic", "Flat", "Bounded"), **ext)
    B = BulkFluxes(101500.0, 1e5, drag=(1.2e-3, 0.2, 299.8), heat=(1.1e-3, 0.2, 300.4), vapor=(1.3e-3, 0.1, 300.4))
    om = oracle.OracleModel(og, surface_pressure=101500.0, potential_temperature=299.1, microphysics="SaturationAdjustment",
                            forcings=ColumnForcings(bulk=B))
    ztop = tf.EXTENT[2][1]
    mask = lambda z: np.exp(-(z - ztop) ** 2 / (2 * (0.15 * ztop) ** 2))
    om.relaxation = {"rw": (0.05 * mask(og.zf), np.zeros(og.Nz + 1))}
    grid = bz.RectilinearGrid(size, topology=(bz.Periodic, bz.Flat, bz.Bounded), **ext)
    ref = bz.ReferenceState(grid, surface_pressure=101500.0, potential_temperature=299.1)
    bcs = {"ρu": bz.FieldBoundaryConditions(bottom=bz.BulkDrag(coefficient=1.2e-3, gustiness=0.2, surface_temperature=299.8)),
           "ρv": bz.FieldBoundaryConditions(bottom=bz.BulkDrag(coefficient=1.2e-3, gustiness=0.2, surface_temperature=299.8)),
           "ρe": bz.FieldBoundaryConditions(bottom=bz.BulkSensibleHeatFlux(coefficient=1.1e-3, gustiness=0.2, surface_temperature=300.4)),
           "ρqᵉ": bz.FieldBoundaryConditions(bottom=bz.BulkVaporFlux(coefficient=1.3e-3, gustiness=0.1, surface_temperature=300.4))}
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5), boundary_conditions=bcs,
                            microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium()),
                            forcing={"ρw": bz.Relaxation(rate=0.05, mask=mask)})
    ic = tf._ic(seed=13)
    om.set(theta=ic["theta"], qt=lambda x, y, z: 0.0185 * np.exp(-z / 2200.0) + 0 * x, u=lambda x, y, z: -8.75 + 1.5e-3 * z + 0 * x, v=ic["v"])
    hm.set(θ=lambda x, z: ic["theta"](x, 0.0, z), qᵗ=lambda x, z: 0.0185 * np.exp(-z / 2200.0) + 0 * x, u=lambda x, z: -8.75 + 1.5e-3 * z + 0 * x,
           v=lambda x, z: ic["v"](x, 0.0, z))
    for _ in range(3):
        om.time_step(3.0)
        hm.time_step(3.0)
    hm.synchronize()
    mom = max(np.abs(og.interior(getattr(om, n), n == "rw")).max() for n in ("ru", "rv", "rw"))
    for n, f in (("ru", hm.momentum["ρu"]), ("rv", hm.momentum["ρv"]), ("rw", hm.momentum["ρw"]), ("rtheta", hm.potential_temperature_density),
                 ("rq", hm.moisture_density), ("T", hm.temperature)):
        want, got = og.interior(getattr(om, n), n == "rw"), f.interior_cpu()
        scale = mom if n in ("ru", "rv", "rw") else max(np.abs(want).max(), 1e-6)
        assert np.abs(got - want).max() < 2e-9 * scale, (n, np.abs(got - want).max() / scale)
