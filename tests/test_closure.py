"""closure = SmagorinskyLilly() (BASELINE configs[2]): analytic pins of the CPU restatement (oracle/closure.py — Oceananigans'
part of the closure is not vendored, "parity unpinned") and HIP-vs-oracle parity through the C ABI.  The first two CPU tests
restate the reference's own closure tests (test/turbulence_closures.jl:61-67, 98-139)."""
import numpy as np
import pytest

from helpers import PROG, push_state, relerr


def _model(oracle, size=(8, 8, 8), L=100.0, **kw):
    from oracle.closure import SmagorinskyLilly
    og = oracle.Grid(size, x=(0, L), y=(0, L), z=(0, L))
    return oracle.OracleModel(og, closure=SmagorinskyLilly(), **kw)


def test_smagorinsky_viscosity_under_mean_shear(oracle):
    """test/turbulence_closures.jl:61-67: rho u = z/100 gives nu_e > 0.  Away from the walls the value is the textbook one:
    Sigma^2 = S^2/2 for u = S z, so nu_e = (C_s Delta)^2 S in neutral stratification."""
    m = _model(oracle)
    m.set(theta=m.ref.theta0, ru=lambda x, y, z: z / 100 + 0 * x + 0 * y, enforce_mass_conservation=False)
    m.update_state()
    assert m.nu_e.max() > 0
    g = m.grid
    u = g.interior(m.u)[:, 0, 0]
    k = 4
    S = (u[k + 1] - u[k - 1]) / 25.0
    # the two faces carry slightly different shear (rho varies): compare with the face-averaged squares
    Sm, Sp = (u[k] - u[k - 1]) / 12.5, (u[k + 1] - u[k]) / 12.5
    want = (0.16 * 12.5) ** 2 * np.sqrt((Sm ** 2 + Sp ** 2) / 2)
    N2 = 0.0        # theta = theta0: neutral
    assert abs(m.nu_e[k, 3, 3] - want) < 1e-12 * want + 1e-3 * want * (N2 != 0)
    assert abs(S - (Sm + Sp) / 2) < 1e-12
    # top cell: the wall face has zero strain, so half of the off-diagonal square is missing
    assert m.nu_e[-1, 0, 0] < m.nu_e[-2, 0, 0]


def test_les_scalar_diffusion_changes_scalars(oracle):
    """test/turbulence_closures.jl:98-139: random momentum triggers nu_e > 0 and the scalars feel diffusion (compared with the
    same state without the closure, since the oracle always advects)."""
    rng = np.random.default_rng(0)
    gauss = lambda z: np.exp(-(z - 50.0) ** 2 / (2 * 10.0 ** 2))
    ic = dict(theta=lambda x, y, z: 288.0 + 10 * gauss(z) + 0 * x + 0 * y, qt=lambda x, y, z: 0.01 + 1e-3 * gauss(z) + 0 * x + 0 * y)
    noise = {n: rng.standard_normal((8 + (n == "rw"), 8, 8)) for n in ("ru", "rv", "rw")}
    out = []
    for closure in (True, False):
        if closure:
            m = _model(oracle)
        else:
            m = oracle.OracleModel(oracle.Grid((8, 8, 8), x=(0, 100.0), y=(0, 100.0), z=(0, 100.0)))
        m.set(ru=noise["ru"], rv=noise["rv"], rw=noise["rw"], **ic)
        m.update_state()
        out.append({n: m.grid.interior(m.G[n]).copy() for n in ("rtheta", "rq")})
    assert np.abs(out[0]["rtheta"] - out[1]["rtheta"]).max() > 0
    assert np.abs(out[0]["rq"] - out[1]["rq"]).max() > 0


def test_constant_viscosity_reduces_to_discrete_laplacians(oracle):
    """With nu_e held constant the restated fluxes must give rho nu (discrete second derivative): scalar d_xx theta and the
    momentum d_yy u from T_12 = -2 nu Sigma_12; both are checked against the exact discrete eigenvalue of a sine."""
    from oracle.closure import add_closure_tendencies
    N, L = 16, 160.0
    m = _model(oracle, size=(N, N, 8), L=L)
    g = m.grid
    kx = 2 * np.pi / L
    m.set(theta=lambda x, y, z: 288.0 + np.sin(kx * x) + 0 * y + 0 * z, u=lambda x, y, z: np.sin(kx * y) + 0 * x + 0 * z,
          enforce_mass_conservation=False)
    m.update_state()
    for n in m.G:
        m.G[n][...] = 0.0
    nu0 = 3.0
    m.nu_e = np.full((8, N, N), nu0)
    add_closure_tendencies(m)
    rho = m.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
    lam = (2 - 2 * np.cos(kx * g.dx)) / g.dx ** 2
    th = g.interior(m.theta)
    np.testing.assert_allclose(g.interior(m.G["rtheta"]), -rho * nu0 * lam * (th - 288.0), rtol=0, atol=1e-12)
    u = g.interior(m.u)
    np.testing.assert_allclose(g.interior(m.G["ru"]), -rho * nu0 * lam * u, rtol=0, atol=1e-12)
    assert np.abs(g.interior(m.G["rv"])).max() < 1e-13       # d_x T_12 vanishes for u(y)


def test_closure_conserves_scalars_and_horizontal_momentum(oracle):
    """Flux form with periodic sides and stress-free / no-flux walls: volume sums of every closure tendency vanish."""
    m = _model(oracle, size=(12, 10, 8), microphysics="SaturationAdjustment")
    rng = np.random.default_rng(1)
    sh = (8, 10, 12)
    m.set(theta=288.0 + rng.standard_normal(sh), qt=0.005 + 1e-3 * rng.random(sh), ru=rng.standard_normal(sh),
          rv=rng.standard_normal(sh), rw=rng.standard_normal((9, 10, 12)))
    m.update_state()
    from oracle.closure import add_closure_tendencies
    for n in m.G:
        m.G[n][...] = 0.0
    add_closure_tendencies(m)
    g = m.grid
    for n in ("ru", "rv", "rtheta", "rq"):
        G = g.interior(m.G[n])
        assert abs(G.sum()) < 1e-12 * np.abs(G).sum(), n
    assert np.abs(g.interior(m.G["rw"], zface=True)[1:-1]).max() > 0


# ---- GPU -----------------------------------------------------------------------------------------------------------------------

EXTENT = ((-3.2e3, 3.2e3), (-2e3, 2e3), (0.0, 3e3))


def _pair(oracle, bz, size=(32, 20, 16), moist=True, forced=False):
    from oracle.closure import SmagorinskyLilly
    from test_forcings import _hip_forcing_kwargs, _oracle_forcings
    og = oracle.Grid(size, x=EXTENT[0], y=EXTENT[1], z=EXTENT[2])
    om = oracle.OracleModel(og, surface_pressure=101500.0, potential_temperature=299.1,
                            microphysics="SaturationAdjustment" if moist else None, closure=SmagorinskyLilly(),
                            forcings=_oracle_forcings(oracle, og) if forced else None)
    grid = bz.RectilinearGrid(size, x=EXTENT[0], y=EXTENT[1], z=EXTENT[2])
    ref = bz.ReferenceState(grid, surface_pressure=101500.0, potential_temperature=299.1)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5), closure=bz.SmagorinskyLilly(),
                            microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium()) if moist else None,
                            **(_hip_forcing_kwargs(bz) if forced else {}))
    return om, hm


def _turbulent_ic(om, seed):
    g = om.grid
    rng = np.random.default_rng(seed)
    sh = (g.Nz, g.Ny, g.Nx)
    x, y, z = g.nodes("ccc")
    base_u = -8.75 + 1.5e-3 * z + 0 * x + 0 * y
    return dict(theta=298.7 + 0.004 * np.maximum(z - 520.0, 0.0) + 0.2 * rng.standard_normal(sh),
                qt=0.0185 * np.exp(-z / 2200.0) * (1 + 0.02 * rng.standard_normal(sh)),
                u=base_u + 0.6 * rng.standard_normal(sh), v=0.6 * rng.standard_normal(sh))


@pytest.mark.gpu
@pytest.mark.parametrize("moist", [False, True])
def test_eddy_viscosity_and_closure_tendencies_match_oracle(oracle, bz, moist):
    from oracle.closure import add_closure_tendencies
    om, hm = _pair(oracle, bz, moist=moist)
    om.set(**_turbulent_ic(om, 11))
    om.update_state()
    push_state(om, hm, names=("ru", "rv", "rw", "rtheta", "rq"))
    bz.update_state_(hm, compute_tendencies=True)
    hm.synchronize()
    g = om.grid
    nu = hm.closure_fields["νₑ"].interior_cpu()
    assert om.nu_e.max() > 0 and (om.nu_e == 0).any()           # sheared cells and stability-capped cells both present
    assert np.abs(nu - om.nu_e).max() < 1e-11 * om.nu_e.max()
    # the closure part alone (G with closure minus G of an identical model without): smooth arithmetic, tight
    om0 = oracle.OracleModel(g, surface_pressure=101500.0, potential_temperature=299.1,
                             microphysics="SaturationAdjustment" if moist else None)
    for n in ("ru", "rv", "rw", "rtheta", "rq"):
        getattr(om0, n)[...] = getattr(om, n)
    om0.update_state()
    strict = "refdiv" in bz.LIB_PATH
    for n, k in PROG.items():
        zf = n == "rw"
        want, base, got = g.interior(om.G[n], zface=zf), g.interior(om0.G[n], zface=zf), hm.G[k].interior_cpu()
        part = want - base
        assert np.abs(part).max() > 0, n
        assert relerr(got, want) < (1e-12 if strict else 5e-9), n
        if strict:
            assert np.abs((got - base) - part).max() < 1e-10 * np.abs(part).max(), n


@pytest.mark.gpu
@pytest.mark.parametrize("whole_step", [True, False])
def test_bomex_stack_time_steps_match_oracle(oracle, bz, whole_step):
    """WENO5 + saturation adjustment + SmagorinskyLilly + the full forcing stack: the physics list of BASELINE configs[2]."""
    om, hm = _pair(oracle, bz, moist=True, forced=True)
    ic = _turbulent_ic(om, 5)
    om.set(**ic)
    hm.set(θ=ic["theta"], qᵗ=ic["qt"], u=ic["u"], v=ic["v"])
    for _ in range(3):
        om.time_step(3.0)
        bz.time_step_(hm, 3.0, whole_step=whole_step)
    hm.synchronize()
    g = om.grid
    mom = max(np.abs(g.interior(getattr(om, n), zface=(n == "rw"))).max() for n in ("ru", "rv", "rw"))
    for n, k in PROG.items():
        want = g.interior(getattr(om, n), zface=(n == "rw"))
        got = hm.prognostic_fields()[k].interior_cpu()
        scale = mom if n in ("ru", "rv", "rw") else np.abs(want).max()
        assert np.abs(got - want).max() / scale < 2e-9, n
    assert om.nu_e.max() > 0.01


@pytest.mark.gpu
def test_closure_and_forcings_on_a_stretched_vertical_grid(oracle, bz):
    """Variable dz: every metric the closure and the subsidence profile use (dzc, dzf, face densities, the first-level flux
    divisor) differs from level to level."""
    from oracle.closure import SmagorinskyLilly
    from test_forcings import _hip_forcing_kwargs, _oracle_forcings
    size = (32, 20, 16)
    zf = 3e3 * (np.linspace(0.0, 1.0, size[2] + 1) ** 1.6)
    og = oracle.Grid(size, x=EXTENT[0], y=EXTENT[1], z=zf)
    om = oracle.OracleModel(og, surface_pressure=101500.0, potential_temperature=299.1, microphysics="SaturationAdjustment",
                            closure=SmagorinskyLilly(), forcings=_oracle_forcings(oracle, og))
    grid = bz.RectilinearGrid(size, x=EXTENT[0], y=EXTENT[1], z=zf)
    ref = bz.ReferenceState(grid, surface_pressure=101500.0, potential_temperature=299.1)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5), closure=bz.SmagorinskyLilly(),
                            microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium()), **_hip_forcing_kwargs(bz))
    om.set(**_turbulent_ic(om, 21))
    om.update_state()
    push_state(om, hm, names=("ru", "rv", "rw", "rtheta", "rq"))
    bz.update_state_(hm, compute_tendencies=True)
    hm.synchronize()
    g = om.grid
    assert np.abs(hm.closure_fields["νₑ"].interior_cpu() - om.nu_e).max() < 1e-11 * om.nu_e.max()
    strict = "refdiv" in bz.LIB_PATH
    for n, k in PROG.items():
        zfc = n == "rw"
        want, got = g.interior(om.G[n], zface=zfc), hm.G[k].interior_cpu()
        assert relerr(got, want) < (1e-12 if strict else 5e-9), n
    ic = _turbulent_ic(om, 22)
    om.set(**ic)
    hm.set(θ=ic["theta"], qᵗ=ic["qt"], u=ic["u"], v=ic["v"])
    for _ in range(2):
        om.time_step(2.0)
        hm.time_step(2.0)
    hm.synchronize()
    mom = max(np.abs(g.interior(getattr(om, n), zface=(n == "rw"))).max() for n in ("ru", "rv", "rw"))
    for n, k in PROG.items():
        want = g.interior(getattr(om, n), zface=(n == "rw"))
        got = hm.prognostic_fields()[k].interior_cpu()
        assert np.abs(got - want).max() / (mom if n in ("ru", "rv", "rw") else np.abs(want).max()) < 2e-9, n


@pytest.mark.gpu
@pytest.mark.parametrize("whole_step", [True, False])
def test_tracers_diffuse_with_the_closure(oracle, bz, whole_step):
    """User tracers are scalars like any other: - div J^c with kappa = nu_e / Pr enters their tendencies (scalar_tendency,
    update_atmosphere_model_state.jl:352-372).  SmagorinskyLilly + saturation adjustment + two tracers, three steps through the whole-step
    seam (divergence applied to rho c with the stage weight) and through the per-operator sequence (to G)."""
    from oracle.closure import SmagorinskyLilly
    size = (32, 20, 16)
    og = oracle.Grid(size, x=EXTENT[0], y=EXTENT[1], z=EXTENT[2])
    om = oracle.OracleModel(og, surface_pressure=101500.0, potential_temperature=299.1, microphysics="SaturationAdjustment",
                            closure=SmagorinskyLilly(), tracers=2)
    grid = bz.RectilinearGrid(size, x=EXTENT[0], y=EXTENT[1], z=EXTENT[2])
    ref = bz.ReferenceState(grid, surface_pressure=101500.0, potential_temperature=299.1)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5), closure=bz.SmagorinskyLilly(),
                            microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium()), tracers=("a", "b"))
    ic = _turbulent_ic(om, 7)
    Lx, Lz = EXTENT[0][1] - EXTENT[0][0], EXTENT[2][1] - EXTENT[2][0]
    a = lambda x, y, z: 1.0 + np.sin(2 * np.pi * x / Lx) * np.exp(-z / (0.5 * Lz)) + 0 * y
    b = lambda x, y, z: np.where(z < 0.4 * Lz, 1.0, 0.0) + 0 * x + 0 * y          # a sharp layer: the diffusion acts on it
    om.set(rc0=a, rc1=b, **ic)
    hm.tracers["a"].set_interior(a)
    hm.tracers["b"].set_interior(b)
    hm.set(θ=ic["theta"], qᵗ=ic["qt"], u=ic["u"], v=ic["v"])
    for _ in range(3):
        om.time_step(3.0)
        bz.time_step_(hm, 3.0, whole_step=whole_step)
    hm.synchronize()
    for n, k in (("rc0", "a"), ("rc1", "b")):
        want = og.interior(getattr(om, n))
        assert np.abs(hm.tracers[k].interior_cpu() - want).max() < 2e-9 * np.abs(want).max(), n
    want = og.interior(om.rtheta)
    assert np.abs(hm.potential_temperature_density.interior_cpu() - want).max() < 2e-9 * np.abs(want).max()
    # the closure did act on the tracer: without it the sharp layer of b evolves differently
    om2 = oracle.OracleModel(og, surface_pressure=101500.0, potential_temperature=299.1, microphysics="SaturationAdjustment", tracers=2)
    om2.set(rc0=a, rc1=b, **ic)
    for _ in range(3):
        om2.time_step(3.0)
    assert np.abs(og.interior(om2.rc1) - og.interior(om.rc1)).max() > 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("moist", [False, True])
def test_marching_closure_kernels_on_whole_tiles(oracle, bz, moist, monkeypatch):
    """Round 5: on grids of whole 64 x 8 tiles the eddy viscosity and the five closure divergences run as z-marching LDS-tiled kernels
    (csrc/bz_closure.hip: k_smagorinsky_march, k_closure_march; the grids of the tests above are narrower than a tile and keep the
    cell-per-thread kernels).  Same expressions in the same order: nu_e and every tendency carry the BITS of the cell-per-thread kernels
    (BZ_NO_CLOSURE_MARCH=1), and both match the oracle as above.  Three chunks of levels, so chunk seams are crossed."""
    size = (64, 16, 48)

    def run(no_march):
        if no_march:
            monkeypatch.setenv("BZ_NO_CLOSURE_MARCH", "1")
        else:
            monkeypatch.delenv("BZ_NO_CLOSURE_MARCH", raising=False)
        om, hm = _pair(oracle, bz, size=size, moist=moist)
        om.set(**_turbulent_ic(om, 5))
        om.update_state()
        push_state(om, hm, names=("ru", "rv", "rw", "rtheta", "rq"))
        bz.update_state_(hm, compute_tendencies=True)
        hm.synchronize()
        return om, hm

    om, a = run(False)
    _, b = run(True)
    nu_a, nu_b = a.closure_fields["νₑ"].interior_cpu(), b.closure_fields["νₑ"].interior_cpu()
    assert np.array_equal(nu_a, nu_b)
    assert np.abs(nu_a - om.nu_e).max() < 1e-11 * om.nu_e.max()
    strict = "refdiv" in bz.LIB_PATH
    for n, k in PROG.items():
        ga, gb = a.G[k].interior_cpu(), b.G[k].interior_cpu()
        assert np.array_equal(ga, gb), n
        want = om.grid.interior(om.G[n], zface=(n == "rw"))
        assert relerr(ga, want) < (1e-12 if strict else 5e-9), n


@pytest.mark.gpu
@pytest.mark.parametrize("size,dt", [((64, 16, 24), 3.0), ((320, 8, 12), 0.5)])
def test_level_sums_ride_on_the_projection_kernel(oracle, bz, size, dt, monkeypatch):
    """Round 5: inside one bz_time_step(s)_anelastic call the horizontal sums of u, v, theta, q^t that SubsidenceForcing averages
    (subsidence_forcing.jl:75-91) are emitted by the projection + diagnosis kernel of the previous stage (one value per wavefront,
    csrc/bz_fused.hip: PDFields::lsum; reduced by csrc/bz_forcing.hip: k_level_reduce) instead of a pass of their own over the four
    arrays; the first stage of a call keeps its own pass.  Rows of one wavefront (three of a block's four slots never written) and
    of five (a second block with one): three steps in one call against the oracle, against the same steps with BZ_NO_FUSE_LEVEL_SUMS=1
    (another summation order: 1e-12, not bits), and bit for bit against three single-step calls replayed from recorded graphs."""
    def run(no_ride, single_calls=0, graph=False):
        if no_ride:
            monkeypatch.setenv("BZ_NO_FUSE_LEVEL_SUMS", "1")
        else:
            monkeypatch.delenv("BZ_NO_FUSE_LEVEL_SUMS", raising=False)
        om, hm = _pair(oracle, bz, size=size, moist=True, forced=True)
        ic = _turbulent_ic(om, 9)
        hm.set(θ=ic["theta"], qᵗ=ic["qt"], u=ic["u"], v=ic["v"])
        if graph:
            hm.graph_enable(True)
        if single_calls:
            for _ in range(single_calls):
                hm.time_step(dt)
        else:
            hm.time_steps(dt, 3)
        hm.synchronize()
        return om, ic, {k: hm.prognostic_fields()[k].interior_cpu().copy() for k in PROG.values()}, hm

    om, ic, a, _ = run(False)
    _, _, b, _ = run(True)
    om.set(**ic)
    for _ in range(3):
        om.time_step(dt)
    g = om.grid
    mom = max(np.abs(g.interior(getattr(om, n), zface=(n == "rw"))).max() for n in ("ru", "rv", "rw"))
    for n, k in PROG.items():
        want = g.interior(getattr(om, n), zface=(n == "rw"))
        scale = mom if n in ("ru", "rv", "rw") else np.abs(want).max()
        assert np.abs(a[k] - want).max() / scale < 2e-9, n
        assert np.abs(a[k] - b[k]).max() / scale < 1e-12, n
    assert any(not np.array_equal(a[k], b[k]) for k in a)          # the riding sums were used (their order of summation differs)
    # single-step calls: stage 1 of every call sums in its own pass — in the order of the riding sums, so that n single-step calls leave the
    # bits of one n-step call (the contract of bz_time_steps_anelastic) —, stages 2 - 3 ride; replayed graphs leave the same bits
    _, _, c, _ = run(False, single_calls=3)
    _, _, d, hd = run(False, single_calls=6, graph=True)
    for _ in range(3):
        om.time_step(dt)
    for n, k in PROG.items():
        want = g.interior(getattr(om, n), zface=(n == "rw"))
        scale = mom if n in ("ru", "rv", "rw") else np.abs(want).max()
        assert np.array_equal(c[k], a[k]), n      # round 6: the first stage of a call sums in the riding order too (k_level_wave_sums)
        assert np.abs(d[k] - want).max() / scale < 4e-9, n
    en, cap, rep = hd.graph_info()
    if en and rep > 0:
        _, _, e, _ = run(False, single_calls=6)
        assert all(np.array_equal(d[k], e[k]) for k in d)
