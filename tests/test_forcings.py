"""The forcing / Coriolis / bottom-flux stack of the BOMEX configuration (BASELINE configs[2], examples/bomex.jl:80-207):
analytic pins of the CPU restatement (oracle/forcings.py) and HIP-vs-oracle parity through the C ABI.

The profiles are the Siebesma et al. (2003) appendix-B forms the example takes from AtmosphericProfilesLibrary (not vendored):
piecewise-linear subsidence, geostrophic wind, drying and radiative cooling."""
import numpy as np
import pytest

from helpers import PROG, push_state, relerr

F0, RHO0, USTAR = 3.76e-5, 1.15, 0.28


def ws_profile(z):
    return -6.5e-3 * z / 1500.0 if z <= 1500.0 else (-6.5e-3 * (1 - (z - 1500.0) / 600.0) if z <= 2100.0 else 0.0)


def ug_profile(z):
    return -10.0 + 1.8e-3 * z


def vg_profile(z):
    return 0.0


def drying_profile(z):
    return -1.2e-8 if z <= 300.0 else (-1.2e-8 * (1 - (z - 300.0) / 200.0) if z <= 500.0 else 0.0)


def cooling_profile(z):        # c_pd dT/dt
    dTdt = -2.0 / 86400.0 if z <= 1500.0 else (-2.0 / 86400.0 * (1 - (z - 1500.0) / 1500.0) if z <= 3000.0 else 0.0)
    return 1005.0 * dTdt


EXTENT = ((-3.2e3, 3.2e3), (-2e3, 2e3), (0.0, 3e3))


def _oracle_forcings(oracle, og, full=True):
    from oracle.forcings import ColumnForcings
    zc, zf = og.zc, og.zf
    col = lambda f, z: np.array([f(v) for v in z])
    kw = dict(Fu=-F0 * col(vg_profile, zc), Fv=F0 * col(ug_profile, zc), Fq=col(drying_profile, zc),
              Fe=col(cooling_profile, zc), w_subsidence=col(ws_profile, zf), coriolis_f=F0)
    if full:
        kw.update(flux_theta=RHO0 * 8e-3, flux_q=RHO0 * 5.2e-5, drag_rho0_ustar2=RHO0 * USTAR ** 2)
    return ColumnForcings(**kw)


def _hip_forcing_kwargs(bz, full=True):
    subsidence = bz.SubsidenceForcing(ws_profile)
    geo = bz.geostrophic_forcings(ug_profile, vg_profile)
    kw = dict(coriolis=bz.FPlane(f=F0),
              forcing={"u": (subsidence, geo.u), "v": (subsidence, geo.v), "θ": subsidence,
                       "qᵉ": (subsidence, bz.Forcing(drying_profile)), "e": bz.Forcing(cooling_profile)})
    if full:
        drag = bz.FieldBoundaryConditions(bottom=bz.FluxBoundaryCondition(bz.FrictionVelocityDrag(RHO0, USTAR)))
        kw["boundary_conditions"] = {"ρθ": bz.FieldBoundaryConditions(bottom=bz.FluxBoundaryCondition(RHO0 * 8e-3)),
                                     "ρqᵉ": bz.FieldBoundaryConditions(bottom=bz.FluxBoundaryCondition(RHO0 * 5.2e-5)),
                                     "ρu": drag, "ρv": drag}
    return kw


def _pair(oracle, bz, size=(32, 20, 16), moist=True, full=True):
    og = oracle.Grid(size, x=EXTENT[0], y=EXTENT[1], z=EXTENT[2])
    om = oracle.OracleModel(og, surface_pressure=101500.0, potential_temperature=299.1,
                            microphysics="SaturationAdjustment" if moist else None, forcings=_oracle_forcings(oracle, og, full))
    grid = bz.RectilinearGrid(size, x=EXTENT[0], y=EXTENT[1], z=EXTENT[2])
    ref = bz.ReferenceState(grid, surface_pressure=101500.0, potential_temperature=299.1)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5),
                            microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium()) if moist else None,
                            **_hip_forcing_kwargs(bz, full))
    return om, hm


def _ic(seed=3):
    rng = np.random.default_rng(seed)
    ph = rng.uniform(0, 2 * np.pi, 4)
    th = lambda x, y, z: 298.7 + 0.004 * np.maximum(z - 520.0, 0.0) + 0.3 * np.sin(2 * np.pi * x / 6.4e3 + ph[0]) * np.exp(-z / 800.0) + 0 * y
    qt = lambda x, y, z: 0.0185 * np.exp(-z / 2200.0) * (1 + 0.08 * np.cos(2 * np.pi * y / 4e3 + ph[1])) + 0 * x
    u = lambda x, y, z: -8.75 + 1.5e-3 * z + 0.5 * np.sin(2 * np.pi * y / 4e3 + ph[2]) + 0 * x
    v = lambda x, y, z: 0.4 * np.cos(2 * np.pi * x / 6.4e3 + ph[3]) + 0 * y + 0 * z
    return dict(theta=th, qt=qt, u=u, v=v)


# ---- CPU: analytic pins of the restatement -------------------------------------------------------------------------------------

def test_subsidence_profile_of_a_linear_mean_is_minus_w_times_slope():
    from oracle.forcings import subsidence_profile
    Nz, dz = 12, 50.0
    zc, zf = (np.arange(Nz) + 0.5) * dz, np.arange(Nz + 1) * dz
    avg = 300.0 + 0.004 * zc
    ws = -1e-3 * np.ones(Nz + 1)
    F = subsidence_profile(ws, avg, np.full(Nz + 1, dz))
    np.testing.assert_allclose(F, 1e-3 * 0.004, rtol=1e-10)          # one-sided ends carry the same slope
    # a z-dependent w_s: interior cells average the two face products, the ends use their single interior face
    ws = np.array([ws_profile(z) for z in zf * 4])
    F = subsidence_profile(ws, avg, np.full(Nz + 1, dz))
    np.testing.assert_allclose(F[3], -0.004 * (ws[3] + ws[4]) / 2, rtol=1e-10)
    np.testing.assert_allclose(F[0], -0.004 * ws[1], rtol=1e-10)
    np.testing.assert_allclose(F[-1], -0.004 * ws[Nz - 1], rtol=1e-10)


def test_geostrophic_balance_and_flux_bcs_in_the_oracle(oracle):
    """u = u_g, v = v_g: Coriolis and the geostrophic forcing cancel in both momentum tendencies (uniform-in-x,y flow has
    no advective tendency); the bottom fluxes land in the first level only, as +J / dz (the reference's
    test/forcing_and_boundary_conditions.jl:89-128 asserts exactly that for a flux function of (rho u, rho v))."""
    from oracle.forcings import ColumnForcings
    og = oracle.Grid((8, 6, 10), x=(0, 800.0), y=(0, 600.0), z=(0, 1000.0))
    zc = og.zc
    ug, vg = -10 + 1.8e-3 * zc, 2.0 + 0 * zc
    F = ColumnForcings(Fu=-F0 * vg, Fv=F0 * ug, coriolis_f=F0, flux_theta=0.0092, flux_q=6e-5, drag_rho0_ustar2=0.09)
    om = oracle.OracleModel(og, forcings=F)
    om.set(u=lambda x, y, z: -10 + 1.8e-3 * z + 0 * x + 0 * y, v=2.0, enforce_mass_conservation=False)
    om.update_state()
    g = og
    rho = om.ref.density[g.Hz:g.Hz + g.Nz]
    assert np.abs(g.interior(om.G["ru"])).max() < 1e-15 * F0 * 10 * rho.max() * 1e3
    assert np.abs(g.interior(om.G["rv"])).max() < 1e-15 * F0 * 10 * rho.max() * 1e3
    G0 = {n: om.G[n].copy() for n in om.G}
    from oracle.forcings import add_flux_bc_tendencies
    add_flux_bc_tendencies(om)
    dz = 100.0
    d = {n: g.interior(om.G[n]) - g.interior(G0[n]) for n in ("ru", "rv", "rtheta", "rq")}
    for n in d:
        assert np.all(d[n][1:] == 0)
    np.testing.assert_allclose(d["rtheta"][0], 0.0092 / dz, rtol=1e-12)
    np.testing.assert_allclose(d["rq"][0], 6e-5 / dz, rtol=1e-12)
    ru0, rv0 = rho[0] * ug[0], rho[0] * 2.0
    sp = np.hypot(ru0, rv0)
    np.testing.assert_allclose(d["ru"][0], -0.09 * ru0 / sp / dz, rtol=1e-10)
    np.testing.assert_allclose(d["rv"][0], -0.09 * rv0 / sp / dz, rtol=1e-10)


def test_energy_forcing_enters_theta_as_F_over_cpm_exner(oracle):
    from oracle.forcings import ColumnForcings
    og = oracle.Grid((4, 4, 8), x=(0, 400.0), y=(0, 400.0), z=(0, 2000.0))
    Fe = np.array([cooling_profile(z) for z in og.zc])
    om = oracle.OracleModel(og, forcings=ColumnForcings(Fe=Fe))
    om.update_state()
    c, r, g = om.constants, om.ref, og
    rho, p = r.density[g.Hz:g.Hz + g.Nz], r.pressure[g.Hz:g.Hz + g.Nz]
    want = rho * Fe / (c.cpd * (p / r.pst) ** (c.Rd / c.cpd))
    np.testing.assert_allclose(g.interior(om.G["rtheta"])[:, 1, 2], want, rtol=1e-12)


# ---- CPU: the reference's own known-answer tests for these forcings, restated on the oracle ----------------------------------------

@pytest.mark.parametrize("name", ["theta", "q", "u"])
def test_reference_subsidence_constant_gradient_known_answer(oracle, name):
    """test/geostrophic_subsidence_forcings.jl:296-339 ("Subsidence forcing gradient"): w_s = 1, phi = Gamma z, one step of
    dt = 1e-2 changes rho phi by rho_r (-dt w_s Gamma) in the bottom and in the top cell (rtol 1e-3) — the one-sided ends of the
    zb-average.  The reference runs it with advection = nothing; the oracle always advects, so phi rides on a resting,
    nearly neutral column (theta0 + Gamma z), where the advective change over one step is negligible at that tolerance."""
    from oracle.forcings import ColumnForcings
    og = oracle.Grid((4, 4, 4), x=(0, 10), y=(0, 10), z=(0, 16))
    ws, Gam, dt = 1.0, 1e-2, 1e-2
    F = ColumnForcings(w_subsidence=ws * np.ones(5), subsidence_on=(name,))
    m = oracle.OracleModel(og, forcings=F)
    lin = lambda x, y, z: Gam * z + 0 * x + 0 * y
    if name == "theta":
        m.set(theta=lambda x, y, z: m.ref.theta0 + lin(x, y, z))
    elif name == "q":
        m.set(theta=m.ref.theta0, qt=lin)
    else:
        m.set(theta=m.ref.theta0, u=lin, enforce_mass_conservation=False)
    key = {"theta": "rtheta", "q": "rq", "u": "ru"}[name]
    g = og
    before = g.interior(getattr(m, key)).copy()
    m.time_step(dt)
    after = g.interior(getattr(m, key))
    rho = m.ref.density[g.Hz:g.Hz + g.Nz]
    dphi = -dt * ws * Gam
    for k in (0, 3):
        assert (after - before)[k, 0, 0] == pytest.approx(rho[k] * dphi, rel=1e-3)


def test_reference_subsidence_multi_step_linear_accumulation(oracle):
    """test/geostrophic_subsidence_forcings.jl:233-294: the constant gradient is preserved, so N = 5 steps change rho theta by
    N rho_r (-dt w_s Gamma) everywhere (1e-3 of the expected change)."""
    from oracle.forcings import ColumnForcings
    og = oracle.Grid((4, 4, 4), x=(0, 10), y=(0, 10), z=(0, 16))
    ws, Gam, dt, N = 1.0, 1e-2, 1e-2, 5
    m = oracle.OracleModel(og, forcings=ColumnForcings(w_subsidence=ws * np.ones(5), subsidence_on=("theta",)))
    m.set(theta=lambda x, y, z: m.ref.theta0 + Gam * z + 0 * x + 0 * y)
    g = og
    before = g.interior(m.rtheta).copy()
    for _ in range(N):
        m.time_step(dt)
    rho = m.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
    expected = N * rho * (-dt * ws * Gam) + 0 * before
    assert np.abs((g.interior(m.rtheta) - before) - expected).max() < 1e-3 * np.abs(expected).max()


def test_reference_geostrophic_smoke_signs(oracle):
    """test/geostrophic_subsidence_forcings.jl:12-38,163-187: u_g = -10, v_g = 0, f = 1e-4, one step of 1e-6 from rest gives
    rho v < 0 everywhere (F_rho_v = +f rho_r u_g), also combined with subsidence."""
    from oracle.forcings import ColumnForcings
    og = oracle.Grid((4, 4, 4), x=(0, 100), y=(0, 100), z=(0, 100))
    for sub in (False, True):
        F = ColumnForcings(Fu=-1e-4 * np.zeros(4), Fv=1e-4 * (-10.0) * np.ones(4), coriolis_f=1e-4,
                           w_subsidence=-0.01 * np.ones(5) if sub else None, subsidence_on=("u", "v"))
        m = oracle.OracleModel(og, forcings=F)
        m.time_step(1e-6)
        assert og.interior(m.rv).max() < 0


# ---- GPU: parity through the C ABI ------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("moist", [False, True])
def test_forcing_tendencies_match_oracle(oracle, bz, moist):
    om, hm = _pair(oracle, bz, moist=moist)
    om.set(**_ic())
    om.update_state()
    push_state(om, hm, names=("ru", "rv", "rw", "rtheta", "rq"))
    bz.update_state_(hm, compute_tendencies=True)
    hm.synchronize()
    g = om.grid
    strict = "refdiv" in bz.LIB_PATH
    for n, k in PROG.items():
        zf = n == "rw"
        want, got = g.interior(om.G[n], zface=zf), hm.G[k].interior_cpu()
        if zf:
            want, got = want[1:-1], got[1:-1]
        assert relerr(got, want) < (1e-12 if strict else 5e-9), n
    # the forcing part alone (difference to an unforced model on the same state) is smooth data: tight without the WENO caveat
    om0 = oracle.OracleModel(g, surface_pressure=101500.0, potential_temperature=299.1,
                             microphysics="SaturationAdjustment" if moist else None)
    for n in ("ru", "rv", "rw", "rtheta", "rq"):
        getattr(om0, n)[...] = getattr(om, n)
    om0.update_state()
    for n, k in PROG.items():
        if n == "rw":
            continue
        forced = g.interior(om.G[n]) - g.interior(om0.G[n])
        assert np.abs(forced).max() > 0
        got = hm.G[k].interior_cpu() - g.interior(om0.G[n])
        assert np.abs(got - forced).max() < 1e-9 * np.abs(forced).max() + 1e-9 * np.abs(g.interior(om0.G[n])).max(), n
    # bottom fluxes
    from oracle.forcings import add_flux_bc_tendencies
    before = {k: hm.G[k].interior_cpu().copy() for k in PROG.values()}
    G0 = {n: g.interior(om.G[n], zface=(n == "rw")).copy() for n in PROG}
    add_flux_bc_tendencies(om)
    bz.compute_flux_bc_tendencies_(hm)
    hm.synchronize()
    for n, k in PROG.items():
        want = g.interior(om.G[n], zface=(n == "rw")) - G0[n]
        got = hm.G[k].interior_cpu() - before[k]
        if n == "rw":
            assert np.all(got == 0)
            continue
        assert np.abs(want).max() > 0
        assert np.abs(got - want).max() < 1e-9 * np.abs(want).max(), n


@pytest.mark.gpu
@pytest.mark.parametrize("whole_step", [True, False])
def test_forced_time_steps_match_oracle(oracle, bz, whole_step):
    om, hm = _pair(oracle, bz, moist=True)
    ic = _ic(seed=7)
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
        assert np.abs(got - want).max() / scale < 1e-9, n
    # the stack did something: compare with an unforced run of the same initial state
    om0 = oracle.OracleModel(g, surface_pressure=101500.0, potential_temperature=299.1, microphysics="SaturationAdjustment")
    om0.set(**ic)
    for _ in range(3):
        om0.time_step(3.0)
    assert np.abs(g.interior(om0.rtheta) - g.interior(om.rtheta)).max() > 1e-6
    assert np.abs(g.interior(om0.ru) - g.interior(om.ru)).max() > 1e-6


@pytest.mark.gpu
def test_forcing_interface_errors(bz):
    grid = bz.RectilinearGrid((16, 16, 8), x=(0, 1e3), y=(0, 1e3), z=(0, 1e3))
    mk = lambda **kw: bz.AtmosphereModel(grid, advection=bz.WENO(order=5), **kw)
    with pytest.raises(ValueError, match="specific"):
        mk(forcing={"ρθ": bz.SubsidenceForcing(ws_profile)})
    with pytest.raises(ValueError, match="coriolis"):
        mk(forcing={"u": bz.geostrophic_forcings(ug_profile, vg_profile).u})
    with pytest.raises(NotImplementedError):
        mk(forcing={"w": bz.Forcing(lambda z: 0.0)})


# ---- bulk aerodynamic bottom conditions ----------------------------------------------------------------------------------------

def test_reference_bulk_drag_known_answer(oracle):
    """test/forcing_and_boundary_conditions.jl:309-343: uniform u = 5, C^D = 1e-3, gustiness 0.1, T0 = 290 gives
    J^u = -rho0 C^D sqrt(U^2 + gustiness^2) U with rho0 = p0 / (R^d T0)."""
    from oracle.forcings import BulkFluxes, ColumnForcings, add_flux_bc_tendencies
    g = oracle.Grid((4, 4, 4), x=(0, 100), y=(0, 100), z=(0, 100))
    m = oracle.OracleModel(g, forcings=ColumnForcings(bulk=BulkFluxes(101325.0, 1e5, drag=(1e-3, 0.1, 290.0))))
    m.set(theta=m.ref.theta0, u=5.0, enforce_mass_conservation=False)
    m.update_state()
    for n in m.G:
        m.G[n][...] = 0.0
    add_flux_bc_tendencies(m)
    rho0 = 101325.0 / (m.constants.Rd * 290.0)
    want = -rho0 * 1e-3 * np.sqrt(25.0 + 0.01) * 5.0
    np.testing.assert_allclose(g.interior(m.G["ru"])[0] * 25.0, want, rtol=1e-13)
    assert np.all(g.interior(m.G["ru"])[1:] == 0) and np.all(g.interior(m.G["rv"]) == 0)


def test_reference_bulk_sensible_heat_vanishes_at_the_surface_equivalent_theta(oracle):
    """test/forcing_and_boundary_conditions.jl:252-276: with p0 != p_st the flux compares theta with the surface-equivalent
    theta0 = T0 (p_st/p0)^(R^d/c_pd); setting the air to exactly that value gives zero flux although theta != T0."""
    from oracle.forcings import BulkFluxes, ColumnForcings, add_flux_bc_tendencies
    g = oracle.Grid((4, 4, 4), x=(0, 100), y=(0, 100), z=(0, 100))
    T0, p0 = 290.0, 101325.0
    m = oracle.OracleModel(g, surface_pressure=p0, forcings=ColumnForcings(bulk=BulkFluxes(p0, 1e5, heat=(1e-3, 0.1, T0))))
    c = m.constants
    theta_s = T0 / (p0 / 1e5) ** (c.Rd / c.cpd)
    assert abs(theta_s - T0) > 0.5
    m.set(theta=theta_s, u=3.0, enforce_mass_conservation=False)
    m.update_state()
    for n in m.G:
        m.G[n][...] = 0.0
    add_flux_bc_tendencies(m)
    assert np.abs(g.interior(m.G["rtheta"])).max() < 1e-12


@pytest.mark.gpu
def test_bulk_surface_fluxes_match_oracle(oracle, bz):
    from oracle.forcings import BulkFluxes, ColumnForcings, add_flux_bc_tendencies
    size = (32, 20, 16)
    og = oracle.Grid(size, x=EXTENT[0], y=EXTENT[1], z=EXTENT[2])
    B = BulkFluxes(101500.0, 1e5, drag=(1.2e-3, 0.2, 299.8), heat=(1.1e-3, 0.2, 300.4), vapor=(1.3e-3, 0.1, 300.4))
    om = oracle.OracleModel(og, surface_pressure=101500.0, potential_temperature=299.1, microphysics="SaturationAdjustment",
                            forcings=ColumnForcings(bulk=B))
    grid = bz.RectilinearGrid(size, x=EXTENT[0], y=EXTENT[1], z=EXTENT[2])
    ref = bz.ReferenceState(grid, surface_pressure=101500.0, potential_temperature=299.1)
    bcs = {"ρu": bz.FieldBoundaryConditions(bottom=bz.BulkDrag(coefficient=1.2e-3, gustiness=0.2, surface_temperature=299.8)),
           "ρv": bz.FieldBoundaryConditions(bottom=bz.BulkDrag(coefficient=1.2e-3, gustiness=0.2, surface_temperature=299.8)),
           "ρθ": bz.FieldBoundaryConditions(bottom=bz.BulkSensibleHeatFlux(coefficient=1.1e-3, gustiness=0.2, surface_temperature=300.4)),
           "ρqᵉ": bz.FieldBoundaryConditions(bottom=bz.BulkVaporFlux(coefficient=1.3e-3, gustiness=0.1, surface_temperature=300.4))}
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5),
                            microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium()), boundary_conditions=bcs)
    ic = _ic(seed=13)
    om.set(**ic)
    om.update_state()
    push_state(om, hm, names=("ru", "rv", "rw", "rtheta", "rq"))
    bz.update_state_(hm, compute_tendencies=True)
    before = {k: hm.G[k].interior_cpu().copy() for k in PROG.values()}
    G0 = {n: og.interior(om.G[n], zface=(n == "rw")).copy() for n in PROG}
    add_flux_bc_tendencies(om)
    bz.compute_flux_bc_tendencies_(hm)
    hm.synchronize()
    for n, k in PROG.items():
        want = og.interior(om.G[n], zface=(n == "rw")) - G0[n]
        got = hm.G[k].interior_cpu() - before[k]
        if n == "rw":
            assert np.all(got == 0)
            continue
        assert np.abs(want).max() > 0, n
        assert np.abs(got - want).max() < 1e-9 * np.abs(want).max(), n
    om.set(**ic)
    hm.set(θ=ic["theta"], qᵗ=ic["qt"], u=ic["u"], v=ic["v"])
    for _ in range(3):
        om.time_step(3.0)
        hm.time_step(3.0)
    hm.synchronize()
    for n, k in PROG.items():
        want = og.interior(getattr(om, n), zface=(n == "rw"))
        got = hm.prognostic_fields()[k].interior_cpu()
        scale = max(np.abs(og.interior(om.ru)).max(), 1e-3) if n in ("ru", "rv", "rw") else np.abs(want).max()
        assert np.abs(got - want).max() / scale < 1e-9, n


@pytest.mark.gpu
@pytest.mark.parametrize("moist", [False, True])
def test_energy_flux_keyed_rho_e_enters_rho_theta_divided_by_the_mixture_heat_capacity(oracle, bz, moist):
    """boundary_conditions = (; ρe = FieldBoundaryConditions(bottom = FluxBoundaryCondition(Q))) in a potential-temperature model
    (BoundaryConditions.jl:218-227, thermodynamic_variable_bcs.jl: J_theta = Q / c_pm): tendency of the lowest level and three steps"""
    from oracle.forcings import ColumnForcings, add_flux_bc_tendencies
    size, Q = (32, 20, 16), 120.0
    og = oracle.Grid(size, x=EXTENT[0], y=EXTENT[1], z=EXTENT[2])
    om = oracle.OracleModel(og, surface_pressure=101500.0, potential_temperature=299.1, microphysics="SaturationAdjustment" if moist else None,
                            forcings=ColumnForcings(flux_energy=Q, flux_q=RHO0 * 5.2e-5))
    grid = bz.RectilinearGrid(size, x=EXTENT[0], y=EXTENT[1], z=EXTENT[2])
    ref = bz.ReferenceState(grid, surface_pressure=101500.0, potential_temperature=299.1)
    bcs = {"ρe": bz.FieldBoundaryConditions(bottom=bz.FluxBoundaryCondition(Q)),
           "ρqᵉ": bz.FieldBoundaryConditions(bottom=bz.FluxBoundaryCondition(RHO0 * 5.2e-5))}
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5), boundary_conditions=bcs,
                            microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium()) if moist else None)
    ic = _ic(seed=21)
    om.set(**ic)
    om.update_state()
    push_state(om, hm, names=("ru", "rv", "rw", "rtheta", "rq"))
    bz.update_state_(hm, compute_tendencies=True)
    before = hm.G["ρθ"].interior_cpu().copy()
    G0 = og.interior(om.G["rtheta"]).copy()
    add_flux_bc_tendencies(om)
    bz.compute_flux_bc_tendencies_(hm)
    hm.synchronize()
    want, got = og.interior(om.G["rtheta"]) - G0, hm.G["ρθ"].interior_cpu() - before
    assert np.abs(want[0]).min() > 0 and np.all(want[1:] == 0) and np.all(got[1:] == 0)
    assert np.abs(got - want).max() < 1e-9 * np.abs(want).max()
    cpd = 1005.0
    assert abs(want[0].mean() * og.dzc[og.Hz] - Q / cpd) < 0.03 * Q / cpd      # Q / c_pm, c_pm within 3 % of the dry value
    om.set(**ic)
    hm.set(θ=ic["theta"], qᵗ=ic["qt"], u=ic["u"], v=ic["v"])
    for _ in range(3):
        om.time_step(3.0)
        hm.time_step(3.0)
    hm.synchronize()
    for n, k in PROG.items():
        want = og.interior(getattr(om, n), zface=(n == "rw"))
        got = hm.prognostic_fields()[k].interior_cpu()
        scale = max(np.abs(og.interior(om.ru)).max(), 1e-3) if n in ("ru", "rv", "rw") else np.abs(want).max()
        assert np.abs(got - want).max() / scale < 1e-9, n
    with pytest.raises(ValueError):
        bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5),
                           boundary_conditions={"ρe": bz.FieldBoundaryConditions(bottom=bz.FluxBoundaryCondition(Q)),
                                                "ρθ": bz.FieldBoundaryConditions(bottom=bz.FluxBoundaryCondition(0.01))})
