"""Pins for the compressible split-explicit oracle: the reference's own known-answer tests restated
(test/acoustic_substepping_components.jl, test/substepper_structural.jl,
test/substepper_rest_state.jl).  None of these needs a GPU."""
import ctypes as C
import math

import numpy as np
import pytest


@pytest.fixture(scope="module")
def oc(oracle):
    from oracle import oracle_compressible
    return oracle_compressible


def test_first_small_step_pressure_gradient_gate(oc):
    """test/acoustic_substepping_components.jl:50-56"""
    f = oc.apply_horizontal_pressure_gradient_substep
    assert f(1, 1)
    assert not f(1, 2)
    assert f(2, 2)
    assert not f(1, 6)
    assert f(6, 6)
    assert f(1, 6, True)


def test_explicit_horizontal_step_known_answer(oracle, oc):
    """test/acoustic_substepping_components.jl:58-93 — Pi = gammaR = 1, p = 2x + 3y, rho theta' = 0,
    dtau = 0.5, gate off  =>  (rho u)'[2,2,2] == -1, (rho v)'[2,2,2] == -1.5 exactly."""
    g = oracle.Grid((4, 4, 4), x=(0, 4), y=(0, 4), z=(0, 4))
    m = oc.CompressibleOracleModel(g, reference_state=False)
    x, y, _ = g.nodes("ccc")
    g.interior(m.p)[...] = 2 * x + 3 * y
    g.interior(m.Pi)[...] = 1.0
    g.interior(m.gR)[...] = 1.0
    from oracle.oracle import _p
    m.lib.og_explicit_horizontal_step(C.byref(m.cg), _p(m.rup), _p(m.rvp), _p(m.p), _p(m.rthp), _p(m.Pi), _p(m.gR),
                                      _p(m.G["ru"]), _p(m.G["rv"]), C.c_double(0.5), C.c_int(0))
    assert g.interior(m.rup)[1, 1, 1] == -1.0
    assert g.interior(m.rvp)[1, 1, 1] == -1.5


def test_acoustic_tridiagonal_coefficients_closed_form(oracle, oc):
    """test/acoustic_substepping_components.jl:95-166"""
    Nz, Lz = 5, 1000.0
    g = oracle.Grid((4, 4, Nz), x=(0, 1), y=(0, 1), z=(0, Lz))
    m = oc.CompressibleOracleModel(g, reference_state=False)
    I = g.interior
    for k in range(1, Nz + 1):
        I(m.Pi)[k - 1, 1, 1] = 0.90 + 0.02 * k
        I(m.thL)[k - 1, 1, 1] = 280 + 3 * k
        I(m.gR)[k - 1, 1, 1] = 390 + 5 * k
    for f in (m.Pi, m.thL, m.gR):
        m._halo_center(f)
    dtn, dnew, grav, dz = 0.7, 0.03, 9.81, Lz / Nz
    m.cg.g = grav
    from oracle.oracle import _p
    fn = m.lib.og_acoustic_coefficient
    fn.restype = C.c_double

    def code(row, which):   # Julia row index -> 0-based row
        return fn(C.byref(m.cg), _p(m.Pi), _p(m.thL), _p(m.gR), C.c_int(1), C.c_int(1), C.c_int(row - 1),
                  C.c_int(which), C.c_double(dtn), C.c_double(dnew))

    Cc = lambda k: I(m.gR)[k - 1, 1, 1] * I(m.Pi)[k - 1, 1, 1]
    th = lambda k: I(m.thL)[k - 1, 1, 1]
    thf = lambda k: th(1) if k == 1 else (th(Nz) if k == Nz + 1 else (th(k) + th(k - 1)) / 2)
    assert code(1, 1) == 1.0
    assert code(1, 2) == 0.0
    for k in range(2, Nz + 1):
        lower = -dtn ** 2 * Cc(k - 1) * thf(k - 1) / dz ** 2 + dtn ** 2 * grav / (2 * dz) - dnew / dz ** 2
        diag = 1 + dtn ** 2 * thf(k) * (Cc(k) + Cc(k - 1)) / dz ** 2 + 2 * dnew / dz ** 2
        assert code(k, 0) == pytest.approx(lower, rel=1e-13)
        assert code(k, 1) == pytest.approx(diag, rel=1e-13)
    for k in range(2, Nz):
        upper = -dtn ** 2 * Cc(k) * thf(k + 1) / dz ** 2 - dtn ** 2 * grav / (2 * dz) - dnew / dz ** 2
        assert code(k, 2) == pytest.approx(upper, rel=1e-13)


def test_compute_acoustic_substeps(oracle, oc):
    """test/acoustic_substepping_components.jl:274-313"""
    c = oracle.Constants()
    g = oracle.Grid((100, 6, 10), x=(0, 100e3), y=(0, 6e3), z=(0, 10e3), halo=(5, 5, 5))
    expected = lambda nu: math.ceil(12 * math.sqrt(1.4 * 287.0 * 300) / (nu * 1000))
    assert oc.compute_acoustic_substeps(g, 12, c, 0.5) == expected(0.5) == 9
    assert oc.compute_acoustic_substeps(g, 12, c, 0.25) == expected(0.25)
    assert oc.compute_acoustic_substeps(g, 12, c, 1.0) == expected(1.0)
    assert oc.compute_acoustic_substeps(g, -12, c, 0.5) == expected(0.5)
    gf = oracle.Grid((100, 10), x=(0, 100e3), z=(0, 10e3), halo=(5, 5), topology=("Periodic", "Flat", "Bounded"))
    assert oc.compute_acoustic_substeps(gf, 12, c, 0.5) == expected(0.5)
    # ProportionalSubsteps (acoustic_substepping.jl:491-495)
    assert oc.stage_substep_count_and_size(6, 1 / 3, 3.0, g, c, 0.5) == (2, 0.5)
    assert oc.stage_substep_count_and_size(6, 0.5, 3.0, g, c, 0.5) == (3, 0.5)
    n, dtau = oc.stage_substep_count_and_size(None, 1.0, 12.0, g, c, 0.5)
    assert n == 9 and dtau == 12.0 / 9


T0_REST, G_REST, CPD_REST = 250.0, 9.80665, 1005.0


def theta_isothermal(z):
    return T0_REST * np.exp(G_REST * z / (CPD_REST * T0_REST))


def rest_model(oracle, oc, size=(8, 8, 32), Lz=30e3, Lh=100e3, **td):
    g = oracle.Grid(size, x=(0, Lh), y=(0, Lh), z=(0, Lz), halo=(5, 5, 5))
    m = oc.CompressibleOracleModel(g, time_discretization=oc.SplitExplicit(**td), surface_pressure=1e5,
                                   standard_pressure=1e5, reference_potential_temperature=theta_isothermal)
    return m


def set_rest(m):
    """test/substepper_structural.jl:53-63: rho_d = rho_ref, rho theta = p_ref / (Rd Pi_ref), u = 0."""
    m.rho_d[...] = m.ref.density[:, None, None]
    m.rtheta[...] = (m.ref.pressure / (m.constants.Rd * np.where(m.ref.exner_function == 0, 1, m.ref.exner_function)))[:, None, None]
    m.update_state()


def test_exner_reference_state_discrete_balance(oracle, oc):
    """test/substepper_rest_state.jl T1 (residual <= 1e-9) and components.jl:508-520 (p decreasing)."""
    m = rest_model(oracle, oc, size=(4, 4, 64))
    g, r = m.grid, m.ref
    Hz, Nz = g.Hz, g.Nz
    p, rho = r.pressure[Hz:Hz + Nz], r.density[Hz:Hz + Nz]
    dz = g.dzf[Hz + 1:Hz + Nz]
    res = (p[1:] - p[:-1]) / dz + m.constants.g * (rho[1:] + rho[:-1]) / 2
    assert np.abs(res).max() <= 1e-9
    assert np.all(np.diff(p) < 0)
    # Value BC halos: Iz(p)[face 0] = p0, Iz(rho)[face 0] = rho0
    assert 0.5 * (r.pressure[Hz - 1] + r.pressure[Hz]) == pytest.approx(1e5, rel=1e-15)
    assert r.rho0 > 0


def test_rest_state_pressure_consistency_and_slow_tendency(oracle, oc):
    """test/substepper_rest_state.jl T2, T3"""
    m = rest_model(oracle, oc, size=(8, 8, 64))
    set_rest(m)
    g = m.grid
    pref = m.ref.pressure[g.Hz:g.Hz + g.Nz][:, None, None]
    assert np.abs(g.interior(m.p) - pref).max() <= 100 * np.finfo(float).eps * pref.max()
    assert np.all(g.interior(m.rho_d) == m.ref.density[g.Hz:g.Hz + g.Nz][:, None, None])
    m.refresh_linearization()
    m.compute_slow_tendencies()
    m.assemble_slow_vertical_momentum()
    assert np.abs(g.interior(m.Gs, True)).max() <= 1e-12


def test_rest_state_structural_invariants(oracle, oc):
    """test/substepper_structural.jl S1, S4, S5: bottom row of the column system, mass conservation and
    closed lid after one outer step at rest (omega = 0.55, no damping)."""
    m = rest_model(oracle, oc, forward_weight=0.55, damping_coefficient=None)
    set_rest(m)
    g = m.grid
    M0 = g.interior(m.rho_d).sum()
    m.time_step(0.5)
    M1 = g.interior(m.rho_d).sum()
    assert abs(M1 - M0) / M0 <= 1e-12
    assert np.abs(g.interior(m.w, True)[-1]).max() <= 1e-12
    assert np.abs(g.interior(m.w, True)[0]).max() == 0.0
    assert m.last_substeps == [1, 1, 1]


def test_rest_atmosphere_stays_quiet(oracle, oc):
    """test/acoustic_substepping_stability.jl:329-353 (shape): a balanced rest state stays at rest
    under default damping / off-centering, w_max < sqrt(eps)."""
    m = rest_model(oracle, oc, size=(8, 8, 16), Lz=10e3, Lh=8e3)
    set_rest(m)
    for _ in range(5):
        m.time_step(6.0)
    g = m.grid
    assert np.isfinite(g.interior(m.rho_d)).all()
    assert np.abs(g.interior(m.w, True)).max() < math.sqrt(np.finfo(float).eps)
    assert np.abs(g.interior(m.u)).max() < math.sqrt(np.finfo(float).eps)
    assert m.last_substeps[-1] >= 2


def test_tiny_dry_bubble_consistency_with_anelastic(oracle, oc):
    """test/acoustic_substepping_stability.jl:192-322 restated: 16x16 (Periodic, Flat, Bounded) bubble,
    pressure-balanced initial density, split-explicit (substeps = 6) vs anelastic after one dt = 0.5 step:
    max updraft within rtol 1.25, updraft centroid within 2 dz; dry mass / rho*theta conserved."""
    c = oracle.Constants()
    kap = c.Rd / c.cpd
    p0 = pst = 1e5
    th0, dth, radius, zb = 300.0, 10.0, 2e3, 3e3
    exner = lambda z: (p0 / pst) ** kap - c.g * z / (c.cpd * th0)
    pref = lambda z: pst * exner(z) ** (1 / kap)
    theta = lambda x, y, z: th0 + dth * np.maximum(0.0, 1.0 - np.sqrt(x ** 2 + (z - zb) ** 2) / radius)
    rho = lambda x, y, z: pref(z) / (c.Rd * theta(x, y, z) * exner(z))
    kw = dict(x=(-8e3, 8e3), z=(0, 8e3), halo=(5, 5), topology=("Periodic", "Flat", "Bounded"))
    g = oracle.Grid((16, 16), **kw)
    m = oc.CompressibleOracleModel(g, time_discretization=oc.SplitExplicit(substeps=6), surface_pressure=p0,
                                   standard_pressure=pst, reference_potential_temperature=th0)
    m.set(rho=rho, theta=theta, qv=0.0)
    M0, H0 = g.interior(m.rho_d).sum(), g.interior(m.rtheta).sum()
    m.time_step(0.5)
    assert m.last_substeps == [2, 3, 6]
    ga = oracle.Grid((16, 16), **kw)
    a = oracle.OracleModel(ga, surface_pressure=p0, potential_temperature=th0, standard_pressure=pst)
    a.set(theta=theta, qt=0.0)
    a.time_step(0.5)

    def diag(mod, gg):
        w = np.maximum(0.0, gg.interior(mod.w, True))
        return w.max(), (w.sum(axis=(1, 2)) * gg.zf).sum() / w.sum()

    (ws, zs), (wa, za) = diag(m, g), diag(a, ga)
    assert np.isfinite(g.interior(m.w, True)).all()
    assert ws > 0 and wa > 0
    assert abs(ws - wa) <= 1.25 * max(ws, wa)
    assert abs(zs - za) <= 2 * (8e3 / 16)
    assert abs(g.interior(m.rho_d).sum() - M0) / M0 < 1e-13
    assert abs(g.interior(m.rtheta).sum() - H0) / H0 < 1e-13
    print("tiny bubble: split-explicit w_max %.4f zW %.1f | anelastic w_max %.4f zW %.1f" % (ws, zs, wa, za))


def test_moist_exner_reference_state_discrete_balance_and_dry_limit(oracle, oc):
    """ExnerReferenceState with vapor_mass_fraction (reference_states.jl:572-672): level-local R_m, c_pm, kappa_m; the discrete
    balance holds to rounding at every interior face, the moist EOS holds level by level, q^v = 0 reproduces the dry column
    bit for bit, and a moist column is lighter than the dry one at the same pressure."""
    g = oracle.Grid((4, 4, 40), x=(0, 4e3), y=(0, 4e3), z=(0, 20e3))
    c = oracle.Constants()
    th = lambda z: 300.0 + 43.0 * (np.minimum(z, 12e3) / 12e3) ** 1.25 + 0.02 * np.maximum(z - 12e3, 0.0)
    qv = lambda z: 0.014 * np.exp(-z / 3e3)
    dry = oc.ExnerReferenceState(g, c, 1e5, th, 1e5)
    dry0 = oc.ExnerReferenceState(g, c, 1e5, th, 1e5, vapor_mass_fraction=0.0)
    moist = oc.ExnerReferenceState(g, c, 1e5, th, 1e5, vapor_mass_fraction=qv)
    Hz, Nz = g.Hz, g.Nz
    for a in ("pressure", "density", "exner_function"):
        assert np.array_equal(getattr(dry, a), getattr(dry0, a))
    p, rho = moist.pressure[Hz:Hz + Nz], moist.density[Hz:Hz + Nz]
    res = (p[1:] - p[:-1]) / g.dzf[Hz + 1:Hz + Nz] + c.g * (rho[1:] + rho[:-1]) / 2
    assert np.abs(res).max() < 1e-10 * c.g * rho.max()
    q = np.array([qv(z) for z in g.zc])
    Rm = (1 - q) * c.Rd + q * c.Rv
    cpm = (1 - q) * c.cpd + q * c.cpv
    T = th(g.zc) * (p / 1e5) ** (Rm / cpm)
    np.testing.assert_allclose(rho, p / (Rm * T), rtol=1e-13)
    assert moist.density[Hz] < dry.density[Hz]
    # the model picks it up: a resting moist column on its own reference has no slow vertical-momentum tendency
    m = oc.CompressibleOracleModel(g, time_discretization=oc.SplitExplicit(substeps=4), surface_pressure=1e5,
                                   reference_potential_temperature=th, reference_vapor_mass_fraction=qv)
    m.set(rho=m.ref.density[Hz:Hz + Nz][:, None, None], theta=lambda x, y, z: th(z) + 0 * x + 0 * y,
          qv=lambda x, y, z: qv(z) + 0 * x + 0 * y, u=0.0, v=0.0, w=0.0)
    m.time_step(2.0)
    assert np.abs(g.interior(m.rw, True)).max() < 1e-8


def test_host_moist_exner_reference_state_equals_oracle(oracle, oc, bz):
    """The host-side column builder (breeze.jl_amd/compressible.py) and the oracle's follow the same recurrence: identical digits."""
    size, ext = (4, 4, 24), dict(x=(0, 4e3), y=(0, 4e3), z=(0, 18e3))
    th = lambda z: 300.0 + 40.0 * (np.minimum(z, 12e3) / 12e3) ** 1.25 + 0.015 * np.maximum(z - 12e3, 0.0)
    qv = lambda z: float(0.012 * np.exp(-z / 2500.0))
    og = oracle.Grid(size, **ext)
    ro = oc.ExnerReferenceState(og, oracle.Constants(), 1e5, th, 1e5, vapor_mass_fraction=qv)
    grid = bz.RectilinearGrid(size, **ext)
    rh = bz.compressible.ExnerReferenceState(grid, surface_pressure=1e5, potential_temperature=th, standard_pressure=1e5,
                                             vapor_mass_fraction=qv)
    Hz, Nz = og.Hz, og.Nz
    sl = slice(Hz - 1, Hz + Nz + 1)
    for a in ("pressure", "density", "exner_function"):
        np.testing.assert_allclose(getattr(rh, a)[sl], getattr(ro, a)[sl], rtol=1e-15, atol=0)


def test_density_based_temperature_inversion_reference_known_answers(oracle, oc):
    """test/compressible_saturation_adjustment.jl:31-44,76-96 (NumericalEarth/Breeze.jl#765): the density-based theta^li -> T
    inversion is the self-consistent fixed point T = (rho R_m T / p_st)^kappa theta + L (rtol 1e-9 with the default Newton
    solver), lies above the non-iterated closed form theta^gamma (rho R_m / p_st)^(gamma-1) + L by about 1.39 kappa L
    (rtol 0.15), and reduces to the closed form without condensate."""
    g = oracle.Grid((4, 4, 4), x=(0, 400.0), y=(0, 400.0), z=(0, 400.0))
    m = oc.CompressibleOracleModel(g, time_discretization=oc.SplitExplicit(substeps=2), surface_pressure=1e5,
                                   reference_potential_temperature=300.0, microphysics="Kessler")
    I = g.interior
    c, t = m.constants, m.tetens
    for qv, ql in ((0.020, 0.0), (0.018, 0.005)):
        rho = 1.0
        I(m.rho_d)[...] = rho * (1 - qv - ql)
        I(m.rq)[...], I(m.rqcl)[...], I(m.rqr)[...] = rho * qv, rho * ql, 0.0
        I(m.rtheta)[...] = I(m.rho_d) * 300.0
        m.update_state(compute_tendencies=False)
        T, th, r = I(m.T)[1, 1, 1], I(m.theta)[1, 1, 1], I(m.rho)[1, 1, 1]
        assert th == pytest.approx(300.0, rel=1e-15) and r == pytest.approx(1.0, rel=1e-15)
        Rm = (1 - qv - ql) * c.Rd + qv * c.Rv
        cpm = (1 - qv - ql) * c.cpd + qv * c.cpv + ql * t.cl
        kap, gam = Rm / cpm, cpm / (cpm - Rm)
        L = t.Ll * ql / cpm
        assert T == pytest.approx((r * Rm * T / 1e5) ** kap * th + L, rel=1e-9)
        T_noniter = th ** gam * (r * Rm / 1e5) ** (gam - 1) + L
        if ql == 0.0:
            assert T == pytest.approx(T_noniter, rel=1e-9)
        else:
            assert T > T_noniter
            assert T - T_noniter == pytest.approx(1.39 * kap * L, rel=0.15)


def test_direct_divergence_damping_properties(oracle, oc):
    """DirectDivergenceDamping (acoustic_substepping.jl:1146-1188) has no known-answer test in the reference; its defining properties on
    the oracle kernel: (1) a horizontally non-divergent theta-flux field is left untouched; (2) for uniform theta_L the correction is
    alpha dx^2 grad(div (rho u)') — a plane compression wave (rho u)' = sin(k x) has its amplitude reduced by the exact discrete factor
    1 - alpha (2 sin(k dx / 2))^2, the Laplacian-diffusion stability bound alpha <~ 0.25 quoted in time_discretizations.jl:262-267."""
    import ctypes as C
    g = oracle.Grid((16, 12, 6), x=(0.0, 1600.0), y=(0.0, 1200.0), z=(0.0, 600.0))
    m = oc.CompressibleOracleModel(g, time_discretization=oc.SplitExplicit(substeps=2, direct_damping=True), reference_potential_temperature=300.0)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    L, cg = m.lib, C.byref(m.cg)
    thL = np.full_like(m.rup, 300.0)
    x = g.xf[None, None, :] + 0 * g.yc[None, :, None]
    kx = 2 * np.pi * 3 / 1600.0
    up, vp, delta = np.zeros_like(m.rup), np.zeros_like(m.rup), np.zeros_like(m.rup)
    g.interior(up)[...] = np.sin(kx * x)
    m._halo_center(up)
    before = g.interior(up).copy()
    L.og_direct_divergence_damping(cg, p(up), p(vp), p(delta), p(thL), C.c_double(0.1))
    factor = 1.0 - 0.1 * (2 * np.sin(kx * g.dx / 2)) ** 2
    assert np.allclose(g.interior(up), factor * before, rtol=0, atol=1e-13)
    assert np.max(np.abs(g.interior(vp))) < 1e-13
    # non-divergent: (rho u)' = f(y) only
    up[...] = 0.0
    g.interior(up)[...] = np.cos(2 * np.pi * g.yc / 1200.0)[None, :, None]
    m._halo_center(up)
    before = g.interior(up).copy()
    L.og_direct_divergence_damping(cg, p(up), p(vp), p(delta), p(thL), C.c_double(0.1))
    assert np.max(np.abs(g.interior(up) - before)) < 1e-14


def test_thermal_damping_length_scale_is_a_fixed_diffusivity(oracle, oc):
    """ThermalDivergenceDamping(length_scale = l) (time_discretizations.jl:215-218, acoustic_substepping.jl:1085-1092): the correction is
    (alpha l^2 / dtau) d[(rho theta)' - (rho theta)'_old] / theta_L in both directions; l = min(dx, dy) reproduces the local default
    (:1100-1110) bit for bit, and the correction scales with l^2."""
    import ctypes as C
    g = oracle.Grid((16, 12, 6), x=(0.0, 1600.0), y=(0.0, 2400.0), z=(0.0, 600.0))      # dx = 100, dy = 200
    m = oc.CompressibleOracleModel(g, time_discretization=oc.SplitExplicit(substeps=2), reference_potential_temperature=300.0)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    L, cg = m.lib, C.byref(m.cg)
    rng = np.random.default_rng(3)
    thL = 300.0 + rng.random(m.rup.shape)
    rthp, rth_old = rng.standard_normal(m.rup.shape), rng.standard_normal(m.rup.shape)

    def run(length_scale):
        up, vp = np.zeros_like(m.rup), np.zeros_like(m.rup)
        L.og_thermal_divergence_damping(cg, p(up), p(vp), p(rthp), p(rth_old), p(thL), C.c_double(0.1), C.c_double(0.5), C.c_double(length_scale))
        return g.interior(up).copy(), g.interior(vp).copy()

    u0, v0 = run(0.0)
    u1, v1 = run(100.0)                       # = min(dx, dy)
    assert np.array_equal(u0, u1) and np.array_equal(v0, v1)
    u2, v2 = run(300.0)
    assert np.allclose(u2, 9.0 * u0, rtol=1e-14, atol=0) and np.allclose(v2, 9.0 * v0, rtol=1e-14, atol=0)
    i, j, k = 5, 4, 2
    n = lambda a, di=0, dj=0: g.interior(a)[k, j + dj, i + di]
    want = -(0.1 * 300.0 ** 2 / 0.5) * (((n(rthp) - n(rth_old)) - (n(rthp, -1) - n(rth_old, -1))) / 100.0) / ((n(thL) + n(thL, -1)) / 2)
    assert abs(u2[k, j, i] - want) < 1e-12 * abs(want)


def test_upper_sponge_coefficients_known_answers(oracle, oc):
    """test/acoustic_substepping_components.jl:476-499 restated on the oracle's profile: LinearRamp, depth 2000 on z = (0, 8000):
    zero at the bottom face, damping_rate at the lid, so the diagonal term is |dtau_new| rate and the rhs term |dtau_old| rate rho_w;
    plus the ramp shapes' end values and monotonicity (time_discretizations.jl:398-433) and the diagonal entering the column
    coefficient exactly as sponge_term_diag (acoustic_substepping.jl:639)."""
    zf = np.linspace(0.0, 8000.0, 9)
    rate, depth, dtn, dto = 0.2, 2000.0, 3.0, 2.0
    prof = oc.upper_sponge_profile(zf, 8000.0, rate, depth, "linear")
    assert prof[0] == 0.0
    assert np.isclose(abs(dtn) * prof[-1], dtn * rate) and np.isclose(abs(dto) * prof[-1] * 4.0, dto * rate * 4.0)
    assert np.all(prof[:6] == 0.0) and np.isclose(prof[7], 0.5 * rate)          # z = 7000: half way up the layer
    for ramp in ("linear", "cubic", "sin2"):
        q = oc.upper_sponge_profile(np.linspace(5000.0, 8000.0, 31), 8000.0, 1.0, 2000.0, ramp)
        assert q[0] == 0.0 and np.isclose(q[-1], 1.0) and np.all(np.diff(q) >= -1e-15)
    assert np.isclose(oc.upper_sponge_profile([7000.0], 8000.0, 1.0, 2000.0, "cubic")[0], 0.5)
    assert np.isclose(oc.upper_sponge_profile([7000.0], 8000.0, 1.0, 2000.0, "sin2")[0], 0.5)
    # one substep loop with and without the sponge on a rest-free state: identical below the layer's influence on the diagonal
    # is not expected (the column system couples), but the sponge must remove vertical-momentum perturbation energy near the lid
    og = oracle.Grid((8, 8, 16), x=(0, 8e3), y=(0, 8e3), z=(0.0, 8e3))
    out = {}
    for key, sp in (("off", None), ("on", (0.5, 3000.0, "cubic"))):
        om = oc.CompressibleOracleModel(og, time_discretization=oc.SplitExplicit(substeps=6, sponge=sp),
                                        reference_potential_temperature=300.0, reference_state=True)
        rho = om.ref.density[og.Hz:og.Hz + og.Nz][:, None, None]
        om.set(rho=rho, theta=lambda x, y, z: 300.0 + 0.5 * np.sin(2 * np.pi * x / 8e3) * np.sin(np.pi * z / 8e3) + 0 * y,
               u=0.0, v=0.0, w=lambda x, y, z: 0.5 * np.sin(np.pi * z / 8e3) + 0 * x + 0 * y)
        rw0 = og.interior(om.rw, True).copy()
        for _ in range(3):
            om.time_step(2.0)
        out[key] = og.interior(om.rw, True) - rw0          # the sponge acts on the perturbation (rho w)' the substeps build up
    top = slice(-5, -1)
    # (every stage restarts the perturbation from the rewind U0 - U^L, so only the change built up inside a stage is damped: a few per cent here)
    assert np.sum(out["on"][top] ** 2) < 0.98 * np.sum(out["off"][top] ** 2)
    assert np.allclose(out["on"][:6], out["off"][:6], rtol=0.2, atol=1e-3 * np.abs(out["off"]).max())
    assert np.all(np.isfinite(out["on"]))


def test_substep_distributions(oracle, oc):
    """stage_substep_count_and_size for the three distributions (acoustic_substepping.jl:468-508): ProportionalSubsteps tiles beta dt
    with ceil(beta N) substeps; ConstantSubstepSize rounds N up to a multiple of 6 and uses dt / N in every stage (beta N integral, so
    every stage is covered exactly); MonolithicFirstStage takes stage 1 in one substep of dt / 3"""
    g = oracle.Grid((16, 16, 8), x=(0, 16e3), y=(0, 16e3), z=(0, 8e3))
    c = oracle.Constants()
    dt = 12.0
    for N in (None, 4, 6, 8, 13):
        for beta in (1 / 3, 1 / 2, 1.0):
            n, dtau = oc.stage_substep_count_and_size(N, beta, dt, g, c, 0.5, "proportional")
            assert abs(n * dtau - beta * dt) < 1e-12
            n, dtau = oc.stage_substep_count_and_size(N, beta, dt, g, c, 0.5, "constant")
            n_raw = N if N is not None else oc.compute_acoustic_substeps(g, dt, c, 0.5)
            Nu = max(6, 6 * -(-n_raw // 6))
            assert Nu % 6 == 0 and Nu >= n_raw and dtau == dt / Nu and n == round(beta * Nu) and abs(n * dtau - beta * dt) < 1e-12
            n, dtau = oc.stage_substep_count_and_size(N, beta, dt, g, c, 0.5, "monolithic_first_stage")
            if beta < 0.4:
                assert (n, dtau) == (1, dt / 3)
            else:
                assert dtau == dt / Nu and n == round(beta * Nu)
    assert oc.stage_substep_count_and_size(8, 1.0, dt, g, c, 0.5, "constant")[0] == 12


def test_coriolis_and_sponges_are_slow_terms_with_closed_forms(oracle, oc):
    """FPlane + density-keyed Relaxation sponges in the compressible slow tendencies (dynamics_kernel_functions.jl:79,99;
    examples/tropical_cyclone_with_rainband.jl:434-514): on a horizontally uniform wind the advective parts vanish, so
    G_rho_u = + f rho v - r(z) rho u,  G_rho_v = - f rho u - r(z) rho v,  G_rho_theta = r(z) (target - rho theta) exactly."""
    f, rate = 5e-4, 0.01
    m = rest_model(oracle, oc, size=(8, 8, 16))
    set_rest(m)
    g = m.grid
    Hz, Nz = g.Hz, g.Nz
    rho = m.ref.density[Hz:Hz + Nz][:, None, None]
    m.set(rho=rho, theta=300.0, u=3.0, v=-2.0, w=0.0)
    r = rate * np.exp(-(g.zc - g.zf[-1]) ** 2 / (2 * 2000.0 ** 2))
    m.coriolis_f = f
    m.relaxation = {"ru": (r, np.zeros(Nz)), "rv": (r, np.zeros(Nz)), "rtheta": (r, (rho * 301.0).ravel())}
    m.compute_slow_tendencies()
    ru, rv, rth = g.interior(m.ru), g.interior(m.rv), g.interior(m.rtheta)
    rc = r[:, None, None]
    scale = np.abs(f * rv).max()
    assert np.abs(g.interior(m.G["ru"]) - (f * rv - rc * ru)).max() < 1e-12 * scale
    assert np.abs(g.interior(m.G["rv"]) - (-f * ru - rc * rv)).max() < 1e-12 * scale
    want = rc * (rho * 301.0 - rth)
    assert np.abs(g.interior(m.G["rtheta"]) - want).max() < 1e-9 * np.abs(want).max()
