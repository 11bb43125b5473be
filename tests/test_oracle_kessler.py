"""Pins for the Kessler column microphysics oracle: the reference's own "Julia vs Fortran" fidelity test
(test/dcmip2016_kessler.jl:262-401) and helper checks (:212-258), restated.  No GPU needed."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def ks(oracle):
    from oracle import kessler
    return kessler


def reference_test_profile(ks):
    """Inputs of test/dcmip2016_kessler.jl:262-330."""
    Nz = 40
    zc = (np.arange(Nz) + 0.5) * (4000.0 / Nz)
    T_surface, p_surface, g, Rd, cpd, lapse = 288.0, 101325.0, 9.81, 287.0, 1003.0, 0.0065
    T = T_surface - lapse * zc
    p = p_surface * (T / T_surface) ** (g / (Rd * lapse))
    rho = p / (Rd * T)
    rv = 0.015 * np.exp(-((zc - 1000) / 1000) ** 2)
    rcl = np.where((zc > 1500) & (zc < 2500), 0.002, 0.0)
    rr = np.where((zc > 1000) & (zc < 2000), 0.0005, 0.0)
    R = 8.314462618
    Md = R / 287.0
    c = ks.TetensConstants(molar_gas_constant=R, dry_air_molar_mass=Md, vapor_molar_mass=Md, dry_air_heat_capacity=cpd,
                           vapor_heat_capacity=cpd, liquid_latent_heat=2500000.0, liquid_heat_capacity=cpd,
                           liquid_temperature_offset=36.0)
    rt = rv + rcl + rr
    return zc, T, p, rho, rv / (1 + rt), rcl / (1 + rt), rr / (1 + rt), c


def test_kernel_restatement_matches_fortran_translation(ks):
    """Physical fidelity: the kernel restatement and the independent DCMIP2016 translation agree to rtol 1e-12."""
    zc, T, p, rho, qv, qcl, qr, c = reference_test_profile(ks)
    mp = ks.KesslerParameters()
    dt, p0 = 10.0, 100000.0
    T_ref, qv_ref, qcl_ref, qr_ref = T.copy(), qv.copy(), qcl.copy(), qr.copy()
    ks.dcmip2016_fortran_reference(T_ref, qv_ref, qcl_ref, qr_ref, rho, p, dt, zc, c, mp, p0)
    theta = np.zeros_like(T)
    for k in range(len(T)):
        ql = qcl[k] + qr[k]
        cpm, Rm = ks.mixture_heat_capacity(qv[k], ql, c), ks.mixture_gas_constant(qv[k], ql, c)
        theta[k] = (T[k] - c.Ll * ql / cpm) / (p[k] / p0) ** (Rm / cpm)
    rtheta, rqv, rqcl, rqr = rho * theta, rho * qv, rho * qcl, rho * qr
    qv_k, qcl_k, qr_k, W, precip, Ns = ks.kessler_column_update(dt, rho, p, p0, zc, theta, rtheta, rqv, rqcl, rqr, mp, c)
    T_k = np.zeros_like(T)
    for k in range(len(T)):
        qv_b, qcl_b, qr_b = rqv[k] / rho[k], rqcl[k] / rho[k], rqr[k] / rho[k]
        th = rtheta[k] / rho[k]
        cpm, Rm = ks.mixture_heat_capacity(qv_b, qcl_b + qr_b, c), ks.mixture_gas_constant(qv_b, qcl_b + qr_b, c)
        T_k[k] = (p[k] / p0) ** (Rm / cpm) * th + c.Ll * (qcl_b + qr_b) / cpm
    np.testing.assert_allclose(T_k, T_ref, rtol=1e-12)
    np.testing.assert_allclose(rqv / rho, qv_ref, rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(rqcl / rho, qcl_ref, rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(rqr / rho, qr_ref, rtol=1e-12, atol=1e-300)
    assert precip >= 0 and Ns >= 1
    assert np.abs(T_ref - T).max() > 1e-3            # the step did something


def test_kessler_helpers(ks):
    """test/dcmip2016_kessler.jl:212-258: terminal velocity range / monotonicity, conversion round trip."""
    mp = ks.KesslerParameters()
    W = ks.terminal_velocity(0.001, 1.0, 1.2, mp)
    assert 0 < W < 20
    assert ks.terminal_velocity(0.0, 1.0, 1.2, mp) == 0.0
    assert ks.terminal_velocity(0.005, 1.0, 1.2, mp) > W
    qv, ql = 0.01, 0.002
    qt = qv + ql
    rv, rl = qv / (1 - qt), ql / (1 - qt)
    qv_b, ql_b = ks._fractions_of_ratios(rv, rl)
    assert qv_b == pytest.approx(qv, rel=1e-10) and ql_b == pytest.approx(ql, rel=1e-10)
    # water is conserved by one step of the column physics apart from surface precipitation
    c = ks.TetensConstants()
    assert ks.saturation_vapor_pressure_tetens(273.15, c) == 610.0


def test_oracle_model_with_kessler_runs_and_makes_rain(oracle):
    """AtmosphereModel(...; microphysics = DCMIP2016KesslerMicrophysics()) on the oracle: supersaturated moist bubble ->
    cloud -> rain; the species stay non-negative and the temperature diagnosis carries the liquid water."""
    g = oracle.Grid((8, 8, 20), x=(0, 4e3), y=(0, 4e3), z=(0, 5e3))
    m = oracle.OracleModel(g, surface_pressure=1e5, potential_temperature=300.0, microphysics="Kessler")
    bubble = lambda x, y, z: np.maximum(0.0, 1.0 - np.sqrt((x - 2e3) ** 2 + (y - 2e3) ** 2 + (z - 1500.0) ** 2) / 1200.0)
    m.set(qt=lambda x, y, z: 0.016 * np.exp(-z / 3000.0) + 0.004 * bubble(x, y, z),
          theta=lambda x, y, z: 300.0 + 0.004 * z + 1.0 * bubble(x, y, z),
          qcl=lambda x, y, z: 0.003 * bubble(x, y, z), qr=lambda x, y, z: 0.001 * bubble(x, y, z))
    I = g.interior
    assert I(m.qcl).max() > 2e-3 and np.allclose(I(m.ql), I(m.qcl) + I(m.qr))
    T0 = I(m.T).copy()
    for _ in range(3):
        m.time_step(5.0)
    for f in (m.rq, m.rqcl, m.rqr):
        assert np.isfinite(I(f)).all() and I(f).min() >= 0.0
    assert I(m.W).max() > 0.5 and np.abs(I(m.T) - T0).max() > 1e-2
    assert set(m.PROGNOSTIC) == {"ru", "rv", "rw", "rtheta", "rq", "rqcl", "rqr"}


def test_compressible_oracle_model_with_kessler(oracle):
    """CompressibleDynamics + DCMIP2016KesslerMicrophysics on the oracle: total density includes the condensates, the EOS
    temperature carries the latent term, a few steps run and total water changes only through surface precipitation."""
    from oracle import oracle_compressible as oc
    g = oracle.Grid((8, 8, 16), x=(0, 4e3), y=(0, 4e3), z=(0, 4e3))
    m = oc.CompressibleOracleModel(g, time_discretization=oc.SplitExplicit(substeps=6), surface_pressure=1e5,
                                   reference_potential_temperature=300.0, microphysics="Kessler")
    bub = lambda x, y, z: np.maximum(0.0, 1.0 - np.sqrt((x - 2e3) ** 2 + (y - 2e3) ** 2 + (z - 1500.0) ** 2) / 1200.0)
    rho = m.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
    m.set(rho=rho, theta=lambda x, y, z: 300.0 + 0.004 * z + 1.0 * bub(x, y, z), u=0.0, v=0.0, w=0.0,
          qv=lambda x, y, z: 0.014 * np.exp(-z / 3000.0) + 0.004 * bub(x, y, z),
          qcl=lambda x, y, z: 0.003 * bub(x, y, z), qr=lambda x, y, z: 0.001 * bub(x, y, z))
    I = g.interior
    np.testing.assert_allclose(I(m.rho), I(m.rho_d) + I(m.rq) + I(m.rqcl) + I(m.rqr), rtol=1e-15)
    # EOS consistency: theta = T (pst/p)^kappa - latent shift
    c, t = m.constants, m.tetens
    qv, ql = I(m.q), I(m.qcl) + I(m.qr)
    cpm = (1 - qv - ql) * c.cpd + qv * c.cpv + ql * t.cl
    Rm = (1 - qv - ql) * c.Rd + qv * c.Rv
    th_back = (I(m.T) - t.Ll * ql / cpm) * (m.pst / I(m.p)) ** (Rm / cpm)
    np.testing.assert_allclose(th_back, I(m.theta), rtol=1e-6)      # Newton abstol 1e-4 K
    for _ in range(2):
        m.time_step(2.0)
    for f in (m.rq, m.rqcl, m.rqr, m.rho_d):
        assert np.isfinite(I(f)).all()
    assert I(m.rqcl).min() >= 0 and I(m.rqr).min() >= 0 and I(m.W).max() > 0.5
