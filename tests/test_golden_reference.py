"""Reference-generated goldens (tests/golden/reference_doctests.json: outputs of Breeze.jl's own jldoctests, which its
test/doctests.jl verifies against the real code) against the CPU oracle's thermodynamics.  Bit-exact where the doctest
prints a full Float64, 10 digits where it prints a rounded value.  No GPU needed."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "reference_doctests.json"), encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="module")
def thermo():
    from oracle import thermo
    return thermo


def test_saturation_specific_humidity_doctests_bit_exact(golden, thermo):
    g = golden["saturation_specific_humidity"]
    c = thermo.ThermoConstants()
    T, p = g["inputs"]["T"], g["inputs"]["p"]
    rho = thermo.density(T, p, 0.0, 0.0, 0.0, c)
    assert thermo.saturation_specific_humidity(T, rho, c, "liquid") == g["PlanarLiquidSurface"]
    assert thermo.saturation_specific_humidity(T, rho, c, "ice") == g["PlanarIceSurface"]
    assert thermo.saturation_specific_humidity(T, rho, c, ("mixed", 0.4)) == g["PlanarMixedPhaseSurface(0.4)"]


def test_pressure_balanced_density_doctest(golden, thermo):
    g = golden["pressure_balanced_density"]
    i = g["inputs"]
    assert thermo.pressure_balanced_density(i["rho_background"], i["theta_background"], i["theta_initial"]) == g["output"]


def test_solver_doctests(golden, thermo):
    n, s = golden["newton_solve"], golden["secant_solve"]
    x = thermo.newton_solve(lambda x: (x * x - 2, 2 * x), n["inputs"]["x0"], reltol=n["inputs"]["reltol"],
                            abstol=n["inputs"]["abstol"], maxiter=n["inputs"]["maxiter"])
    assert round(x, n["inputs"]["round_digits"]) == n["output"]
    i = s["inputs"]
    x = thermo.secant_solve(lambda x: x * x - 2, i["x1"], i["x2"], i["scale"], reltol=i["reltol"], abstol=i["abstol"],
                            maxiter=i["maxiter"])
    assert round(x, i["round_digits"]) == s["output"]


def test_warm_phase_adjustment_recovers_constructed_saturated_states(thermo):
    """test/saturation_adjustment.jl:43-97 restated for the potential-temperature state: build a saturated parcel with known
    (T2, qv+, ql), compute its theta, and require the adjustment to return it (atol = 10 sqrt(1e-6), solver abstol 1e-6)."""
    c = thermo.ThermoConstants()
    pst, pr = 1e5, 101325.0 * (1 - 9.81 * 0.5 / (1005.0 * 288.0 * (101325.0 / 1e5) ** (c.Rd / c.cpd))) ** (c.cpd / c.Rd)
    atol = 10 * (1e-6) ** 0.5
    checked = 0
    for T2 in (280.0, 300.0, 320.0):
        for qt in (1e-2, 3e-2, 5e-2):
            qs = thermo.adjustment_saturation_specific_humidity(T2, pr, qt, c, "liquid")
            if qt > qs:
                ql = qt - qs
                cpm, Rm = thermo.mixture_heat_capacity(qs, ql, 0.0, c), thermo.mixture_gas_constant(qs, ql, 0.0, c)
                theta = (T2 - c.Ll * ql / cpm) / (pr / pst) ** (Rm / cpm)
                T, qv, qliq = thermo.adjust_warm_phase(theta, qt, pr, pst, c, abstol=1e-6)
                assert abs(T - T2) <= atol and abs(qv - qs) <= atol and abs(qliq - ql) <= atol
                checked += 1
    assert checked >= 4
    # unsaturated parcel: temperature is the dry-exner one, no liquid
    T, qv, ql = thermo.adjust_warm_phase(300.0, 1e-3, pr, pst, c)
    assert ql == 0.0 and qv == 1e-3 and T == thermo.theta_state_temperature(300.0, 1e-3, 0.0, pr, pst, c)


def test_field_level_adjustment_matches_scalar_restatement(oracle, thermo):
    """The C field kernel of the oracle (og_compute_thermo_sa) against oracle/thermo.py cell by cell, on a column that is
    unsaturated aloft and cloudy below; warm bubble with SaturationAdjustment runs and keeps q^e = qv + ql."""
    import numpy as np
    g = oracle.Grid((8, 8, 16), x=(0, 8e3), y=(0, 8e3), z=(0, 4e3))
    m = oracle.OracleModel(g, potential_temperature=295.0, microphysics="SaturationAdjustment")
    qt = lambda x, y, z: 0.018 * np.exp(-z / 2500.0) * (1 + 0.1 * np.sin(2 * np.pi * x / 8e3)) + 0 * y
    m.set(qt=qt, theta=lambda x, y, z: 295.0 + 0.003 * z + 0 * x + 0 * y)
    I = g.interior
    c = thermo.ThermoConstants()
    pr = m.ref.pressure[g.Hz:g.Hz + g.Nz]
    n_cloudy = 0
    for k in range(g.Nz):
        for i in (0, 3):
            T, qv, ql = thermo.adjust_warm_phase(I(m.theta)[k, 2, i], I(m.q)[k, 2, i], pr[k], m.ref.pst, c)
            assert abs(T - I(m.T)[k, 2, i]) <= 1e-12 * T
            assert abs(qv - I(m.qv)[k, 2, i]) <= 1e-15 and abs(ql - I(m.ql)[k, 2, i]) <= 1e-15
            n_cloudy += ql > 0
    assert 0 < n_cloudy < 2 * g.Nz
    np.testing.assert_allclose(I(m.qv) + I(m.ql), I(m.q), rtol=1e-15)
    for _ in range(3):
        m.time_step(2.0)
    assert np.isfinite(I(m.T)).all() and (I(m.ql) >= 0).all()
    np.testing.assert_allclose(I(m.qv) + I(m.ql), I(m.q), rtol=1e-14)


def test_density_based_saturation_adjustment_reference_known_answers(thermo):
    """test/compressible_saturation_adjustment.jl:54-74: a supersaturated all-vapour parcel (theta = 300 K, rho = 1 kg/m3,
    q^t = 0.020) condenses onto the saturation curve evaluated at its OWN density (atol 1e-5), theta^li is conserved by the
    self-consistent inversion, and a subsaturated parcel (q^t = 0.002) is left alone."""
    c = thermo.ThermoConstants()
    T, qv, ql = thermo.adjust_warm_phase_density(300.0, 0.020, 1.0, 1e5, c)
    assert ql > 0
    assert qv == pytest.approx(thermo.saturation_specific_humidity(T, 1.0, c, "liquid"), abs=1e-5)
    Rm, cpm = thermo.mixture_gas_constant(qv, ql, 0.0, c), thermo.mixture_heat_capacity(qv, ql, 0.0, c)
    theta_back = (T - c.Ll * ql / cpm) * (1e5 / (1.0 * Rm * T)) ** (Rm / cpm)
    assert theta_back == pytest.approx(300.0, rel=1e-9)
    Td, qvd, qld = thermo.adjust_warm_phase_density(300.0, 0.002, 1.0, 1e5, c)
    assert qld == 0.0 and qvd == 0.002


def test_moist_compressible_update_state_is_density_consistent(oracle):
    """test/compressible_saturation_adjustment.jl:114-148: 8^3 cells over 2 km, theta_ref(z) = 300 exp(g z / (c_p 300)),
    rho = reference density, theta = 300, q^t = 0.030: condensation occurs and in every saturated cell q^v sits on the
    saturation curve at the cell's own total density (atol 1e-4); one tiny step stays finite."""
    from oracle import oracle_compressible as oc
    from oracle import thermo as th
    g = oracle.Grid((8, 8, 8), x=(0, 2e3), y=(0, 2e3), z=(0, 2e3))
    thref = lambda z: 300.0 * np.exp(9.80616 * z / (1005 * 300.0))
    m = oc.CompressibleOracleModel(g, time_discretization=oc.SplitExplicit(), surface_pressure=1e5,
                                   reference_potential_temperature=thref, microphysics="SaturationAdjustment")
    rho = m.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
    m.set(rho=rho, theta=300.0, qv=0.030, u=0.0, v=0.0, w=0.0)
    I = g.interior
    T, p, r, ql, qv = I(m.T), I(m.p), I(m.rho), I(m.ql), I(m.qv)
    assert np.isfinite(T).all() and (T > 0).all() and np.isfinite(p).all() and (p > 0).all()
    assert ql.max() > 0
    tc = th.ThermoConstants()
    for idx in np.ndindex(T.shape):
        if ql[idx] > 1e-6:
            assert qv[idx] == pytest.approx(th.saturation_specific_humidity(T[idx], r[idx], tc, "liquid"), abs=1e-4)
    m.time_step(1e-3)
    assert np.isfinite(I(m.T)).all()


# ---- model-level doctest outputs: reference-generated numbers for the reference column + temperature diagnosis -----------------

def _sig(x, digits=6):
    """Tolerance of a value printed with `digits` significant digits (Julia's Field summary prints 6)."""
    return 0.6 * 10.0 ** (np.floor(np.log10(abs(x))) - (digits - 1))


def _column_model(oracle, Nz, microphysics=None):
    g = oracle.Grid((4, 4, Nz), x=(0, 1.0), y=(0, 1.0), z=(-1000.0, 0.0))     # extent = (., ., 1e3): z in (-1000, 0)
    return oracle.OracleModel(g, microphysics=microphysics)                         # every default: p0 = 101325, theta0 = 288, p_st = 1e5


def test_static_energy_doctest_pins_reference_column_and_temperature(oracle, golden):
    """StaticEnergy(model) after set!(model, θ = 300) on the default model: e = c_pm T + g z with T = Pi(p_r(z)) theta — the printed
    max / min / mean (6 significant digits) pin the reference pressure column (ReferenceState defaults, adiabatic hydrostatic
    closed form) and the theta -> T diagnosis of the oracle to reference-generated numbers."""
    gd = golden["model_diagnostics"]["static_energy"]
    m = _column_model(oracle, 8)
    m.set(theta=300.0)
    g, c = m.grid, m.constants
    T = g.interior(m.T)[:, 0, 0]
    e = c.cpd * T + c.g * g.zc
    for key, val in (("max", e.max()), ("min", e.min()), ("mean", e.mean())):
        assert abs(val - gd[key]) <= _sig(gd[key]), (key, val, gd[key])


def test_virtual_potential_temperature_doctest(oracle, golden):
    """VirtualPotentialTemperature after set!(model, θ = 300, qᵗ = 0.01): (T / Pi_d)(1 + (R_v/R_d - 1) q^v)
    (potential_temperatures.jl:574-579); printed with 5 significant digits."""
    gd = golden["model_diagnostics"]["virtual_potential_temperature"]
    m = _column_model(oracle, 8)
    m.set(qt=0.01, theta=300.0)
    g, c, r = m.grid, m.constants, m.ref
    T, q = g.interior(m.T)[:, 0, 0], g.interior(m.q)[:, 0, 0]
    p = r.pressure[g.Hz:g.Hz + g.Nz]
    thv = T / (p / r.pst) ** (c.Rd / c.cpd) * (1 + (c.Rv / c.Rd - 1) * q)
    for key, val in (("max", thv.max()), ("min", thv.min()), ("mean", thv.mean())):
        assert abs(val - gd[key]) <= 0.006, (key, val, gd[key])


def test_relative_humidity_doctest_pins_saturation_adjustment_state(oracle, golden, thermo):
    """RelativeHumidity on the default SaturationAdjustment model, θ = 300, qᵗ = 0.005 (subsaturated), 128 levels: p^v / p^v+ with
    p^v = rho q^v R_v T, rho = p_r / (R_m T) (microphysics_diagnostics.jl:139-170).  Six printed digits pin the adjusted
    temperature, the Clausius-Clapeyron saturation pressure and the reference column together."""
    gd = golden["model_diagnostics"]["relative_humidity"]
    m = _column_model(oracle, 128, microphysics="SaturationAdjustment")
    m.set(qt=0.005, theta=300.0)
    g, c, r = m.grid, m.constants, m.ref
    T, qv, ql = (g.interior(f)[:, 0, 0] for f in (m.T, m.qv, m.ql))
    assert np.all(ql == 0)
    p = r.pressure[g.Hz:g.Hz + g.Nz]
    rho = p / (((1 - qv - ql) * c.Rd + qv * c.Rv) * T)
    tc = thermo.ThermoConstants()
    ps = np.array([thermo.saturation_vapor_pressure(t, tc, "liquid") for t in T])
    rh = rho * qv * c.Rv * T / ps
    for key, val in (("max", rh.max()), ("min", rh.min()), ("mean", rh.mean())):
        assert abs(val - gd[key]) <= _sig(gd[key]), (key, val, gd[key])


def test_dewpoint_temperature_doctest(oracle, golden, thermo):
    """DewpointTemperature on the default SaturationAdjustment model, θ = 300, qᵗ = 0.01, 8 levels: secant inversion of the
    saturation vapour pressure (vapor_saturation.jl:313-331: guesses T and T - 20 (1 - ℋ), SecantSolver(reltol = 1e-4, abstol = 0,
    maxiter = 10) scaled by p^v) of p^v = rho q^v R_v T.  Six printed digits."""
    gd = golden["model_diagnostics"]["dewpoint_temperature"]
    m = _column_model(oracle, 8, microphysics="SaturationAdjustment")
    m.set(qt=0.01, theta=300.0)
    g, c, r = m.grid, m.constants, m.ref
    T, qv, ql = (g.interior(f)[:, 0, 0] for f in (m.T, m.qv, m.ql))
    p = r.pressure[g.Hz:g.Hz + g.Nz]
    tc = thermo.ThermoConstants()
    out = []
    for Tk, qvk, qlk, pk in zip(T, qv, ql, p):
        rho = pk / (((1 - qvk - qlk) * c.Rd + qvk * c.Rv) * Tk)
        pv = rho * qvk * c.Rv * Tk
        ps1 = thermo.saturation_vapor_pressure(Tk, tc, "liquid")
        if ps1 - pv <= 0:
            out.append(Tk)
            continue
        T2 = Tk - (1 - pv / ps1) * 20
        out.append(thermo.secant_solve(lambda x: thermo.saturation_vapor_pressure(x, tc, "liquid") - pv, Tk, T2, pv,
                                       reltol=1e-4, abstol=0.0, maxiter=10))
    Td = np.array(out)
    for key, val in (("max", Td.max()), ("min", Td.min()), ("mean", Td.mean())):
        assert abs(val - gd[key]) <= _sig(gd[key]), (key, val, gd[key])


def _theta_column(oracle):
    """The default model of the potential-temperature doctests after set!(model, θ = 300, qᵗ = 0.01): T, q, p_r on 8 levels."""
    m = _column_model(oracle, 8)
    m.set(qt=0.01, theta=300.0)
    g, c, r = m.grid, m.constants, m.ref
    return g.interior(m.T)[:, 0, 0], g.interior(m.q)[:, 0, 0], r.pressure[g.Hz:g.Hz + g.Nz], r.pst, c


def _check(val, gd, digits_tol):
    for key, x in (("max", val.max()), ("min", val.min()), ("mean", val.mean())):
        assert abs(x - gd[key]) <= digits_tol, (key, x, gd[key])


def test_plain_and_liquid_ice_potential_temperature_doctests(oracle, golden):
    """PotentialTemperature = T / Pi_m and LiquidIcePotentialTemperature = theta (1 - L q^l / (c_pm T)) of a vapour-only state
    (potential_temperatures.jl:556-573): both must return the 300 K that was set — the theta -> T -> theta round trip through the
    oracle's moist Exner function."""
    T, q, p, pst, c = _theta_column(oracle)
    qd = 1.0 - q
    Rm, cpm = qd * c.Rd + q * c.Rv, qd * c.cpd + q * c.cpv
    theta = T / (p / pst) ** (Rm / cpm)
    _check(theta, golden["model_diagnostics"]["potential_temperature"], 1e-9)
    _check(theta * (1.0 - 0.0 / (cpm * T)), golden["model_diagnostics"]["liquid_ice_potential_temperature"], 1e-9)


def test_equivalent_potential_temperature_doctests(oracle, golden, thermo):
    """EquivalentPotentialTemperature (Emanuel 1994 eq. 4.5.11 as written in potential_temperatures.jl:580-605):
    theta_e = T (p_st/p)^(R_d/c_pm) exp(L_l(T) q^v / (c_pm T)) H^(-R_v q^v / c_pm), H = p^v / p^v+ over liquid with
    rho = p / (R_m T); and the stability-equivalent flavour, identical for q^l = 0.  Six printed digits (326.162 / 325.851 / 326.006)
    pin the diagnosed temperature, the reference pressure column, L_l(T) and the Clausius-Clapeyron pressure together."""
    T, q, p, pst, c = _theta_column(oracle)
    tc = thermo.ThermoConstants()
    qd = 1.0 - q
    Rm, cpm = qd * c.Rd + q * c.Rv, qd * c.cpd + q * c.cpv
    rho = p / (Rm * T)
    ps = np.array([thermo.saturation_vapor_pressure(t, tc, "liquid") for t in T])
    H = rho * q * c.Rv * T / ps
    Ll = tc.Ll + (tc.cpv - tc.cl) * (T - tc.T_energy)
    theta_e = T * (pst / p) ** (c.Rd / cpm) * np.exp(Ll * q / (cpm * T)) * H ** (-c.Rv * q / cpm)
    _check(theta_e, golden["model_diagnostics"]["equivalent_potential_temperature"], 6e-4)
    theta_b = theta_e * (T / tc.T_energy) ** (tc.cl * 0.0 / cpm)
    _check(theta_b, golden["model_diagnostics"]["stability_equivalent_potential_temperature"], 6e-4)
