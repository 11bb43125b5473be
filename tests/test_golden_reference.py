"""Reference-generated goldens (tests/golden/reference_doctests.json: outputs of Breeze.jl's own jldoctests, which its
test/doctests.jl verifies against the real code) against the CPU oracle's thermodynamics.  Bit-exact where the doctest
prints a full Float64, 10 digits where it prints a rounded value.  No GPU needed."""
import json
import os

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
