"""
oracle/thermo.py — scalar Float64 restatement of the Breeze.Thermodynamics / Breeze.Solvers / Breeze.Microphysics pieces
behind warm-phase saturation adjustment.  TEST INFRASTRUCTURE ONLY.

PARITY STATUS: **pinned by reference-generated values** — the jldoctest outputs of the reference's own docstrings
(executed by its test/doctests.jl), extracted by tools/make_golden_doctests.py into tests/golden/reference_doctests.json:
saturation_specific_humidity over liquid / ice / mixed surfaces, pressure_balanced_density, newton_solve and secant_solve.

Reference lines (relative to /root/reference):
  ThermodynamicConstants defaults, CondensedPhase      src/Thermodynamics/thermodynamics_constants.jl:87-93,182-217,263-274
  mixture_gas_constant / mixture_heat_capacity        src/Thermodynamics/thermodynamics_constants.jl:315-316,341-378
  saturation_vapor_pressure (Clausius-Clapeyron)      src/Thermodynamics/clausius_clapeyron.jl:59-68
  saturation_specific_humidity, adjustment_*          src/Thermodynamics/vapor_saturation.jl:23-36,93-97,250-256
  LiquidIcePotentialTemperatureState temperature      src/Thermodynamics/dynamic_states.jl:31-58,125-128,143-148
  newton_solve / secant_solve                         src/Solvers.jl:190-212,243-262
  adjust_thermodynamic_state (warm phase)             src/Microphysics/saturation_adjustment.jl:82-86,168-235
  pressure_balanced_density                           src/Thermodynamics/reference_states.jl:125-160
"""
import math


class ThermoConstants:
    def __init__(self, molar_gas_constant=8.314462618, gravitational_acceleration=9.81,
                 energy_reference_temperature=273.15, triple_point_temperature=273.16, triple_point_pressure=611.657,
                 dry_air_molar_mass=0.02897, dry_air_heat_capacity=1005.0, vapor_molar_mass=0.018015,
                 vapor_heat_capacity=1850.0, liquid_latent_heat=2500800.0, liquid_heat_capacity=4181.0,
                 ice_latent_heat=2834000.0, ice_heat_capacity=2108.0):
        self.R, self.g = molar_gas_constant, gravitational_acceleration
        self.T_energy, self.Ttr, self.ptr = energy_reference_temperature, triple_point_temperature, triple_point_pressure
        self.Md, self.cpd, self.Mv, self.cpv = dry_air_molar_mass, float(dry_air_heat_capacity), vapor_molar_mass, float(vapor_heat_capacity)
        self.Ll, self.cl, self.Li, self.ci = float(liquid_latent_heat), float(liquid_heat_capacity), float(ice_latent_heat), float(ice_heat_capacity)
        self.Rd, self.Rv = self.R / self.Md, self.R / self.Mv


def mixture_gas_constant(qv, ql, qi, c):
    qd = 1 - (qv + ql + qi)
    return qd * c.Rd + qv * c.Rv


def mixture_heat_capacity(qv, ql, qi, c):
    qd = 1 - (qv + ql + qi)
    return qd * c.cpd + qv * c.cpv + ql * c.cl + qi * c.ci


def _surface(c, surface):
    """(specific_heat_difference, absolute_zero_latent_heat) of 'liquid' | 'ice' | ('mixed', liquid_fraction)."""
    dcl, dci = c.cpv - c.cl, c.cpv - c.ci
    L0l, L0i = c.Ll - dcl * c.T_energy, c.Li - dci * c.T_energy
    if surface == "liquid":
        return dcl, L0l
    if surface == "ice":
        return dci, L0i
    lam = surface[1]
    return lam * dcl + (1 - lam) * dci, lam * L0l + (1 - lam) * L0i


def saturation_vapor_pressure(T, c, surface="liquid"):
    dc, L0 = _surface(c, surface)
    return c.ptr * (T / c.Ttr) ** (dc / c.Rv) * math.exp((1 / c.Ttr - 1 / T) * L0 / c.Rv)


def saturation_specific_humidity(T, rho, c, surface="liquid"):
    return saturation_vapor_pressure(T, c, surface) / (rho * c.Rv * T)


def density(T, p, qv, ql, qi, c):
    return p / (mixture_gas_constant(qv, ql, qi, c) * T)


def adjustment_saturation_specific_humidity(T, pr, qt, c, surface="liquid"):
    ps = saturation_vapor_pressure(T, c, surface)
    eps = c.Rd / c.Rv
    return eps * (1 - qt) * ps / (pr - ps)


def pressure_balanced_density(rho_background, theta_background, theta_initial):
    return rho_background * theta_background / theta_initial


def newton_solve(residual_and_derivative, x, reltol=0.0, abstol=1e-4, maxiter=8):
    dx, it = x, 0
    while abs(dx) > max(abstol, reltol * abs(x)) and it < maxiter:
        r, dr = residual_and_derivative(x)
        dx = -r / dr
        x += dx
        it += 1
    return x


def secant_solve(residual, x1, x2, scale, reltol=0.0, abstol=1e-4, maxiter=20):
    r1, r2, it = residual(x1), residual(x2), 0
    while abs(r2) > max(abstol, reltol * abs(scale)) and it < maxiter:
        denom = r2 - r1
        s = (x2 - x1) / denom if denom != 0 else math.inf
        valid = math.isfinite(s)
        s = s if valid else 0.0
        x1, r1 = x2, r2
        x2 -= r2 * s
        r2 = residual(x2)
        r2 = r2 if valid else 0.0
        it += 1
    return x2


def theta_state_temperature(theta, qv, ql, pr, pst, c):
    Rm, cpm = mixture_gas_constant(qv, ql, 0.0, c), mixture_heat_capacity(qv, ql, 0.0, c)
    return (pr / pst) ** (Rm / cpm) * theta + (c.Ll * ql + c.Li * 0.0) / cpm


def adjust_warm_phase(theta, qt, pr, pst, c, abstol=1e-4, maxiter=20):
    """adjust_thermodynamic_state for LiquidIcePotentialTemperatureState + WarmPhaseEquilibrium -> (T, qv, ql)."""
    if theta == 0:
        return 0.0, qt, 0.0
    T1 = theta_state_temperature(theta, qt, 0.0, pr, pst, c)
    rho1 = density(T1, pr, qt, 0.0, 0.0, c)
    if qt <= saturation_specific_humidity(T1, rho1, c, "liquid"):
        return T1, qt, 0.0

    def adjust(T):
        qs = adjustment_saturation_specific_humidity(T, pr, qt, c, "liquid")
        ql = max(0, qt - qs)
        return qt - ql, ql

    qv1, ql1 = adjust(T1)
    dT = (c.Ll * ql1 + c.Li * 0.0) / mixture_heat_capacity(qv1, ql1, 0.0, c)
    T2 = T1 + max(0.01, dT / 2)

    def residual(T):
        qv, ql = adjust(T)
        return T - theta_state_temperature(theta, qv, ql, pr, pst, c)

    Ts = secant_solve(residual, T1, T2, T2, abstol=abstol, maxiter=maxiter)
    qv, ql = adjust(Ts)
    return theta_state_temperature(theta, qv, ql, pr, pst, c), qv, ql


# ---- density-based liquid-ice potential temperature state (CompressibleDynamics) -----------------------------------------------
#   LiquidIceDensityState temperature (Newton on T = (rho R_m T / p_st)^kappa theta + L)   src/Thermodynamics/dynamic_states.jl:161-232
#   saturated_density_residual, adjust_thermodynamic_state(::LiquidIceDensityState, ::SA)    src/Microphysics/saturation_adjustment.jl:236-301
# Pinned by the reference's known-answer tests test/compressible_saturation_adjustment.jl:31-66 (tests/test_golden_reference.py).

def density_state_temperature(theta, qv, ql, rho, pst, c, abstol=1e-4, maxiter=8):
    Rm, cpm = mixture_gas_constant(qv, ql, 0.0, c), mixture_heat_capacity(qv, ql, 0.0, c)
    kap, gam = Rm / cpm, cpm / (cpm - Rm)
    L = (c.Ll * ql + c.Li * 0.0) / cpm
    T = theta ** gam * (rho * Rm / pst) ** (gam - 1.0) + L

    def rd(T):
        Phi = (rho * Rm * T / pst) ** kap * theta
        return T - Phi - L, 1.0 - kap * Phi / T

    return newton_solve(rd, T, abstol=abstol, maxiter=maxiter)


def saturated_density_residual(T, theta0, rho, qt, pst, c):
    qs = saturation_specific_humidity(T, rho, c, "liquid")
    ql = max(0, qt - qs)
    qv = qt - ql
    Rm, cpm = mixture_gas_constant(qv, ql, 0.0, c), mixture_heat_capacity(qv, ql, 0.0, c)
    kap = Rm / cpm
    L = (c.Ll * ql + c.Li * 0.0) / cpm
    p = rho * Rm * T
    theta = (T - L) * (pst / p) ** kap
    return theta - theta0, (qv, ql)


def adjust_warm_phase_density(theta, qt, rho, pst, c, abstol=1e-4, maxiter=20, newton_abstol=1e-4, newton_maxiter=8):
    """adjust_thermodynamic_state(LiquidIceDensityState, SaturationAdjustment(WarmPhaseEquilibrium)) -> (T, qv, ql)."""
    if theta == 0:
        return 0.0, qt, 0.0
    T1 = density_state_temperature(theta, qt, 0.0, rho, pst, c, newton_abstol, newton_maxiter)
    if qt <= saturation_specific_humidity(T1, rho, c, "liquid"):
        return T1, qt, 0.0
    _, (qv1, ql1) = saturated_density_residual(T1, theta, rho, qt, pst, c)
    dT = (c.Ll * ql1 + c.Li * 0.0) / mixture_heat_capacity(qv1, ql1, 0.0, c)
    T2 = T1 + max(0.01, dT / 2)
    residual = lambda T: saturated_density_residual(T, theta, rho, qt, pst, c)[0]
    Ts = secant_solve(residual, T1, T2, T2, abstol=abstol, maxiter=maxiter)
    _, (qv, ql) = saturated_density_residual(Ts, theta, rho, qt, pst, c)
    return density_state_temperature(theta, qv, ql, rho, pst, c, newton_abstol, newton_maxiter), qv, ql
