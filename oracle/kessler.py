"""
oracle/kessler.py — CPU restatement of the DCMIP2016 Kessler column microphysics.
TEST INFRASTRUCTURE ONLY.

  kessler_column_update        restates the reference kernel `_microphysical_update!`
                               (/root/reference/src/Microphysics/dcmip2016_kessler.jl:618-858) with its helpers
                               kessler_terminal_velocity (:397-402), cloud_to_rain_production (:420-430),
                               step_kessler_microphysics (:519-562), mass-fraction / mixing-ratio conversions (:570-612),
                               Tetens saturation vapour pressure (src/Thermodynamics/tetens_formula.jl:112-119).
  dcmip2016_fortran_reference  restates the *independent* translation of the DCMIP2016 Fortran routine that the
                               reference's own test carries (test/dcmip2016_kessler.jl:34-209) and compares with the
                               kernel at rtol 1e-12 (:262-401).

PARITY STATUS: pinned the way the reference pins it — the two independently written implementations must agree to 1e-12
on the reference test's profile (tests/test_oracle_kessler.py).  All arithmetic is Float64 in the written operation order.
"""
import math

import numpy as np


class TetensConstants:
    """ThermodynamicConstants with saturation_vapor_pressure = TetensFormula(...) (tetens_formula.jl:72-85)."""

    def __init__(self, molar_gas_constant=8.314462618, dry_air_molar_mass=0.02897, vapor_molar_mass=0.018015,
                 dry_air_heat_capacity=1005.0, vapor_heat_capacity=1850.0, liquid_latent_heat=2500800.0,
                 liquid_heat_capacity=4181.0, reference_saturation_vapor_pressure=610.0, reference_temperature=273.15,
                 liquid_coefficient=17.27, liquid_temperature_offset=35.85):
        self.Rd, self.Rv = molar_gas_constant / dry_air_molar_mass, molar_gas_constant / vapor_molar_mass
        self.cpd, self.cpv = float(dry_air_heat_capacity), float(vapor_heat_capacity)
        self.Ll, self.cl = float(liquid_latent_heat), float(liquid_heat_capacity)
        self.psat_ref, self.T_ref = float(reference_saturation_vapor_pressure), float(reference_temperature)
        self.a_liquid, self.dT_liquid = float(liquid_coefficient), float(liquid_temperature_offset)


class KesslerParameters:
    """DCMIP2016KesslerMicrophysics defaults (dcmip2016_kessler.jl:154-169)."""

    def __init__(self, **kw):
        self.dcmip_temperature_scale = 237.3
        self.terminal_velocity_coefficient = 36.34
        self.density_scale = 0.001
        self.terminal_velocity_exponent = 0.1364
        self.autoconversion_rate = 0.001
        self.autoconversion_threshold = 0.001
        self.accretion_rate = 2.2
        self.accretion_exponent = 0.875
        self.evaporation_ventilation_coefficient_1 = 1.6
        self.evaporation_ventilation_coefficient_2 = 124.9
        self.evaporation_ventilation_exponent_1 = 0.2046
        self.evaporation_ventilation_exponent_2 = 0.525
        self.diffusivity_coefficient = 2.55e8
        self.thermal_conductivity_coefficient = 5.4e5
        self.substep_cfl = 0.8
        for k, v in kw.items():
            assert hasattr(self, k), k
            setattr(self, k, float(v))


def saturation_vapor_pressure_tetens(T, c):
    return c.psat_ref * math.exp(c.a_liquid * (T - c.T_ref) / (T - c.dT_liquid))


def saturation_specific_humidity(T, rho, c):
    return saturation_vapor_pressure_tetens(T, c) / (rho * c.Rv * T)


def mixture_gas_constant(qv, ql, c):
    return (1 - (qv + ql + 0.0)) * c.Rd + qv * c.Rv


def mixture_heat_capacity(qv, ql, c):
    return (1 - (qv + ql + 0.0)) * c.cpd + qv * c.cpv + ql * c.cl + 0.0


def terminal_velocity(rr, rho, rho1, mp):
    return mp.terminal_velocity_coefficient * (rr * mp.density_scale * rho) ** mp.terminal_velocity_exponent * math.sqrt(rho1 / rho)


def cloud_to_rain_production(rcl, rr, dt, mp):
    A = max(0, mp.autoconversion_rate * (rcl - mp.autoconversion_threshold))
    denom = 1 + dt * mp.accretion_rate * rr ** mp.accretion_exponent
    return rcl - (rcl - dt * A) / denom


def step_kessler(rv, rcl, rr, drW, T, rho, p, dt, mp, c, f5, dT):
    dP = cloud_to_rain_production(rcl, rr, dt, mp)
    rcl = max(0, rcl - dP)
    rr = max(0, rr + dP + drW)
    qs = saturation_specific_humidity(T, rho, c)
    rs = qs / (1 - qs)
    dsat = (rv - rs) / (1 + rs * f5 / (T - dT) ** 2)
    rhok = mp.density_scale * rho
    rhorr = rhok * rr
    Vev = (mp.evaporation_ventilation_coefficient_1 + mp.evaporation_ventilation_coefficient_2 * rhorr ** mp.evaporation_ventilation_exponent_1) * rhorr ** mp.evaporation_ventilation_exponent_2
    Dth = mp.diffusivity_coefficient / (p * rs) + mp.thermal_conductivity_coefficient
    drs = max(0, rs - rv)
    E = Vev / Dth * drs / (rhok * rs + 1e-20)
    dEmax = max(0, -dsat - rcl)
    dE = min(min(dt * E, dEmax), rr)
    dC = max(dsat, -rcl)
    rv = max(0, rv - dC + dE)
    rcl = rcl + dC
    rr = rr - dE
    return rv, rcl, rr, dC - dE


def _fractions_of_ratios(rv, rl):
    """MoistureMassFractions(MoistureMixingRatio(rv, rl)): q = r * inv(1 + r_t)."""
    inv = 1.0 / (1 + (rv + rl + 0.0))
    return rv * inv, rl * inv


def _temperature(theta, rv, rcl, rr, p, pst, c):
    qv, ql = _fractions_of_ratios(rv, rcl + rr)
    cpm, Rm = mixture_heat_capacity(qv, ql, c), mixture_gas_constant(qv, ql, c)
    Pi = (p / pst) ** (Rm / cpm)
    return Pi * theta + c.Ll * ql / cpm, Pi, ql, cpm


def kessler_column_update(dt, rho, p, pst, zc, theta, rtheta, rqv, rqcl, rqr, mp, c):
    """One column of `_microphysical_update!`.  rho, p, zc, theta, rtheta, rqv, rqcl, rqr: arrays over k (modified in place:
    theta, rtheta, rqv, rqcl, rqr).  Returns (qv, qcl, qr, W, precipitation_rate, n_substeps)."""
    Nz = len(rho)
    f5 = c.a_liquid * mp.dcmip_temperature_scale * c.Ll / c.cpd
    dT = c.dT_liquid
    rho1 = rho[0]
    rv, rcl, rr, W = np.zeros(Nz), np.zeros(Nz), np.zeros(Nz), np.zeros(Nz)
    max_dt = dt
    for k in range(Nz):
        r = rho[k]
        qv = rqv[k] / r
        qcl, qr = max(0, rqcl[k] / r), max(0, rqr[k] / r)
        qv = max(0, qv)
        inv_qd = 1.0 / (1 - (qv + (qcl + qr) + 0.0))
        rvk = qv * inv_qd
        rt = rvk + (qcl + qr) * inv_qd + 0.0 * inv_qd
        rv[k], rcl[k], rr[k] = rvk, qcl * (1 + rt), qr * (1 + rt)
        W[k] = terminal_velocity(rr[k], r, rho1, mp)
        if k < Nz - 1:
            with np.errstate(divide="ignore"):
                max_dt = min(max_dt, np.float64(mp.substep_cfl * (zc[k + 1] - zc[k])) / np.float64(W[k]))
    Ns = max(1, math.ceil(dt / max_dt))
    inv_Ns = 1.0 / float(Ns)
    dts = dt * inv_Ns
    Psurf = 0.0
    for m in range(1, Ns + 1):
        rt1 = rv[0] + rcl[0] + rr[0]
        Psurf += rr[0] / (1 + rt1) * W[0]
        for k in range(Nz):
            r, pk = rho[k], p[k]
            Tk, _, _, _ = _temperature(theta[k], rv[k], rcl[k], rr[k], pk, pst, c)
            rhok = mp.density_scale * r
            if k < Nz - 1:
                dz = zc[k + 1] - zc[k]
                rhok1 = mp.density_scale * rho[k + 1]
                drW = dts * (rhok1 * rr[k + 1] * W[k + 1] - rhok * rr[k] * W[k]) / (rhok * dz)
            else:
                dz_half = (zc[k] - zc[k - 1]) / 2
                drW = -dts * rr[k] * W[k] / dz_half
            rv[k], rcl[k], rr[k], drl = step_kessler(rv[k], rcl[k], rr[k], drW, Tk, r, pk, dts, mp, c, f5, dT)
            T = Tk + c.Ll / c.cpd * drl
            qv, ql = _fractions_of_ratios(rv[k], rcl[k] + rr[k])
            cpm, Rm = mixture_heat_capacity(qv, ql, c), mixture_gas_constant(qv, ql, c)
            Pi = (pk / pst) ** (Rm / cpm)
            theta[k] = (T - c.Ll * ql / cpm) / Pi
            rtheta[k] = r * theta[k]
        if m < Ns:
            for k in range(Nz):
                W[k] = terminal_velocity(rr[k], rho[k], rho1, mp)
    qv_out, qcl_out, qr_out = np.zeros(Nz), np.zeros(Nz), np.zeros(Nz)
    for k in range(Nz):
        rl = rcl[k] + rr[k]
        qv, _ = _fractions_of_ratios(rv[k], rl)
        rt = rv[k] + rl + 0.0
        qcl, qr = rcl[k] / (1 + rt), rr[k] / (1 + rt)
        rqv[k], rqcl[k], rqr[k] = rho[k] * qv, rho[k] * qcl, rho[k] * qr
        qv_out[k], qcl_out[k], qr_out[k] = qv, qcl, qr
    return qv_out, qcl_out, qr_out, W, Psurf * inv_Ns, Ns


def dcmip2016_fortran_reference(T, qv, qcl, qr, rho, p, dt, z, c, mp, p0=100000.0):
    """The independent translation held by the reference's test (test/dcmip2016_kessler.jl:34-209); arrays modified in place."""
    Nz = len(T)
    f5 = c.a_liquid * mp.dcmip_temperature_scale * c.Ll / c.cpd
    T_offset = c.dT_liquid
    theta = np.zeros(Nz)
    for k in range(Nz):
        ql = qcl[k] + qr[k]
        cpm, Rm = mixture_heat_capacity(qv[k], ql, c), mixture_gas_constant(qv[k], ql, c)
        theta[k] = (T[k] - c.Ll * ql / cpm) / (p[k] / p0) ** (Rm / cpm)
    rv, rcl, rr, W = np.zeros(Nz), np.zeros(Nz), np.zeros(Nz), np.zeros(Nz)
    rho1, max_dt = rho[0], dt
    for k in range(Nz):
        qt = qv[k] + qcl[k] + qr[k]
        rv[k], rcl[k], rr[k] = qv[k] / (1 - qt), qcl[k] / (1 - qt), qr[k] / (1 - qt)
        W[k] = terminal_velocity(rr[k], rho[k], rho1, mp)
        if k < Nz - 1 and W[k] > 0:
            max_dt = min(max_dt, mp.substep_cfl * (z[k + 1] - z[k]) / W[k])
    Ns = max(1, math.ceil(dt / max_dt))
    dts = dt / Ns
    for s in range(1, Ns + 1):
        zk = z[0]
        for k in range(Nz):
            rt = rv[k] + rcl[k] + rr[k]
            qv_l, ql_l = rv[k] / (1 + rt), (rcl[k] + rr[k]) / (1 + rt)
            cpm, Rm = mixture_heat_capacity(qv_l, ql_l, c), mixture_gas_constant(qv_l, ql_l, c)
            T[k] = (p[k] / p0) ** (Rm / cpm) * theta[k] + c.Ll * ql_l / cpm
            if k < Nz - 1:
                dz = z[k + 1] - zk
                drW = dts * (rho[k + 1] * rr[k + 1] * W[k + 1] - rho[k] * rr[k] * W[k]) / (rho[k] * dz)
                zk = z[k + 1]
            else:
                drW = -dts * rr[k] * W[k] / (0.5 * (z[k] - z[k - 1]))
            A = max(0.0, mp.autoconversion_rate * (rcl[k] - mp.autoconversion_threshold))
            denom = 1.0 + dts * mp.accretion_rate * rr[k] ** mp.accretion_exponent
            dP = rcl[k] - (rcl[k] - dts * A) / denom
            rcl_new = max(0.0, rcl[k] - dP)
            rr_new = max(0.0, rr[k] + dP + drW)
            qs = saturation_specific_humidity(T[k], rho[k], c)
            rs = qs / (1 - qs)
            dsat = (rv[k] - rs) / (1 + rs * f5 / (T[k] - T_offset) ** 2)
            rhok = rho[k] * mp.density_scale
            rhorr = rhok * rr_new
            Vev = (mp.evaporation_ventilation_coefficient_1 + mp.evaporation_ventilation_coefficient_2 * rhorr ** mp.evaporation_ventilation_exponent_1) * rhorr ** mp.evaporation_ventilation_exponent_2
            Dth = mp.diffusivity_coefficient / (p[k] * rs) + mp.thermal_conductivity_coefficient
            drs = max(0.0, rs - rv[k])
            E = Vev / Dth * drs / (rhok * rs + 1e-20)
            dEmax = max(0.0, -dsat - rcl_new)
            dE = min(min(dts * E, dEmax), rr_new)
            dC = max(dsat, -rcl_new)
            rv_new = max(0.0, rv[k] - dC + dE)
            rcl_f, rr_f = rcl_new + dC, rr_new - dE
            T_new = T[k] + (c.Ll / c.cpd) * (dC - dE)
            rt_new = rv_new + rcl_f + rr_f
            qv_n, ql_n = rv_new / (1 + rt_new), (rcl_f + rr_f) / (1 + rt_new)
            cpm_n, Rm_n = mixture_heat_capacity(qv_n, ql_n, c), mixture_gas_constant(qv_n, ql_n, c)
            theta[k] = (T_new - c.Ll * ql_n / cpm_n) / (p[k] / p0) ** (Rm_n / cpm_n)
            rv[k], rcl[k], rr[k] = rv_new, rcl_f, rr_f
        if s < Ns:
            for k in range(Nz):
                W[k] = terminal_velocity(rr[k], rho[k], rho1, mp)
    for k in range(Nz):
        rt = rv[k] + rcl[k] + rr[k]
        qv[k], qcl[k], qr[k] = rv[k] / (1 + rt), rcl[k] / (1 + rt), rr[k] / (1 + rt)
        ql = qcl[k] + qr[k]
        cpm, Rm = mixture_heat_capacity(qv[k], ql, c), mixture_gas_constant(qv[k], ql, c)
        T[k] = (p[k] / p0) ** (Rm / cpm) * theta[k] + c.Ll * ql / cpm
