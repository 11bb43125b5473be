"""
oracle/forcings.py — CPU restatement of the forcing / Coriolis / bottom-flux terms of the BOMEX configuration
(BASELINE configs[2], examples/bomex.jl:80-207) for the anelastic OracleModel.  TEST INFRASTRUCTURE ONLY.

PARITY STATUS: **pinned by the reference's own known-answer tests, restated in tests/test_forcings.py** — the reference holds
no golden arrays for these terms, its tests check closed-form expectations, and the oracle reproduces each of them:
  test/geostrophic_subsidence_forcings.jl:296-339  one step of w_s = 1 on phi = Gamma z changes rho phi by rho_r (-dt w_s Gamma)
                                                   in the bottom and top cells (theta, q^v, u; rtol 1e-3)
  test/geostrophic_subsidence_forcings.jl:233-294  five steps accumulate linearly (1e-3)
  test/geostrophic_subsidence_forcings.jl:12-38,163-187  u_g = -10: rho v < 0 after one step, alone and with subsidence
  test/forcing_and_boundary_conditions.jl:89-128   a bottom FluxBoundaryCondition J lands as +J/dz in G[.,.,1] with the field
                                                   dependencies taken at the boundary location
The Coriolis and boundary-flux arithmetic itself lives in Oceananigans (not vendored) and is restated from its published
operators; beyond those anchors the digits are unpinned.  Breeze-side formulas are followed line by line:

  SubsidenceForcing  F = -zb-average(w_s dz(avg phi))     src/Forcings/subsidence_forcing.jl:75-91,104-126
  geostrophic_forcings  F_u = -f v_g, F_v = +f u_g         src/Forcings/geostrophic_forcings.jl (GeostrophicForcing call)
  SpecificForcing  rho at the target location x F          src/Forcings/specific_forcing.jl:61-74
  energy forcing in the theta tendency  F_rho_e/(c_pm Pi)  src/PotentialTemperatureFormulations/potential_temperature_tendency.jl:86-104
  Coriolis  -x_f_cross_U / -y_f_cross_U                    src/AtmosphereModels/dynamics_kernel_functions.jl:79,99
                                                           (Oceananigans FPlane: x = -f xy-average(rho v), y = +f xy-average(rho u))
  compute_forcings! before the tendencies                  src/AtmosphereModels/update_atmosphere_model_state.jl:52,81-86
  compute_flux_bc_tendencies! before every RK substep      src/AtmosphereModels/update_atmosphere_model_state.jl:418-434,
                                                           src/TimeSteppers/ssp_runge_kutta_3.jl (time_step!)
                                                           (Oceananigans: G[i,j,1] += J Az / V for a bottom flux J)
  bulk drag of the example  J = -rho0 u*^2 rho_u/|rho U|   examples/bomex.jl:95-101 (rho v / rho u interpolated to the BC location)
"""
import numpy as np


class ColumnForcings:
    """Horizontally uniform forcing profiles (length Nz, cell centres; None = absent) + f-plane + bottom fluxes."""

    def __init__(self, Fu=None, Fv=None, Ftheta=None, Fq=None, Fe=None, w_subsidence=None,
                 subsidence_on=("u", "v", "theta", "q"), coriolis_f=0.0, flux_theta=0.0, flux_q=0.0,
                 drag_rho0_ustar2=0.0, bulk=None, drag_epsilon=0.0, flux_energy=0.0):
        self.Fu, self.Fv, self.Ftheta, self.Fq, self.Fe = Fu, Fv, Ftheta, Fq, Fe
        self.w_subsidence = w_subsidence              # Nz+1 faces
        self.subsidence_on = tuple(subsidence_on) if w_subsidence is not None else ()
        self.f = float(coriolis_f)
        self.flux_theta, self.flux_q, self.drag = float(flux_theta), float(flux_q), float(drag_rho0_ustar2)
        self.drag_eps = float(drag_epsilon)           # benchmarking/src/convective_boundary_layer.jl:142-146
        # a constant energy flux keyed rho e in a potential-temperature model: J_theta = Q / c_pm of the lowest cell
        # (EnergyFluxBoundaryCondition, src/BoundaryConditions/thermodynamic_variable_bcs.jl; BoundaryConditions.jl:218-227)
        self.flux_energy = float(flux_energy)
        self.bulk = bulk                              # BulkFluxes or None
        self.sub = {}


class BulkFluxes:
    """BulkDrag / BulkSensibleHeatFlux / BulkVaporFlux with constant coefficients and unfiltered fields
    (src/BoundaryConditions/bulk_drag.jl:114-135, bulk_scalar_fluxes.jl:82-90,123-137,206-232, BoundaryConditions.jl:64-85):
      J^u = -rho0 C^D U~ u,  J^theta = -rho0 C^T U~ (theta - theta0),  J^v = -rho0 C^v U~ (q^v - q^v+(T0, rho0)),
      U~ = sqrt(U^2 + gustiness^2) with U^2 interpolated to the flux location, rho0 = p0 / (R^d T0) (reference_states.jl:73-76),
      theta0 = T0 / (p0/p_st)^(R^d/c_pd) (dynamic_states.jl:111-124)."""

    def __init__(self, surface_pressure, standard_pressure=1e5, drag=None, heat=None, vapor=None):
        # each of drag / heat / vapor: None or (coefficient, gustiness, surface_temperature)
        self.p0, self.pst = float(surface_pressure), float(standard_pressure)
        self.drag_params, self.heat, self.vapor = drag, heat, vapor


def subsidence_profile(ws, avg, dzf):
    """-zb-average of w_s[k] * (avg[k]-avg[k-1])/dzf[k]: one-sided at the bottom and top cells."""
    Nz = avg.shape[0]
    wdz = np.zeros(Nz + 1)
    wdz[1:Nz] = ws[1:Nz] * ((avg[1:] - avg[:-1]) / dzf[1:Nz])
    out = np.empty(Nz)
    out[1:Nz - 1] = (wdz[2:Nz] + wdz[1:Nz - 1]) / 2
    out[0] = wdz[1]
    out[Nz - 1] = wdz[Nz - 1]
    return -out


def compute_forcings(m):
    """compute_forcing!(::SubsidenceForcing): Average(specific field, dims=(1,2))."""
    F, g = m.forcings, m.grid
    dzf = m.grid.dzf[g.Hz:g.Hz + g.Nz + 1]
    spec = {"u": m.u, "v": m.v, "theta": m.theta, "q": m.q}
    F.sub = {}
    for name in F.subsidence_on:
        avg = g.interior(spec[name]).mean(axis=(1, 2))
        F.sub[name] = subsidence_profile(np.asarray(F.w_subsidence, dtype=np.float64), avg, dzf)


def _xy_to_fc(m, f):
    """xy-average of a (c,f,c) field to (f,c,c) on the interior: (i-1,j),(i,j),(i-1,j+1),(i,j+1)."""
    g = m.grid
    z = slice(g.Hz, g.Hz + g.Nz)
    y0, y1, x0, x1 = g.Hy, g.Hy + g.Ny, g.Hx, g.Hx + g.Nx
    dy = 0 if g.Ny == 1 and g.Hy == 0 else 1
    dx = 0 if g.Nx == 1 and g.Hx == 0 else 1
    a = lambda jo, io: f[z, y0 + jo:y1 + jo, x0 + io:x1 + io]
    yc_i = (a(0, 0) + a(dy, 0)) / 2
    yc_im = (a(0, -dx) + a(dy, -dx)) / 2
    return (yc_im + yc_i) / 2


def _xy_to_cf(m, f):
    """xy-average of a (f,c,c) field to (c,f,c): (i,j-1),(i+1,j-1),(i,j),(i+1,j)."""
    g = m.grid
    z = slice(g.Hz, g.Hz + g.Nz)
    y0, y1, x0, x1 = g.Hy, g.Hy + g.Ny, g.Hx, g.Hx + g.Nx
    dy = 0 if g.Ny == 1 and g.Hy == 0 else 1
    dx = 0 if g.Nx == 1 and g.Hx == 0 else 1
    a = lambda jo, io: f[z, y0 + jo:y1 + jo, x0 + io:x1 + io]
    xc_j = (a(0, 0) + a(0, dx)) / 2
    xc_jm = (a(-dy, 0) + a(-dy, dx)) / 2
    return (xc_jm + xc_j) / 2


def add_forcing_tendencies(m):
    """Coriolis + forcing terms of x/y momentum, theta and moisture tendencies (after the advective part)."""
    F, g, c, r = m.forcings, m.grid, m.constants, m.ref
    I = g.interior
    rho = r.density[g.Hz:g.Hz + g.Nz][:, None, None]
    col = lambda p: np.asarray(p, dtype=np.float64)[:, None, None]
    G = m.G
    if F.f != 0.0:
        I(G["ru"])[...] -= -F.f * _xy_to_fc(m, m.rv)
        I(G["rv"])[...] -= F.f * _xy_to_cf(m, m.ru)
    for name, key, static in (("u", "ru", F.Fu), ("v", "rv", F.Fv), ("theta", "rtheta", F.Ftheta), ("q", "rq", F.Fq)):
        total = None
        if name in F.sub:
            total = rho * col(F.sub[name])
        if static is not None:
            total = rho * col(static) if total is None else total + rho * col(static)
        if total is not None:
            I(G[key])[...] += total
    if F.Fe is not None:
        if m.microphysics == "SaturationAdjustment":
            qv, ql = I(m.qv), I(m.ql)
            cl = m._sa.cl
        else:
            qv, ql, cl = I(m.q), 0.0, 0.0
        qd = 1.0 - (qv + ql + 0.0)
        Rm = qd * c.Rd + qv * c.Rv
        cpm = qd * c.cpd + qv * c.cpv + ql * cl + 0.0
        pr = r.pressure[g.Hz:g.Hz + g.Nz][:, None, None]
        Pi = (pr / r.pst) ** (Rm / cpm)
        I(G["rtheta"])[...] += (rho * col(F.Fe) + 0.0) / (cpm * Pi)


def add_relaxation_tendencies(m):
    """Sponge layers: Oceananigans' Relaxation, F = (rate mask(z)) (target(z) - field), as the reference's examples attach it
    (examples/rico.jl:103-105,164; neutral_atmospheric_boundary_layer.jl:103-136).  m.relaxation = {key: (rate_column, target_column)}:
    density keys "ru", "rv", "rw", "rtheta", "rq" relax the prognostic density (G += F); the specific keys "u", "v", "w" are specific
    forcings, G += rho_r F with the reference density at the field's location (src/Forcings/specific_forcing.jl:61-74).  Columns at the
    field's vertical location (rw / w: Nz + 1 faces; the wall faces carry no tendency)."""
    g = m.grid
    for key, (rate, target) in (getattr(m, "relaxation", None) or {}).items():
        specific = key in ("u", "v", "w")
        if specific:
            r = m.ref
            rho_c = r.density[g.Hz:g.Hz + g.Nz]
            rho_f = 0.5 * (r.density[g.Hz - 1:g.Hz + g.Nz] + r.density[g.Hz:g.Hz + g.Nz + 1])
        name = {"u": "ru", "v": "rv", "w": "rw"}.get(key, key)
        zface = name == "rw"
        field = g.interior(getattr(m, key), zface)
        G = g.interior(m.G[name], zface)
        rate, target = np.asarray(rate, dtype=np.float64)[:, None, None], np.asarray(target, dtype=np.float64)[:, None, None]
        F = rate * (target - field)
        if specific:
            F = (rho_f if zface else rho_c)[:, None, None] * F
        if zface:
            G[1:g.Nz] += F[1:g.Nz]
        elif name == "ru" and g.topo[0] == 1:
            G[:, :, 1:] += F[:, :, 1:]
        elif name == "rv" and g.topo[1] == 1:
            G[:, 1:, :] += F[:, 1:, :]
        else:
            G[...] += F


def add_field_forcing(m):
    """Forcing(f(x, y, z)) on the thermodynamic variable (examples/tropical_cyclone_with_rainband.jl:419-432,511-514): m.field_forcing =
    (F on the interior, specific).  Keyed theta it is a specific forcing, G_rho_theta += rho F with the coupling density — rho_r(z) of the
    anelastic model, the dry density of the compressible one (specific_forcing.jl:61-74, compressible_dynamics.jl:385); keyed rho theta, G += F."""
    g = m.grid
    F, specific = m.field_forcing
    if specific:
        rho = g.interior(m.rho_d) if hasattr(m, "rho_d") else m.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
        F = rho * F
    g.interior(m.G["rtheta"])[...] += F


def add_flux_bc_tendencies(m):
    """compute_flux_bc_tendencies!: bottom FluxBoundaryConditions, G[.,.,1] += J / dz_1."""
    F, g = m.forcings, m.grid
    dz = g.dzc[g.Hz]
    k = g.Hz
    I = g.interior
    if F.flux_theta != 0.0:
        I(m.G["rtheta"])[0] += F.flux_theta * 1.0 / dz
    if F.flux_q != 0.0:
        I(m.G["rq"])[0] += F.flux_q * 1.0 / dz
    if getattr(F, "flux_energy", 0.0) != 0.0:
        c = m.constants
        if m.microphysics == "SaturationAdjustment":
            qv, ql, cl = I(m.qv)[0], I(m.ql)[0], m._sa.cl
        else:
            qv, ql, cl = I(m.q)[0], 0.0, 0.0
        cpm = (1.0 - (qv + ql)) * c.cpd + qv * c.cpv + ql * cl
        I(m.G["rtheta"])[0] += (F.flux_energy / cpm) * 1.0 / dz
    if F.drag != 0.0:
        ru, rv = I(m.ru)[0], I(m.rv)[0]
        rv_fc = _xy_to_fc(m, m.rv)[0]
        ru_cf = _xy_to_cf(m, m.ru)[0]
        Ju = -F.drag * ru / np.sqrt(ru ** 2 + rv_fc ** 2 + F.drag_eps)
        Jv = -F.drag * rv / np.sqrt(ru_cf ** 2 + rv ** 2 + F.drag_eps)
        I(m.G["ru"])[0] += Ju * 1.0 / dz
        I(m.G["rv"])[0] += Jv * 1.0 / dz
    del k
    if F.bulk is not None:
        _add_bulk_fluxes(m, F.bulk, dz)


def _add_bulk_fluxes(m, B, dz):
    from .thermo import ThermoConstants, saturation_specific_humidity
    g, c = m.grid, m.constants
    I = g.interior
    z = g.Hz
    y0, y1, x0, x1 = g.Hy, g.Hy + g.Ny, g.Hx, g.Hx + g.Nx
    u, v = m.u[z], m.v[z]                       # first level with halos
    fy = 0 if (g.Ny == 1 and g.Hy == 0) else 1      # a Flat direction has no neighbours: its averages collapse onto the cell
    fx = 0 if (g.Nx == 1 and g.Hx == 0) else 1
    sq = lambda a, jo, io: a[y0 + jo * fy:y1 + jo * fy, x0 + io * fx:x1 + io * fx] ** 2
    if B.drag_params is not None:
        C, gust, T0 = B.drag_params
        rho0 = B.p0 / (c.Rd * T0)
        # wind_speed2 at (f,c): u^2 + xy-average of v^2; at (c,f): xy-average of u^2 + v^2
        v2_fc = ((sq(v, 0, -1) + sq(v, 1, -1)) / 2 + (sq(v, 0, 0) + sq(v, 1, 0)) / 2) / 2
        u2_cf = ((sq(u, -1, 0) + sq(u, -1, 1)) / 2 + (sq(u, 0, 0) + sq(u, 0, 1)) / 2) / 2
        ui, vi = u[y0:y1, x0:x1], v[y0:y1, x0:x1]
        Ju = -rho0 * C * np.sqrt(ui ** 2 + v2_fc + gust ** 2) * ui
        Jv = -rho0 * C * np.sqrt(u2_cf + vi ** 2 + gust ** 2) * vi
        I(m.G["ru"])[0] += Ju * 1.0 / dz
        I(m.G["rv"])[0] += Jv * 1.0 / dz
    U2c = (sq(u, 0, 0) + sq(u, 0, 1)) / 2 + (sq(v, 0, 0) + sq(v, 1, 0)) / 2
    if B.heat is not None:
        C, gust, T0 = B.heat
        rho0 = B.p0 / (c.Rd * T0)
        theta0 = T0 / (B.p0 / B.pst) ** (c.Rd / c.cpd)
        th = I(m.theta)[0]
        I(m.G["rtheta"])[0] += (-rho0 * C * np.sqrt(U2c + gust ** 2) * (th - theta0)) * 1.0 / dz
    if B.vapor is not None:
        C, gust, T0 = B.vapor
        rho0 = B.p0 / (c.Rd * T0)
        qv0 = saturation_specific_humidity(T0, rho0, ThermoConstants(), "liquid")
        qv = I(m.qv)[0] if m.microphysics == "SaturationAdjustment" else I(m.q)[0]
        I(m.G["rq"])[0] += (-rho0 * C * np.sqrt(U2c + gust ** 2) * (qv - qv0)) * 1.0 / dz
