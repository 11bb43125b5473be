"""
oracle/closure.py — CPU restatement of `closure = SmagorinskyLilly()` for the anelastic OracleModel (BASELINE configs[2];
SURVEY §8f rank 2).  TEST INFRASTRUCTURE ONLY.

PARITY STATUS: **parity unpinned**.  The Breeze side is followed line by line — the density-weighted stress / flux divergences
  dynamic stresses  T_ij = rho x (kinematic viscous flux)      src/TurbulenceClosures/TurbulenceClosures.jl:66-101
  scalar fluxes     J = rho x (kinematic diffusive flux)       src/TurbulenceClosures/TurbulenceClosures.jl:44-58
  tendency terms    - d_j T_1j, - d_j T_2j, - d_j T_3j, - div J  src/AtmosphereModels/dynamics_kernel_functions.jl:80,100,128,157;
                                                                 src/PotentialTemperatureFormulations/potential_temperature_tendency.jl:102
  buoyancy gradient N^2 = g dz(log theta_v)                    src/AtmosphereModels/atmosphere_model_buoyancy.jl:46-68
  compute_closure_fields! inside compute_auxiliary_variables!  src/AtmosphereModels/update_atmosphere_model_state.jl:218
— but the eddy viscosity and the kinematic fluxes are Oceananigans' (Project.toml:43 pins 0.110.14; not vendored).  They are
restated from its published algorithm (Smagorinsky 1963 / Lilly 1962 as documented by Oceananigans):
  nu_e = varsigma (C_s Delta)^2 sqrt(2 Sigma^2),  C_s = 0.16,  Delta = (dx dy dz)^(1/3),
  varsigma = sqrt(1 - min(1, C_b N^2+ / Sigma^2)) (0 where Sigma^2 = 0),  C_b = 1,  N^2 = centre average of the face values,
  Sigma^2 = Sigma_ij Sigma_ij at cell centres with the off-diagonal squares averaged from their edges,
  viscous flux_ij = -2 nu Sigma_ij with nu averaged to the flux location,  diffusive flux = -(nu / Pr) grad c,  Pr = 1,
  velocity halos: periodic in x, y; no-flux (zero gradient) for u, v across the bottom / top; w = 0 on both.
The reference's own closure tests (test/turbulence_closures.jl:61-67,98-139) assert only nu_e > 0 under shear and that scalars
change under diffusion; tests/test_closure.py repeats those and adds analytic checks (pure shear, Fickian decay, conservation).
"""
import numpy as np


class SmagorinskyLilly:
    def __init__(self, C=0.16, Cb=1.0, Pr=1.0):
        self.C, self.Cb, self.Pr = float(C), float(Cb), float(Pr)


def _pad_y(g, a, yface=False):
    """one row on each side in y: the periodic images, or — walls in y (topology (Periodic, Bounded, Bounded)) — the no-flux copy of the
    wall row for a field that is a centre in y, zeros for a y-face field (the row above is the wall face Ny; the row below is never read)"""
    if g.topo[1] == 1:      # oracle.BOUNDED
        if yface:
            z = np.zeros_like(a[:, :1, :])
            return np.concatenate([z, a, z], axis=1)
        return np.concatenate([a[:, :1, :], a, a[:, -1:, :]], axis=1)
    return np.concatenate([a[:, -1:, :], a, a[:, :1, :]], axis=1)


def _pad_x(g, a, xface=False):
    """the same in x (topology (Bounded, Flat, Bounded))"""
    if g.topo[0] == 1:
        if xface:
            z = np.zeros_like(a[:, :, :1])
            return np.concatenate([z, a, z], axis=2)
        return np.concatenate([a[:, :, :1], a, a[:, :, -1:]], axis=2)
    return np.concatenate([a[:, :, -1:], a, a[:, :, :1]], axis=2)


def _pad_center(g, f, yface=False, xface=False):
    """(Nz+2, Ny+2, Nx+2): periodic or walled in x and y; zero gradient in z."""
    a = f[g.Hz:g.Hz + g.Nz, g.Hy:g.Hy + g.Ny, g.Hx:g.Hx + g.Nx]
    a = _pad_x(g, _pad_y(g, a, yface), xface)
    return np.concatenate([a[:1], a, a[-1:]], axis=0)


def _pad_w(g, f):
    """(Nz+1, Ny+2, Nx+2) faces 0..Nz, periodic in x, periodic or walled in y; the wall faces hold 0."""
    a = f[g.Hz:g.Hz + g.Nz + 1, g.Hy:g.Hy + g.Ny, g.Hx:g.Hx + g.Nx].copy()
    a[0] = 0.0
    a[-1] = 0.0
    return _pad_x(g, _pad_y(g, a))


def _columns(m):
    g, r = m.grid, m.ref
    Hz, Nz = g.Hz, g.Nz
    dzc = g.dzc[Hz:Hz + Nz]                  # centre k
    dzf = g.dzf[Hz:Hz + Nz + 1]              # face k (spacing between centres k-1 and k)
    rho = r.density[Hz:Hz + Nz]
    rho_f = 0.5 * (r.density[Hz - 1:Hz + Nz] + r.density[Hz:Hz + Nz + 1])      # faces 0..Nz
    p_ext = r.pressure[Hz - 1:Hz + Nz + 1]   # centres -1..Nz (first halo cells of the reference column)
    return dzc, dzf, rho, rho_f, p_ext


def strain(m):
    """Sigma_11, Sigma_22, Sigma_33 at centres of the padded box; Sigma_12 (ffc), Sigma_13 (fcf), Sigma_23 (cff).
    Padded index convention: centre arrays [k+1, j+1, i+1]; x-face i and y-face j share the index of the cell to their right."""
    g = m.grid
    dzc, dzf, *_ = _columns(m)
    u, v, w = _pad_center(g, m.u, xface=True), _pad_center(g, m.v, yface=True), _pad_w(g, m.w)
    dx, dy = g.dx, g.dy
    Nz, Ny, Nx = g.Nz, g.Ny, g.Nx
    C = lambda a, dk=0, dj=0, di=0: a[1 + dk:1 + dk + Nz, 1 + dj:1 + dj + Ny, 1 + di:1 + di + Nx]
    S11 = (C(u, di=1) - C(u)) / dx
    S22 = (C(v, dj=1) - C(v)) / dy
    wi = w[:, 1:1 + Ny, 1:1 + Nx]
    S33 = (wi[1:] - wi[:-1]) / dzc[:, None, None]
    # Sigma_12 at corners (i, j), i in 0..Nx, j in 0..Ny (face indices) -> arrays (Nz, Ny+1, Nx+1)
    uy = (u[1:1 + Nz, 1:2 + Ny, 1:2 + Nx] - u[1:1 + Nz, 0:1 + Ny, 1:2 + Nx]) / dy
    vx = (v[1:1 + Nz, 1:2 + Ny, 1:2 + Nx] - v[1:1 + Nz, 1:2 + Ny, 0:1 + Nx]) / dx
    S12 = (uy + vx) / 2
    # Sigma_13 at (x-face i in 0..Nx, z-face k in 0..Nz) -> (Nz+1, Ny, Nx+1); zero on the walls
    uz = (u[1:2 + Nz, 1:1 + Ny, 1:2 + Nx] - u[0:1 + Nz, 1:1 + Ny, 1:2 + Nx]) / dzf[:, None, None]
    wx = (w[:, 1:1 + Ny, 1:2 + Nx] - w[:, 1:1 + Ny, 0:1 + Nx]) / dx
    S13 = (uz + wx) / 2
    vz = (v[1:2 + Nz, 1:2 + Ny, 1:1 + Nx] - v[0:1 + Nz, 1:2 + Ny, 1:1 + Nx]) / dzf[:, None, None]
    wy = (w[:, 1:2 + Ny, 1:1 + Nx] - w[:, 0:1 + Ny, 1:1 + Nx]) / dy
    S23 = (vz + wy) / 2
    return S11, S22, S33, S12, S13, S23


def buoyancy_frequency(m):
    """N^2 at centres: average of g dz(log theta_v) on the two faces (atmosphere_model_buoyancy.jl:46-68)."""
    g, c, r = m.grid, m.constants, m.ref
    dzc, dzf, rho, rho_f, p_ext = _columns(m)
    T = _pad_center(g, m.T)[:, 1:-1, 1:-1]
    qv_field = m.qv if m.microphysics == "SaturationAdjustment" else m.q
    qv = _pad_center(g, qv_field)[:, 1:-1, 1:-1]
    Rm = (1.0 - (qv + 0.0 + 0.0)) * c.Rd + qv * c.Rv
    thv = Rm / c.Rd * T * (r.pst / p_ext[:, None, None]) ** (c.Rd / c.cpd)
    lg = np.log(thv)
    dzb = c.g * ((lg[1:] - lg[:-1]) / dzf[:, None, None])       # faces 0..Nz
    return (dzb[:-1] + dzb[1:]) / 2


def eddy_viscosity(m):
    """nu_e on the interior (Nz, Ny, Nx)."""
    g, cl = m.grid, m.closure
    dzc, *_ = _columns(m)
    S11, S22, S33, S12, S13, S23 = strain(m)
    sq12, sq13, sq23 = S12 ** 2, S13 ** 2, S23 ** 2
    a12 = ((sq12[:, :-1, :-1] + sq12[:, :-1, 1:]) / 2 + (sq12[:, 1:, :-1] + sq12[:, 1:, 1:]) / 2) / 2
    a13 = ((sq13[:-1, :, :-1] + sq13[:-1, :, 1:]) / 2 + (sq13[1:, :, :-1] + sq13[1:, :, 1:]) / 2) / 2
    a23 = ((sq23[:-1, :-1, :] + sq23[:-1, 1:, :]) / 2 + (sq23[1:, :-1, :] + sq23[1:, 1:, :]) / 2) / 2
    Sig2 = (S11 ** 2 + S22 ** 2 + S33 ** 2) + 2 * a12 + 2 * a13 + 2 * a23
    N2 = buoyancy_frequency(m)
    N2p = np.maximum(0.0, N2)
    with np.errstate(divide="ignore", invalid="ignore"):
        sig2 = 1.0 - np.minimum(1.0, cl.Cb * N2p / Sig2)
        stab = np.where(Sig2 == 0, 0.0, np.sqrt(sig2))
    delta = np.cbrt(g.dx * g.dy * dzc)[:, None, None]
    return (stab * cl.C ** 2) * delta ** 2 * np.sqrt(2 * Sig2)


def compute_closure_fields(m):
    m.nu_e = eddy_viscosity(m)


def add_closure_tendencies(m):
    """G_rho_u -= d_j T_1j etc. and G_rho_c -= div J for theta and moisture."""
    g, cl = m.grid, m.closure
    dzc, dzf, rho, rho_f, _ = _columns(m)
    Nz, Ny, Nx = g.Nz, g.Ny, g.Nx
    dx, dy = g.dx, g.dy
    S11, S22, S33, S12, S13, S23 = strain(m)
    nu = m.nu_e
    # nu padded periodically in x, y and by zero gradient in z: np[k+1, j+1, i+1]
    nup = _pad_x(g, _pad_y(g, nu))
    nup = np.concatenate([nup[:1], nup, nup[-1:]], axis=0)
    r3 = rho[:, None, None]
    rf3 = rho_f[:, None, None]
    dz3 = dzc[:, None, None]
    # viscosity at the flux locations
    nu_ffc = ((nup[1:-1, 0:Ny + 1, 0:Nx + 1] + nup[1:-1, 0:Ny + 1, 1:Nx + 2]) / 2 +
              (nup[1:-1, 1:Ny + 2, 0:Nx + 1] + nup[1:-1, 1:Ny + 2, 1:Nx + 2]) / 2) / 2          # (Nz, Ny+1, Nx+1)
    nu_fcf = ((nup[0:Nz + 1, 1:-1, 0:Nx + 1] + nup[0:Nz + 1, 1:-1, 1:Nx + 2]) / 2 +
              (nup[1:Nz + 2, 1:-1, 0:Nx + 1] + nup[1:Nz + 2, 1:-1, 1:Nx + 2]) / 2) / 2          # (Nz+1, Ny, Nx+1)
    nu_cff = ((nup[0:Nz + 1, 0:Ny + 1, 1:-1] + nup[0:Nz + 1, 1:Ny + 2, 1:-1]) / 2 +
              (nup[1:Nz + 2, 0:Ny + 1, 1:-1] + nup[1:Nz + 2, 1:Ny + 2, 1:-1]) / 2) / 2          # (Nz+1, Ny+1, Nx)
    # dynamic stresses
    T11 = r3 * (-2 * nu * S11)
    T22 = r3 * (-2 * nu * S22)
    T33 = r3 * (-2 * nu * S33)
    T12 = r3 * (-2 * nu_ffc * S12)
    T13 = rf3 * (-2 * nu_fcf * S13)
    T23 = rf3 * (-2 * nu_cff * S23)
    Ax, Ay, Az = dy * dz3, dx * dz3, dx * dy
    Vc = dx * dy * dz3
    # x momentum at faces i = 0..Nx-1: d_x T11 between centres i-1 and i (periodic), d_y T12, d_z T13
    T11m = np.roll(T11, 1, axis=2)
    div_u = (Ax * T11 - Ax * T11m) + (Ay * T12[:, 1:, :-1] - Ay * T12[:, :-1, :-1]) + (Az * T13[1:, :, :-1] - Az * T13[:-1, :, :-1])
    I = g.interior
    if g.topo[0] == 1:      # walls in x: the wall face i = 0 is never updated
        I(m.G["ru"])[:, :, 1:] -= (div_u / Vc)[:, :, 1:]
    else:
        I(m.G["ru"])[...] -= div_u / Vc
    T22m = np.roll(T22, 1, axis=1)
    div_v = (Ax * T12[:, :-1, 1:] - Ax * T12[:, :-1, :-1]) + (Ay * T22 - Ay * T22m) + (Az * T23[1:, :-1, :] - Az * T23[:-1, :-1, :])
    if g.topo[1] == 1:      # walls in y: the wall face j = 0 is never updated (its rolled T22 is meaningless)
        I(m.G["rv"])[:, 1:, :] -= (div_v / Vc)[:, 1:, :]
    else:
        I(m.G["rv"])[...] -= div_v / Vc
    # z momentum at interior faces k = 1..Nz-1
    dzf3 = dzf[1:Nz, None, None]
    Axf, Ayf = dy * dzf3, dx * dzf3
    Vf = dx * dy * dzf3
    div_w = (Axf * T13[1:Nz, :, 1:] - Axf * T13[1:Nz, :, :-1]) + (Ayf * T23[1:Nz, 1:, :] - Ayf * T23[1:Nz, :-1, :]) + \
            (Az * T33[1:] - Az * T33[:-1])
    g.interior(m.G["rw"], zface=True)[1:Nz] -= div_w / Vf
    # scalars
    # every scalar of the model diffuses with kappa = nu_e / Pr (scalar_tendency: - div J^c for the thermodynamic variable, the moisture
    # and each user tracer alike; update_atmosphere_model_state.jl:352-372)
    scalars = [("rtheta", m.theta), ("rq", m.q)] + [(f"rc{t}", getattr(m, f"c{t}")) for t in range(getattr(m, "n_tracers", 0))]
    for name, field in scalars:
        c = _pad_center(g, field)
        kap = nup / cl.Pr
        kx = (kap[1:-1, 1:-1, 0:Nx + 1] + kap[1:-1, 1:-1, 1:Nx + 2]) / 2                    # x faces 0..Nx
        ky = (kap[1:-1, 0:Ny + 1, 1:-1] + kap[1:-1, 1:Ny + 2, 1:-1]) / 2
        kz = (kap[0:Nz + 1, 1:-1, 1:-1] + kap[1:Nz + 2, 1:-1, 1:-1]) / 2
        Jx = r3 * (-kx * ((c[1:-1, 1:-1, 1:Nx + 2] - c[1:-1, 1:-1, 0:Nx + 1]) / dx))
        Jy = r3 * (-ky * ((c[1:-1, 1:Ny + 2, 1:-1] - c[1:-1, 0:Ny + 1, 1:-1]) / dy))
        Jz = rf3 * (-kz * ((c[1:Nz + 2, 1:-1, 1:-1] - c[0:Nz + 1, 1:-1, 1:-1]) / dzf[:, None, None]))
        div = (Ax * Jx[:, :, 1:] - Ax * Jx[:, :, :-1]) + (Ay * Jy[:, 1:, :] - Ay * Jy[:, :-1, :]) + (Az * Jz[1:] - Az * Jz[:-1])
        I(m.G[name])[...] -= div / Vc
