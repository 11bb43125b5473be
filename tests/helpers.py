"""Shared helpers for the parity tests: matched oracle / HIP models on the same seeded inputs."""
import numpy as np


def bubble_theta(theta0, g, N2=1e-6, dtheta=10.0, r0=2e3, zc=3000.0):
    def f(x, y, z):
        r = np.sqrt(x ** 2 + y ** 2 + (z - zc) ** 2)
        return theta0 * np.exp(N2 * z / g) + dtheta * np.maximum(0.0, 1.0 - r / r0)
    return f


def make_pair(orc, bz, size, halo=(3, 3, 3), extent=((-10e3, 10e3), (-10e3, 10e3), (0.0, 10e3)), theta0=300.0,
              z_faces=None, formulation="LiquidIcePotentialTemperature"):
    """Return (oracle model, HIP model) on identical grids / reference states."""
    z = z_faces if z_faces is not None else extent[2]
    og = orc.Grid(size, x=extent[0], y=extent[1], z=z, halo=halo)
    om = orc.OracleModel(og, potential_temperature=theta0, formulation=formulation)
    grid = bz.RectilinearGrid(size, x=extent[0], y=extent[1], z=z, halo=halo)
    ref = bz.ReferenceState(grid, potential_temperature=theta0)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5), formulation=formulation)
    return om, hm


ORACLE_TO_HIP = {
    "ru": lambda m: m.momentum["ρu"], "rv": lambda m: m.momentum["ρv"], "rw": lambda m: m.momentum["ρw"],
    "rtheta": lambda m: m.potential_temperature_density, "rq": lambda m: m.moisture_density,
    "u": lambda m: m.velocities["u"], "v": lambda m: m.velocities["v"], "w": lambda m: m.velocities["w"],
    "theta": lambda m: m.potential_temperature, "q": lambda m: m.specific_moisture, "T": lambda m: m.temperature,
    "phi": lambda m: m.dynamics.pressure_anomaly,
}
PROG = {"ru": "ρu", "rv": "ρv", "rw": "ρw", "rtheta": "ρθ", "rq": "ρq"}


def push_state(om, hm, names=None):
    """Copy oracle parent arrays (halos included) into the HIP model's fields, bit for bit."""
    import torch
    for n in (names or ORACLE_TO_HIP):
        ORACLE_TO_HIP[n](hm).parent.copy_(torch.from_numpy(getattr(om, n)))
    for n, k in PROG.items():
        hm.G[k].parent.copy_(torch.from_numpy(om.G[n]))
        hm.U0[k].parent.copy_(torch.from_numpy(om.U0[n]))


def randomize(om, seed, amp_u=5.0, amp_theta=2.0, amp_q=5e-3):
    """Seeded smooth + rough perturbation of every prognostic field of the oracle model (interior),
    followed by update_state so that halos / diagnostics are consistent."""
    g = om.grid
    rng = np.random.default_rng(seed)
    x, y, z = g.nodes("ccc")
    Lx, Ly, Lz = g.Nx * g.dx, g.Ny * g.dy, g.zf[-1] - g.zf[0]

    def field(shape, amp):
        smooth = np.sin(2 * np.pi * x / Lx + 0.3) * np.cos(2 * np.pi * y / Ly - 0.2) * np.sin(np.pi * (z - g.zf[0]) / Lz)
        return amp * (np.broadcast_to(smooth, shape) * 0.7 + 0.3 * rng.standard_normal(shape))

    Hz, Nz = g.Hz, g.Nz
    rho_c = om.ref.density[Hz:Hz + Nz][:, None, None]
    sh = (g.Nz, g.Ny, g.Nx)
    g.interior(om.ru)[...] = rho_c * field(sh, amp_u)
    g.interior(om.rv)[...] = rho_c * field(sh, amp_u)
    wi = np.zeros((g.Nz + 1, g.Ny, g.Nx))
    wi[1:-1] = 0.5 * (field(sh, amp_u)[1:] + field(sh, amp_u)[:-1])
    g.interior(om.rw, True)[...] = wi
    g.interior(om.rtheta)[...] = rho_c * (om.ref.theta0 + field(sh, amp_theta))
    g.interior(om.rq)[...] = rho_c * np.abs(field(sh, amp_q))
    om.update_state(compute_tendencies=False)


def relerr(a, b):
    scale = np.max(np.abs(b))
    return np.max(np.abs(a - b)) / (scale if scale > 0 else 1.0)
