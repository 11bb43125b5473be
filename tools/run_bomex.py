#!/usr/bin/env python
"""tools/run_bomex.py — a short BOMEX integration on the device as an end-to-end sanity check of the configs[2] physics list
(WENO5 + saturation adjustment + SmagorinskyLilly + Coriolis/geostrophic/subsidence/drying/radiative forcings + bottom fluxes):
prints the horizontally averaged state every --every steps so that cloud base / cloud fraction / w_max can be compared with the
Siebesma et al. (2003) intercomparison (cloud base ~500-600 m, cloud cover of a few per cent after the spin-up hour).

    python tools/run_bomex.py --size 128 128 96 --dt 2.0 --steps 2700 --every 300
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, nargs=3, default=[128, 128, 96])
    ap.add_argument("--dt", type=float, default=2.0)
    ap.add_argument("--steps", type=int, default=2700)
    ap.add_argument("--every", type=int, default=300)
    a = ap.parse_args()
    import torch
    import breeze_jl_amd as bz
    from test_forcings import _hip_forcing_kwargs
    Nx, Ny, Nz = a.size
    grid = bz.RectilinearGrid((Nx, Ny, Nz), x=(0.0, 6400.0), y=(0.0, 6400.0), z=(0.0, 3000.0))
    ref = bz.ReferenceState(grid, surface_pressure=101500.0, potential_temperature=299.1)
    m = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5), closure=bz.SmagorinskyLilly(),
                           microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium()),
                           **_hip_forcing_kwargs(bz, full=True))
    rng = np.random.default_rng(0)
    nt, nq = rng.standard_normal((Nz, Ny, Nx)), rng.standard_normal((Nz, Ny, Nx))

    def theta(x, y, z):
        base = np.where(z < 520.0, 298.7, np.where(z < 1480.0, 298.7 + (z - 520.0) * (302.4 - 298.7) / 960.0,
                        np.where(z < 2000.0, 302.4 + (z - 1480.0) * (308.2 - 302.4) / 520.0, 308.2 + (z - 2000.0) * 3.65e-3)))
        return base + 0.1 * nt * (z < 1600.0)

    def qt(x, y, z):
        base = np.where(z < 520.0, 17.0 + z * (16.3 - 17.0) / 520.0, np.where(z < 1480.0, 16.3 + (z - 520.0) * (10.7 - 16.3) / 960.0,
                        np.where(z < 2000.0, 10.7 + (z - 1480.0) * (4.2 - 10.7) / 520.0, 4.2 + (z - 2000.0) * (-1.2e-3)))) * 1e-3
        return base + 2.5e-5 * nq * (z < 1600.0)

    u = lambda x, y, z: np.where(z < 700.0, -8.75, -8.75 + (z - 700.0) * 1.8e-3) + 0 * x + 0 * y
    m.set(θ=theta, qᵗ=qt, u=u)
    zc = np.asarray(grid.zᶜ)
    t0 = time.perf_counter()
    rows = []
    for n in range(1, a.steps + 1):
        m.time_step(a.dt)
        if n % a.every == 0 or n == a.steps:
            ql = m.microphysical_fields["qˡ"].interior
            w = m.velocities["w"].interior
            cloudy_col = (ql > 1e-8).any(dim=0).double().mean().item()
            prof = ql.mean(dim=(1, 2)).cpu().numpy()
            cloudy_levels = zc[prof > 1e-9]
            row = {"step": n, "t_min": n * a.dt / 60.0, "w_max": float(w.abs().max().item()),
                   "cloud_cover": cloudy_col, "ql_max_g_per_kg": float(ql.max().item() * 1e3),
                   "cloud_base_m": float(cloudy_levels.min()) if cloudy_levels.size else None,
                   "cloud_top_m": float(cloudy_levels.max()) if cloudy_levels.size else None,
                   "theta_mean_lowest": float(m.potential_temperature.interior[0].mean().item()),
                   "finite": bool(torch.isfinite(w).all().item()), "wall_s": time.perf_counter() - t0}
            rows.append(row)
            print(json.dumps(row), flush=True)
    return rows


if __name__ == "__main__":
    main()
