#!/usr/bin/env python
"""tools/run_supercell.py — the splitting-supercell case of BASELINE configs[4] (examples/splitting_supercell.jl; Klemp et al. 2015)
on the device as an end-to-end sanity check of CompressibleDynamics split-explicit WS-RK3 + DCMIP2016 Kessler (or the anelastic core):
168 km x 168 km x 20 km box, theta / RH / shear sounding of the example, 3 K bubble, dt = 2 s.  WENO-5 instead of the example's
WENO-9.  Prints w_max, rain and cloud maxima and the surface precipitation every --every steps.

    python tools/run_supercell.py --dynamics compressible --size 168 168 40 --steps 3600 --every 450
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

G, CPD, RD, RV = 9.81, 1005.0, 8.314462618 / 0.02897, 8.314462618 / 0.018015
TH0, THP, ZP, TP, QVMAX = 300.0, 343.0, 12000.0, 213.0, 0.014
ZS, US, UC = 5000.0, 30.0, 15.0
P0 = PST = 1e5


def theta_background(z):
    z = np.asarray(z, dtype=np.float64)
    tht = TH0 + (THP - TH0) * (np.minimum(z, ZP) / ZP) ** 1.25
    ths = THP * np.exp(G / (CPD * TP) * (z - ZP))
    return np.where(z <= ZP, tht, ths)


def _hydrostatic(z):
    """dry hydrostatic T, rho under theta_background: Exner integrated from the surface (fine trapezoid)."""
    zs = np.linspace(0.0, float(z), max(2, int(z / 5.0) + 2))
    integrand = G / (CPD * theta_background(zs))
    Pi = (P0 / PST) ** (RD / CPD) - np.trapz(integrand, zs)
    T = float(theta_background(z)) * Pi
    p = PST * Pi ** (CPD / RD)
    return T, p / (RD * T)


def qv_background(z):
    H = (1 - 0.75 * (z / ZP) ** 1.25) if z <= ZP else 0.25
    T, rho = _hydrostatic(z)
    psat = 610.0 * np.exp(17.27 * (T - 273.15) / (T - 35.85))          # TetensFormula defaults
    return min(H * psat / (rho * RV * T), QVMAX)


def u_background(z):
    ul = US * (z / ZS) - UC
    ut = (-0.8 + 3 * (z / ZS) - 1.25 * (z / ZS) ** 2) * US - UC
    return np.where(z < ZS - 1000, ul, np.where(np.abs(z - ZS) <= 1000, ut, US - UC))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dynamics", choices=("compressible", "anelastic"), default="compressible")
    ap.add_argument("--size", type=int, nargs=3, default=[168, 168, 40])
    ap.add_argument("--dt", type=float, default=2.0)
    ap.add_argument("--steps", type=int, default=3600)
    ap.add_argument("--every", type=int, default=450)
    a = ap.parse_args()
    import torch
    import breeze_jl_amd as bz
    Nx, Ny, Nz = a.size
    Lx = Ly = 168e3
    grid = bz.RectilinearGrid((Nx, Ny, Nz), x=(0.0, Lx), y=(0.0, Ly), z=(0.0, 20e3))
    tc = bz.ThermodynamicConstants(saturation_vapor_pressure=bz.TetensFormula())
    zc = np.asarray(grid.zᶜ)
    qv_col = np.array([qv_background(z) for z in zc])
    qv_of = lambda z: float(np.interp(z, zc, qv_col))

    def theta_i(x, y, z):
        r = np.sqrt((x - Lx / 2) ** 2 + (y - Ly / 2) ** 2)
        R = np.sqrt((r / 10e3) ** 2 + ((z - 1500.0) / 1500.0) ** 2)
        return theta_background(z) + np.where(R < 1, 3.0 * np.cos(np.pi * R / 2) ** 2, 0.0)

    qv_i = lambda x, y, z: qv_col[:, None, None] + 0 * x + 0 * y
    u_i = lambda x, y, z: u_background(z) + 0 * x + 0 * y
    if a.dynamics == "compressible":
        dyn = bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(), surface_pressure=P0, standard_pressure=PST,
                                      reference_potential_temperature=lambda z: theta_background(z),
                                      reference_vapor_mass_fraction=qv_of)
        m = bz.CompressibleAtmosphereModel(grid, dyn, advection=bz.WENO(order=5), thermodynamic_constants=tc,
                                           microphysics=bz.DCMIP2016KesslerMicrophysics())
        Hz = grid.Hz
        rho_ref = m.dynamics.reference_state.density[Hz:Hz + Nz][:, None, None]
        # pressure_balanced_density (src/Thermodynamics/reference_states.jl:125-160): rho theta unchanged by the bubble
        rho_i = lambda x, y, z: rho_ref * theta_background(z) / theta_i(x, y, z)
        m.set(ρ=rho_i, θ=theta_i, qᵗ=qv_i, u=u_i, v=0.0, w=0.0)
    else:
        ref = bz.ReferenceState(grid, tc, surface_pressure=P0, potential_temperature=TH0, standard_pressure=PST)
        m = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5), thermodynamic_constants=tc,
                               microphysics=bz.DCMIP2016KesslerMicrophysics())
        m.set(θ=theta_i, qᵗ=qv_i, u=u_i)
    μ = m.microphysical_fields
    rho_surface = P0 / (RD * TH0)
    t0 = time.perf_counter()
    for n in range(1, a.steps + 1):
        m.time_step(a.dt)
        if n % a.every == 0 or n == a.steps:
            w = m.velocities["w"].interior
            P = μ["precipitation_rate"]
            row = {"dynamics": a.dynamics, "step": n, "t_min": n * a.dt / 60.0, "w_max": float(w.max().item()),
                   "w_min": float(w.min().item()), "qcl_max_g_per_kg": float(μ["qᶜˡ"].interior.max().item() * 1e3),
                   "qr_max_g_per_kg": float(μ["qʳ"].interior.max().item() * 1e3),
                   # precipitation_rate is q^r W^r at the surface (m/s per unit mass fraction, dcmip2016_kessler.jl:657-666):
                   # times the surface air density it is kg m^-2 s^-1 = mm/s of rain
                   "precip_max_mm_per_h": float(P.max().item() * rho_surface * 3600.0),
                   "finite": bool(torch.isfinite(w).all().item()), "wall_s": time.perf_counter() - t0}
            print(json.dumps(row), flush=True)
            if not row["finite"]:
                break


if __name__ == "__main__":
    main()
