#!/usr/bin/env python
"""tools/bench_compressible.py — timing of the compressible split-explicit step (second milestone).

Workload: the dry thermal bubble of bench.py with pressure-balanced density, CompressibleDynamics +
SplitExplicitTimeDiscretization defaults (omega = 0.65, acoustic_cfl = 0.5, ThermalDivergenceDamping 0.1,
ProportionalSubsteps), WENO-5, Float64.  Prints one JSON line: cells/s, ms/step, substeps per step, per-kernel-group
milliseconds (HIP events on the launch stream) and the HBM fraction of the acoustic substep (contract words of
SURVEY §8d: 58 words/cell/substep; this build moves fewer).

    python tools/bench_compressible.py --size 512 512 256 --dt 1.0 --steps 5 --warmup 2
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, nargs=3, default=[512, 512, 256])
    ap.add_argument("--dt", type=float, default=1.0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--substeps", type=int, default=0)
    ap.add_argument("--substep-float32", action="store_true", help="substep_floattype = Float32 inside the Float64 model")
    ap.add_argument("--kessler", action="store_true",
                    help="BASELINE configs[4] physics: DCMIP2016 Kessler on the 168 km x 168 km x 20 km supercell box "
                         "(examples/splitting_supercell.jl:88-96), moist sounding + warm bubble")
    ap.add_argument("--moist", action="store_true", help="vapour set in the benchmark bubble (every stage accumulates, moisture tendency evaluated)")
    a = ap.parse_args()
    import torch
    import breeze_jl_amd as bz
    Nx, Ny, Nz = a.size
    if a.kessler:
        grid = bz.RectilinearGrid((Nx, Ny, Nz), x=(0.0, 168e3), y=(0.0, 168e3), z=(0.0, 20e3))
    else:
        grid = bz.RectilinearGrid((Nx, Ny, Nz), x=(-10e3, 10e3), y=(-10e3, 10e3), z=(0.0, 10e3))
    td = bz.SplitExplicitTimeDiscretization(substeps=a.substeps or None)
    dyn = bz.CompressibleDynamics(td, surface_pressure=1e5, reference_potential_temperature=300.0)
    mkw = {}
    if a.kessler:
        mkw = dict(thermodynamic_constants=bz.ThermodynamicConstants(saturation_vapor_pressure=bz.TetensFormula()),
                   microphysics=bz.DCMIP2016KesslerMicrophysics())
    import numpy as _np
    m = bz.CompressibleAtmosphereModel(grid, dyn, advection=bz.WENO(order=5), substep_floattype=_np.float32 if a.substep_float32 else None, **mkw)
    c = m.thermodynamic_constants
    Rd, cpd, g = 8.314462618 / c.dry_air_molar_mass, c.dry_air_heat_capacity, c.gravitational_acceleration
    kap = Rd / cpd

    def theta(x, y, z):
        r = np.sqrt(x ** 2 + y ** 2 + (z - 3000.0) ** 2)
        return 300.0 + 10.0 * np.maximum(0.0, 1.0 - r / 2000.0) + 0 * z

    def rho(x, y, z):
        ex = 1.0 - g * z / (cpd * 300.0)
        return 1e5 * ex ** (1 / kap) / (Rd * theta(x, y, z) * ex)

    if a.kessler:
        def bubble(x, y, z):     # 3 K bubble at (84 km, 84 km, 1.5 km), radii 10 km x 1.5 km (the example's shape)
            r = np.sqrt(((x - 84e3) / 10e3) ** 2 + ((y - 84e3) / 10e3) ** 2 + ((z - 1500.0) / 1500.0) ** 2)
            return 3.0 * np.cos(np.pi / 2 * np.minimum(r, 1.0)) ** 2

        def theta(x, y, z):      # noqa: F811 — neutral column (the model's reference) + bubble
            return 300.0 + bubble(x, y, z)

        def rho(x, y, z):        # noqa: F811 — pressure_balanced_density against the reference column
            Hz = grid.Hz
            col = m.dynamics.reference_state.density[Hz:Hz + Nz]
            return col[:, None, None] * 300.0 / theta(x, y, z)

        qv = lambda x, y, z: 0.014 * np.exp(-z / 2500.0) + 0 * x + 0 * y
        m.set(ρ=rho, θ=theta, u=lambda x, y, z: 0.0015 * np.minimum(z, 5e3) + 0 * x + 0 * y, v=0.0, w=0.0, qᵗ=qv)
    else:
        m.set(ρ=rho, θ=theta, u=0.0, v=0.0, w=0.0, qᵗ=(lambda x, y, z: 4e-3 * np.exp(-z / 2500.0) + 0 * x + 0 * y) if a.moist else 0.0)
    for _ in range(a.warmup):
        m.time_step(a.dt)
    m.synchronize()
    m.profile_enable(True)
    m.profile_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        m.time_step(a.dt)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    m.profile_enable(False)
    prof = m.profile()
    ms = (t1 - t0) / a.steps * 1e3
    nsub = [m.stage_substeps(a.dt, b)[0] for b in (1 / 3, 1 / 2, 1.0)]
    cells = Nx * Ny * Nz
    per = {k: v[0] / a.steps for k, v in sorted(prof.items())}
    sub_ms = sum(per.get(k, 0.0) for k in ("acoustic_horizontal", "acoustic_column_forward", "acoustic_horizontal+column_forward",
                                               "acoustic_column_backward"))
    per_sub = sub_ms / sum(nsub)
    w = m.velocities["w"].interior
    if a.kessler:
        μ = m.microphysical_fields
        extra = {"physics": "DCMIP2016 Kessler", "qcl_max": float(μ["qᶜˡ"].interior.max().item()),
                 "qr_max": float(μ["qʳ"].interior.max().item())}
    else:
        extra = {}
    out = {"metric": "grid-cells advanced/sec, compressible split-explicit WS-RK3 step", "value": cells / (ms * 1e-3),
           "unit": "cells/s", "ms_per_step": ms, "grid": [Nx, Ny, Nz], "dt": a.dt, "substeps_per_stage": nsub,
           "dtype": "f64", "kernels_ms_per_step": per, "acoustic_ms_per_substep": per_sub,
           "acoustic_substep_contract_GBs": cells * 58 * 8 / (per_sub * 1e-3) / 1e9,
           "acoustic_substep_frac_of_8TBs": cells * 58 * 8 / (per_sub * 1e-3) / 8e12,
           "finite": bool(torch.isfinite(w).all().item()), "w_max": float(w.max().item())}
    out.update(extra)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
