#!/usr/bin/env python
"""tools/long_run_check.py — a few hundred steps of the headline case (256^3 here), of its Float32 twin and of the CBL case with WENO9 through
the multi-step seam: finite fields, conserved rho theta (flux-form advection inside periodic / closed boundaries; the CBL gains its surface
flux), bounded vertical velocity, discrete divergence at round-off.  A robustness run, not a parity test (those are tests/)."""
import json
import sys
import os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import breeze_jl_amd as bz

out = []
for ft, name in ((np.float64, "f64"), (np.float32, "f32")):
    grid = bz.RectilinearGrid((256, 256, 256), x=bench.EXTENT[0], y=bench.EXTENT[1], z=bench.EXTENT[2], float_type=ft)
    m = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, surface_pressure=101325, potential_temperature=300)),
                           advection=bz.WENO(order=5))
    m.set(θ=bench.bubble)
    s0 = float(m.potential_temperature_density.interior.sum(dtype=torch.float64))
    rec = {"case": f"dry bubble 256^3 {name}", "steps": 0, "checks": []}
    for chunk in range(6):
        m.time_steps(0.5, 50)      # dz = 39 m, |w| reaches 30 m/s: CFL 0.4
        m.synchronize()
        rec["steps"] += 50
        w = m.velocities["w"].interior
        s1 = float(m.potential_temperature_density.interior.sum(dtype=torch.float64))
        rec["checks"].append({"step": rec["steps"], "finite": bool(torch.isfinite(w).all() and torch.isfinite(m.temperature.interior).all()),
                              "max_abs_w": float(w.abs().max()), "rho_theta_drift": abs(s1 - s0) / abs(s0), "max_abs_divergence": float(m.max_abs_divergence())})
    out.append(rec)
    del m
    torch.cuda.empty_cache()
m = bz.benchmarks.convective_boundary_layer((256, 256, 128), float_type=np.float32, advection=bz.WENO(order=9))
rec = {"case": "CBL 256x256x128 f32 WENO9", "steps": 0, "checks": []}
for chunk in range(6):
    m.time_steps(0.05, 100)
    m.synchronize()
    rec["steps"] += 100
    w = m.velocities["w"].interior
    rec["checks"].append({"step": rec["steps"], "finite": bool(torch.isfinite(w).all() and torch.isfinite(m.temperature.interior).all()),
                          "max_abs_w": float(w.abs().max()), "max_abs_divergence": float(m.max_abs_divergence())})
out.append(rec)
print(json.dumps(out, indent=1))
