#!/usr/bin/env python
"""tools/bench_bomex.py — timing of the BOMEX-shaped anelastic step (BASELINE configs[2], examples/bomex.jl).

Workload: 256 x 256 x 128 cells over 6.4 km x 6.4 km x 3 km (the example's 25 m x 25 m x 23.4 m... grid scaled to the config
size), AnelasticDynamics + WENO-5 + SaturationAdjustment(WarmPhaseEquilibrium) + the example's forcing stack: f-plane Coriolis,
geostrophic forcing, subsidence of u, v, theta, q^e (horizontal averages every stage), drying, radiative cooling through the
energy forcing, bottom sensible / latent / drag fluxes, and closure = SmagorinskyLilly() — the physics list of the configuration
(WENO5 instead of the example's WENO9).  Float64.  One JSON line with per-kernel milliseconds.

    python tools/bench_bomex.py --size 256 256 128 --dt 2.0 --steps 10 --warmup 3 [--no-forcing]
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
    ap.add_argument("--size", type=int, nargs=3, default=[256, 256, 128])
    ap.add_argument("--dt", type=float, default=2.0)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--single-calls", action="store_true", help="one bz_time_step_anelastic call per step instead of one bz_time_steps_anelastic call")
    ap.add_argument("--graph", action="store_true", help="replay recorded steps (hipGraph, bz_graph_enable) in the timed pass")
    ap.add_argument("--no-forcing", action="store_true")
    ap.add_argument("--no-closure", action="store_true")
    ap.add_argument("--order", type=int, default=5, choices=(5, 7, 9), help="WENO order (examples/bomex.jl:204 uses 9; generic kernels)")
    ap.add_argument("--float32", action="store_true", help="eltype(grid) = Float32, the precision of examples/bomex.jl (libbreeze_hip_f32.so)")
    a = ap.parse_args()
    import torch
    import breeze_jl_amd as bz
    from test_forcings import _hip_forcing_kwargs
    Nx, Ny, Nz = a.size
    grid = bz.RectilinearGrid((Nx, Ny, Nz), x=(0.0, 6400.0), y=(0.0, 6400.0), z=(0.0, 3000.0),
                              float_type=np.float32 if a.float32 else np.float64, **({"halo": (5, 5, 5)} if a.order != 5 else {}))
    ref = bz.ReferenceState(grid, surface_pressure=101500.0, potential_temperature=299.1)
    kw = {} if a.no_forcing else _hip_forcing_kwargs(bz, full=True)
    if not a.no_closure:
        kw["closure"] = bz.SmagorinskyLilly()
    m = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=a.order),
                           microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium()), **kw)
    rng = np.random.default_rng(0)
    noise_t = rng.standard_normal((Nz, Ny, Nx))
    noise_q = rng.standard_normal((Nz, Ny, Nx))

    def theta(x, y, z):      # Siebesma et al. (2003) initial profile + the example's noise below 1600 m
        base = np.where(z < 520.0, 298.7, np.where(z < 1480.0, 298.7 + (z - 520.0) * (302.4 - 298.7) / 960.0,
                        np.where(z < 2000.0, 302.4 + (z - 1480.0) * (308.2 - 302.4) / 520.0, 308.2 + (z - 2000.0) * 3.65e-3)))
        return base + 0.1 * noise_t * (z < 1600.0)

    def qt(x, y, z):
        base = np.where(z < 520.0, 17.0 + z * (16.3 - 17.0) / 520.0, np.where(z < 1480.0, 16.3 + (z - 520.0) * (10.7 - 16.3) / 960.0,
                        np.where(z < 2000.0, 10.7 + (z - 1480.0) * (4.2 - 10.7) / 520.0, 4.2 + (z - 2000.0) * (-1.2e-3)))) * 1e-3
        return base + 2.5e-5 * noise_q * (z < 1600.0)

    u = lambda x, y, z: np.where(z < 700.0, -8.75, -8.75 + (z - 700.0) * 1.8e-3) + 0 * x + 0 * y
    m.set(θ=theta, qᵗ=qt, u=u)
    for _ in range(a.warmup):
        m.time_step(a.dt)
    m.synchronize()
    # timed region: unprofiled (the HIP events of the kernel table below sit between the launches of a step and cost ~0.3 ms of it), the
    # reference's benchmark loop (benchmarking/src/timestepping.jl:11-16: many_time_steps!) unless --single-calls, as bench.py's headline
    def run(n):
        if a.single_calls:
            for _ in range(n):
                m.time_step(a.dt)
        else:
            m.time_steps(a.dt, n)

    if a.graph:
        m.graph_enable(True)
        for _ in range(3):
            run(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.steps)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    graph_info = m.graph_info() if a.graph else None
    if a.graph:
        m.graph_enable(False)
    # kernel table: a second pass with HIP events around the kernel groups
    m.profile_enable(True)
    m.profile_reset()
    run(a.steps)
    torch.cuda.synchronize()
    m.profile_enable(False)
    prof = m.profile()
    ms = (t1 - t0) / a.steps * 1e3
    cells = Nx * Ny * Nz
    ql = m.microphysical_fields["qˡ"].interior
    w = m.velocities["w"].interior
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from accounting import HBM_PEAK_GBS, compulsory_words, roofline_block, step_compulsory_words
    word = 4 if a.float32 else 8
    kern = {k: {"avg_ms": v[0] / v[1], "launches": v[1], "total_ms": v[0]} for k, v in prof.items() if v[1]}
    # project_and_diagnose of a saturation-adjustment model writes q^v, q^l as well
    words_of = lambda k: compulsory_words("project_and_diagnose+saturation_adjustment" if k == "project_and_diagnose" else k)      # noqa: E731
    known = [k for k in kern if words_of(k)]
    dom = max(known, key=lambda k: kern[k]["total_ms"]) if known else None
    roofline = roofline_block(dom, kern[dom]["avg_ms"], cells, word, words=words_of(dom)) if dom else None
    step_words = sum(words_of(k) * v["launches"] / a.steps for k, v in kern.items() if words_of(k))
    step_gbs = cells / (ms * 1e-3) * step_words * word / 1e9
    out = {"metric": "grid-cells advanced/sec, BOMEX-shaped anelastic SSP-RK3 step (WENO + saturation adjustment + SmagorinskyLilly + forcing stack)", "weno_order": a.order,
           "value": cells / (ms * 1e-3), "unit": "cells/s", "ms_per_step": ms, "grid": [Nx, Ny, Nz], "dt": a.dt,
           "stepping": ("one call per step" if a.single_calls else "bz_time_steps_anelastic (many_time_steps!)") + (", recorded steps replayed (hipGraph)" if a.graph else ""),
           "graph_info": graph_info, "kernel_times": "HIP events of a second pass of %d steps after the timed region" % a.steps, "forcing": not a.no_forcing, "closure": None if a.no_closure else "SmagorinskyLilly", "dtype": "f32" if a.float32 else "f64",
           "kernels_ms_per_step": {k: v[0] / a.steps for k, v in sorted(prof.items())},
           "kernel_launches_per_step": {k: v[1] / a.steps for k, v in sorted(prof.items())},
           # one accounting (tools/accounting.py): the dominant kernel group and the whole step in COMPULSORY bytes against 8 TB/s.  At this size
           # (8.4 M cells: 67 MB per Float64 array) every array of a kernel fits the 256 MB Infinity Cache, so the HBM roof is not what bounds
           # these kernels — the fractions say how far the launch / latency / issue floor of a small grid sits below it
           "roofline": roofline,
           "step_roofline": {"bound": "hbm", "achieved": step_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": step_gbs / HBM_PEAK_GBS,
                             "bytes": "compulsory", "compulsory_words_per_cell_step": step_words, "unpriced_groups": sorted(k for k in kern if not words_of(k))},
           "step_contract_frac_of_8TBs": cells * (1000 if a.float32 else 2000) / (ms * 1e-3) / 8e12,
           "finite": bool(torch.isfinite(w).all().item()), "w_max": float(w.abs().max().item()),
           "cloud_fraction": float((ql > 0).double().mean().item())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
