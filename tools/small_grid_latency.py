#!/usr/bin/env python
"""tools/small_grid_latency.py — per-step wall time of small (launch-bound) problems: BASELINE configs[0] (256 x 256 2-D dry bubble,
topology (Periodic, Flat, Bounded)), small 3-D boxes, and a small compressible box.  Prints one JSON line per case with ms/step,
Mcells/s and the number of profile groups launched per step.  With BZ_GRAPH=1 / 0 in the environment the same cases run with /
without hipGraph replay of the step (csrc/bz_graph.hip).

    python tools/small_grid_latency.py [--steps 200]
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


def timed(step, sync, steps, warmup=10):
    for _ in range(warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--only", default=None, help="substring of the case name")
    a = ap.parse_args()
    import torch
    import breeze_jl_amd as bz
    from helpers import bubble_theta
    cases = []
    g2 = bz.RectilinearGrid((256, 256), x=(-10e3, 10e3), z=(0.0, 10e3), topology=("Periodic", "Flat", "Bounded"))
    cases.append(("configs[0] dry bubble 256x256 (Periodic,Flat,Bounded)", g2, 0.5))
    for n in (32, 64, 128):
        cases.append((f"dry bubble {n}^3", bz.RectilinearGrid((n, n, n), x=(-10e3, 10e3), y=(-10e3, 10e3), z=(0.0, 10e3)), 0.5))
    for name, grid, dt in cases:
        if a.only and a.only not in name:
            continue
        ref = bz.ReferenceState(grid, potential_temperature=300.0)
        m = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5))
        th = bubble_theta(300.0, 9.81, r0=2e3, zc=3e3)
        if grid.topology[1] == "Flat":
            m.set(θ=lambda x, z: th(x, 0 * x, z))
        else:
            m.set(θ=th)
        ms = timed(lambda: m.time_step(dt), torch.cuda.synchronize, a.steps)
        cells = grid.Nx * grid.Ny * grid.Nz
        w = float(m.velocities["w"].interior.abs().max())
        print(json.dumps({"case": name, "ms_per_step": round(ms, 4), "Mcells_per_s": round(cells / ms / 1e3, 1),
                          "graph": os.environ.get("BZ_GRAPH", "default"), "max_abs_w": w}), flush=True)
        del m
    if a.only and a.only not in "compressible dry bubble 64^3":
        return
    # small compressible box: 6 acoustic substeps per stage-3, the launch count per step is what matters
    grid = bz.RectilinearGrid((64, 64, 64), x=(-10e3, 10e3), y=(-10e3, 10e3), z=(0.0, 10e3))
    dyn = bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(substeps=6), surface_pressure=1e5, reference_potential_temperature=300.0)
    cm = bz.CompressibleAtmosphereModel(grid, dyn, advection=bz.WENO(order=5))
    cm.set(θ=bubble_theta(300.0, 9.81, r0=2e3, zc=3e3))
    ms = timed(lambda: cm.time_step(0.5), torch.cuda.synchronize, a.steps)
    print(json.dumps({"case": "compressible dry bubble 64^3", "ms_per_step": round(ms, 4), "Mcells_per_s": round(64 ** 3 / ms / 1e3, 1),
                      "graph": os.environ.get("BZ_GRAPH", "default")}), flush=True)


if __name__ == "__main__":
    main()
