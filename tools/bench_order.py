#!/usr/bin/env python
"""tools/bench_order.py — the dry-bubble step at N^3 with WENO(order), halo 5 for orders 7 / 9 (the measurement behind DESIGN.md's
"WENO orders 7 and 9" table): python tools/bench_order.py --size 256 --order 9 [--float32]"""
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
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--order", type=int, default=9)
    ap.add_argument("--float32", action="store_true")
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    import torch
    import breeze_jl_amd as bz
    N = a.size
    grid = bz.RectilinearGrid((N, N, N), x=(-10e3, 10e3), y=(-10e3, 10e3), z=(0.0, 10e3), halo=(5, 5, 5) if a.order != 5 else (3, 3, 3),
                              float_type=np.float32 if a.float32 else np.float64)
    m = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, potential_temperature=300)), advection=bz.WENO(order=a.order))

    def theta(x, y, z):
        r = np.sqrt(x ** 2 + y ** 2 + (z - 3000.0) ** 2)
        return 300.0 * np.exp(1e-6 * z / 9.81) + 10.0 * np.maximum(0.0, 1.0 - r / 2e3)

    m.set(θ=theta)
    for _ in range(3):
        m.time_step(1.0)
    m.profile_reset(); m.profile_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        m.time_step(1.0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    prof = {k: v[0] / a.steps for k, v in sorted(m.profile().items(), key=lambda kv: -kv[1][0])}
    print(json.dumps({"grid": [N, N, N], "weno_order": a.order, "dtype": "f32" if a.float32 else "f64", "ms_per_step": ms,
                      "value": N ** 3 / (ms * 1e-3), "unit": "cells/s", "kernels_ms_per_step": prof}))


if __name__ == "__main__":
    main()
