#!/usr/bin/env python
"""tools/bench_slab_shapes.py — kernel time of one rank of the STRONG 512^3 split at W = 1, 2, 4, 8: the library-owned slab
step on 512 x (512 / W) x 512 slabs, world 1 on one GPU (y halos = the slab's own periodic rows; no message leaves the rank), so the numbers are the
rank-local kernels at the slab shapes the 8-GPU launch will hand out (64-row slabs at W = 8: four 16-row tiles of the y-momentum kernel).
One JSON line per shape: ms per step, cells per second of the rank, kernel groups."""
import json
import os
import sys
import time
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    import torch
    import breeze_jl_amd as bz
    from breeze_jl_amd import distributed as bz_dist
    steps, warm = 10, 3
    for W in (1, 2, 4, 8):
        Ny = 512 // W
        G = bz.RectilinearGrid((512, Ny, 512), x=(-10e3, 10e3), y=(-10e3 / W, 10e3 / W), z=(0.0, 10e3))

        def bubble(x, y, z):
            r = np.sqrt(x ** 2 + y ** 2 + (z - 3000.0) ** 2)
            return 300.0 * np.exp(1e-6 * z / 9.81) + 10.0 * np.maximum(0.0, 1.0 - r / 2000.0)

        m = bz_dist.SlabAtmosphereModel(G, 0, 1, advection=bz.WENO(order=5), surface_pressure=101325, potential_temperature=300,
                                        device="cuda:0", transport="local:" + uuid.uuid4().hex)
        m.set(θ=bubble)
        for _ in range(warm):
            m.time_step(1.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.time_steps(1.0, steps)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        m.profile_enable(True)
        m.profile_reset()
        m.time_steps(1.0, 3)
        torch.cuda.synchronize()
        prof = {k: round(v[0] / 3, 3) for k, v in m.profile().items()}
        m.profile_enable(False)
        print(json.dumps({"world": W, "slab": [512, Ny, 512], "ms_per_step": round(ms, 3), "rank_Gcells_per_s": round(512 * Ny * 512 / ms / 1e6, 3),
                          "job_Gcells_per_s_if_no_comm": round(512 ** 3 / ms / 1e6, 3), "kernels_ms_per_step": prof}))
        del m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
