#!/usr/bin/env python
"""tools/f32_line.py [N] — the Float32 leg of bench.py alone (A/B runs of the Float32 twin: BREEZE_HIP_F32_LIB=...)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import breeze_jl_amd as bz
r = bench.float32_run(bz, "cuda:0", int(sys.argv[1]) if len(sys.argv) > 1 else 512, steps=10, warmup=3)
k, n = r["kernels_ms_per_step"], r["kernel_launches_per_step"]
print(f"f32 {r['ms_per_step']:.2f} ms/step | " + " ".join(f"{a.split('_tend')[0].replace('poisson_', 'p_')}={v / n[a]:.3f}" for a, v in k.items()))
