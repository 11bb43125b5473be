"""Print per-step deviation HIP vs oracle for the parity-test configurations (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import breeze_jl_amd as bz
from oracle import oracle as orc
from helpers import PROG, bubble_theta, make_pair

for size, dt in (((32, 20, 16), 2.0), ((64, 8, 32), 1.0)):
    om, hm = make_pair(orc, bz, size)
    th = bubble_theta(300.0, om.constants.g)
    om.set(theta=th, u=3.0, v=-2.0)
    hm.set(θ=th, u=3.0, v=-2.0)
    for step in range(4):
        om.time_step(dt); hm.time_step(dt); hm.synchronize()
        out = []
        for n, k in PROG.items():
            got = hm.prognostic_fields()[k].interior_cpu()
            want = om.grid.interior(getattr(om, n), zface=(n == "rw"))
            out.append(f"{n}:{np.max(np.abs(got-want)):.2e}/{np.max(np.abs(want)):.2e}")
        print(size, dt, step, " ".join(out), flush=True)
