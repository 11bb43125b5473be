import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import breeze_jl_amd as bz
from oracle import oracle as orc
from helpers import PROG, make_pair, push_state, randomize
for size in ((32, 20, 16), (16, 8, 8)):
    om, hm = make_pair(orc, bz, size)
    randomize(om, seed=11); om.compute_tendencies()
    push_state(om, hm, names=("ru", "rv", "rw", "rtheta", "rq", "u", "v", "w", "theta", "q", "T"))
    for k in hm.G.values(): k.parent.zero_()
    bz.compute_tendencies_(hm); hm.synchronize()
    for n, k in PROG.items():
        zf = n == "rw"
        want = om.grid.interior(om.G[n], zface=zf); got = hm.G[k].interior_cpu()
        if zf: want, got = want[1:-1], got[1:-1]
        bad = ~np.isfinite(got)
        err = np.nanmax(np.abs(got - want)) / np.max(np.abs(want) + 1e-300)
        print(size, n, "nan count", bad.sum(), "err(non-nan)", err, "first nan idx", np.argwhere(bad)[:4].tolist())
