"""Compare the lean whole-step seam with the generation-4 seam field by field (max abs / relative difference)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import breeze_jl_amd as bz
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import bubble_theta
size = (32, 20, 16)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
runs = []
for lean in (True, False):
    if lean: os.environ.pop("BZ_NO_LEAN", None)
    else: os.environ["BZ_NO_LEAN"] = "1"
    grid = bz.RectilinearGrid(size, x=(-10e3, 10e3), y=(-10e3, 10e3), z=(0, 10e3))
    m = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, potential_temperature=300)), advection=bz.WENO())
    m.set(θ=bubble_theta(300.0, 9.81), u=3.0, v=-2.0)
    for _ in range(steps): m.time_step(1.5)
    m.synchronize(); runs.append(m)
a, b = runs
F = {"ρu": lambda m: m.momentum["ρu"], "ρv": lambda m: m.momentum["ρv"], "ρw": lambda m: m.momentum["ρw"], "ρθ": lambda m: m.potential_temperature_density,
     "ρq": lambda m: m.moisture_density, "T": lambda m: m.temperature, "u": lambda m: m.velocities["u"]}
for n, g in F.items():
    x, y = g(a).interior_cpu(), g(b).interior_cpu()
    d = np.abs(x - y)
    print(n, "max abs diff", d.max(), "rel", d.max() / max(np.abs(y).max(), 1e-30), "n differing", int((d > 0).sum()), "of", d.size)
