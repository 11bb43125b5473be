import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import breeze_jl_amd as bz
from breeze_jl_amd.distributed import SlabAtmosphereModel
def bubble(x, y, z):
    r = np.sqrt(x**2 + y**2 + (z-3000.0)**2)
    return 300.0*np.exp(1e-6*z/9.81) + 10*np.maximum(0, 1-r/2000.0)
for cls in ("whole", "slab"):
    g = bz.RectilinearGrid((1024, 128, 512), x=(-20e3, 20e3), y=(-2.5e3, 2.5e3), z=(0.0, 10e3))
    if cls == "whole":
        ref = bz.ReferenceState(g, surface_pressure=101325, potential_temperature=300)
        m = bz.AtmosphereModel(g, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5))
    else:
        m = SlabAtmosphereModel(g, 0, 1, advection=bz.WENO(order=5), surface_pressure=101325, potential_temperature=300, device="cuda:0")
    m.set(θ=bubble)
    for _ in range(2): m.time_step(1.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): m.time_step(1.0)
    torch.cuda.synchronize(); dt = (time.perf_counter()-t0)/5
    w = m.momentum["ρw"].parent
    print(cls, "1024x128x512 ms/step", round(dt*1e3, 2), "Gcells/s", round(1024*128*512/dt/1e9, 3), "finite", bool(torch.isfinite(w).all()))
    del m; torch.cuda.empty_cache()
