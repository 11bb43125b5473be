import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import breeze_jl_amd as bz
from breeze_jl_amd.distributed import SlabAtmosphereModel
from torch.profiler import profile, ProfilerActivity
N = 512
EXT = ((-10e3, 10e3), (-10e3, 10e3), (0.0, 10e3))
g = bz.RectilinearGrid((N, N, N), x=EXT[0], y=EXT[1], z=EXT[2])
m = SlabAtmosphereModel(g, 0, 1, advection=bz.WENO(order=5), surface_pressure=101325, potential_temperature=300, device="cuda:0")
def bubble(x, y, z):
    r = np.sqrt(x**2 + y**2 + (z-3000.0)**2)
    return 300.0*np.exp(1e-6*z/9.81) + 10*np.maximum(0, 1-r/2000.0)
m.set(θ=bubble)
for _ in range(2): m.time_step(1.0)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3): m.time_step(1.0)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=70))
