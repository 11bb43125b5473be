#!/bin/sh
# builds tools/tendbench (standalone A/B harness; links all library sources statically)
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result $EXTRA \
  tools/tendbench.hip breeze.jl_amd/csrc/bz_context.hip breeze.jl_amd/csrc/bz_halo.hip breeze.jl_amd/csrc/bz_state.hip \
  breeze.jl_amd/csrc/bz_poisson.hip breeze.jl_amd/csrc/bz_tendency3.hip \
  breeze.jl_amd/csrc/bz_fused.hip breeze.jl_amd/csrc/bz_slab.hip breeze.jl_amd/csrc/bz_step.hip -lhipfft -o ${OUT:-tools/tendbench}
