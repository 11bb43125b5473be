#!/bin/bash
# A/B of the XCD band order of the order-7 / 9 kernels (lib/var_nogx.so, lib/var32_nogx.so = -DGEN_XCD=0): CBL 512x512x256 Float32 WENO9, 256^3 Float64 bubble WENO9
export TMPDIR=/tmp
L=$PWD/breeze.jl_amd/lib
for rep in 1 2; do for name in base nogx; do
if [ $name = base ]; then A=$L/libbreeze_hip.so; B=$L/libbreeze_hip_f32.so; else A=$L/var_nogx.so; B=$L/var32_nogx.so; fi
BREEZE_HIP_LIB=$A BREEZE_HIP_F32_LIB=$B python bench.py --workload cbl --cbl-order 9 --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('$name cbl9', round(d['ms_per_step'],2), {a.split('_tend')[0]:round(b/3,3) for a,b in k.items() if 'tendency' in a and 'forcing' not in a})"
BREEZE_HIP_LIB=$A BREEZE_HIP_F32_LIB=$B python tools/bench_order.py --size 256 --order 9 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name bubble256 f64 weno9', round(d['ms_per_step'],2))"
done; done
