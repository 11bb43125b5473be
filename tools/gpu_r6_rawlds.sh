#!/bin/bash
# BZ5_RAW_LDS resolved with a median of fresh processes (moist 512^3 step; alternating)
set -u
export TMPDIR=/tmp
L=$PWD/breeze.jl_amd/lib
for rep in 1 2 3 4 5; do
for v in "X=0" "BREEZE_HIP_LIB=$L/libbreeze_hip_var_rawlds.so"; do
env $v timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-compressible --no-float32 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); m=d['moist_variant']
print('${v##*/}', 'dry', round(d['ms_per_step'],3), 'moist', round(m['ms_per_step'],3), 'scalar pair moist', round(m['kernels_ms_per_step'].get('scalar_tendencies+rk3+thermo',0),3))"
done; done
