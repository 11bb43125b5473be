#!/bin/bash
# A/B of the edge-flux batch length of the lean kernels (lib/var_eb16.so ... = -DBZ5_EB=16 | 8), Float64 and Float32 legs
export TMPDIR=/tmp
L=$PWD/breeze.jl_amd/lib
for rep in 1 2; do for name in base eb16 eb8; do
if [ $name = base ]; then A=$L/libbreeze_hip.so; B=$L/libbreeze_hip_f32.so; else A=$L/var_$name.so; B=$L/var32_$name.so; fi
BREEZE_HIP_LIB=$A BREEZE_HIP_F32_LIB=$B python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-compressible --no-moist-variant 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; f=d['float32']; kf=f['kernels_ms_per_step']
g=lambda k: ' '.join('%.3f'%(k[n]/3) for n in ('x_momentum_tendency+rk3+velocity','y_momentum_tendency+rk3+velocity','z_momentum_tendency+rk3+velocity','scalar_tendencies+rk3+thermo'))
print('$name f64', round(d['ms_per_step'],2), g(k), '| f32', round(f['ms_per_step'],2), g(kf))"
done; done
