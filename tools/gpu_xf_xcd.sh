#!/bin/bash
# A/B of the XCD band order of the x transforms (lib/var_noxf.so, lib/var32_noxf.so = -DXF_XCD=0): Float64 and Float32 legs of the headline
export TMPDIR=/tmp
L=$PWD/breeze.jl_amd/lib
for rep in 1 2; do for name in base noxf; do
if [ $name = base ]; then A=$L/libbreeze_hip.so; B=$L/libbreeze_hip_f32.so; else A=$L/var_noxf.so; B=$L/var32_noxf.so; fi
BREEZE_HIP_LIB=$A BREEZE_HIP_F32_LIB=$B python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-compressible --no-moist-variant 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; f=d['float32']; kf=f['kernels_ms_per_step']
g=lambda k,n: round(k[n]/3,3)
print('$name f64', round(d['ms_per_step'],2), 'xfwd', g(k,'poisson_source_term+fft_x'), 'xinv', g(k,'poisson_fft_x_inverse'), '| f32', round(f['ms_per_step'],2), 'xfwd', g(kf,'poisson_source_term+fft_x'), 'xinv', g(kf,'poisson_fft_x_inverse'), 'tri', g(kf,'poisson_tridiagonal'))"
done; done
