#!/bin/bash
# A/B: XCD band order of the projection kernels in the Float32 build (lib/var32_bands32.so = -DBZ_STREAM_BANDS_F32=1)
export TMPDIR=/tmp
L=$PWD/breeze.jl_amd/lib
for rep in 1 2; do for name in base bands32; do
B=$L/libbreeze_hip_f32.so; [ $name = bands32 ] && B=$L/var32_bands32.so
BREEZE_HIP_F32_LIB=$B python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-compressible --no-moist-variant 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['float32']; kf=f['kernels_ms_per_step']
print('$name f32', round(f['ms_per_step'],2), 'project', round(kf['project_momentum']/2.9,3))"
done; done
