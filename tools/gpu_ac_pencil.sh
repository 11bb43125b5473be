#!/bin/bash
# A/B of the row-group / z-pencil block order of the acoustic stage kernels (lib/var_nopencil.so = -DAC_PENCIL=0)
export TMPDIR=/tmp
L=$PWD/breeze.jl_amd/lib
for rep in 1 2 3; do for name in base nopencil; do
A=$L/libbreeze_hip.so; [ $name = nopencil ] && A=$L/var_nopencil.so
BREEZE_HIP_LIB=$A python tools/bench_compressible.py --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('$name', round(d['ms_per_step'],1), 'init', round(k['acoustic_stage_init']/3,3), 'end', round(k['acoustic_stage_end+update_state'],3), 'end+lin', round(k['acoustic_stage_end+update_state+linearization']/2,3), 'bwd', round(k['acoustic_column_backward']/33,3))"
done; done
