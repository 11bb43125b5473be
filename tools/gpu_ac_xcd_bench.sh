#!/bin/bash
# A/B of BZ_AC_XCD inside the default bench command (the compressible leg runs after the anelastic legs, on a warm GPU)
export TMPDIR=/tmp
for rep in 1 2; do for v in 0 1; do
BZ_AC_XCD=$v python bench.py --no-cpu-baseline --no-float32 --no-moist-variant 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); sm=d['second_milestone']; k=sm['kernels_ms_per_step']
print('xcd=$v headline', round(d['ms_per_step'],2), 'cmp', round(sm['ms_per_step'],1), 'fwd/launch', round(sm['roofline']['avg_launch_ms'],3), 'bwd', round(k['acoustic_column_backward'],1), 'f32sub', round(sm['substep_floattype_float32']['ms_per_step'],1))"
done; done
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | head -6
