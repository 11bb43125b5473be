export TMPDIR=/tmp
run() { ( cd $1; python bench.py --no-cpu-baseline --no-float32 --no-moist-variant 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); sm=d['second_milestone']; k=sm['kernels_ms_per_step']
print('$2 headline', round(d['ms_per_step'],2), 'cmp', round(sm['ms_per_step'],1), 'fwd/launch', round(sm['roofline']['avg_launch_ms'],3), 'bwd/launch', round(k['acoustic_column_backward']/33,3), 'stage_init', round(k['acoustic_stage_init'],2), 'f32sub', round(sm['substep_floattype_float32']['ms_per_step'],1))" ); }
for rep in 1 2; do run _old_build old; run . new; done
