export TMPDIR=/tmp
for rep in 1 2 3; do for v in 0 1; do
BZ_AC_XCD=$v python tools/bench_compressible.py --steps 4 --warmup 2 --substep-float32 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('f32sub xcd=$v', round(d['ms_per_step'],2), round(k['acoustic_horizontal+column_forward'],2), round(k['acoustic_column_backward'],2))"
done; done
