#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_dryavg; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_compressible.py tests/test_distributed.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
line() {
python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d.get('kernels_ms_per_step',{})
print('$1', round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items() if 'acoustic' in a})"
}
for rep in 1 2; do
for v in "X=0" "BZ_NO_DRY_SHORTCUT=1"; do
env $v timeout 300 python tools/bench_compressible.py --steps 4 --warmup 2 2>$O/err.log | tail -1 | line "[$v]" || tail -5 $O/err.log
done; done
for v in "X=0" "BZ_NO_DRY_SHORTCUT=1"; do
env $v timeout 300 python tools/bench_compressible.py --steps 4 --warmup 2 --substep-float32 2>$O/err.log | tail -1 | line "[f32 storage $v]" || tail -5 $O/err.log
done
