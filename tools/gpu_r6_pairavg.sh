#!/bin/bash
# <u>, <v> accumulated in pairs of substeps (AcParams::acc_mode) against every substep: tests, then alternating fresh processes on one box
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_pairavg; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_compressible.py tests/test_distributed.py tests/test_lateral_boundaries.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
line() {
python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d.get('kernels_ms_per_step',{})
print('$1', round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items() if 'forward' in a or 'backward' in a})"
}
for rep in 1 2 3; do
for v in "BZ_AC_PAIR_AVG=1" "BZ_AC_PAIR_AVG=0"; do
env $v timeout 300 python tools/bench_compressible.py --steps 4 --warmup 2 2>$O/err.log | tail -1 | line "[dry $v]" || tail -5 $O/err.log
env $v timeout 300 python tools/bench_compressible.py --steps 4 --warmup 2 --moist 2>$O/err.log | tail -1 | line "[moist $v]" || tail -5 $O/err.log
done; done
