#!/bin/bash
# buffer rotation of the compressible whole-step seam (no store_initial_state! copies, stage epilogues of stages 1 and 3 out of place) against the
# copying form: the whole GPU suite, then alternating fresh processes on one box
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_rotate; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
line() {
python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d.get('kernels_ms_per_step',{})
sub=sum(v for a,v in k.items() if 'forward' in a or 'backward' in a)
print('$1', round(d['ms_per_step'],2), 'outside the substep loop', round(d['ms_per_step']-sub,2), {a:round(b,2) for a,b in k.items() if 'stage' in a or 'recover' in a or 'store' in a})"
}
for rep in 1 2 3; do
for v in "BZ_AC_ROTATE=1" "BZ_AC_ROTATE=0"; do
env $v timeout 300 python tools/bench_compressible.py --steps 4 --warmup 2 2>$O/err.log | tail -1 | line "[dry $v]" || tail -5 $O/err.log
env $v timeout 300 python tools/bench_compressible.py --steps 4 --warmup 2 --moist 2>$O/err.log | tail -1 | line "[moist $v]" || tail -5 $O/err.log
done; done
for v in "BZ_AC_ROTATE=1" "BZ_AC_ROTATE=0" "BZ_AC_ROTATE=1" "BZ_AC_ROTATE=0"; do
env $v timeout 300 python bench.py --workload config4 --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[config4 $v]', round(d['ms_per_step'],2))"
done
