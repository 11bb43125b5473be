#!/bin/bash
# Round 6: the rewritten forward acoustic sweep (k_ac_forward2) — parity, then A/B of the variants on the 512x512x256 compressible step.
#   bash tools/gpu_r6_fwd2.sh [TAG]
set -u
export TMPDIR=/tmp
TAG=${1:-r6_fwd2}
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_compressible.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
line() {
python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d.get('kernels_ms_per_step',{})
print('$1', round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items() if 'acoustic' in a})"
}
for rep in 1 2; do
for v in "BZ_AC_FWD2=0" "BZ_AC_FWD2=1 BZ_AC_PFOLD=0 BZ_AC_MW=2" "BZ_AC_FWD2=1 BZ_AC_PFOLD=0 BZ_AC_MW=3" "BZ_AC_FWD2=1 BZ_AC_PFOLD=0 BZ_AC_MW=4" "BZ_AC_PFOLD=1 BZ_AC_MW=2" "BZ_AC_PFOLD=1 BZ_AC_MW=3" "BZ_AC_PFOLD=1 BZ_AC_MW=4"; do
env $v timeout 300 python tools/bench_compressible.py --steps 4 --warmup 2 2>$O/err.log | tail -1 | line "[$v]" || tail -5 $O/err.log
done; done
for v in "BZ_AC_FWD2=0" "BZ_AC_PFOLD=1 BZ_AC_MW=3" "BZ_AC_PFOLD=1 BZ_AC_MW=4"; do
env $v timeout 300 python tools/bench_compressible.py --steps 4 --warmup 2 --substep-float32 2>$O/err.log | tail -1 | line "[f32 storage $v]" || tail -5 $O/err.log
done
