#!/bin/bash
# Round 6: k_ac_forward2 with DPP wavefront shifts (BZ_AC_CFG bit 4)
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_dpp; mkdir -p $O
./tools/dpp_check
timeout 1500 python -m pytest tests/test_gpu_compressible.py -m gpu -x -q -k "round6 or substep_loop or time_steps" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
line() {
python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d.get('kernels_ms_per_step',{})
print('$1', round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items() if 'forward' in a or 'backward' in a})"
}
VARS=("BZ_AC_CFG=6 BZ_AC_BX=64" "BZ_AC_CFG=22 BZ_AC_BX=64" "BZ_AC_CFG=13 BZ_AC_BX=128" "BZ_AC_CFG=29 BZ_AC_BX=128" "BZ_AC_CFG=29 BZ_AC_BX=64" "BZ_AC_CFG=29 BZ_AC_BX=256" "BZ_AC_CFG=28 BZ_AC_BX=64")
for rep in 1 2 3; do
for v in "${VARS[@]}"; do
env $v timeout 300 python tools/bench_compressible.py --steps 4 --warmup 2 2>$O/err.log | tail -1 | line "[$v]" || tail -5 $O/err.log
done; done
for v in "BZ_AC_CFG=13 BZ_AC_BX=128" "BZ_AC_CFG=29 BZ_AC_BX=128"; do
env $v timeout 300 python tools/bench_compressible.py --steps 4 --warmup 2 --substep-float32 2>$O/err.log | tail -1 | line "[f32 storage $v]" || tail -5 $O/err.log
done
