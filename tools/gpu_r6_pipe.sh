#!/bin/bash
# Round 6: k_ac_forward2 with the next level's loads in flight (BZ_AC_CFG bit 3)
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_pipe; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_compressible.py -m gpu -x -q -k "round6 or substep_loop or time_steps" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
line() {
python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d.get('kernels_ms_per_step',{})
print('$1', round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items() if 'forward' in a or 'backward' in a})"
}
VARS=("BZ_AC_CFG=6 BZ_AC_BX=64" "BZ_AC_CFG=5 BZ_AC_BX=128" "BZ_AC_CFG=13 BZ_AC_BX=128" "BZ_AC_CFG=13 BZ_AC_BX=64" "BZ_AC_CFG=12 BZ_AC_BX=64" "BZ_AC_CFG=8 BZ_AC_BX=64" "BZ_AC_CFG=13 BZ_AC_BX=256")
for rep in 1 2 3; do
for v in "${VARS[@]}"; do
env $v timeout 300 python tools/bench_compressible.py --steps 4 --warmup 2 2>$O/err.log | tail -1 | line "[$v]" || tail -5 $O/err.log
done; done
i=0
for v in "BZ_AC_CFG=13 BZ_AC_BX=128" "BZ_AC_CFG=12 BZ_AC_BX=64"; do
i=$((i+1))
env $v timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/f$i -- python tools/bench_compressible.py --steps 1 > $O/f$i.log 2>&1
python tools/pmc_summary.py $O/f$i.json $O/f$i > /dev/null 2>&1
python - $O/f$i.json "$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if 'forward2<false, true' in k and 'FETCH_SIZE' in v: print(sys.argv[2], round(v['FETCH_SIZE']*2*1024/1e9,2), 'GB fetched')
PY
done
find $O -name "*.csv" -size +4M -delete; find $O -name "*.db" -delete
