#!/bin/bash
# Round 6: block shape of k_ac_forward2 and a PMC pass of the compressible step.   bash tools/gpu_r6_shapes.sh
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_shapes; mkdir -p $O
line() {
python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d.get('kernels_ms_per_step',{})
print('$1', round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items() if 'forward' in a or 'backward' in a})"
}
for rep in 1 2; do
for v in "BZ_AC_BX=64 BZ_AC_MW=2" "BZ_AC_BX=128 BZ_AC_MW=2" "BZ_AC_BX=256 BZ_AC_MW=2" "BZ_AC_BX=64 BZ_AC_MW=3" "BZ_AC_BX=128 BZ_AC_MW=3" "BZ_AC_BX=256 BZ_AC_MW=3" "BZ_AC_BX=256 BZ_AC_MW=3 BZ_AC_XCD=0"; do
env $v timeout 300 python tools/bench_compressible.py --steps 4 --warmup 2 2>$O/err.log | tail -1 | line "[$v]" || tail -5 $O/err.log
done; done
BZ_AC_MW=2 bash tools/gpu_pmc_compressible.sh $O/pmc_mw2 > $O/pmc_mw2.log 2>&1
BZ_AC_MW=3 bash tools/gpu_pmc_compressible.sh $O/pmc_mw3 > $O/pmc_mw3.log 2>&1
python - <<'PY'
import json
for t in ("mw2","mw3"):
    try:
        d=json.load(open(f"gpurun_out/r6_shapes/pmc_{t}/pmc_summary.json"))
    except Exception as e:
        print(t, e); continue
    for k,v in d.items():
        if "forward2" in k or "backward" in k or "stage_init" in k:
            print(t, k[:60], {c: (round(x,1) if isinstance(x,float) else x) for c,x in v.items() if c in ("FETCH_SIZE","WRITE_SIZE","SQ_WAVES","SQ_WAVE_CYCLES","SQ_WAIT_ANY","SQ_ACTIVE_INST_VALU","SQ_BUSY_CYCLES","vgpr","dispatches","GRBM_GUI_ACTIVE")})
PY
