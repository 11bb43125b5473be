#!/bin/bash
# SQ / TA counters of the order-9 kernels on the CBL case (Float32): bash tools/gpu_pmc_weno9.sh
set -u
export TMPDIR=/tmp
O=gpurun_out/w9pmc_r04; mkdir -p $O
B="python bench.py --workload cbl --cbl-order 9 --steps 3 --warmup 1 --no-cpu-baseline"
run() { name=$1; shift; timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $O/$name -- $B > $O/$name.log 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
run ta1 TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run ta2 GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
python tools/pmc_summary.py $O/pmc_summary.json $O/sq1 $O/sq2 $O/ta1 $O/ta2 > $O/pmc_summary.log 2>&1
find $O -name "*.csv" -size +8M -delete; find $O -name "*.db" -delete
python - $O/pmc_summary.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if 'tendency' in k and v.get('dispatches',0)>=3:
        print(k[:70]); print('   ', {a:(round(b) if isinstance(b,(int,float)) else b) for a,b in v.items()})
PY
tail -3 $O/ta1.log $O/ta2.log
