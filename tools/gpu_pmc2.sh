#!/bin/bash
# memory-path PMC passes (TLB, L1->L2 latency, TA) over a short bench run:  bash tools/gpu_pmc2.sh OUTDIR [env assignments...]
set -u
export TMPDIR=/tmp
O=$1; shift
mkdir -p $O
B="env $* python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-compressible --no-float32"
run() { name=$1; shift; timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $O/$name -- $B > $O/$name.log 2>&1; }
run tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
run lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum
run stall TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
run sq3 SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_SALU
python tools/pmc_summary.py $O/pmc_summary.json $O/tlb $O/lat $O/stall $O/ta $O/sq3 > $O/pmc_summary.log 2>&1
find $O -name "*.csv" -size +8M -delete
find $O -name "*.db" -delete
cat $O/pmc_summary.log
