#!/bin/bash
# SQ-only PMC passes (stall attribution of the tendency kernels): bash tools/gpu_pmc_sq.sh OUTDIR [bench args]
set -u
export TMPDIR=/tmp
O=$1; shift
mkdir -p $O
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-compressible --no-float32 $*"
run() { name=$1; shift; timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $O/$name -- $B > $O/$name.log 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
run sq3 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_VALU
python tools/pmc_summary.py $O/pmc_summary.json $O/sq1 $O/sq2 $O/sq3 > $O/pmc_summary.log 2>&1
find $O -name "*.csv" -size +8M -delete
find $O -name "*.db" -delete
cat $O/pmc_summary.log
