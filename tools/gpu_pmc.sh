#!/bin/bash
# PMC passes over a short bench run (one pass per counter group; --kernel-trace only, as gpurun requires).
#   bash tools/gpu_pmc.sh OUTDIR [bench args...]
set -u
export TMPDIR=/tmp
O=$1; shift
mkdir -p $O
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-compressible --no-float32 $*"
run() { name=$1; shift; timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $O/$name -- $B > $O/$name.log 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
python tools/pmc_summary.py $O/pmc_summary.json $O/sq1 $O/sq2 $O/tcc $O/fetch $O/write $O/grbm > $O/pmc_summary.log 2>&1
find $O -name "*.csv" -size +8M -delete
find $O -name "*.db" -delete
cat $O/pmc_summary.log
