#!/bin/bash
# PMC passes + rocprofv3 kernel stats over the compressible split-explicit bench (tools/bench_compressible.py)
set -u
export TMPDIR=/tmp
O=$1; shift
mkdir -p $O
B="python tools/bench_compressible.py --steps 1 $*"
timeout 600 python tools/bench_compressible.py --steps 3 $* > $O/bench.json 2> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $O/$name -- $B > $O/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run grbm GRBM_GUI_ACTIVE
python tools/pmc_summary.py $O/pmc_summary.json $O/fetch $O/write $O/sq1 $O/grbm > $O/pmc_summary.log 2>&1
find $O -name "*.csv" -size +4M -delete
find $O -name "*.db" -delete
cat $O/pmc_summary.log; head -c 600 $O/bench.json
