#!/bin/bash
# round profile set: bench line, rocprofv3 kernel stats of the same command, PMC passes (SQ, TCC, FETCH/WRITE)
#   bash tools/gpu_profile_round.sh TAG
set -u
export TMPDIR=/tmp
TAG=$1
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py > $O/bench_prof.json 2> $O/stats.log
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
bash tools/gpu_pmc.sh $O/pmc > $O/pmc.log 2>&1
bash tools/gpu_pmc2.sh $O/pmc2 > $O/pmc2.log 2>&1
find $O -name "*.csv" -size +4M -delete
head -c 400 $O/bench.json; echo; head -5 $O/kernel_stats.csv
