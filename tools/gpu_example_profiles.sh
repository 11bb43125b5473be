#!/bin/bash
# rocprofv3 kernel statistics of the example-configuration bench lines (BOMEX and supercell shapes, WENO9 + Float32 as the examples run them)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exprof; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bomex -- python $R/tools/bench_bomex.py --order 9 --float32 > $O/bomex.json 2> $O/bomex.log
cp $(find $O/bomex -name "*kernel_stats.csv" | head -1) $O/bomex_weno9_f32_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4 -- python $R/bench.py --workload config4 --config4-order 9 --config4-float32 --steps 10 --warmup 2 > $O/c4.json 2> $O/c4.log
cp $(find $O/c4 -name "*kernel_stats.csv" | head -1) $O/config4_weno9_f32_kernel_stats.csv
rm -rf $O/bomex $O/c4
head -8 $O/bomex_weno9_f32_kernel_stats.csv | cut -c1-150; head -8 $O/config4_weno9_f32_kernel_stats.csv | cut -c1-150
