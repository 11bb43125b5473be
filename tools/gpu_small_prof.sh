#!/bin/bash
# kernel-trace of the launch-bound small cases: where the 0.7-0.9 ms floor goes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/small
for c in "bubble 64^3" "configs[0]"; do
  tag=$(echo "$c" | tr -dc 'a-z0-9')
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/small/$tag -- python $R/tools/small_grid_latency.py --steps 100 --only "$c" > $R/gpurun_out/small/$tag.log 2>&1
  f=$(find $R/gpurun_out/small/$tag -name '*kernel_stats.csv' | head -1)
  cp "$f" $R/gpurun_out/small/${tag}_kernel_stats.csv
  find $R/gpurun_out/small/$tag -name '*kernel_trace.csv' -exec cp {} $R/gpurun_out/small/${tag}_kernel_trace.csv \;
  rm -rf $R/gpurun_out/small/$tag
  tail -2 $R/gpurun_out/small/$tag.log
  head -25 $R/gpurun_out/small/${tag}_kernel_stats.csv | cut -c1-160
done
