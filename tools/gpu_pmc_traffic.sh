#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes only (L2-fabric bytes per launch of every kernel): bash tools/gpu_pmc_traffic.sh OUTDIR [bench args]
set -u
export TMPDIR=/tmp
O=$1; shift
mkdir -p $O
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-compressible $*"      # the Float32 leg rides along (its kernels carry _f32 names)
run() { name=$1; shift; timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $O/$name -- $B > $O/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
python tools/pmc_summary.py $O/pmc_summary.json $O/fetch $O/write > $O/pmc_summary.log 2>&1
python tools/pmc_to_traffic.py $O/pmc_summary.json > $O/pmc_traffic.json
find $O -name "*.csv" -size +8M -delete
find $O -name "*.db" -delete
python - $O/pmc_traffic.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))["per_kernel_group"]
for k,v in sorted(d.items()): print(f"{k:36s} read {v['read_bytes']/1e9:6.2f} GB  write {v['write_bytes']/1e9:6.2f} GB  total {v['hbm_bytes_per_launch']/1e9:6.2f} GB   ({v['kernel']})")
PY
