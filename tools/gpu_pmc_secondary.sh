#!/bin/bash
# PMC traffic passes (FETCH_SIZE, WRITE_SIZE; separate passes) of the secondary workloads at their own grids -> profiles/r06_pmc_traffic_<tag>.json
#   bash tools/gpu_pmc_secondary.sh OUTDIR
set -u
export TMPDIR=/tmp
O=$1; mkdir -p $O
one() {
  tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/${tag}_fetch -- python bench.py "$@" > $O/${tag}_fetch.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/${tag}_write -- python bench.py "$@" > $O/${tag}_write.log 2>&1
  python tools/pmc_summary.py $O/${tag}_summary.json $O/${tag}_fetch $O/${tag}_write > /dev/null 2>&1
  python tools/pmc_to_traffic.py $O/${tag}_summary.json > $O/pmc_traffic_${tag}.json
  python - $O/pmc_traffic_${tag}.json $tag <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for sec in ("per_kernel_group","per_kernel_group_float32"):
    for g,v in d.get(sec,{}).items(): print(sys.argv[2], sec[-7:], g, round(v["hbm_bytes_per_launch"]/1e9,3), "GB")
PY
}
one cbl_weno5 --workload cbl --cbl-order 5 --steps 3 --warmup 1
one cbl_weno9 --workload cbl --cbl-order 9 --steps 3 --warmup 1
one config4 --workload config4 --steps 2 --warmup 1
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
