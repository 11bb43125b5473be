#!/bin/bash
# bench lines only for env variants:  bash tools/gpu_bench_only.sh TAG "ENV.." "ENV.." ...
set -u
export TMPDIR=/tmp
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
idx=0
for envs in "$@"; do
  idx=$((idx+1))
  env $envs timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-compressible > $O/bench_$idx.json 2> $O/bench_$idx.err
  python - "$O/bench_$idx.json" "$envs" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    k=d['kernels_ms_per_step']
    print(f"[{sys.argv[2]}] {d['ms_per_step']:.2f} ms/step finite={d['finite']} | " + " ".join(f"{n.split('_tend')[0].replace('poisson_','p_')}={v/3:.2f}" for n,v in k.items()))
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
