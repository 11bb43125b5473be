#!/bin/bash
# short Float64 bench lines for the environments given as arguments ("" = default)
#   bash tools/gpu_env_bench.sh TAG "ENV1=.. ENV2=.." "ENV3=.." ...
set -u
export TMPDIR=/tmp
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
idx=0
for envs in "$@"; do
  idx=$((idx+1))
  env BZ_X=0 $envs timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-compressible --no-float32 > $O/bench_$idx.json 2> $O/bench_$idx.err
  python - "$O/bench_$idx.json" "$envs" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    k=d['kernels_ms_per_step']; l=d.get('kernel_launches_per_step',{})
    print(f"[{sys.argv[2]}] {d['ms_per_step']:.2f} ms/step finite={d['finite']} | " + " ".join(f"{n.replace('_tendency','').replace('poisson_','p_').replace('_momentum','').replace('+rk3','').replace('+velocity','').replace('+thermo','')}={v/max(l.get(n,1),1):.2f}" for n,v in k.items()))
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
