#!/bin/bash
# A/B of environment variants on the headline bench: bash tools/gpu_env_ab.sh REPS "ENV.." "ENV.." ...   (X=0 = default)
set -u
export TMPDIR=/tmp
REPS=$1; shift
O=gpurun_out/env_ab; mkdir -p $O
for rep in $(seq $REPS); do
for envs in "$@"; do
  env $envs timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-compressible > $O/bench.json 2> $O/bench.err
  python - "$O/bench.json" "$envs" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    k=d['kernels_ms_per_step']
    f32=d.get('float32_variant'); f32=f32.get('ms_per_step',0) if isinstance(f32,dict) else 0
    print(f"[{sys.argv[2]}] {d['ms_per_step']:.2f} ms/step moist {d.get('moist_variant',{}).get('ms_per_step',0):.2f} f32 {f32:.2f} | " + " ".join(f"{n.replace('poisson_','p_').replace('_tendency+rk3+velocity','')}={v:.2f}" for n,v in k.items()))
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done; done
