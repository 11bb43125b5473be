#!/bin/bash
# A/B bench line(s) without the pytest leg: bash tools/gpu_ab.sh TAG "ENV.." ...   (prints per-launch ms of every kernel group, f64 and f32)
set -u
export TMPDIR=/tmp
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
idx=0
for envs in "$@"; do
  idx=$((idx+1))
  env $envs timeout 400 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-compressible ${BENCH_ARGS:-} > $O/bench_$idx.json 2> $O/bench_$idx.err
  python - "$O/bench_$idx.json" "$envs" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    def line(tag, d):
        k=d['kernels_ms_per_step']; n=d.get('kernel_launches_per_step') or {}
        print(f"[{sys.argv[2]}] {tag} {d['ms_per_step']:.2f} ms/step | " + " ".join(f"{a.split('_tend')[0].replace('poisson_','p_')}={v/(n.get(a,3) if a in n else (1 if 'diagnose' in a else 2 if a=='project_momentum' else 3)):.3f}" for a,v in k.items()))
    line('f64', d)
    if 'float32' in d: line('f32', d['float32'])
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
