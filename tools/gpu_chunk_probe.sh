#!/bin/bash
# chunk-size probe of the L3-resident Poisson pipeline
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-chunk_probe}; mkdir -p $O
BZ_POISSON_CHUNK=16 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_chunk16.log 2>&1; echo "chunk16 parity: $(tail -1 $O/pytest_chunk16.log)"
for envs in "BZ_X=0" "BZ_POISSON_CHUNK=8" "BZ_POISSON_CHUNK=16" "BZ_POISSON_CHUNK=32" "BZ_POISSON_CHUNK=64" "BZ_POISSON_CHUNK=128"; do
  env $envs timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-compressible --no-float32 > $O/bench_$envs.json 2> $O/bench_$envs.err
  python - "$O/bench_$envs.json" "$envs" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    k=d['kernels_ms_per_step']
    print(f"[{sys.argv[2]}] {d['ms_per_step']:.2f} ms/step finite={d['finite']} | " + " ".join(f"{n.replace('_tendency','').replace('poisson_','p_').replace('_momentum','')}={v:.2f}" for n,v in k.items()))
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
