#!/bin/bash
# round-4 first contact: multi-step seam tests + core parity, then bench A/B (multi-step seam vs single steps)
set -u
export TMPDIR=/tmp
O=gpurun_out/r4_first; mkdir -p $O
timeout 1500 python -m pytest tests/test_multi_step.py tests/test_gpu_parity.py tests/test_comm.py tests/test_cbl.py -m gpu -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
for mode in "" "--single-steps"; do
  timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-compressible $mode > $O/bench_${mode:-multi}.json 2> $O/bench_${mode:-multi}.err
  python - "$O/bench_${mode:-multi}.json" "${mode:-multi}" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    def line(tag, d):
        k=d['kernels_ms_per_step']; n=d.get('kernel_launches_per_step') or {}
        print(f"[{sys.argv[2]}] {tag} {d['ms_per_step']:.2f} ms/step | " + " ".join(f"{a.split('_tend')[0].replace('poisson_','p_')}={v:.3f}" for a,v in k.items()))
    line('f64', d)
    if 'float32' in d: line('f32', d['float32'])
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
