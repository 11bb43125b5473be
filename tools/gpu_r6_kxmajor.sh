#!/bin/bash
# Round 6: kx-major half spectrum + chunked middle of the Poisson solve: parity, then A/B on the headline.
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_kxmajor; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_float32.py tests/test_multi_step.py tests/test_dry_shortcut.py -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for rep in 1 2; do
for envs in "BZ_POISSON_KXMAJOR=0" "BZ_POISSON_KX_CHUNK_MB=200 BZ_POISSON_KX_PAD=0" "BZ_POISSON_KX_CHUNK_MB=200 BZ_POISSON_KX_PAD=1" "BZ_POISSON_KX_CHUNK_MB=256 BZ_POISSON_KX_PAD=1" "BZ_POISSON_KX_CHUNK_MB=360 BZ_POISSON_KX_PAD=1" "BZ_POISSON_KX_CHUNK_MB=200 BZ_POISSON_KX_PAD=3"; do
  env $envs timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-compressible > $O/bench.json 2> $O/bench.err
  python - "$O/bench.json" "$envs" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    k=d['kernels_ms_per_step']
    print(f"[{sys.argv[2]}] {d['ms_per_step']:.2f} ms/step moist {d.get('moist_variant',{}).get('ms_per_step',0):.2f} f32 {d.get('float32_variant',{}).get('ms_per_step',0) if isinstance(d.get('float32_variant'),dict) else 0} | " + " ".join(f"{n.replace('poisson_','p_')}={v:.2f}" for n,v in k.items() if 'poisson' in n))
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done; done
