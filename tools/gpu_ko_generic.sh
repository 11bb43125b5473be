#!/bin/bash
# timing knock-outs of the WENO9 generic kernels on the CBL case (Float32): run on the GPU box after tools/gpu_variant.sh build32 ... (lib/var32_<name>.so)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for name in base "$@"; do
  lib=$PWD/breeze.jl_amd/lib/var32_$name.so; [ $name = base ] && lib=$PWD/breeze.jl_amd/lib/libbreeze_hip_f32.so
  BREEZE_HIP_F32_LIB=$lib timeout 300 python bench.py --workload cbl --cbl-order 9 --steps 20 --warmup 3 --no-cpu-baseline > /tmp/k_$name.json 2>/tmp/k_$name.err
  python - /tmp/k_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]); k=d['kernels_ms_per_step']
    print(f"[{sys.argv[2]:>6s}] {d['ms_per_step']:.2f} ms/step " + " ".join(f"{a.replace('_tendency','').replace('+rk3','')}={v/3:.3f}" for a,v in sorted(k.items()) if v > 0.3))
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
done
