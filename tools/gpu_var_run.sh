#!/bin/bash
# run on the GPU box: bash tools/gpu_var_run.sh TAG name1 name2 ...   — one short headline bench per lib/var_<name>.so (and the default library first),
# per-launch ms of the four tendency kernels and the step; BENCH_ARGS adds bench.py flags (e.g. --moist)
set -u
export TMPDIR=/tmp
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for name in base "$@"; do
  lib=$PWD/breeze.jl_amd/lib/var_$name.so; [ $name = base ] && lib=$PWD/breeze.jl_amd/lib/libbreeze_hip.so
  BREEZE_HIP_LIB=$lib timeout 300 python bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-compressible --no-float32 ${BENCH_ARGS:-} > $O/b_$name.json 2> $O/b_$name.err
  python - $O/b_$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d['kernels_ms_per_step']
    mv=d.get('moist_variant') or {}
    mk=mv.get('kernels_ms_per_step') or {}
    s=f"[{sys.argv[2]:>10s}] {d['ms_per_step']:.2f} ms/step scalar={k['scalar_tendencies+rk3+thermo']/3:.3f} u={k['x_momentum_tendency+rk3+velocity']/3:.3f} v={k['y_momentum_tendency+rk3+velocity']/3:.3f} w={k['z_momentum_tendency+rk3+velocity']/3:.3f}"
    if mk: s+=f" | moist {mv.get('ms_per_step',0):.2f} scalar={mk['scalar_tendencies+rk3+thermo']/3:.3f} w={mk['z_momentum_tendency+rk3+velocity']/3:.3f}"
    print(s)
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
done
