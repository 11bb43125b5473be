#!/bin/bash
# timing knock-outs of the scalar-pair kernel: builds lib/ko_<mask>.so HERE (cross-compile), then on the GPU box runs bench per library
#   build:  bash tools/gpu_knockout.sh build 2 4 8 ...      run (on the box): bash tools/gpu_knockout.sh run 2 4 8 ...
set -u
mode=$1; shift
cd "$(dirname "$0")/.."
C=breeze.jl_amd/csrc
if [ "$mode" = build ]; then
  for m in "$@"; do
    ( extra=""; kk=$m; case $m in stub*) extra="-DBZ_WENO_STUB"; kk=${m#stub};; esac
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value -Wno-unused-result -DBZ_KO=${kk:-0} $extra -c $C/bz_tendency5.hip -o $C/build/ko_$m.o &&
      objs=$(ls $C/build/bz_*.o | grep -v bz_tendency5.o) &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 $objs $C/build/ko_$m.o -shared -L/opt/rocm/lib -lhipfft -ldl -lpthread -Wl,-rpath,/opt/rocm/lib -o breeze.jl_amd/lib/ko_$m.so ) &
  done
  wait
  ls -la breeze.jl_amd/lib/
else
  export TMPDIR=/tmp
  O=gpurun_out/knockout; mkdir -p $O
  for m in base "$@"; do
    lib=breeze.jl_amd/lib/ko_$m.so; [ $m = base ] && lib=breeze.jl_amd/lib/libbreeze_hip.so
    BREEZE_HIP_LIB=$PWD/$lib timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-compressible --no-float32 --single-steps > $O/b_$m.json 2> $O/b_$m.err
    python - $O/b_$m.json $m <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d['kernels_ms_per_step']
    print(f"[{sys.argv[2]:>8s}] {d['ms_per_step']:.2f} ms/step scalar={k['scalar_tendencies+rk3+thermo']/3:.3f} u={k['x_momentum_tendency+rk3+velocity']/3:.3f} v={k['y_momentum_tendency+rk3+velocity']/3:.3f} w={k['z_momentum_tendency+rk3+velocity']/3:.3f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
  done
fi
