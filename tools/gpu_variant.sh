#!/bin/bash
# A/B of compile-time variants of ONE source file:  build: bash tools/gpu_variant.sh build FILE.hip name1 "-DFLAGS1" name2 "-DFLAGS2" ...
#   (links lib/var_<name>.so from the default objects + the variant object)    run on the GPU box: bash tools/gpu_variant.sh run "<bench command with {LIB}>" name1 name2 ...
set -u
mode=$1; shift
cd "$(dirname "$0")/.."
C=breeze.jl_amd/csrc
if [ "$mode" = build32 ]; then      # the same for the Float32 twin: FILE.hip is taken from csrc/build/f32 (generated sources), lib/var32_<name>.so
  f=$1; shift; base=${f%.hip}
  while [ $# -gt 1 ]; do
    name=$1; flags=$2; shift; shift
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value -Wno-unused-result -DBZ_WENO_ONE_DIVISION=2 -fno-slp-vectorize $flags -c $C/build/f32/$f -o $C/build/f32/var_$name.o &&
      objs=$(ls $C/build/f32/bz_*.o | grep -v "/$base.o") &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 $objs $C/build/f32/var_$name.o -shared -L/opt/rocm/lib -lhipfft -ldl -lpthread -Wl,-rpath,/opt/rocm/lib -o breeze.jl_amd/lib/var32_$name.so ) &
  done
  wait; ls breeze.jl_amd/lib/
elif [ "$mode" = build ]; then
  f=$1; shift; base=${f%.hip}
  while [ $# -gt 1 ]; do
    name=$1; flags=$2; shift; shift
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value -Wno-unused-result $flags -c $C/$f -o $C/build/var_$name.o &&
      objs=$(ls $C/build/bz_*.o | grep -v "/$base.o") &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 $objs $C/build/var_$name.o -shared -L/opt/rocm/lib -lhipfft -ldl -lpthread -Wl,-rpath,/opt/rocm/lib -o breeze.jl_amd/lib/var_$name.so ) &
  done
  wait; ls breeze.jl_amd/lib/
else
  export TMPDIR=/tmp
  cmd=$1; shift
  for name in base "$@"; do
    lib=$PWD/breeze.jl_amd/lib/var_$name.so; [ $name = base ] && lib=$PWD/breeze.jl_amd/lib/libbreeze_hip.so
    echo "== $name"; BREEZE_HIP_LIB=$lib bash -c "$cmd" 2>&1 | tail -${TAIL:-3}
  done
fi
