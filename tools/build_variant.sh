#!/bin/bash
# one-off library variants for A/B runs: bash tools/build_variant.sh NAME SOURCE.hip -DFLAG=..   ->  breeze.jl_amd/lib/libbreeze_hip_var_NAME.so
# (the named source compiled with the extra flags, every other object taken from the default build; select with BREEZE_HIP_LIB)
set -e
cd "$(dirname "$0")/../breeze.jl_amd/csrc"
NAME=$1; SRC=$2; shift 2
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value "$@" -c $SRC -o build/var_${NAME}.o
OBJS=$(ls build/bz_*.o | grep -v "build/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 build/var_${NAME}.o $OBJS -shared -L/opt/rocm/lib -lhipfft -ldl -lpthread -Wl,-rpath,/opt/rocm/lib -o ../lib/libbreeze_hip_var_${NAME}.so
echo ../lib/libbreeze_hip_var_${NAME}.so
