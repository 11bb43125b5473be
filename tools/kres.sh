#!/bin/bash
# tools/kres.sh FILE.hip [filter] [extra flags] — register / LDS / scratch table of the kernels of one source file (Float64 build flags)
f=$1; flt=${2:-}; shift; shift
d=$(mktemp -d); cd "$(dirname "$0")/../breeze.jl_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value -Wno-unused-result "$@" --cuda-device-only -S $f -o $d/k.s 2>/dev/null
python ../../tools/kernel_resources.py $d/k.s "$flt"
echo "asm: $d/k.s"
