#!/bin/bash
# A/B of the XCD-contiguous block order of k_scalar_pair_lds (lib/var_noxcd4.so = -DBZ4_XCD=0) on the BOMEX step
export TMPDIR=/tmp
for r in 1 2 3; do for name in base noxcd4; do
lib=$PWD/breeze.jl_amd/lib/var_$name.so; [ $name = base ] && lib=$PWD/breeze.jl_amd/lib/libbreeze_hip.so
BREEZE_HIP_LIB=$lib python tools/bench_bomex.py --steps 30 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],3), round(d['kernels_ms_per_step']['scalar_tendencies+rk3'],3))"
done; done
