#!/bin/bash
# tile-height probe of the lean kernels: parity of the lean seam under each setting, then short bench lines
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-ty_probe}; mkdir -p $O
for envs in "BZ_SCALAR_TY=4" "BZ_LEAN_TY=4"; do
  env $envs timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lean or time_steps or whole_step" > $O/pytest_$envs.log 2>&1; echo "$envs: $(tail -1 $O/pytest_$envs.log)"
done
bash tools/gpu_quick.sh ${1:-ty_probe} "" "BZ_SCALAR_TY=4" "BZ_LEAN_TY=4"
