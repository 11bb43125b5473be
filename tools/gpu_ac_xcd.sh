#!/bin/bash
# A/B of the forward acoustic sweep's XCD mapping (BZ_AC_XCD=0: launch order, 1: bands of tile rows): bash tools/gpu_ac_xcd.sh
export TMPDIR=/tmp
for rep in 1 2; do
for v in 0 1; do
echo "== BZ_AC_XCD=$v"
BZ_AC_XCD=$v python tools/bench_compressible.py --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d.get('kernels_ms_per_step',{})
print(round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items() if 'acoustic' in a})"
done; done
BZ_AC_XCD=1 python tools/bench_compressible.py --steps 4 --warmup 2 --substep-float32 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('f32 substeps xcd=1', round(d['ms_per_step'],2))"
BZ_AC_XCD=0 python tools/bench_compressible.py --steps 4 --warmup 2 --substep-float32 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('f32 substeps xcd=0', round(d['ms_per_step'],2))"
