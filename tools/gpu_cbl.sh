#!/bin/bash
# The reference's CI benchmark matrix (.github/workflows/Benchmarks.yml:34-45) on one MI355X: convective boundary layer,
# Float32, 5 warm-up + 150 timed steps, three grids x WENO5 / WENO9.  One JSON line per case into OUTDIR.
#   bash tools/gpu_cbl.sh OUTDIR [sizes...]
set -u
export TMPDIR=/tmp
O=$1; shift
mkdir -p $O
SIZES=${@:-256x256x128 512x512x256 768x768x256}
for s in $SIZES; do
  for o in 5 9; do
    timeout 900 python bench.py --workload cbl --cbl-size $s --cbl-order $o --warmup 5 --steps 150 > $O/cbl_${s}_weno${o}_f32.json 2> $O/cbl_${s}_weno${o}.err
    python - $O/cbl_${s}_weno${o}_f32.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(f"{d['config']['grid']} {d['config']['workload'].split('WENO')[1][:1]} {d['dtype']}: {d['ms_per_step']:.3f} ms/step = {d['value']/1e9:.3f} Gcells/s  finite={d['finite']}  top: " +
          ", ".join(f"{k}={v:.2f}" for k,v in sorted(d['kernels_ms_per_step'].items(), key=lambda kv:-kv[1])[:6]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  done
done
