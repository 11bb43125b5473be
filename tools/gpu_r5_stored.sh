#!/bin/bash
# round 5: full GPU test suite, then A/B of the stored-velocity sixth-generation kernels (BZ_NO_K6_STORED=1 = the kernels they replace) on the
# BOMEX step (configs[2], both precisions) and the compressible 512 x 512 x 256 step
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r5_stored}; mkdir -p $O
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
fi
for v in "" "BZ_NO_K6_STORED=1"; do
  tag=${v:+old}; tag=${tag:-new}
  env $v python tools/bench_bomex.py --steps 10 --warmup 3 > $O/bomex_f64_$tag.json 2> $O/bomex_f64_$tag.err
  env $v python tools/bench_bomex.py --steps 10 --warmup 3 --float32 > $O/bomex_f32_$tag.json 2> $O/bomex_f32_$tag.err
  env $v python tools/bench_compressible.py --steps 3 --warmup 1 > $O/cmp_$tag.json 2> $O/cmp_$tag.err
  python - $O $tag <<'PY'
import json,sys
O,tag=sys.argv[1:3]
for f in ("bomex_f64","bomex_f32","cmp"):
    try:
        d=json.loads(open(f"{O}/{f}_{tag}.json").read().strip().splitlines()[-1])
        k=d.get("kernels_ms_per_step",{})
        pick={n:round(v,3) for n,v in k.items() if "momentum" in n}
        print(f"[{tag}] {f}: {d.get('ms_per_step')} ms/step {pick}")
    except Exception as e:
        print(tag,f,"FAILED",e, open(f"{O}/{f}_{tag}.err").read()[-600:])
PY
done
