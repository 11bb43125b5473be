#!/bin/bash
# five fresh processes each, one box: the headline leg (512^3 dry bubble, Float64) and the compressible milestone (dry path and with vapour) -> medians
set -u
export TMPDIR=/tmp
O=gpurun_out/five; mkdir -p $O
for r in 1 2 3 4 5; do
  timeout 300 python bench.py --no-cpu-baseline --no-float32 --no-compressible --no-moist-variant 2>/dev/null | tail -1 > $O/head_$r.json
  timeout 300 python bench.py --milestone-only 2>/dev/null | tail -1 > $O/mile_$r.json
done
python - $O <<'PY'
import json, sys, statistics as st
O = sys.argv[1]
h = [json.load(open(f"{O}/head_{r}.json"))["ms_per_step"] for r in range(1, 6)]
m = [json.load(open(f"{O}/mile_{r}.json")) for r in range(1, 6)]
d = [x["ms_per_step"] for x in m]
w = [(x.get("moist_variant") or {}).get("ms_per_step") for x in m]
ns = [x.get("non_substep_ms_per_step") for x in m]
f = lambda v: " ".join(f"{x:.2f}" for x in v)
print("headline 512^3 Float64, ms per step:        ", f(h), " median", round(st.median(h), 2))
print("compressible 512x512x256 dry, ms per step:  ", f(d), " median", round(st.median(d), 2))
print("  ... outside the substep loop:             ", f(ns), " median", round(st.median(ns), 2))
if all(w): print("compressible with vapour set, ms per step:  ", f(w), " median", round(st.median(w), 2))
PY
