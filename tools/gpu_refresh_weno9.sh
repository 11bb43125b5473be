#!/bin/bash
# refresh of the WENO9 lines of the round-4 evidence set after the z-ring commit (same commands as tools/gpu_profile_r4.sh)
export TMPDIR=/tmp
O=gpurun_out/r04_weno9; mkdir -p $O
timeout 600 python bench.py --workload cbl --cbl-order 9 --steps 50 --warmup 5 > $O/cbl_weno9.json 2> $O/cbl_weno9.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cbl9 -- python bench.py --workload cbl --cbl-order 9 --steps 20 --warmup 3 > /dev/null 2> $O/stats_cbl9.log
cp $(find $O/stats_cbl9 -name "*kernel_stats.csv" | head -1) $O/cbl_weno9_kernel_stats.csv 2>/dev/null
timeout 300 python tools/bench_bomex.py --order 9 2>/dev/null | tail -1 > $O/bomex_weno9_f64.json
timeout 300 python tools/bench_bomex.py --order 9 --float32 2>/dev/null | tail -1 > $O/bomex_weno9_f32.json
timeout 300 python tools/bench_order.py --size 256 --order 9 2>/dev/null | tail -1 > $O/bubble256_weno9_f64.json
find $O -name "*.csv" -size +6M -delete; find $O -name "*.db" -delete
python - $O <<'PY'
import json, sys, os
for f in ("cbl_weno9.json", "bomex_weno9_f64.json", "bomex_weno9_f32.json", "bubble256_weno9_f64.json"):
    d = json.loads([l for l in open(os.path.join(sys.argv[1], f)).read().splitlines() if l.startswith("{")][-1]); print(f, d.get("ms_per_step"))
PY
