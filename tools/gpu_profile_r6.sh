#!/bin/bash
# round-6 evidence set, all from ONE box: the default bench line, rocprofv3 kernel statistics of the same command, FETCH_SIZE / WRITE_SIZE
# passes of the same command (anelastic Float64 + Float32 legs and the compressible milestone), the CBL lines of the reference's CI
# sizes' middle grid (WENO5 / WENO9, Float32) with their kernel statistics, config4, and the roofline tables built from them.
#   bash tools/gpu_profile_r5.sh TAG        -> gpurun_out/TAG/*
set -u
export TMPDIR=/tmp
TAG=${1:-r06}
O=gpurun_out/$TAG; mkdir -p $O
echo "== bench (default command)"
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
echo "== rocprofv3 kernel statistics of the default command"
# (the profiled runs keep the compressible milestone in the process rocprofv3 watches: BZ_BENCH_MILESTONE_INPROCESS=1)
BZ_BENCH_MILESTONE_INPROCESS=1 timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.log
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
echo "== PMC traffic passes (separate passes: FETCH_SIZE, WRITE_SIZE)"
export BZ_BENCH_MILESTONE_INPROCESS=1
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"      # the moist leg rides along: the general bodies are kernels of their own since round 5
timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
python tools/pmc_summary.py $O/pmc_summary.json $O/pmc_fetch $O/pmc_write > $O/pmc_summary.log 2>&1
python tools/pmc_to_traffic.py $O/pmc_summary.json > $O/pmc_traffic.json
unset BZ_BENCH_MILESTONE_INPROCESS
echo "== SQ / traffic passes of the compressible step (k_ac_forward2: wait fractions, waves, fetched bytes)"
bash tools/gpu_pmc_compressible.sh $O/pmc_cmp > $O/pmc_cmp.log 2>&1
cp $O/pmc_cmp/pmc_summary.json $O/pmc_compressible.json 2>/dev/null
cp $O/pmc_cmp/kernel_stats.csv $O/compressible_kernel_stats.csv 2>/dev/null
echo "== SQ passes of the headline kernels (LDS phases of the x transforms and the tridiagonal kernel: VERDICT r04 item 5)"
bash tools/gpu_pmc_sq.sh $O/pmc_sq --no-moist-variant > $O/pmc_sq.log 2>&1
cp $O/pmc_sq/pmc_summary.json $O/pmc_sq_headline.json 2>/dev/null
echo "== CBL 512x512x256 Float32, WENO5 and WENO9 (+ kernel statistics)"
for ord in 5 9; do
  timeout 600 python bench.py --workload cbl --cbl-order $ord --steps 50 --warmup 5 > $O/cbl_weno${ord}.json 2> $O/cbl_weno${ord}.err
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cbl$ord -- python bench.py --workload cbl --cbl-order $ord --steps 20 --warmup 3 > /dev/null 2> $O/stats_cbl$ord.log
  cp $(find $O/stats_cbl$ord -name "*kernel_stats.csv" | head -1) $O/cbl_weno${ord}_kernel_stats.csv 2>/dev/null
done
echo "== config4 (compressible + Kessler, 512x512x128)"
timeout 600 python bench.py --workload config4 --steps 5 --warmup 2 > $O/config4.json 2> $O/config4.err
echo "== BOMEX 256x256x128 (physics list of examples/bomex.jl) and the 256^3 bubble with WENO9"
timeout 300 python tools/bench_bomex.py --steps 30 2>/dev/null | tail -1 > $O/bomex_weno5_f64.json
timeout 300 python tools/bench_bomex.py --steps 30 --float32 2>/dev/null | tail -1 > $O/bomex_weno5_f32.json
timeout 300 python tools/bench_bomex.py --steps 30 --order 9 2>/dev/null | tail -1 > $O/bomex_weno9_f64.json
timeout 300 python tools/bench_bomex.py --steps 30 --order 9 --float32 2>/dev/null | tail -1 > $O/bomex_weno9_f32.json
timeout 300 python tools/bench_order.py --size 256 --order 9 2>/dev/null | tail -1 > $O/bubble256_weno9_f64.json
echo "== slab driver on one GPU: world 1, and with every message sent to itself"
timeout 600 python bench.py --slab --steps 10 --warmup 3 --no-cpu-baseline --no-compressible --no-float32 > $O/slab_world1.json 2> $O/slab_world1.err
BZ_COMM_SELF_MESSAGES=1 timeout 600 python bench.py --slab --steps 10 --warmup 3 --no-cpu-baseline --no-compressible --no-float32 > $O/slab_world1_self_messages.json 2> $O/slab_self.err
echo "== GPU tests of the same build"
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
echo "== tables"
python tools/csv_roofline.py $O/kernel_stats.csv > $O/csv_roofline.md 2>> $O/table.err
python tools/roofline_table.py $O/bench.json $O/pmc_traffic.json > $O/roofline_table.md 2> $O/table.err
python tools/roofline_table.py $O/bench.json $O/pmc_traffic.json --float32 > $O/roofline_table_f32.md 2>> $O/table.err
find $O -name "*.csv" -size +6M -delete
find $O -name "*.db" -delete
ls -la $O | head -40
python - $O <<'PY'
import json, sys, os
O = sys.argv[1]
for f in ("bench.json", "cbl_weno5.json", "cbl_weno9.json", "config4.json", "slab_world1.json", "slab_world1_self_messages.json", "bomex_weno5_f64.json", "bomex_weno5_f32.json", "bomex_weno9_f64.json", "bomex_weno9_f32.json", "bubble256_weno9_f64.json"):
    try:
        d = json.loads([l for l in open(os.path.join(O, f)).read().splitlines() if l.startswith("{")][-1])
        print(f, d.get("ms_per_step"), d.get("value"), (d.get("roofline") or {}).get("kernel"), (d.get("roofline") or {}).get("frac"))
    except Exception as e:
        print(f, "FAILED", e)
PY
