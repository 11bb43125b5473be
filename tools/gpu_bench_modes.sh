#!/bin/bash
# every bench.py mode one GPU can run, short: prints the key numbers of each JSON line (debugging the bench line itself)
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-bench_modes}; mkdir -p $O
run() { name=$1; shift; timeout 900 "$@" > $O/$name.json 2> $O/$name.err || { echo "$name FAILED rc=$?"; tail -5 $O/$name.err; }
  python - $O/$name.json $name <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
except Exception as e:
    print(sys.argv[2], "no JSON line", e); sys.exit(0)
def rf(r): return None if not r else f"{r['kernel']}: frac {r['frac']:.3f} traffic/compulsory {r.get('traffic_over_compulsory')}"
print(f"[{sys.argv[2]}] {d.get('ms_per_step')} ms/step value {d.get('value')} scaling {d.get('scaling')} | roofline {rf(d.get('roofline'))} | step {d.get('step_roofline', {}).get('frac')} contract {d.get('step_roofline', {}).get('contract_frac')}")
for k in ("float32", "second_milestone"):
    if k in d:
        e = d[k]
        print(f"   {k}: {e.get('ms_per_step')} ms/step roofline {rf(e.get('roofline'))} step {e.get('step_roofline', {}).get('frac') if 'step_roofline' in e else None} substep {e.get('acoustic_substep_roofline', {}).get('frac')} non-substep {e.get('non_substep_ms_per_step')} f32sub {e.get('substep_floattype_float32', {}).get('ms_per_step')}")
for k in ("comm_ms_per_step", "exposed_comm_ms_per_step", "transport", "error"):
    if k in d: print("   ", k, d[k])
PY
}
run default python bench.py --steps 10 --warmup 3 --no-cpu-baseline
run slab1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-compressible --no-float32 --slab
BZ_COMM_SELF_MESSAGES=1 run slab1_self python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-compressible --no-float32 --slab
run cbl python bench.py --workload cbl --steps 20 --warmup 3
run cbl9 python bench.py --workload cbl --cbl-order 9 --steps 10 --warmup 3
run config4 python bench.py --workload config4 --steps 3 --warmup 1
