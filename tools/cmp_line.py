#!/usr/bin/env python
"""stdin: the JSON line of tools/bench_compressible.py -> one summary line"""
import json, sys
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1])
k = d.get("kernels_ms_per_step", {})
sub = sum(v for n, v in k.items() if n.startswith("acoustic_horizontal") or n.startswith("acoustic_column"))
print(f"{d['ms_per_step']:.1f} ms/step  substep {d.get('acoustic_ms_per_substep', 0):.3f} ms  loop {sub:.1f}  rest {d['ms_per_step'] - sub:.1f} | " +
      " ".join(f"{n.replace('acoustic_', 'ac_').replace('_tendency', '')}={v:.2f}" for n, v in sorted(k.items())))
