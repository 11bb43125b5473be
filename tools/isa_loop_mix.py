#!/usr/bin/env python
"""tools/isa_loop_mix.py FILE.s KERNEL_SUBSTRING — instruction mix of the largest loop (label .. backward branch) of a kernel."""
import collections
import re
import sys

text = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r"^(\w*" + re.escape(key) + r"\w*):\s*; @", text, re.M)
name = m.group(1)
body = text[m.end():text.index(".Lfunc_end", m.end())]
lines = body.split("\n")
labels = {}
for i, ln in enumerate(lines):
    lm = re.match(r"^(\.LBB\d+_\d+):", ln)
    if lm:
        labels[lm.group(1)] = i
best = None
for i, ln in enumerate(lines):
    bm = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", ln)
    if bm:
        t = bm.group(1) or bm.group(2)
        if t in labels and labels[t] < i and (best is None or i - labels[t] > best[1] - best[0]):
            best = (labels[t], i)
lo, hi = best
cls = collections.Counter()
ops = collections.Counter()
for ln in lines[lo:hi + 1]:
    ln = ln.strip()
    if not ln or ln.startswith((";", ".")) or ln.endswith(":"):
        continue
    op = ln.split()[0]
    ops[op] += 1
    if op.startswith("v_") and "f64" in op: c = "valu_f64"
    elif op.startswith("v_cndmask"): c = "valu_cndmask"
    elif op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")): c = "valu_lane"
    elif op.startswith("v_"): c = "valu_other"
    elif op.startswith("s_waitcnt"): c = "s_waitcnt"
    elif op.startswith(("s_load", "s_buffer")): c = "smem"
    elif op.startswith("s_"): c = "salu"
    elif op.startswith("ds_"): c = "lds"
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c = "vmem"
    else: c = "other"
    cls[c] += 1
print(name, "loop lines", lo, hi, "instructions", sum(cls.values()))
for c, n in cls.most_common():
    print(f"  {c:14s} {n}")
print("  top ops:", ", ".join(f"{o}={n}" for o, n in ops.most_common(28)))
