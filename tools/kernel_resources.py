#!/usr/bin/env python
"""tools/kernel_resources.py FILE.s [filter] — VGPR / AGPR / SGPR / LDS / scratch / occupancy of every kernel in an amdgcn assembly
file (hipcc -save-temps), read from the .amdhsa_ / metadata comments the compiler emits."""
import re
import subprocess
import sys

text = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
pat = re.compile(r"^\s*; Kernel info:|^(\w+):\s*; @(\w+)|; (NumVgprs|NumAgprs|TotalNumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize): (\d+)|; codeLenInByte = (\d+)", re.M)
cur, rows = None, {}
for m in re.finditer(r"^(\w+):\s*; @\w+|^; (\w+): (\d+)|^; codeLenInByte = (\d+)", text, re.M):
    if m.group(1):
        cur = m.group(1)
        rows[cur] = {}
    elif cur and m.group(2):
        rows[cur][m.group(2)] = int(m.group(3))
    elif cur and m.group(4):
        rows[cur]["code"] = int(m.group(4))
for k, v in rows.items():
    if "NumVgprs" not in v:
        continue
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", k], capture_output=True, text=True).stdout.strip()
    except OSError:
        name = k
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    if flt and flt not in name:
        continue
    g = lambda key: v.get(key, 0)      # noqa: E731
    print(f"{name:48s} vgpr {g('NumVgprs'):4d} agpr {g('NumAgprs'):3d} sgpr {g('TotalNumSgprs'):4d} lds {g('LDSByteSize'):6d} "
          f"scratch {g('ScratchSize'):4d} occ {g('Occupancy'):2d} code {g('code'):6d}")
