#!/usr/bin/env python
"""tools/pmc_summary.py — per-kernel averages of rocprofv3 --pmc passes.

    python tools/pmc_summary.py OUT.json DIR [DIR ...]
Every DIR is the -d output directory of one `rocprofv3 --pmc ... --kernel-trace` pass (counters of different passes may
overlap; the last one wins).  Output: {kernel (template arguments kept, argument list dropped): {counter: mean per dispatch,
"dispatches": n, "vgpr": ..., "lds_bytes": ..., "scratch": ...}}.  Units are the counters' own (see
/opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE in KiB with FETCH_SIZE counting 64 B per 128-B request on gfx950;
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    depth, out = 0, []
    for ch in name:
        if ch == "(" and depth == 0:
            break
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        out.append(ch)
    s = "".join(out).strip()
    return s[5:] if s.startswith("void ") else s


def main():
    out_path, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    meta = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    a = acc[k][row["Counter_Name"]]
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
                    meta[k] = {"vgpr": int(row["VGPR_Count"]), "agpr": int(row["Accum_VGPR_Count"]), "sgpr": int(row["SGPR_Count"]),
                               "lds_bytes": int(row["LDS_Block_Size"]), "scratch": int(row["Scratch_Size"]),
                               "workgroup": int(row["Workgroup_Size"]), "grid": int(row["Grid_Size"])}
    res = {}
    for k, counters in acc.items():
        res[k] = dict(meta[k])
        for c, (s, n) in counters.items():
            res[k][c] = s / n
            res[k]["dispatches"] = n
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    print(f"{len(res)} kernels -> {out_path}")


if __name__ == "__main__":
    main()
