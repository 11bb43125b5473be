#!/usr/bin/env python
"""tools/pmc_to_traffic.py PMC_SUMMARY.json > profiles/rNN_pmc_traffic.json — HBM-side bytes per launch of every kernel group of
bench.py from a tools/pmc_summary.py file (FETCH_SIZE / WRITE_SIZE passes).  Units per /opt/skills/guides/MI355X_MICROARCH.md:
both counters in KiB; on gfx950 FETCH_SIZE tallies 64 B per 128-B request for wide coalesced reads, so read bytes = 2 x FETCH_SIZE
x 1024 (calibrated in round 1 on the 15-array RK update: expected 16.7 GB, 2 x FETCH_SIZE = 17.3 GB).  The counters sit at the
L2-fabric boundary: Infinity-Cache hits are included."""
import json
import sys

import os

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from accounting import ADDITIVE_KERNELS, kernel_group, kernel_variant      # noqa: E402  rocprofv3 kernel name -> kernel group of bench.py (one table for every tool)


def main():
    d = json.load(open(sys.argv[1]))
    out = {"note": __doc__.split("—", 1)[1].strip(), "source": sys.argv[1], "per_kernel_group": {}, "per_kernel_group_float32": {}}
    for k, v in d.items():
        if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        rd, wr = 2.0 * v["FETCH_SIZE"] * 1024.0, v["WRITE_SIZE"] * 1024.0
        # kernels of the Float32 twin carry an _f32 suffix on their name (tools/gen_f32_sources.py); rocFFT's single-precision plans say _sp_
        group, f32 = kernel_group(k)
        if group is None:
            continue
        # round 5: the dry and the general body of the lean scalar-pair / z-momentum kernels are separate kernels.  The group's own key keeps
        # the body the headline runs (dry); the general body goes under "<group> (general body)".  Guarded launches of the body that does
        # not apply return at once: a few KB of traffic — not a row of the table
        var = kernel_variant(k)
        if var is not None and rd + wr < 1e7:
            continue
        if var == "general":
            group = group + " (general body)"
        # the acoustic kernels instantiated for substep_floattype = Float32 inside the Float64 library (template argument `float`) move
        # fewer bytes: kept apart so that the Float64-storage figures are not averaged with them
        sub32 = (not f32) and k.startswith("k_ac_") and k.rstrip().endswith("float>")
        per = out.setdefault("per_kernel_group_substep_float32", {}) if sub32 else out["per_kernel_group_float32" if f32 else "per_kernel_group"]
        e = per.setdefault(group, {"kernel": k if not k.startswith("fft_rtc_") else "rocFFT batched 1-D C2C plan along y", "read_bytes": 0.0, "write_bytes": 0.0,
                                   "hbm_bytes_per_launch": 0.0, "kernels": 0})
        # several instantiations of one group (e.g. the damped / undamped forward sweep) are averaged; rocFFT plans made of several kernels add up
        import re
        base = k.replace("_f32", "", 1)
        if k.startswith("fft_rtc_") or any(re.search(p, base) for p in ADDITIVE_KERNELS):
            e["read_bytes"] += rd; e["write_bytes"] += wr; e["hbm_bytes_per_launch"] += rd + wr
        else:
            n = e["kernels"]
            e["read_bytes"] = (e["read_bytes"] * n + rd) / (n + 1); e["write_bytes"] = (e["write_bytes"] * n + wr) / (n + 1)
            e["hbm_bytes_per_launch"] = e["read_bytes"] + e["write_bytes"]
            e["kernels"] = n + 1
    json.dump(out, sys.stdout, indent=1)
    sys.stdout.write("\n")


if __name__ == "__main__":
    main()
