#!/usr/bin/env python
"""tools/roofline_table.py — per-kernel roofline table (markdown) from a bench.py JSON line and the PMC traffic file.

    python tools/roofline_table.py profiles/r01_bench512_v12_default.json profiles/r01_pmc_traffic.json > profiles/r01_roofline_table.md
Columns: HIP-event ms per step and per launch, algorithmic bytes per launch (SURVEY §8d words x 8 B x cells), the rate that
implies against the 8 TB/s HBM peak, and the HBM bytes the kernel really moved (rocprofv3 PMC passes) with the rate that implies.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    bench, pmc = sys.argv[1], sys.argv[2]
    f32 = "--float32" in sys.argv[3:]          # the Float32 leg of the same bench line (b["float32"]), 4-byte words
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from accounting import CONTRACT_WORDS as WORDS_PER_CELL, compulsory_words
    b = json.load(open(bench))
    p = json.load(open(pmc)).get("per_kernel_group_float32" if f32 else "per_kernel_group", {})
    word = 4 if f32 else 8
    if f32:
        top = b
        b = dict(b["float32"])
        b["config"] = top["config"]
        b.setdefault("kernel_launches_per_step", top.get("kernel_launches_per_step", {}))
    cells = 1
    for n in b["config"]["grid_per_gpu"] if "grid_per_gpu" in b["config"] else b["config"]["grid"]:
        cells *= n
    launches = b.get("kernel_launches_per_step", {})
    dry = bool(b.get("dry_path"))
    print(f"# Per-kernel roofline, {b['config']['workload']}" + (" — the Float32 leg (libbreeze_hip_f32.so)" if f32 else "") + "\n")
    sr = b['step_roofline']
    print(f"Step: {b['ms_per_step']:.2f} ms, {b['value'] / 1e9:.3f} Gcells/s; step fraction of the 8 TB/s roofline: {sr['frac']:.3f} in compulsory bytes "
          f"({sr.get('compulsory_bytes_per_cell_step', 0):.0f} B per cell and step); SURVEY §8(d) contract bytes ({sr.get('contract_bytes_per_cell_step', 0):.0f} B per cell and step): "
          + (f"{sr['contract_frac']:.3f}" if sr.get('contract_frac') is not None else "above the roof at this step rate — the fused step no longer moves the unfused array lists") +
          f".  Source: `{os.path.basename(bench)}` (HIP events on the launch stream), "
          f"`{os.path.basename(pmc)}` (rocprofv3 PMC, HBM-side bytes per launch).\n")
    print("Columns: `compulsory` = every distinct 3-D array the (fused) kernel must read or write, once — what `bench.py`'s `roofline.achieved` is priced in; "
          "`contract` = SURVEY §8(d)'s words of the unfused kernel list the fused kernel replaces (the step figure of 250 words per cell and step); "
          "`PMC` = bytes seen at the L2-fabric boundary (FETCH_SIZE / WRITE_SIZE passes).\n")
    print("| kernel group | launches/step | ms/step | ms/launch | compulsory words/cell | compulsory GB/launch | compulsory TB/s (frac of 8) | contract words/cell | PMC GB/launch | PMC / compulsory | PMC TB/s (frac of 8) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    tot = 0.0
    for name, ms in sorted(b["kernels_ms_per_step"].items(), key=lambda kv: -kv[1]):
        tot += ms
        n = launches.get(name, 3)
        per_launch = ms / n
        w = WORDS_PER_CELL.get(name)
        alg = w * word * cells / 1e9 if w else None
        traffic = p.get(name, {}).get("hbm_bytes_per_launch")
        cw = compulsory_words(name, dry)
        comp = cw * word * cells / 1e9 if cw else None
        row = [name, f"{n:.3g}", f"{ms:.2f}", f"{per_launch:.3f}", f"{cw:.2f}" if cw else "—",
               f"{comp:.2f}" if comp else "—", f"{comp / per_launch:.2f} ({comp / per_launch / 8:.2f})" if comp else "—",
               f"{w}" if w else "—",
               f"{traffic / 1e9:.2f}" if traffic else "—", f"{traffic / 1e9 / comp:.2f}" if (traffic and comp) else "—",
               f"{traffic / 1e9 / per_launch:.2f} ({traffic / 1e9 / per_launch / 8:.2f})" if traffic else "—"]
        print("| " + " | ".join(row) + " |")
    print(f"\nSum of kernel time {tot:.2f} ms of the {b['ms_per_step']:.2f} ms step (the rest is launch gaps and two memsets).")


if __name__ == "__main__":
    main()
