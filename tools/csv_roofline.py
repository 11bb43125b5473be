#!/usr/bin/env python
"""tools/csv_roofline.py KERNEL_STATS.csv [cells] — recompute every kernel's roofline fraction from a rocprofv3 `--kernel-trace --stats`
summary alone:   frac = compulsory words (tools/accounting.py) x cells x sizeof(word) / AverageNs / 8 TB/s.
Rows of the lean scalar-pair / z-momentum kernels are priced by the body they are (dry: rho q skipped; general), which the kernel name
says since round 5; rows that average under 20 us are guarded launches of the body that did not apply (they return at once) and are
listed as such.  cells defaults to 512^3; kernels of the compressible leg are priced on CMP_CELLS (default 512 x 512 x 256).
    python tools/csv_roofline.py KERNEL_STATS.csv [CELLS [CMP_CELLS]]"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from accounting import HBM_PEAK_GBS, compulsory_words, kernel_group, kernel_variant      # noqa: E402


def main():
    path = sys.argv[1]
    cells = int(sys.argv[2]) if len(sys.argv) > 2 else 512 ** 3
    cells_cmp = int(sys.argv[3]) if len(sys.argv) > 3 else 512 * 512 * 256      # the compressible leg of the default bench command
    rows = list(csv.DictReader(open(path)))
    print(f"# Roofline fractions recomputed from `{os.path.basename(path)}` ({cells} cells per launch)\n")
    print("| kernel | body | calls | average ms | compulsory words/cell | compulsory TB/s | frac of 8 TB/s |")
    print("|---|---|---|---|---|---|---|")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        name = r["Name"].strip()
        if name.startswith("void "):
            name = name[5:]
        group, f32 = kernel_group(name)
        if group is None:
            continue
        avg_ms = float(r["AverageNs"]) * 1e-6
        var = kernel_variant(name)
        short = name.split("(")[0]
        if var is not None and avg_ms < 0.02:
            print(f"| `{short}` | {var} (guarded launches that returned at once) | {r['Calls']} | {avg_ms:.4f} | — | — | — |")
            continue
        w = compulsory_words(group.split(" | ")[0], dry=(var == "dry"))
        if w is None:
            continue
        word = 4 if f32 else 8
        # substep_floattype = Float32 inside a Float64 model (k_ac_*<..., float>): the working fields of the acoustic loop are 4-byte words,
        # (rho w)', right-hand side, factors, averaged velocities and every model field stay 8-byte (csrc/bz_compressible.hip: AcFieldsT)
        mixed = None
        if (not f32) and name.startswith("k_ac_") and ", float>" in name.split("(")[0]:
            mixed = {"acoustic_horizontal+column_forward": 10 * 4 + 13 * 8, "acoustic_column_backward": 5 * 4 + 5 * 8}.get(group.split(" | ")[0], 0)
        compressible = group.startswith(("acoustic", "update_state", "refresh_lin", "density+", "kessler", "store_initial")) or group in ("x_momentum_tendency", "y_momentum_tendency", "z_momentum_tendency")
        ncell = cells_cmp if compressible else cells
        if mixed == 0:
            print(f"| `{short}` | Float32 working fields, mixed word sizes (compressible leg) | {r['Calls']} | {avg_ms:.3f} | {w:.2f} | not priced | — |")
            continue
        tbs = (mixed if mixed else w * word) * ncell / (avg_ms * 1e-3) / 1e12
        if mixed:
            var = f"Float32 working fields: {mixed} B per cell"
        print(f"| `{short}` | {var or ''}{' (compressible leg)' if compressible else ''} | {r['Calls']} | {avg_ms:.3f} | {w:.2f} | {tbs:.2f} | {tbs * 1e3 / HBM_PEAK_GBS:.3f} |")


if __name__ == "__main__":
    main()
