#!/usr/bin/env python
"""tools/pmc_to_traffic.py PMC_SUMMARY.json > profiles/rNN_pmc_traffic.json — HBM-side bytes per launch of every kernel group of
bench.py from a tools/pmc_summary.py file (FETCH_SIZE / WRITE_SIZE passes).  Units per /opt/skills/guides/MI355X_MICROARCH.md:
both counters in KiB; on gfx950 FETCH_SIZE tallies 64 B per 128-B request for wide coalesced reads, so read bytes = 2 x FETCH_SIZE
x 1024 (calibrated in round 1 on the 15-array RK update: expected 16.7 GB, 2 x FETCH_SIZE = 17.3 GB).  The counters sit at the
L2-fabric boundary: Infinity-Cache hits are included."""
import json
import sys

GROUPS = {
    "k5_scalar_pair<8>": "scalar_tendencies+rk3+thermo", "k5_u<8>": "x_momentum_tendency+rk3+velocity",
    "k6_u<8>": "x_momentum_tendency+rk3+velocity", "k6_v<8>": "y_momentum_tendency+rk3+velocity", "k6_w<8>": "z_momentum_tendency+rk3+velocity",
    "k6_u<8, false>": "x_momentum_tendency+rk3+velocity", "k6_v<8, false>": "y_momentum_tendency+rk3+velocity",
    # round 3, after the walls-in-y instantiations: the WY template argument shows in the names
    "k5_scalar_pair<8, false>": "scalar_tendencies+rk3+thermo", "k6_u<8, false, false>": "x_momentum_tendency+rk3+velocity",
    "k6_v<8, false, false>": "y_momentum_tendency+rk3+velocity", "k6_w<8, false>": "z_momentum_tendency+rk3+velocity",
    "k6_v<16, false, false>": "y_momentum_tendency+rk3+velocity",      # 64 x 16 tiles of the Float64 y-momentum kernel
    "k_tridiag_coop": "poisson_tridiagonal", "k_tridiag_coop<64>": "poisson_tridiagonal", "k_tridiag_coop<64, 8, true>": "poisson_tridiagonal", "k_x_forward<1>": "poisson_source_term+fft_x", "k_x_inverse": "poisson_fft_x_inverse",
    "k_x_forward<1, 1>": "poisson_source_term+fft_x", "k_x_inverse<1>": "poisson_fft_x_inverse",
    "k5_v<8>": "y_momentum_tendency+rk3+velocity", "k5_w<8>": "z_momentum_tendency+rk3+velocity",
    "k_project_lean": "project_momentum", "k_project_diagnose<0>": "project_and_diagnose",
    "k_poisson_source_rows": "poisson_source_term", "k_tridiag_solve": "poisson_tridiagonal", "k_tridiag_lds": "poisson_tridiagonal",
    "k_scalar_pair_lds<8>": "scalar_tendencies+rk3", "k_u_tend_lds<8>": "x_momentum_tendency+rk3",
    "k_v_tend_lds<8>": "y_momentum_tendency+rk3", "k_w_tend_lds<8, 0>": "z_momentum_tendency+rk3",
}
FFT = {"fwd": "poisson_fft_y_forward", "back": "poisson_fft_y_inverse"}      # library y transforms of the transposed spectrum


def main():
    d = json.load(open(sys.argv[1]))
    out = {"note": __doc__.split("—", 1)[1].strip(), "source": sys.argv[1], "per_kernel_group": {}, "per_kernel_group_float32": {}}
    for k, v in d.items():
        if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        rd, wr = 2.0 * v["FETCH_SIZE"] * 1024.0, v["WRITE_SIZE"] * 1024.0
        # kernels of the Float32 twin carry an _f32 suffix on their name (tools/gen_f32_sources.py); rocFFT's single-precision plans say _sp_
        f32 = "_f32" in k.split("<")[0].split("(")[0] or (k.startswith("fft_rtc_") and "_sp_" in k)
        per = out["per_kernel_group_float32" if f32 else "per_kernel_group"]
        base = k.replace("_f32", "", 1) if f32 else k
        if base in GROUPS:
            per[GROUPS[base]] = {"kernel": k, "read_bytes": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr}
        elif k.startswith("fft_rtc_"):
            g = FFT["fwd" if "_fwd_" in k else "back"]
            e = per.setdefault(g, {"kernel": "rocFFT batched 1-D C2C plan along y", "read_bytes": 0.0, "write_bytes": 0.0, "hbm_bytes_per_launch": 0.0})
            e["read_bytes"] += rd; e["write_bytes"] += wr; e["hbm_bytes_per_launch"] += rd + wr
    json.dump(out, sys.stdout, indent=1)
    sys.stdout.write("\n")


if __name__ == "__main__":
    main()
