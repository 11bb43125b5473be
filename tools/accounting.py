"""tools/accounting.py — the ONE roofline accounting of this repository.

bench.py (every `roofline` block of every workload), tools/roofline_table.py and tools/pmc_to_traffic.py import these tables, so a
fraction printed in a bench line can be recomputed from

        frac = COMPULSORY words x cells x sizeof(word) / (rocprofv3 AverageNs of the kernel) / 8 TB/s

with the kernel statistics committed under profiles/ (KERNEL_GROUPS maps rocprofv3 kernel names onto the groups used here).

Three byte counts, never mixed:
  compulsory  every distinct 3-D array a (fused) kernel has to read or write, once — halo frames, re-reads, column tables and boundary
              planes excluded.  `roofline.achieved` / `roofline.frac` are priced in these everywhere.  Kernels whose argument list
              depends on the RK stage (the lean tendency kernels read U0 in stages 2 and 3 only) are priced per stage and averaged over
              the launches of a step.
  contract    SURVEY.md §8(d): the words of the reference's UNFUSED kernel list that a fused kernel replaces (a stage sums to 80
              words, the step to 250).  Reported beside the compulsory figure as words / bytes — never as a fraction of the roof for a
              single kernel (a fused kernel that got fast exceeds 1.0 in contract bytes, which says nothing); the whole-step contract
              fraction ("the contract figure stays fixed") is kept under `step_roofline.contract_frac`.
  traffic     bytes seen by the FETCH_SIZE / WRITE_SIZE counters at the L2-fabric boundary (rocprofv3 PMC passes, corrected as
              /opt/skills/guides/MI355X_MICROARCH.md prescribes; Infinity-Cache hits included), from profiles/rNN_pmc_traffic.json —
              a reference figure of the same build, not a measurement of the timed run; the line says which file.
"""
import re

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# ---- contract words per cell per launch ------------------------------------------------------------------------------
CONTRACT_WORDS = {
    "ssp_rk3_substep": 20, "store_initial_state": 10, "poisson_source_term": 4,
    "poisson_fft_forward": 4, "poisson_tridiagonal": 2, "poisson_fft_inverse": 4,
    "make_pressure_correction": 7, "compute_velocities": 6,
    "compute_auxiliary_thermodynamic_variables": 5,
    "x_momentum_tendency": 5, "y_momentum_tendency": 5, "z_momentum_tendency": 7,
    "potential_temperature_tendency": 6, "moisture_tendency": 5,
    "scalar_tendencies": 11, "momentum_tendencies": 17, "tendencies": 28,
    "ssp_rk3_substep+store_initial_state": 30, "project_and_diagnose": 18,
    "x_momentum_tendency+rk3": 9, "y_momentum_tendency+rk3": 9, "z_momentum_tendency+rk3": 11,
    "scalar_tendencies+rk3": 19,
    "x_momentum_tendency+rk3+velocity": 11, "y_momentum_tendency+rk3+velocity": 11, "z_momentum_tendency+rk3+velocity": 13,
    "scalar_tendencies+rk3+thermo": 24, "project_momentum": 7,
    "poisson_source_term+fft_forward": 8, "poisson_fft_inverse+project_momentum": 11,
    "poisson_fft_inverse+project_and_diagnose": 22,
    "poisson_source_term+fft_x": 6, "poisson_fft_y_forward": 2, "poisson_fft_y_inverse": 2,
    "poisson_fft_x+project_momentum": 9, "poisson_fft_x_inverse": 2,
    "poisson_fft_y+tridiagonal": 6, "poisson_fft_x_forward": 2,      # round 6: the three middle passes of the solve, chunk by chunk of wavenumbers
}
A_STEP_CONTRACT_WORDS = 250          # 3 stages x 80 + 10
ACOUSTIC_SUBSTEP_CONTRACT_WORDS = 58

# ---- compulsory words per cell per launch --------------------------------------------------------------------------------------------
# a tuple = (stage 1, stage 2, stage 3) of the SSP-RK3 step; the lean seam's first stage reads no U0 (the state arrays ARE U0:
# bz_step.hip, buffer rotation) and nothing stores one
COMPULSORY_WORDS = {
    # lean whole-step seam of the anelastic model (bz_tendency5_kernels.h, bz_fused.hip, bz_xfft_kernels.h, bz_poisson.hip)
    "scalar_tendencies+rk3+thermo": (7, 9, 9),         # R rho_u, rho_v, rho_w, rho_theta, rho_q [+ U0 x 2]; W rho_theta, rho_q
    "x_momentum_tendency+rk3+velocity": (4, 5, 5),     # R rho_u, rho_v, rho_w [+ U0]; W predictor
    "y_momentum_tendency+rk3+velocity": (4, 5, 5),
    "z_momentum_tendency+rk3+velocity": (6, 7, 7),     # + R rho_theta, rho_q (buoyancy: T is derived, not read)
    "project_momentum": 7,                             # R phi, predictor x 3; W momentum x 3
    "project_and_diagnose": 16,                        # R phi, predictor x 3, rho_theta, rho_q; W momentum x 3, u, v, w, theta, q, T, phi
    "poisson_source_term+fft_x": 4,                    # R predictor x 3; W half spectrum
    "poisson_fft_y_forward": 2, "poisson_tridiagonal": 2, "poisson_fft_y_inverse": 2, "poisson_fft_x_inverse": 2,
    "poisson_fft_y+tridiagonal": 6,                    # round 6: R + W of the half spectrum by each of the three passes (a chunk's three passes back to back: what reaches HBM is less)
    "poisson_fft_x_forward": 2,
    # fused-RK tier (generic WENO 7 / 9 kernels, saturation adjustment, closures): tendency + RK update in one pass, stored diagnostics
    "x_momentum_tendency+rk3": (6, 6, 6),              # R rho_u, rho_v, rho_w, u [+ U0 R / W]; W predictor
    "y_momentum_tendency+rk3": (6, 6, 6),
    "z_momentum_tendency+rk3": (8, 8, 8),              # + T, q
    "scalar_tendencies+rk3": (11, 11, 11),             # R u, v, w, theta, q, rho_theta, rho_q, U0 x 2 (R or W); W rho_theta, rho_q
    "potential_temperature_tendency+rk3": (7, 7, 7), "moisture_tendency+rk3": (7, 7, 7),
    "project_and_diagnose+saturation_adjustment": 18,  # project_and_diagnose + W q^v, q^l (bz_fused.hip: k_project_diagnose<1>)
    # closure = SmagorinskyLilly and the column-forcing stack of BASELINE configs[2] (bz_closure.hip, bz_forcing.hip)
    "smagorinsky_viscosity": 6,                        # R u, v, w, T, q^v; W nu_e
    "closure_tendencies": 16,                          # R u, v, w, nu_e, theta, q; R + W predictor x 3, rho_theta, rho_q
    "forcing_tendencies": 10,                          # R u, v (Coriolis / subsidence gradients are column data); R + W predictor x 2, rho_theta, rho_q
    "subsidence_averages": 4,                          # R u, v, theta, q (horizontal means; one word per cell each)
    "subsidence_averages_from_wave_sums": 0.0625,      # R one sum per wavefront (64 cells) of u, v, theta, q, left by project_and_diagnose
    "flux_bc_tendencies": 0.0,                         # bottom plane only: no 3-D array
    # per-operator kernels (one array list per operator, as SURVEY §8d counts them)
    "x_momentum_tendency": 5, "y_momentum_tendency": 5, "z_momentum_tendency": 7, "potential_temperature_tendency": 6,
    "moisture_tendency": 6, "scalar_tendencies": 11, "scalar_tendency": 5, "ssp_rk3_substep": 20, "ssp_rk3_substep+store_initial_state": 20,
    "make_pressure_correction": 7, "compute_velocities": 6, "compute_auxiliary_thermodynamic_variables": 5, "poisson_source_term": 4,
    "poisson_fft_forward": 4, "poisson_fft_inverse": 4,
    # compressible split-explicit path (bz_compressible.hip)
    # round 6: the horizontal gradient of p^L is folded into the slow momentum tendencies once per stage (stages of >= 5 substeps), so the
    # forward sweep reads Gp_ru, Gp_rv instead of G_ru, G_rv AND p: 22 words (23 in a stage that keeps p^L in the substep)
    "acoustic_horizontal+column_forward": 22,          # R rho', (rho theta)' x 2 levels, theta_L, C, (rho u)', (rho v)', G x 4, (rho w)', G^s, <u>, <v>; W (rho u)', (rho v)', <u>, <v>, both predictors, rhs
    "acoustic_column_backward": 10,                    # R theta_L, rhs, predictors x 2, factors, <w>; W rho', (rho theta)', (rho w)', <w>
    # round 6: the stage's first sweeps form the initial perturbations themselves and nothing is copied into U0 (buffer rotation): the kernel
    # is left with R p, rho, G_rho_w, G_rho_u, G_rho_v; W G^s, Gp_ru, Gp_rv (a stage that keeps p^L in the substep: R p, rho, G_rho_w; W G^s)
    "acoustic_stage_init": 8,
    "acoustic_stage_end+update_state": 37,             # R U x 5, perturbations x 5, (rho theta)'_old, theta_L, <u v w>, U0_q, G_q; W U x 5 (rho_d below), <u v w>, rho, u, v, w, theta, q, T, p
    "acoustic_stage_end+update_state+linearization": 41,
    "acoustic_recover_density": 3,
    "acoustic_finalize": 17, "acoustic_recover": 18, "update_state": 14, "update_state+linearization": 18, "refresh_linearization": 8,
    "density+potential_temperature_tendency": 10,      # R rho, u, v, w, theta, rho_u, rho_v, rho_w; W G_rho_theta, G_rho
    "kessler_species_tendencies": 6,
    "store_initial_state": 12,
}
# Dry runs (rho q identically zero, found by the moisture scan that opens every step call — bz_step.hip: bzi_scan_moisture): the two lean
# kernels that would touch rho q skip it altogether, so their compulsory array lists shrink; bench.py prices a run in these when its
# profile shows the scan and the workload set no moisture (and reports the moist variant of the same workload beside it)
COMPULSORY_WORDS_DRY = {
    "scalar_tendencies+rk3+thermo": (5, 6, 6),         # R rho_u, rho_v, rho_w, rho_theta [+ U0]; W rho_theta
    "z_momentum_tendency+rk3+velocity": (5, 6, 6),     # R rho_u, rho_v, rho_w, rho_theta [+ U0]; W predictor
    "moisture_scan": 1,                                # R rho_q (once per step CALL)
}
ACOUSTIC_SUBSTEP_COMPULSORY_WORDS = 32                 # forward 22 + backward 10 (fused substep, thermal divergence damping, p^L gradient folded)


def acoustic_substep_words(nsub, dry, fold_min=5, pair_avg=True):
    """Mean compulsory words per cell of the forward and the backward sweep over the substeps of one step, from the substep counts of the
    three stages.  A stage of fewer than fold_min substeps keeps p^L in the substep (+1 word, bz_compressible.hip: AcStage::pfold); in a
    DRY whole step (rho q identically zero, found by the opening scan) stages 1 and 2 carry no time-average accumulators: the forward sweep
    neither reads nor writes <u>, <v> (-4 words), the backward sweep <w> (-2) (AcParams::skip_avg_if_dry).  pair_avg: in a stage that does
    accumulate, <u>, <v> are added two substeps at a time (AcParams::acc_mode): of the substeps 2 .. N, floor((N - 1) / 2) leave the
    accumulators alone (-4 words each)."""
    tot = float(sum(nsub))
    fwd = sum(n * ((22 if n >= fold_min else 23) - (4 if (dry and s < 2) else 0)) - (0 if (dry and s < 2) or not pair_avg else 4 * ((n - 1) // 2))
              for s, n in enumerate(nsub)) / tot
    # <w> in the backward sweep: a pair's first substep leaves the accumulator alone (-2), its second reads the (rho w)' it overwrites (+1)
    bwd = sum(n * (10 - (2 if (dry and s < 2) else 0)) - (0 if (dry and s < 2) or not pair_avg else (n - 1) // 2) for s, n in enumerate(nsub)) / tot
    return fwd, bwd


def compulsory_words(group, dry=False):
    """Mean compulsory words per cell per launch of a kernel group (None: unknown group).  Stage-dependent groups are averaged over the
    three stages — a step launches each of them once per stage.  dry: the run took the zero-moisture path of the lean kernels."""
    w = COMPULSORY_WORDS_DRY.get(group) if dry else None
    if w is None:
        w = COMPULSORY_WORDS.get(group)
    if w is None:
        return None
    if isinstance(w, tuple):
        return sum(w) / float(len(w))
    return float(w)


def step_compulsory_words(launches_per_step, groups=None, dry=False):
    """Compulsory words per cell and step of a set of kernel groups from their launches per step (bench.py's own counters)."""
    tot = 0.0
    for name, n in launches_per_step.items():
        if groups is not None and name not in groups:
            continue
        w = compulsory_words(name, dry)
        if w is not None:
            tot += w * n
    return tot


def roofline_block(group, avg_launch_ms, cells, word_bytes, traffic_bytes=None, traffic_source=None, words=None, dry=False):
    """The `roofline` object of a bench line for one kernel group: priced in compulsory bytes; contract words beside it; PMC traffic
    (bytes per launch) and its ratio to the compulsory bytes when a committed PMC file has the group."""
    w = words if words is not None else compulsory_words(group, dry)
    if w is None or not avg_launch_ms:
        return None
    nbytes = w * word_bytes * cells
    achieved = nbytes / (avg_launch_ms * 1e-3) / 1e9
    out = {"bound": "hbm", "kernel": group, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "traffic": traffic_bytes, "traffic_source": traffic_source,
           "traffic_over_compulsory": (traffic_bytes / nbytes) if traffic_bytes else None,
           "traffic_frac": (traffic_bytes / (avg_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic_bytes else None,
           "avg_launch_ms": avg_launch_ms, "bytes": "compulsory: every distinct 3-D array of the fused kernel once (tools/accounting.py)",
           "compulsory_words_per_cell": w, "compulsory_bytes_per_launch": nbytes, "dry_path": bool(dry and group in COMPULSORY_WORDS_DRY),
           "contract_words_per_cell": CONTRACT_WORDS.get(group),
           "contract_bytes_per_launch": CONTRACT_WORDS[group] * word_bytes * cells if group in CONTRACT_WORDS else None}
    return out


def load_traffic(root, group, f32=False, files=("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json"), tag=None):
    """HBM-side bytes per launch of a kernel group from the newest committed PMC file that has it -> (bytes, 'profiles/<file>').
    tag: a secondary workload's own file (profiles/r06_pmc_traffic_<tag>.json: PMC passes of that workload at its own grid) and nothing else —
    bytes per launch scale with the grid, the headline's files do not apply."""
    import json
    import os
    if tag is not None:
        files = ("r06_pmc_traffic_%s.json" % tag,)
    for fn in files:
        try:
            with open(os.path.join(root, "profiles", fn)) as fh:
                d = json.load(fh)
            return d["per_kernel_group_float32" if f32 else "per_kernel_group"][group]["hbm_bytes_per_launch"], "profiles/" + fn
        except (OSError, KeyError, ValueError):
            continue
    return None, None


# ---- rocprofv3 kernel name -> kernel group (profiles/rNN_*kernel_stats.csv rows) ------------------------------------------------------
# regular expressions on the demangled name with the `_f32` suffix of the Float32 twin removed; first match wins
KERNEL_GROUPS = [
    (r"^k5_scalar_pair<", "scalar_tendencies+rk3+thermo"),
    # stored-velocity instantiations of the sixth-generation kernels (round 5: k6_u / k6_v<TY, MF, WY, ST = true>, k6_w<.., BM = 0 .. 3>): the
    # per-operator groups (in the default bench command they are the slow tendencies of the compressible leg)
    (r"^k6_u<\d+, \w+, \w+, true>", "x_momentum_tendency"), (r"^k6_v<\d+, \w+, \w+, true>", "y_momentum_tendency"),
    (r"^k6_w<\d+, \w+, \w+, \w+, [0-3]>", "z_momentum_tendency"),
    (r"^k6_u<", "x_momentum_tendency+rk3+velocity"), (r"^k6_v<", "y_momentum_tendency+rk3+velocity"), (r"^k6_w<", "z_momentum_tendency+rk3+velocity"),
    (r"^k_closure_march<", "closure_tendencies"), (r"^k_closure_tendencies", "closure_tendencies"),
    (r"^k_smagorinsky_march<", "smagorinsky_viscosity"), (r"^k_smagorinsky_viscosity", "smagorinsky_viscosity"),
    (r"^k_project_lean", "project_momentum"), (r"^k_project_diagnose<", "project_and_diagnose"),
    (r"^k_x_forward<1", "poisson_source_term+fft_x"), (r"^k_x_inverse", "poisson_fft_x_inverse"), (r"^k_tridiag_", "poisson_tridiagonal"),
    (r"^fft_rtc_fwd_", "poisson_fft_y_forward"), (r"^fft_rtc_back_", "poisson_fft_y_inverse"),
    (r"^k_ac_column_forward<\w+, true", "acoustic_horizontal+column_forward"), (r"^k_ac_forward2<", "acoustic_horizontal+column_forward"),
    (r"^k_ac_column_backward<", "acoustic_column_backward"),
    (r"^k_ac_stage_init<", "acoustic_stage_init"), (r"^k_ac_stage_end<\w+, true", "acoustic_stage_end+update_state+linearization"),
    (r"^k_ac_stage_end<\w+, false", "acoustic_stage_end+update_state"), (r"^k_ac_recover_density<", "acoustic_recover_density"),
    (r"^k_ac_finalize<", "acoustic_finalize"), (r"^k_ac_recover<", "acoustic_recover"),
    (r"^k_cmp_diagnose<true, true", "update_state+linearization"), (r"^k_cmp_diagnose<true, false", "update_state"),
    (r"^k_cmp_linearization", "refresh_linearization"), (r"^k_scalar_(tendency_rho3d|rho3d_lds)", "density+potential_temperature_tendency | moisture_tendency"),
    (r"^k_u_tend_lds<", "x_momentum_tendency"), (r"^k_v_tend_lds<", "y_momentum_tendency"), (r"^k_w_tend_lds<", "z_momentum_tendency"),
    (r"^k_scalar_pair_lds<", "scalar_tendencies+rk3"),
    # generic WENO 7 / 9 kernels of the fused-RK tier (bz_tendency_generic.hip): the one-pass marching kernel k_tendency_m<R, KIND, ..>
    # (KIND 0 scalar, 1 / 2 x / y momentum) and the two passes of the z-momentum kernel (flux pass + divergence pass: their bytes ADD)
    (r"^k_tendency_m<\d+, 0,", "potential_temperature_tendency+rk3"), (r"^k_tendency_m<\d+, 1,", "x_momentum_tendency+rk3"),
    (r"^k_tendency_m<\d+, 2,", "y_momentum_tendency+rk3"), (r"^k_tendency_m<\d+, 3,", "z_momentum_tendency+rk3"),
    (r"^k_w_tendency_g<\d+, [12],", "z_momentum_tendency+rk3"), (r"^k_apply_forcings<", "forcing_tendencies"),
]
# kernel groups made of several launches whose bytes add up (tools/pmc_to_traffic.py): the passes of a two-pass generic kernel
ADDITIVE_KERNELS = (r"^k_w_tendency_g<", r"^k_u_tendency_g<\d+, [12]", r"^k_v_tendency_g<\d+, [12]", r"^k_scalar_tendency_g<\d+, [12]")


def kernel_variant(name):
    """Which body a lean scalar-pair / z-momentum kernel row is (round 5: separate kernels, k5_scalar_pair<TY, WY, DRYQ, GUARD> and
    k6_w<TY, WY, DRYQ, GUARD>): 'dry' (rho q identically zero: priced in COMPULSORY_WORDS_DRY), 'general', or None for every other kernel.
    A guarded launch (GUARD = true) of the body that does not apply returns at once: such rows average a few microseconds."""
    head = name.split("(")[0]
    head = head[5:] if head.startswith("void ") else head
    if re.match(r"^k6_w(?:_f32)?<\s*\d+\s*,\s*\w+\s*,\s*\w+\s*,\s*\w+\s*,\s*[0-3]\s*>", head):
        return None      # stored-velocity instantiations: no moisture-scan dispatch
    m = re.match(r"^(?:k5_scalar_pair|k6_w)(?:_f32)?<\s*\d+\s*,\s*(?:true|false)\s*,\s*(true|false)", head)
    if not m:
        return None
    return "dry" if m.group(1) == "true" else "general"


def kernel_group(name):
    """(group, is_float32) of a rocprofv3 kernel name, or (None, False)."""
    head = name.split("(")[0]
    f32 = "_f32" in head.split("<")[0] or (head.startswith("fft_rtc_") and "_sp_" in head)
    base = head.replace("_f32", "", 1) if "_f32" in head.split("<")[0] else head
    for pat, group in KERNEL_GROUPS:
        if re.search(pat, base):
            return group, f32
    return None, f32
