#!/usr/bin/env python
"""bench.py — grid cells advanced per second by the anelastic WENO5 + Poisson SSP-RK3 step.

Metric and protocol mirror the reference's benchmark_time_stepping
(/root/reference/benchmarking/src/utils.jl:40-172: warm-up steps, device sync, timed steps, sync,
grid_points_per_second = Nx*Ny*Nz / time_per_step).  Workload: BASELINE.json configs[1], the dry
thermal bubble on a 512^3 RectilinearGrid (SURVEY.md §8d "C2"), Float64, fixed dt = 1 s,
deterministic synthetic initial state resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W [--size 512] [--scaling strong|weak] [--workload bubble|config3] [--preflight]

Prints ONE JSON line (rank 0).  N > 1: one rank per GPU, y-slab decomposition with the RCCL communicator inside the C library
(breeze.jl_amd/csrc/bz_comm.hip: y-halo exchange + FFT all-to-all):
  --scaling strong  (default) the size^3 domain of the N = 1 run is split N ways — BASELINE.md §4: "1/2/4/8-GPU cells/s for 512^3"
  --scaling weak    every rank owns a size^3 slab of a size x (size N) x size periodic domain (explicit only)
  --workload config3  BASELINE.json configs[3]: 1024 x 1024 x 512 split over the N ranks (8 ranks -> 1024 x 128 x 512 each)
`value` is the aggregate over ranks; the line carries `scaling`, `transport`, `comm_ms_per_step`, `exposed_comm_ms_per_step`.
Before any timed region a multi-rank run goes through the PREFLIGHT (also available alone as `--preflight`): communicator bootstrap on
all ranks, a rank-coded halo exchange through the library, and two steps of a 64^3-per-rank bubble compared with the same domain
stepped on one GPU (which exercises both all-to-alls of the pressure solve) — any failure ends the run with a JSON "error" line and
a non-zero exit code.  There is no transport fallback: `--transport rccl` (default) or `--transport torch` (explicit) or an error.
If the ranks are not there yet (WORLD_SIZE unset, as in `python bench.py --gpus 8`), the script launches itself under
torch.distributed.run and relays the ranks' line.  `--replicas` (explicit only) runs N independent copies.
Roofline accounting: tools/accounting.py (compulsory bytes everywhere; contract words beside; PMC traffic from profiles/).
"""
import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

sys.path.insert(0, os.path.join(ROOT, "tools"))
from accounting import (A_STEP_CONTRACT_WORDS, ACOUSTIC_SUBSTEP_COMPULSORY_WORDS, acoustic_substep_words, ACOUSTIC_SUBSTEP_CONTRACT_WORDS, COMPULSORY_WORDS,      # noqa: E402
                        CONTRACT_WORDS, HBM_PEAK_GBS, compulsory_words, load_traffic, roofline_block, step_compulsory_words)

WORDS_PER_CELL = CONTRACT_WORDS      # (older tools import the contract table under this name)
A_STEP_WORDS = A_STEP_CONTRACT_WORDS
METRIC = "grid-cells advanced/sec (tendency+Poisson step), 512^3 anelastic"


def bubble(x, y, z):
    """theta_i = theta0 exp(N^2 z / g) + 10 max(0, 1 - r/2000), bubble centre (0, 0, 3000 m)."""
    r = np.sqrt(x ** 2 + y ** 2 + (z - 3000.0) ** 2)
    return 300.0 * np.exp(1e-6 * z / 9.81) + 10.0 * np.maximum(0.0, 1.0 - r / 2000.0)


EXTENT = ((-10e3, 10e3), (-10e3, 10e3), (0.0, 10e3))


def kernel_table(profile):
    return {name: {"avg_ms": ms / n, "launches": n, "total_ms": ms} for name, (ms, n) in profile.items() if n}


def took_dry_path(kernels):
    """The run's profile shows the moisture scan of the lean seam and the workload set no moisture: the scalar-pair and z-momentum kernels
    skipped rho q altogether (csrc/bz_step.hip: bzi_scan_moisture) — their compulsory array lists are the dry ones."""
    return "moisture_scan" in kernels


def dominant_roofline(kernels, cells, word_bytes, f32=False, with_traffic=True, exclude=("comm_", "moisture_scan"), dry=False, general_body=False,
                      traffic_tag=None):
    """`roofline` of the kernel group with the largest share of the timed region, among the groups tools/accounting.py can price."""
    known = [k for k in kernels if compulsory_words(k) is not None and not k.startswith(exclude)]
    if not known:
        return None
    dom = max(known, key=lambda k: kernels[k]["total_ms"])
    traffic, src = (None, None)
    if with_traffic:
        # the general body of the lean scalar-pair / z-momentum kernels has PMC rows of its own (tools/pmc_to_traffic.py)
        if general_body:
            traffic, src = load_traffic(ROOT, dom + " (general body)", f32)
        if traffic is None:
            traffic, src = load_traffic(ROOT, dom, f32, tag=traffic_tag)
    return roofline_block(dom, kernels[dom]["avg_ms"], cells, word_bytes, traffic, src, dry=dry)


def moist_variant(model, dt, steps=5):
    """The general path beside the dry headline: the same model and grid with a moisture field set (q^t = 5 g/kg decaying with height),
    `steps` steps through the same seam.  The dry thermal bubble of BASELINE.json carries q^t = 0, where the lean kernels skip rho q
    after the moisture scan; this leg shows what a moist model of the same size costs (same kernels, nothing skipped)."""
    import torch
    model.set(qᵗ=lambda x, y, z: 5e-3 * np.exp(-z / 2500.0) + 0 * x + 0 * y)
    model.time_step(dt)
    model.profile_reset()
    model.profile_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.time_steps(dt, steps, diagnose_last=True)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    model.profile_enable(False)
    k = kernel_table(model.profile())
    g = model.grid
    cells = g.Nx * g.Ny * g.Nz
    # the moist leg has a roofline of its own: its dominant kernel and the whole step, priced in the GENERAL array
    # lists (rho q read, advected and written; tools/accounting.py COMPULSORY_WORDS).  Since round 5 the dry and the general bodies are
    # separate kernels (k5_scalar_pair<8, false, DRYQ, GUARD>, k6_w<...>), so the rows of a rocprofv3 run separate as well
    return {"ms_per_step": ms, "steps": steps, "moisture": "q^t = 5e-3 exp(-z / 2500 m)",
            "stepping": "bz_time_steps_anelastic(n = K, diagnose_last = 1), the headline's seam and K",
            "value": cells / (ms * 1e-3), "unit": "cells/s",
            "roofline": dominant_roofline(k, cells, 8, with_traffic=(cells == 512 ** 3), dry=False, general_body=True),
            "step_roofline": step_roofline(k, steps, cells / (ms * 1e-3), 8, dry=False),
            "kernels_ms_per_step": {n: v["total_ms"] / steps for n, v in sorted(k.items())},
            "kernel_launches_per_step": {n: v["launches"] / steps for n, v in sorted(k.items())},
            "finite": bool(torch.isfinite(model.moisture_density.interior).all().item())}


def step_roofline(kernels, steps, cells_per_s, word_bytes, dry=False):
    """Whole-step figure: compulsory bytes of the launches that ran (bench.py's own launch counters x tools/accounting.py words) over the
    step time; the fixed contract figure of SURVEY §8(d) (250 words per cell and step) beside it."""
    launches = {k: v["launches"] / float(steps) for k, v in kernels.items()}
    words = step_compulsory_words(launches, dry=dry)
    achieved = cells_per_s * words * word_bytes / 1e9
    contract = cells_per_s * A_STEP_CONTRACT_WORDS * word_bytes / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "per": "GPU",
            "bytes": "compulsory", "compulsory_words_per_cell_step": words, "compulsory_bytes_per_cell_step": words * word_bytes,
            "contract_words_per_cell_step": A_STEP_CONTRACT_WORDS, "contract_bytes_per_cell_step": A_STEP_CONTRACT_WORDS * word_bytes,
            "contract_achieved": contract,
            # a fraction above 1 only says that the fused kernels no longer move the reference's unfused array lists: reported as a number of
            # words saved, not as a fraction of the roof
            "contract_frac": (contract / HBM_PEAK_GBS) if contract <= HBM_PEAK_GBS else None,
            "contract_note": None if contract <= HBM_PEAK_GBS else
            f"the contract figure ({A_STEP_CONTRACT_WORDS} words per cell and step) priced at this step rate would exceed the 8 TB/s roof "
            f"({contract / HBM_PEAK_GBS:.3f}): the fused step moves {words:.0f} compulsory words instead"}


def compressible_milestone(bz, device, steps=2, substep_float32=False):
    """Second milestone, reported beside the headline metric, never as `value`: the compressible
    split-explicit WS-RK3 step (acoustic substep loop) on a 512 x 512 x 256 bubble, Float64, dt = 1 s.
    substep_float32: the same model with substep_floattype = Float32 (acoustic_substepping.jl:199-235): the substepper's ten working
    fields stored as Float32, arithmetic and every model field Float64 — reported under `substep_floattype_float32`.
    The bubble is dry (q^t = 0): the step takes the dry path (no moisture tendency; since round 6 no time-average accumulators in stages
    1 - 2, whose only reader is that tendency) — `dry_path`; the same model with vapour set is timed beside it (`moist_variant`)."""
    import torch
    Nx, Ny, Nz = 512, 512, 256
    grid = bz.RectilinearGrid((Nx, Ny, Nz), x=EXTENT[0], y=EXTENT[1], z=EXTENT[2])
    dyn = bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(), surface_pressure=1e5, reference_potential_temperature=300.0)
    m = bz.CompressibleAtmosphereModel(grid, dyn, advection=bz.WENO(order=5), device=device,
                                       substep_floattype=np.float32 if substep_float32 else None)
    c = m.thermodynamic_constants
    Rd, cpd, g = 8.314462618 / c.dry_air_molar_mass, c.dry_air_heat_capacity, c.gravitational_acceleration

    def theta(x, y, z):
        return 300.0 + 10.0 * np.maximum(0.0, 1.0 - np.sqrt(x ** 2 + y ** 2 + (z - 3000.0) ** 2) / 2000.0) + 0 * z

    def rho(x, y, z):
        ex = 1.0 - g * z / (cpd * 300.0)
        return 1e5 * ex ** (cpd / Rd) / (Rd * theta(x, y, z) * ex)

    m.set(ρ=rho, θ=theta, u=0.0, v=0.0, w=0.0, qᵗ=0.0)
    cells = Nx * Ny * Nz
    nsub = [m.stage_substeps(1.0, b)[0] for b in (1 / 3, 1 / 2, 1.0)]

    def timed(dry):
        m.time_step(1.0)
        m.profile_reset()
        m.profile_enable(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m.time_step(1.0)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        m.profile_enable(False)
        prof = m.profile()
        sub_ms = sum(prof[k][0] for k in prof if k.startswith("acoustic_horizontal") or k.startswith("acoustic_column")) / steps
        per_sub = sub_ms / sum(nsub)
        kernels = kernel_table(prof)
        # the substep pair (forward sweep with the horizontal step + backward sweep) priced in compulsory words (tools/accounting.py:
        # acoustic_substep_words — 22 + 10, less the accumulators a dry stage 1 / 2 skips and the <u>, <v> words of the substeps that leave
        # them to their pair's second substep; the reference's unfused kernel list moves 58 contract words per substep); with
        # substep_floattype = Float32 the working-field words are 4 bytes and every substep accumulates
        wf, wb = acoustic_substep_words(nsub, dry, pair_avg=not substep_float32)
        fwd = kernels.get("acoustic_horizontal+column_forward", {}).get("avg_ms", 0.0)
        t_f, src_f = load_traffic(ROOT, "acoustic_horizontal+column_forward")
        t_b, _ = load_traffic(ROOT, "acoustic_column_backward")
        pair_traffic = (t_f + t_b) if (t_f and t_b and not substep_float32 and not dry) else None
        sub_bytes = ((wf + wb) * 8 - (4 * 16 if substep_float32 else 0)) * cells      # f32 storage: 16 of the words are working-field words
        sub_roof = {"bound": "hbm", "kernel": "acoustic substep (column forward + backward)", "achieved": sub_bytes / (per_sub * 1e-3) / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sub_bytes / (per_sub * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "traffic": pair_traffic, "traffic_source": src_f if pair_traffic else None,
                    "traffic_over_compulsory": pair_traffic / sub_bytes if pair_traffic else None,
                    "bytes": "compulsory", "compulsory_words_per_cell": wf + wb, "compulsory_bytes_per_substep": sub_bytes,
                    "contract_words_per_cell": ACOUSTIC_SUBSTEP_CONTRACT_WORDS, "forward_sweep_ms": fwd,
                    "forward_sweep_frac": (wf * 8 - (4 * 10 if substep_float32 else 0)) * cells / (fwd * 1e-3) / 1e9 / HBM_PEAK_GBS if fwd else None}
        return {"value": cells / (ms * 1e-3), "unit": "cells/s", "ms_per_step": ms, "acoustic_substep_ms": per_sub,
                "substep_loop_ms_per_step": sub_ms, "non_substep_ms_per_step": ms - sub_ms, "acoustic_substep_roofline": sub_roof,
                "kernels_ms_per_step": {k: v["total_ms"] / steps for k, v in sorted(kernels.items())}, "_kernels": kernels}

    dry = os.environ.get("BZ_NO_DRY_SHORTCUT") is None
    r = timed(dry)
    kernels = r.pop("_kernels")
    out = {"metric": "grid-cells advanced/sec, compressible split-explicit WS-RK3 step", "grid": [Nx, Ny, Nz], "dt": 1.0, "substeps_per_stage": nsub,
           "dry_path": dry, **r,
           "roofline": roofline_block("acoustic_horizontal+column_forward", kernels["acoustic_horizontal+column_forward"]["avg_ms"], cells, 8,
                                      words=acoustic_substep_words(nsub, dry, pair_avg=not substep_float32)[0]) if not substep_float32 else None,
           "finite": bool(torch.isfinite(m.velocities["w"].interior).all().item())}
    # like for like: the same model with vapour set (moisture tendency evaluated, every stage accumulates the time-averaged velocities)
    try:
        m.set(qᵗ=lambda x, y, z: 5e-3 * np.exp(-z / 2500.0) + 0 * x + 0 * y)
        mv = timed(False)
        mv.pop("_kernels")
        mv["moisture"] = "q^t = 5e-3 exp(-z / 2500 m)"
        mv["finite"] = bool(torch.isfinite(m.velocities["w"].interior).all().item())
        out["moist_variant"] = mv
    except Exception as exc:      # noqa: BLE001
        out["moist_variant"] = {"error": repr(exc)}
    del m
    torch.cuda.empty_cache()
    if not substep_float32:
        try:
            r = compressible_milestone(bz, device, steps, substep_float32=True)
            out["substep_floattype_float32"] = {k: r[k] for k in ("value", "ms_per_step", "acoustic_substep_ms", "substep_loop_ms_per_step",
                                                                  "non_substep_ms_per_step", "acoustic_substep_roofline", "finite", "dry_path")}
            out["substep_floattype_float32"]["moist_variant_ms_per_step"] = (r.get("moist_variant") or {}).get("ms_per_step")
            out["substep_floattype_float32"]["tolerance"] = "2e-6 of the field scale after three steps against the Float64 oracle (tests/test_gpu_compressible.py)"
        except Exception as exc:      # noqa: BLE001
            out["substep_floattype_float32"] = {"error": repr(exc)}
    return out


def milestone_in_fresh_process(bz, device):
    """The compressible milestone timed in a process of its own (python bench.py --milestone-only).  The same library on the same box steps
    8 - 10 % slower when its ~45 arrays of 0.56 GB are allocated after the anelastic legs of this process have allocated and freed theirs
    (1.1 GB arrays) than in a fresh process — round 5 saw the same as 157 - 160 ms standalone against 166 - 176 ms inside this command, with
    the difference sitting in the plain streaming kernels: what differs is where the driver places the arrays, not the
    code measured.  A fresh process gives the placement every stand-alone user of the library gets.  BZ_BENCH_MILESTONE_INPROCESS=1 (or a
    child that fails) keeps the measurement in this process; the line says which it was."""
    if os.environ.get("BZ_BENCH_MILESTONE_INPROCESS") != "1":
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--milestone-only"], capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode == 0 and line:
                out = json.loads(line[-1])
                out["measured_in"] = "a fresh process (python bench.py --milestone-only)"
                return out
        except Exception:      # noqa: BLE001
            pass
    out = compressible_milestone(bz, device)
    out["measured_in"] = "the process of the headline run, after its models were freed"
    return out


def float32_run(bz, device, N, steps=10, warmup=2, single_steps=False):
    """The same workload with eltype(grid) = Float32 (lib/libbreeze_hip_f32.so) — what the reference's own GPU benchmarks run
    (benchmarking/src/convective_boundary_layer.jl:59,70).  Reported beside the Float64 headline, never as `value`
    ("Float32 run reported separately"); every roofline block in 4-byte words (tools/accounting.py)."""
    import torch
    grid = bz.RectilinearGrid((N, N, N), x=EXTENT[0], y=EXTENT[1], z=EXTENT[2], float_type=np.float32)
    ref = bz.ReferenceState(grid, surface_pressure=101325, potential_temperature=300)
    m = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5), device=device)
    m.set(θ=bubble)
    for _ in range(warmup):
        m.time_step(1.0)
    m.profile_reset()
    m.profile_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if single_steps:
        for _ in range(steps):
            m.time_step(1.0)
    else:
        m.time_steps(1.0, steps, diagnose_last=True)      # the multi-step seam, as the Float64 headline run
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    m.profile_enable(False)
    kernels = kernel_table(m.profile())
    cells = N ** 3
    rate = cells * steps / el
    out = {"dtype": "f32", "value": rate, "unit": "cells/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "grid": [N, N, N],
           "roofline": dominant_roofline(kernels, cells, 4, f32=True, dry=took_dry_path(kernels)),
           "step_roofline": step_roofline(kernels, steps, rate, 4, dry=took_dry_path(kernels)),
           "dry_path": took_dry_path(kernels),
           "kernels_ms_per_step": {k: v["total_ms"] / steps for k, v in sorted(kernels.items())},
           "kernel_launches_per_step": {k: v["launches"] / steps for k, v in sorted(kernels.items())},
           "finite": bool(torch.isfinite(m.momentum["ρw"].parent).all().item()),
           "tolerance_vs_float64_oracle": "1e-4 after three steps, 2e-5 per tendency (tests/test_float32.py)"}
    del m
    torch.cuda.empty_cache()
    return out


def _mem_available_gb():
    try:
        with open("/proc/meminfo") as fh:
            for line in fh:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) / 1048576.0
    except OSError:
        pass
    return 0.0


def cpu_baseline_full_size(n=512):
    """The same oracle on the HEADLINE size (512^3, ~45 GB of host arrays) when the box has the memory: all host cores up to 128
    threads, one warm-up step and one timed step (~20-40 s of CPU work).  Reported beside the bounded 256^3 sample."""
    if _mem_available_gb() < 96.0:
        return {"skipped": f"MemAvailable {_mem_available_gb():.0f} GB < 96 GB"}
    from oracle import oracle as orc
    cores = len(os.sched_getaffinity(0))
    threads = max(1, min(128, cores))
    os.environ["OMP_NUM_THREADS"] = str(threads)
    g = orc.Grid((n, n, n), x=EXTENT[0], y=EXTENT[1], z=EXTENT[2])
    m = orc.OracleModel(g, potential_temperature=300.0)
    m.fft_workers = threads
    m.set(theta=bubble)
    m.time_step(1.0)
    t0 = time.perf_counter()
    m.time_step(1.0)
    dt = time.perf_counter() - t0
    return {"value": n ** 3 / dt, "unit": "cells/s", "cores": threads, "host_cores_available": cores, "kind": "port", "s_per_step": dt,
            "sample": f"dry thermal bubble {n}^3 Float64 (the headline size), 1 step after 1 warm-up, C/OpenMP oracle + pocketfft on {threads} threads"}


def cpu_baseline(n, budget_s=25.0):
    """The CPU oracle ("port": this repo's C/OpenMP restatement, not Breeze CPU() — Julia is not installed)
    timed on this box's host cores on a bounded sample of the same workload: the bubble at n^3 (default 256^3, an eighth
    of the 512^3 domain at the same spacing ratio), as many steps as fit in `budget_s` seconds after one warm-up step.
    Threads: one per physical core up to 64 (the stencil kernels stop scaling beyond that on a 16 M-cell grid, and a
    256-thread team on a 2 M-cell grid measured *slower* in round 1); the horizontal transforms run pocketfft on the same
    number of threads.  Both the all-cores and the chosen count are printed."""
    from oracle import oracle as orc
    cores = len(os.sched_getaffinity(0))
    threads = max(1, min(64, cores // 2 if cores >= 16 else cores))
    os.environ["OMP_NUM_THREADS"] = str(threads)
    g = orc.Grid((n, n, n), x=EXTENT[0], y=EXTENT[1], z=EXTENT[2])
    m = orc.OracleModel(g, potential_temperature=300.0)
    m.fft_workers = threads
    m.set(theta=bubble)
    t0 = time.perf_counter()
    m.time_step(1.0)                       # warm-up (also the first-step update_state)
    first = time.perf_counter() - t0
    steps = int(max(2, min(20, budget_s / max(first, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(steps):
        m.time_step(1.0)
    dt = (time.perf_counter() - t0) / steps
    return {"value": n ** 3 / dt, "unit": "cells/s", "cores": threads, "host_cores_available": cores, "kind": "port",
            "s_per_step": dt,
            "sample": f"dry thermal bubble {n}^3 Float64, {steps} steps after 1 warm-up, "
                      f"C/OpenMP oracle on {threads} threads + pocketfft on {threads} threads"}


def cbl_run(args, bz, device):
    """--workload cbl: the reference's own published benchmark case (BreezeBenchmarks), one GPU.  Physics and set-up of
    /root/reference/benchmarking/src/convective_boundary_layer.jl:59-185 (host mirror: breeze.jl_amd/benchmarks.py): dry convective
    boundary layer, AnelasticDynamics, FPlane + geostrophic forcing + u* drag + 0.35 K m/s surface heat flux, halo 5, Float32 unless
    --cbl-float64, WENO(order = --cbl-order), dt = 0.05 s (benchmarking/src/utils.jl:18).  The CI protocol is 5 warm-up + 150 timed
    steps on 256x256x128, 512x512x256 and 768x768x256 with WENO5 and WENO9 (.github/workflows/Benchmarks.yml:34-45);
    tools/gpu_cbl.sh runs that matrix.  The metric is the reference's grid_points_per_second (benchmarking/src/utils.jl:141).
    An NVIDIA L4 produced the external BreezeBenchmarks table: a context number, not a target, and none of it is in BASELINE.md."""
    import torch
    Nx, Ny, Nz = (int(v) for v in args.cbl_size.lower().split("x"))
    f32 = not args.cbl_float64
    dt = 0.05
    pbb = args.cbl_topology == "PBB"      # (Periodic, Bounded, Bounded): benchmarking/run_benchmarks.jl:130 (operator-by-operator tier, cosine transform in y)
    m = bz.benchmarks.convective_boundary_layer((Nx, Ny, Nz), float_type=np.float32 if f32 else np.float64,
                                                advection=bz.WENO(order=args.cbl_order), device=device,
                                                topology=(bz.Periodic, bz.Bounded if pbb else bz.Periodic, bz.Bounded))
    for _ in range(args.warmup):
        m.time_step(dt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if args.single_steps:
        for _ in range(args.steps):
            m.time_step(dt)
    else:
        m.time_steps(dt, args.steps, diagnose_last=True)      # many_time_steps! (benchmarking/src/timestepping.jl:11-16)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # per-kernel times from a second, untimed pass: the HIP events around ~40 launches per step cost ~0.2 ms per step, 11 % of the step at
    # 256 x 256 x 128 (1.86 ms without them, 2.09 ms with them)
    psteps = min(args.steps, 20)
    m.profile_reset()
    m.profile_enable(True)
    if args.single_steps:
        for _ in range(psteps):
            m.time_step(dt)
    else:
        m.time_steps(dt, psteps, diagnose_last=True)
    torch.cuda.synchronize()
    m.profile_enable(False)
    cells, word = Nx * Ny * Nz, 4 if f32 else 8
    kernels = kernel_table(m.profile())
    # PMC traffic of this workload at its own grid (profiles/r06_pmc_traffic_cbl_weno<order>.json: the 512 x 512 x 256 Float32 PPB case)
    own = (Nx, Ny, Nz) == (512, 512, 256) and f32 and args.cbl_topology == "PPB"
    roofline = dominant_roofline(kernels, cells, word, f32=f32, with_traffic=own or (Nx, Ny, Nz) == (512, 512, 512), dry=took_dry_path(kernels),
                                 traffic_tag=("cbl_weno%d" % args.cbl_order) if own else None)
    rate = cells * args.steps / elapsed
    out = {"metric": "grid points per second (time_step!), convective boundary layer benchmark case",
           "value": rate, "unit": "cells/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32" if f32 else "f64", "data": "synthetic",
           "config": {"workload": f"BreezeBenchmarks convective_boundary_layer {Nx}x{Ny}x{Nz} (benchmarking/src/convective_boundary_layer.jl): "
                                  f"AnelasticDynamics, WENO{args.cbl_order}, halo 5, topology {args.cbl_topology}, FPlane + geostrophic forcing + u* drag + surface heat flux, "
                                  f"{'Float32' if f32 else 'Float64'}, dt={dt}s", "grid": [Nx, Ny, Nz], "dt": dt, "parallelism": "single GPU"},
           "roofline": roofline,
           "step_roofline": step_roofline(kernels, psteps, rate, word, dry=took_dry_path(kernels)),
           "dry_path": took_dry_path(kernels),
           "kernels_ms_per_step": {k: v["total_ms"] / psteps for k, v in sorted(kernels.items())},
           "kernel_launches_per_step": {k: v["launches"] / psteps for k, v in sorted(kernels.items())},
           "kernel_times": f"HIP events of a second pass of {psteps} steps after the timed region",
           "finite": bool(torch.isfinite(m.momentum["ρw"].parent).all().item()),
           "w_max": float(m.velocities["w"].interior.abs().max().item())}
    print(json.dumps(out), flush=True)
    return 0


def tendency_run(args, bz, device):
    """--workload scalar_tendency | model_tendency: the reference's tendency micro-benchmarks (BASELINE.md row 4;
    .github/workflows/Benchmarks.yml:88-109: 256x256x128, Float32, WENO 5 / 7 / 9, 320 repeats).
      scalar_tendency  benchmarking/src/scalar_tendency.jl:16-61 — one launch of Gc = -div_Uc(c) with U = (1, 0, 0),
                       c = sin(2 pi x) sin(2 pi y) on a unit box, halo = cld(order + 1, 2); here bz_compute_scalar_tendency of the
                       anelastic model on the same box (the reference density of a 1 m deep box is constant to 1e-4).
      model_tendency   benchmarking/src/model_tendency.jl:25-45 — compute_tendencies!(model) of a CompressibleDynamics model at rest
                       (theta = 300, rho = 1) on a 1 km box; here bz_compute_slow_tendencies.
    The metric is the reference's grid_points_per_second of one evaluation (tendency_profiling.jl:37-124)."""
    import torch
    Nx, Ny, Nz = (int(v) for v in args.tend_size.lower().split("x"))
    f32 = not args.tend_float64
    ft = np.float32 if f32 else np.float64
    order = args.tend_order
    h = max(3, (order + 2) // 2)      # cld(order + 1, 2); this library's kernels want >= 3
    scalar = args.workload == "scalar_tendency"
    L = 1.0 if scalar else 1e3
    grid = bz.RectilinearGrid((Nx, Ny, Nz), x=(0, L), y=(0, L), z=(0, L), halo=(h, h, h), float_type=ft)
    if scalar:
        m = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, potential_temperature=300.0)),
                               advection=bz.WENO(order=order), device=device)
        m.velocities["u"].parent.fill_(1.0)
        ccc = (bz.Center, bz.Center, bz.Center)
        c, Gc = bz.Field(grid, ccc, device), bz.Field(grid, ccc, device)
        c.set_interior(lambda x, y, z: np.sin(2 * np.pi * x) * np.sin(2 * np.pi * y) + 0 * z)
        bz.fill_halo_regions_(m, c)
        run = lambda: bz.compute_scalar_tendency_(m, c, Gc)      # noqa: E731
        words, kernel = 5, "scalar_tendency"      # u, v, w, c read; Gc written
        finite = lambda: bool(torch.isfinite(Gc.interior).all().item())      # noqa: E731
    else:
        dyn = bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(), surface_pressure=1e5, reference_potential_temperature=300.0)
        m = bz.CompressibleAtmosphereModel(grid, dyn, advection=bz.WENO(order=order), device=device)
        m.set(ρ=1.0, θ=300.0, u=0.0, v=0.0, w=0.0, qᵗ=0.0)
        run = lambda: bz.compressible.compute_slow_tendencies_(m)      # noqa: E731
        words, kernel = None, "compute_slow_tendencies"
        finite = lambda: True      # noqa: E731
    for _ in range(max(1, args.warmup)):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    psteps = min(args.steps, 50)      # per-kernel HIP events in a second, untimed pass
    m.profile_reset()
    m.profile_enable(True)
    for _ in range(psteps):
        run()
    torch.cuda.synchronize()
    m.profile_enable(False)
    cells, word = Nx * Ny * Nz, 4 if f32 else 8
    kernels = {k: {"avg_ms": ms / n, "launches": n, "total_ms": ms} for k, (ms, n) in m.profile().items() if n}
    roofline = roofline_block(kernel, kernels[kernel]["avg_ms"], cells, word, words=words) if (words and kernel in kernels) else None
    out = {"metric": f"grid points per second, one {'scalar tendency' if scalar else 'compute_tendencies!'} evaluation (BreezeBenchmarks {args.workload})",
           "value": cells * args.steps / elapsed, "unit": "cells/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32" if f32 else "f64", "data": "synthetic",
           "config": {"workload": f"BreezeBenchmarks {args.workload} {Nx}x{Ny}x{Nz}, WENO{order}, halo {h}, {'Float32' if f32 else 'Float64'} "
                                  f"(benchmarking/src/{args.workload}.jl)", "grid": [Nx, Ny, Nz], "parallelism": "single GPU"},
           "roofline": roofline,
           "kernels_ms_per_evaluation": {k: v["total_ms"] / psteps for k, v in sorted(kernels.items())},
           "finite": finite()}
    print(json.dumps(out), flush=True)
    return 0


def supercell_model(bz, size, device, order=5, f32=False, slab=None):
    """BASELINE configs[4]: the splitting-supercell shape — CompressibleDynamics + SplitExplicitTimeDiscretization defaults +
    DCMIP2016 Kessler microphysics on the example's 168 km x 168 km x 20 km box (/root/reference/examples/splitting_supercell.jl:88-96),
    moist column + 3 K warm bubble + sheared wind, initial condition set.  slab = (rank, world, transport): this rank's y-slab."""
    Nx, Ny, Nz = size
    gkw = {"float_type": np.float32} if f32 else {}
    if order != 5:
        gkw["halo"] = (5, 5, 5)
    G = bz.RectilinearGrid((Nx, Ny, Nz), x=(0.0, 168e3), y=(0.0, 168e3), z=(0.0, 20e3), **gkw)
    dyn = bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(), surface_pressure=1e5, reference_potential_temperature=300.0)
    mkw = dict(thermodynamic_constants=bz.ThermodynamicConstants(saturation_vapor_pressure=bz.TetensFormula()),
               microphysics=bz.DCMIP2016KesslerMicrophysics())
    if slab:
        rank, world, transport = slab
        m = bz.compressible.SlabCompressibleModel(G, rank, world, dyn, advection=bz.WENO(order=order), device=device, transport=transport, **mkw)
    else:
        m = bz.CompressibleAtmosphereModel(G, dyn, advection=bz.WENO(order=order), device=device, **mkw)
    Hz = m.grid.Hz
    col = m.dynamics.reference_state.density[Hz:Hz + Nz][:, None, None]

    def theta(x, y, z):      # neutral column + the example's bubble: 3 K at (84 km, 84 km, 1.5 km), radii 10 km x 1.5 km
        r = np.sqrt(((x - 84e3) / 10e3) ** 2 + ((y - 84e3) / 10e3) ** 2 + ((z - 1500.0) / 1500.0) ** 2)
        return 300.0 + 3.0 * np.cos(np.pi / 2 * np.minimum(r, 1.0)) ** 2

    m.set(ρ=lambda x, y, z: col * 300.0 / theta(x, y, z), θ=theta, v=0.0, w=0.0,
          u=lambda x, y, z: 0.0015 * np.minimum(z, 5e3) + 0 * x + 0 * y,
          qᵗ=lambda x, y, z: 0.014 * np.exp(-z / 2500.0) + 0 * x + 0 * y)
    return m


def config4_run(args, bz, rank, world, dist, device, fail):
    """--workload config4: BASELINE configs[4], the splitting-supercell shape — CompressibleDynamics, split-explicit WS-RK3 with
    acoustic substeps, DCMIP2016 Kessler microphysics, 512 x 512 x 128 cells on the example's 168 km x 168 km x 20 km box
    (/root/reference/examples/splitting_supercell.jl:88-96), moist column + 3 K warm bubble + sheared wind, Float64, dt = 2 s —
    on one GPU or split into `world` y-slabs with the library-owned communicator (bz_comm.hip: per-substep halo exchange of
    (rho theta)' and (rho v)', per-stage exchange of the rest).  A second milestone: never the headline `value` of the default run."""
    import torch
    Nx, Ny, Nz, dt = 512, 512, 128, 2.0
    f32 = bool(getattr(args, "config4_float32", False))      # the example's own precision (splitting_supercell.jl:86), single GPU
    order = int(getattr(args, "config4_order", 5))           # splitting_supercell.jl:279 uses WENO(order = 9): generic kernels, 5-cell halos
    slabs = world > 1 or args.slab
    transport = args.transport
    try:
        m = supercell_model(bz, (Nx, Ny, Nz), device, order=order, f32=f32, slab=(rank, world, transport) if slabs else None)
        for _ in range(max(1, args.warmup)):
            m.time_step(dt)
        m.profile_reset()
        m.profile_enable(True)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        sent0 = m.comm_info()[1] if slabs and transport == "rccl" else 0
        t0 = time.perf_counter()
        for _ in range(args.steps):
            m.time_step(dt)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        m.profile_enable(False)
        finite = bool(torch.isfinite(m.velocities["w"].parent).all().item())
        if dist is not None:
            t = torch.tensor([elapsed, 0.0 if finite else 1.0], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, finite = t[0].item(), t[1].item() == 0.0
    except Exception as exc:      # noqa: BLE001
        fail(f"config4 run failed: {exc!r}")
    if rank == 0:
        prof = {k: v[0] / args.steps for k, v in sorted(m.profile().items())}
        kernels = kernel_table(m.profile())
        nsub = [m.stage_substeps(dt, b)[0] for b in (1 / 3, 1 / 2, 1.0)]
        out = {"metric": "grid-cells advanced/sec (compressible split-explicit + Kessler step), 512x512x128",
               "value": Nx * Ny * Nz * args.steps / elapsed, "unit": "cells/s", "n_gpus": world, "steps": args.steps,
               "warmup": max(1, args.warmup), "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f32" if f32 else "f64", "data": "synthetic",
               "config": {"workload": "BASELINE.json configs[4]: splitting-supercell shape 512x512x128, CompressibleDynamics + "
                                      f"SplitExplicitTimeDiscretization defaults + DCMIP2016 Kessler, WENO{order}, dt=2s" +
                                      (", Float32 (libbreeze_hip_f32.so)" if f32 else ""),
                          "grid": [Nx, Ny, Nz], "dt": dt, "substeps_per_stage": nsub,
                          "parallelism": "single GPU" if not slabs else
                          f"{world} y-slabs of {Nx}x{Ny // world}x{Nz}, halo exchanges over " +
                          ("RCCL inside the C library (bz_comm.hip)" if transport == "rccl" else "torch.distributed (RCCL backend)")},
               "roofline": dominant_roofline(kernels, Nx * (Ny // world) * Nz, 4 if f32 else 8, f32=f32, with_traffic=(world == 1 and not f32 and order == 5),
                                             traffic_tag="config4"),
               "kernels_ms_per_step": prof, "finite": finite,
               "comm_ms_per_step": sum(v for k, v in prof.items() if k.startswith("comm_")),
               "transport": transport if slabs else None,
               "note": "second milestone (SURVEY.md §8 a15-a17); the headline metric is the default workload"}
        if slabs and transport == "rccl":
            out["comm_bytes_sent_per_step_per_gpu"] = (m.comm_info()[1] - sent0) // args.steps
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return 0


# ---------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without a rendezvous environment starts its own ranks
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def error_line(args, message, **extra):
    out = {"metric": METRIC, "value": None, "unit": "cells/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": None, "higher_is_better": True, "scaling": args.scaling or ("strong" if args.gpus > 1 else "weak"), "vs_baseline": None, "dtype": "f64",
           "data": "synthetic", "config": {"workload": args.workload, "parallelism": f"{args.gpus} y-slabs"},
           "error": message}
    out.update(extra)
    return json.dumps(out)


def launch_self(args, argv):
    """Re-execute this script under torch.distributed.run with one rank per GPU and relay the ranks' JSON line."""
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, env=env, text=True, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)        # exactly the process group started above
        except ProcessLookupError:
            pass
        out, _ = proc.communicate()
        print(error_line(args, f"ranks did not finish within {args.launch_timeout} s (hang in a collective?)"))
        return 1
    line = None
    for ln in (out or "").splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                json.loads(ln)
                line = ln
            except ValueError:
                pass
    if line is None:
        print(error_line(args, f"ranks exited with code {proc.returncode} without a result line"))
        return proc.returncode or 1
    print(line)
    return 0 if (proc.returncode == 0 and "\"error\"" not in line) else (proc.returncode or 1)


def comm_selfcheck(decomp, device, Nz=4):
    """Pre-flight check of the transport on small tensors, before the model commits to it: one y-halo exchange and both
    spectral transposes with rank-coded contents, verified against what the neighbours must have sent.  Raises on mismatch."""
    import torch
    d = decomp
    Sy, Sx = d.Ny + 2 * d.Hy, d.Nx + 6
    f = torch.zeros((Nz, Sy, Sx), dtype=torch.float64, device=device)
    rows = torch.arange(d.Ny, dtype=torch.float64, device=device)
    f[:, d.Hy:d.Hy + d.Ny, :] = (1000.0 * d.rank + rows)[None, :, None]
    d.exchange_y_halos([f])
    lo = 1000.0 * d.lower + torch.arange(d.Ny - d.Hy, d.Ny, dtype=torch.float64, device=device)
    hi = 1000.0 * d.upper + torch.arange(0, d.Hy, dtype=torch.float64, device=device)
    if not (torch.equal(f[0, :d.Hy, 0], lo) and torch.equal(f[-1, d.Hy + d.Ny:, -1], hi)):
        raise RuntimeError(f"rank {d.rank}: y-halo exchange delivered wrong rows")
    # transposes: R[k, j, kx] = global row + 1000 kx  ->  S[k, kx_local, J] must equal J + 1000 (kx0 + kx_local)
    jj = (d.rank * d.Ny + torch.arange(d.Ny, device=device, dtype=torch.float64))[None, :, None]
    kk = torch.arange(d.nxh, device=device, dtype=torch.float64)[None, None, :]
    R = (jj + 1000.0 * kk).expand(Nz, d.Ny, d.nxh).to(torch.complex128).contiguous()
    save_Nz, d.Nz = d.Nz, Nz
    try:
        S = d.to_kx_slabs(R)
        J = torch.arange(d.Ny_global, device=device, dtype=torch.float64)[None, None, :]
        kx = d.kx0 + torch.arange(d.nkx, device=device, dtype=torch.float64)[None, :, None]
        want = torch.where(kx < d.nxh, J + 1000.0 * kx, torch.zeros((), dtype=torch.float64, device=device)).expand(Nz, d.nkx, d.Ny_global)
        if not torch.equal(S.real, want):
            raise RuntimeError(f"rank {d.rank}: transpose to kx-slabs delivered wrong blocks")
        back = d.to_y_slabs(S)
        if not torch.equal(back[:, :, :d.nxh].real, R.real):
            raise RuntimeError(f"rank {d.rank}: transpose back to y-slabs delivered wrong blocks")
    finally:
        d.Nz = save_Nz


def library_halo_selfcheck(model):
    """Rank-coded rows through bz_comm_exchange_y_halos (the library's own transport), verified against what the ring neighbours
    must have delivered.  Raises on mismatch."""
    import ctypes as C
    import torch
    g, r, W = model.grid, model.rank, model.world
    f = model.temperature.parent
    keep = f.clone()
    rows = torch.arange(g.Ny, dtype=torch.float64, device=f.device)
    f.zero_()
    f[:, g.Hy:g.Hy + g.Ny, :] = (1000.0 * r + rows)[None, :, None]
    ptrs = (C.c_void_p * 1)(f.data_ptr())
    levels = (C.c_int32 * 1)(f.shape[0])
    model._check(model._lib.bz_comm_exchange_y_halos(model._ctx, ptrs, levels, 1), "bz_comm_exchange_y_halos")
    model.synchronize()
    lo = 1000.0 * ((r - 1) % W) + torch.arange(g.Ny - g.Hy, g.Ny, dtype=torch.float64, device=f.device)
    hi = 1000.0 * ((r + 1) % W) + torch.arange(0, g.Hy, dtype=torch.float64, device=f.device)
    ok = torch.equal(f[0, :g.Hy, 0], lo) and torch.equal(f[-1, g.Hy + g.Ny:, -1], hi)
    f.copy_(keep)
    if not ok:
        raise RuntimeError(f"rank {r}: the library's y-halo exchange delivered wrong rows")


def preflight(args, bz, rank, world, dist, device, transport="rccl", per_rank=64, steps=2, tol=1e-10):
    """Everything a multi-rank run needs, proven on a small problem BEFORE any timed region:
      1. the slab model builds on every rank — communicator bootstrap inside the C library (ncclCommInitRank over the id broadcast
         through torch.distributed) or the torch.distributed transport;
      2. a rank-coded y-halo exchange through that transport delivers the ring neighbours' rows;
      3. `steps` steps of a per_rank^3-per-rank dry bubble (the lean distributed step: halo exchanges under the interior tiles, both
         all-to-alls of every pressure solve, the phi row) agree with the SAME global domain stepped by the single-GPU seam on this
         rank's own GPU to `tol` of the field scale — every rank compares its own slab.
    Returns a summary dict; raises on any failure (the caller turns that into a JSON error line and a non-zero exit code)."""
    import torch
    from breeze_jl_amd.distributed import SlabAtmosphereModel
    n = per_rank
    G = (n, n * world, n)
    Ly = (EXTENT[1][1] - EXTENT[1][0]) * world
    ggrid = bz.RectilinearGrid(G, x=EXTENT[0], y=(EXTENT[1][0], EXTENT[1][0] + Ly), z=EXTENT[2])

    def ic(x, y, z):      # one bubble per 20 km of y, off-centre so that it straddles a slab edge for world > 1
        yy = np.mod(y - EXTENT[1][0] + 7.5e3, EXTENT[1][1] - EXTENT[1][0]) + EXTENT[1][0]
        return bubble(x, yy, z)

    t0 = time.perf_counter()
    m = SlabAtmosphereModel(ggrid, rank, world, advection=bz.WENO(order=5), surface_pressure=101325, potential_temperature=300,
                            device=device, transport=transport)
    if transport == "rccl":
        library_halo_selfcheck(m)
    elif world > 1:
        comm_selfcheck(m.decomp, device)
    m.set(θ=ic, u=3.0, v=-2.0)
    m.time_steps(1.0, steps, diagnose_last=True)
    m.synchronize() if hasattr(m, "synchronize") else torch.cuda.synchronize()
    ref = bz.AtmosphereModel(ggrid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(ggrid, surface_pressure=101325, potential_temperature=300)),
                             advection=bz.WENO(order=5), device=device)
    ref.set(θ=ic, u=3.0, v=-2.0)
    ref.time_steps(1.0, steps, diagnose_last=True)
    ref.synchronize()
    worst = 0.0
    pairs = (("ρu", m.momentum["ρu"], ref.momentum["ρu"]), ("ρv", m.momentum["ρv"], ref.momentum["ρv"]), ("ρw", m.momentum["ρw"], ref.momentum["ρw"]),
             ("ρθ", m.potential_temperature_density, ref.potential_temperature_density), ("T", m.temperature, ref.temperature))
    mom = max(float(ref.momentum[k].interior.abs().max()) for k in ("ρu", "ρv", "ρw"))
    for name, a, b in pairs:
        mine = a.interior
        want = b.interior[:, rank * n:(rank + 1) * n, :]
        scale = mom if name.startswith("ρ") and name != "ρθ" else float(want.abs().max())
        err = float((mine - want).abs().max()) / max(scale, 1e-30)
        worst = max(worst, err)
        if not (err <= tol):
            raise RuntimeError(f"rank {rank}: preflight parity of {name} against the single-GPU step: {err:.3e} > {tol:g}")
    if dist is not None:
        t = torch.tensor([worst], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        worst = t.item()
    out = {"status": "ok", "transport": transport, "world": world, "grid": list(G), "steps": steps, "tolerance": tol,
           "max_relative_deviation_from_single_gpu": worst, "halo_selfcheck": "ok", "seconds": time.perf_counter() - t0}
    del m, ref
    torch.cuda.empty_cache()
    return out


def problem(args, world):
    """Global grid size, per-rank slab and the label of the workload."""
    N = args.size
    if args.workload == "config3":
        G = (1024, 1024, 512)
        label = ("dry thermal bubble 1024x1024x512 RectilinearGrid (Periodic,Periodic,Bounded), halo 3, AnelasticDynamics + WENO5 + "
                 "SSP-RK3, Float64, dt=1s, y-slabs (BASELINE.json configs[3])")
        scaling = "strong"
    elif (args.scaling or "strong") == "strong" or world == 1:
        G = (N, N, N)
        label = (f"dry thermal bubble {N}^3 RectilinearGrid (Periodic,Periodic,Bounded), halo 3, AnelasticDynamics + WENO5 + "
                 "SSP-RK3, Float64, dt=1s (BASELINE.json configs[1])")
        scaling = "strong" if world > 1 else (args.scaling or "weak")      # one GPU: the two notions coincide; the contract's default label
    else:
        G = (N, N * world, N)
        label = (f"dry thermal bubble {N}^3 per GPU: {N}x{N * world}x{N} RectilinearGrid (Periodic,Periodic,Bounded), halo 3, "
                 "AnelasticDynamics + WENO5 + SSP-RK3, Float64, dt=1s (BASELINE.json configs[1] per rank)")
        scaling = "weak"
    if G[1] % world:
        raise ValueError(f"Ny = {G[1]} is not divisible by {world} ranks")
    return G, label, scaling


def run_rank(args):
    import torch
    import breeze_jl_amd as bz

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    cpu_selftest = args.selftest_launcher
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        timeout = datetime.timedelta(seconds=args.collective_timeout)
        if cpu_selftest:
            dist.init_process_group("gloo", timeout=timeout)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"), timeout=timeout)
    device = "cpu" if cpu_selftest else f"cuda:{local_rank}"
    if not cpu_selftest and os.environ.get("BZ_BENCH_OWN_STREAM") == "1":
        # experiments: run on a non-blocking stream of our own instead of the legacy default stream (which synchronises implicitly with
        # every blocking stream, e.g. one created with a CU mask)
        torch.cuda.set_device(local_rank)
        torch.cuda.set_stream(torch.cuda.Stream())

    def fail(message):
        """Every rank reports; rank 0 prints the JSON error line.  Never falls back to a different measurement."""
        print(f"[bench rank {rank}] {message}", file=sys.stderr, flush=True)
        if rank == 0:
            print(error_line(args, message), flush=True)
        os._exit(1)          # peers may be blocked in a collective: leave without the process-group teardown

    if rank == 0 and world > 1:
        # torch.distributed.run sends SIGTERM to the surviving ranks when one dies: still leave a result line
        signal.signal(signal.SIGTERM, lambda *_: (print(error_line(args, "terminated by the launcher: another rank failed"),
                                                        flush=True), os._exit(1)))

    if cpu_selftest:
        # launcher self-test (tests/test_bench_launcher.py): rendezvous, the slab communication pattern on CPU tensors under gloo, the
        # resolved defaults of a multi-rank run (scaling, grid, transport), the preflight's control flow (its communication check runs
        # on CPU tensors; --selftest-fail injects a failure on the last rank) and the result / error line — no model, no GPU
        from breeze_jl_amd.distributed import SlabDecomposition
        d = SlabDecomposition(20, 6, 4, 3, rank, world)
        try:
            comm_selfcheck(d, "cpu")
            if args.selftest_fail and rank == world - 1:
                raise RuntimeError("injected preflight failure (--selftest-fail)")
        except Exception as exc:   # noqa: BLE001
            fail(f"preflight failed: {exc!r}")
        ok = torch.ones(1, dtype=torch.int32)
        if dist is not None:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            dist.barrier()
        if rank == 0:
            G, label, scaling = problem(args, world)
            print(json.dumps({"launcher_selftest": "ok", "n_gpus": world, "backend": "gloo", "scaling": scaling, "grid": list(G),
                              "grid_per_gpu": [G[0], G[1] // world, G[2]], "transport": args.transport,
                              "preflight": {"status": "ok", "mode": "cpu selftest: slab exchange + both transposes on CPU tensors"},
                              "preflight_only": bool(args.preflight)}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return 0

    if args.preflight:      # stand-alone preflight: one JSON line, exit code 0 / 1
        try:
            summary = preflight(args, bz, rank, world, dist, device, transport=args.transport)
        except Exception as exc:      # noqa: BLE001
            fail(f"preflight failed: {exc!r}")
        if rank == 0:
            print(json.dumps({"preflight": summary, "n_gpus": world}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return 0

    if args.workload == "config4":
        return config4_run(args, bz, rank, world, dist, device, fail)
    if args.workload in ("scalar_tendency", "model_tendency"):
        if world != 1:
            fail(f"--workload {args.workload} is a single-GPU micro-benchmark")
        return tendency_run(args, bz, device)
    if args.workload == "cbl":
        if world > 1:
            fail("--workload cbl is the reference's single-GPU benchmark case")
        return cbl_run(args, bz, device)
    dt = 1.0
    G, label, scaling = problem(args, world)
    use_slabs = (world > 1 and not args.replicas) or (world == 1 and args.slab)
    parallelism = "single GPU"
    decomp = None
    if use_slabs:
        from breeze_jl_amd.distributed import SlabAtmosphereModel
        Ly = (EXTENT[1][1] - EXTENT[1][0]) * (G[1] / G[0]) if args.workload != "config3" else (EXTENT[1][1] - EXTENT[1][0])
        ggrid = bz.RectilinearGrid(G, x=EXTENT[0], y=(EXTENT[1][0], EXTENT[1][0] + Ly), z=EXTENT[2])

        def bubbles(x, y, z):          # one bubble per 20 km of y
            yy = np.mod(y - EXTENT[1][0], EXTENT[1][1] - EXTENT[1][0]) + EXTENT[1][0]
            return bubble(x, yy, z)

        # Transport: "rccl" = the communicator inside the C library (bz_comm.hip: the whole step is one C call, halo exchange overlapped
        # with interior tiles); "torch" = the Python orchestration over torch.distributed (explicit only).  No fallback: whatever was
        # asked for either passes the preflight and carries the run, or the run ends with an error line.
        transport = args.transport
        preflight_summary = None
        if world > 1 and not args.no_preflight:
            try:
                preflight_summary = preflight(args, bz, rank, world, dist, device, transport=transport)
            except Exception as exc:      # noqa: BLE001
                fail(f"preflight failed: {exc!r}")
        try:
            model = SlabAtmosphereModel(ggrid, rank, world, advection=bz.WENO(order=5), surface_pressure=101325,
                                        potential_temperature=300, device=device, transport=transport)
            if transport == "rccl":
                library_halo_selfcheck(model)
            elif world > 1:
                comm_selfcheck(model.decomp, device)
            model.set(θ=bubbles)
            model.time_step(dt)
            torch.cuda.synchronize()
            decomp = model.decomp if transport == "torch" else None
        except Exception as exc:      # noqa: BLE001
            fail(f"slab driver failed: {exc!r}")
        per = (G[0], G[1] // world, G[2])
        parallelism = (f"{world} y-slabs of {per[0]}x{per[1]}x{per[2]}, halo exchange + FFT all-to-all over " +
                       ("RCCL inside the C library (bz_comm.hip)" if transport == "rccl" else "torch.distributed (RCCL backend)"))
    else:
        local = G if world == 1 else (args.size,) * 3
        grid = bz.RectilinearGrid(local, x=EXTENT[0], y=EXTENT[1], z=EXTENT[2])
        ref = bz.ReferenceState(grid, surface_pressure=101325, potential_temperature=300)
        model = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5), device=device)
        model.set(θ=bubble)          # u = v = w = 0, dry
        if world > 1:
            parallelism = f"{world} independent replicas (--replicas)"
            G, scaling = (args.size, args.size * world, args.size), "weak"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    try:
        for _ in range(args.warmup):
            model.time_step(dt)
        model.profile_reset()
        model.profile_enable(True)           # HIP events on the kernels' own stream, over the timed region
        if decomp is not None:
            decomp.profile_reset()
            decomp.profile = True
        barrier()
        t0 = time.perf_counter()
        # the reference's many_time_steps! (benchmarking/src/timestepping.jl:11-16): K steps with nothing reading the model in between,
        # issued through the multi-step seam (bz_time_steps_anelastic): every step but the last skips the diagnosis pass, the last one
        # leaves every field of the model current.  --single-steps issues K separate time_step! calls instead.
        if hasattr(model, "time_steps") and not args.single_steps:
            model.time_steps(dt, args.steps, diagnose_last=True)
        else:
            for _ in range(args.steps):
                model.time_step(dt)
        barrier()
        elapsed = time.perf_counter() - t0
        model.profile_enable(False)
        comm_ms = decomp.comm_ms() / args.steps if decomp is not None else 0.0
        comm_bytes = decomp.comm_bytes // args.steps if decomp is not None else 0
        if use_slabs and decomp is None:      # library transport: HIP-event groups "comm_*" of the context + its byte counter
            prof_now = model.profile()
            comm_ms = sum(ms for k, (ms, n) in prof_now.items() if k.startswith("comm_")) / args.steps
            comm_bytes = model.comm_info()[1] // (args.steps + args.warmup + 1)
        if decomp is not None:
            decomp.profile = False
        finite = bool(torch.isfinite(model.momentum["ρw"].parent).all().item())
        if dist is not None:
            t = torch.tensor([elapsed, comm_ms, 0.0 if finite else 1.0], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, comm_ms, finite = t[0].item(), t[1].item(), t[2].item() == 0.0
    except Exception as exc:          # noqa: BLE001
        fail(f"timed region failed: {exc!r}")

    cells = G[0] * G[1] * G[2]                      # all ranks together
    cells_rank = cells // world
    ms_per_step = 1e3 * elapsed / args.steps
    value = cells * args.steps / elapsed

    prof = model.profile()
    if rank == 0:
        kernels = {}
        for name, (ms, n) in prof.items():
            if n:
                kernels[name] = {"avg_ms": ms / n, "launches": n, "total_ms": ms}
        dry = took_dry_path(kernels)
        roofline = dominant_roofline(kernels, cells_rank, 8, with_traffic=(cells_rank == 512 ** 3), dry=dry)
        kernel_ms = sum(v["total_ms"] for k, v in kernels.items() if not k.startswith("comm_")) / args.steps
        out = {
            "metric": METRIC,
            "value": value, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": label, "grid": list(G), "grid_per_gpu": [G[0], G[1] // world, G[2]], "dt": dt,
                       "parallelism": parallelism},
            "roofline": roofline,
            "step_roofline": step_roofline(kernels, args.steps, cells_rank * args.steps / elapsed, 8, dry=dry),
            "dry_path": dry,
            "dry_path_note": "the workload's moisture is identically zero (BASELINE.json: DRY thermal bubble): after the moisture scan of the step call the "
                             "scalar-pair and z-momentum kernels skip rho q (same bits as the general path: tests/test_dry_shortcut.py); `moist_variant` "
                             "times the general path on the same grid" if dry else None,
            "stepping": "K separate time_step! calls" if (args.single_steps or not hasattr(model, "time_steps")) else
                        "bz_time_steps_anelastic(n = K, diagnose_last = 1): the reference's many_time_steps! loop in one call; every step but the last skips the diagnosis pass",
            "kernels_ms_per_step": {k: v["total_ms"] / args.steps for k, v in sorted(kernels.items())},
            "kernel_launches_per_step": {k: v["launches"] / args.steps for k, v in sorted(kernels.items())},
            "finite": finite,
        }
        if use_slabs:
            # the scalar-pair kernel runs on the library's side stream whenever messages are in flight (bz_comm.hip: fork): the main
            # stream's busy time excludes it, and what the main stream is not busy with is what the exchanges (and launch gaps) expose
            forked = transport == "rccl" and (world > 1 or os.environ.get("BZ_COMM_SELF_MESSAGES") == "1") and os.environ.get("BZ_COMM_NO_SIDE_SCALAR") != "1"
            side_ms = kernels.get("scalar_tendencies+rk3+thermo", {}).get("total_ms", 0.0) / args.steps if forked else 0.0
            out["comm_ms_per_step"] = comm_ms                       # inside point-to-point batches (slowest rank; HIP events on the stream they run on)
            out["compute_ms_per_step"] = kernel_ms                  # library kernels of rank 0 (HIP events), all streams
            out["exposed_comm_ms_per_step"] = max(0.0, ms_per_step - (kernel_ms - side_ms))
            out["exposed_comm_definition"] = "ms_per_step - kernel time of the main stream (all kernel groups but comm_*, and but the scalar-pair kernel when it runs on the side stream): exchanges, waits and launch gaps the main stream could not hide"
            out["comm_bytes_sent_per_step_per_gpu"] = comm_bytes
            out["transport"] = transport
            if preflight_summary:
                out["preflight"] = preflight_summary
        if world == 1 and not use_slabs and args.workload == "bubble" and not args.no_moist_variant:
            try:
                out["moist_variant"] = moist_variant(model, dt, steps=args.steps)      # the same K as the headline: one diagnosis pass per K steps in both legs
            except Exception as exc:       # never let the side measurement take the headline line down
                out["moist_variant"] = {"error": repr(exc)}
        if world == 1 and not args.no_compressible and not use_slabs and args.workload == "bubble":
            try:
                del model
                torch.cuda.empty_cache()
                out["second_milestone"] = milestone_in_fresh_process(bz, device)
            except Exception as exc:       # never let the side measurement take the headline line down
                out["second_milestone"] = {"error": repr(exc)}
        if world == 1 and not args.no_float32 and not use_slabs and args.workload == "bubble":
            try:
                out["float32"] = float32_run(bz, device, args.size, single_steps=args.single_steps)
            except Exception as exc:
                out["float32"] = {"error": repr(exc)}
        if world == 1 and not args.no_cpu_baseline:
            # the headline key is the HEADLINE size (512^3, one warm-up + one timed step, ~40 s of CPU work) when the host has the memory
            # (VERDICT r04 hygiene item); otherwise — or with --no-cpu-full-size — the bounded 256^3 sample
            full = None
            if args.workload == "bubble" and args.size == 512 and not args.no_cpu_full_size:
                try:
                    full = cpu_baseline_full_size(512)
                except Exception as exc:      # never let the side measurement take the headline line down
                    full = {"error": repr(exc)}
            if full is not None and "value" in full:
                out["cpu_baseline"] = full
            else:
                out["cpu_baseline"] = cpu_baseline(args.cpu_size, args.cpu_budget)
                if full is not None:
                    out["cpu_baseline_full_size"] = full
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="N > 1: strong (default) = the size^3 domain split N ways (BASELINE.md §4); weak = size^3 per GPU (explicit only)")
    ap.add_argument("--preflight", action="store_true",
                    help="run only the multi-rank preflight (communicator bootstrap, rank-coded halo exchange, 2-step 64^3-per-rank parity against the "
                         "single-GPU step) and print its JSON line; exit code 1 with an \"error\" line on failure")
    ap.add_argument("--no-preflight", action="store_true", help="N > 1: skip the preflight that precedes the timed region")
    ap.add_argument("--selftest-fail", action="store_true", help="with --selftest-launcher: inject a preflight failure on the last rank")
    ap.add_argument("--cbl-size", default="512x512x256", help="--workload cbl: NxxNyxNz (CI sizes: 256x256x128, 512x512x256, 768x768x256)")
    ap.add_argument("--cbl-order", type=int, default=5, choices=(5, 7, 9), help="--workload cbl: WENO order (CI: 5 and 9)")
    ap.add_argument("--cbl-float64", action="store_true", help="--workload cbl in Float64 (the reference benchmarks Float32)")
    ap.add_argument("--cbl-topology", choices=("PPB", "PBB"), default="PPB", help="--workload cbl: (Periodic, Periodic, Bounded) or (Periodic, Bounded, Bounded)")
    ap.add_argument("--tend-size", default="256x256x128", help="--workload scalar_tendency / model_tendency: NxxNyxNz (CI: 256x256x128)")
    ap.add_argument("--tend-order", type=int, default=5, choices=(5, 7, 9), help="--workload scalar_tendency / model_tendency: WENO order (CI: 5, 7, 9)")
    ap.add_argument("--tend-float64", action="store_true", help="the tendency micro-benchmarks in Float64 (the reference runs Float32)")
    ap.add_argument("--workload", choices=("bubble", "config3", "config4", "cbl", "scalar_tendency", "model_tendency"), default="bubble",
                    help="bubble: the headline workload (configs[1]); config3: 1024 x (128 N) x 512 slabs; config4: compressible + "
                         "Kessler 512x512x128 (second milestone, split over the ranks)")
    ap.add_argument("--single-steps", action="store_true", help="K separate time_step! calls (full diagnosis after every step) instead of the multi-step seam")
    ap.add_argument("--no-moist-variant", action="store_true", help="skip the moist run of the same grid reported under `moist_variant`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-full-size", action="store_true", help="skip the 512^3 leg of the CPU baseline (runs only when the host has >= 96 GB free)")
    ap.add_argument("--replicas", action="store_true", help="N>1: independent replicas instead of the slab decomposition (explicit only)")
    ap.add_argument("--slab", action="store_true", help="N=1: run the slab driver (world 1) instead of the whole-step seam")
    ap.add_argument("--transport", choices=("rccl", "torch"), default="rccl",
                    help="slab runs: the C library's RCCL communicator (default) or the Python orchestration over torch.distributed; never switched silently")
    ap.add_argument("--cpu-size", type=int, default=256)
    ap.add_argument("--cpu-budget", type=float, default=25.0, help="seconds of CPU work for the cpu_baseline leg")
    ap.add_argument("--no-float32", action="store_true", help="skip the Float32 run reported under `float32`")
    ap.add_argument("--config4-order", type=int, default=5, choices=(5, 7, 9), help="--workload config4: WENO order (the example uses 9)")
    ap.add_argument("--config4-float32", action="store_true", help="--workload config4 in Float32, the example's own precision (one GPU)")
    ap.add_argument("--milestone-only", action="store_true", help="run only the compressible milestone and print its JSON object (what the default command spawns)")
    ap.add_argument("--no-compressible", action="store_true",
                    help="skip the short compressible split-explicit measurement reported under `second_milestone`")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="self-launched ranks are killed after this many seconds")
    ap.add_argument("--collective-timeout", type=float, default=300.0, help="process-group timeout (seconds)")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU/gloo: exercise the launcher and the slab communication pattern without a GPU")
    args = ap.parse_args()
    if args.milestone_only:
        import breeze_jl_amd as bz
        print(json.dumps(compressible_milestone(bz, "cuda:0")), flush=True)
        return 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_self(args, sys.argv[1:])
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
