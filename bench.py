#!/usr/bin/env python
"""bench.py — grid cells advanced per second by the anelastic WENO5 + Poisson SSP-RK3 step.

Metric and protocol mirror the reference's benchmark_time_stepping
(/root/reference/benchmarking/src/utils.jl:40-172: warm-up steps, device sync, timed steps, sync,
grid_points_per_second = Nx*Ny*Nz / time_per_step).  Workload: BASELINE.json configs[1], the dry
thermal bubble on a 512^3 RectilinearGrid (SURVEY.md §8d "C2"), Float64, fixed dt = 1 s,
deterministic synthetic initial state resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W [--size 512] [--scaling weak|strong] [--workload bubble|config3]

Prints ONE JSON line (rank 0).  N > 1: one rank per GPU over RCCL, y-slab decomposition
(breeze.jl_amd/distributed.py: y-halo exchange + FFT transposes):
  --scaling weak    (default) every rank owns a size^3 slab of a size x (size N) x size periodic domain
  --scaling strong  the size^3 domain of the N = 1 run is split N ways ("1/2/4/8-GPU cells/s for 512^3")
  --workload config3  BASELINE.json configs[3]: 1024 x 1024 x 512 split over the N ranks (8 ranks -> 1024 x 128 x 512 each)
`value` is the aggregate over ranks; `comm_ms_per_step` / `compute_ms_per_step` split the step of the slowest rank.
If the ranks are not there yet (WORLD_SIZE unset, as in `python bench.py --gpus 8`), the script launches itself under
torch.distributed.run and relays the ranks' line; a failed multi-GPU run prints a JSON line with an "error" field and
exits non-zero — it never silently measures something else.  `--replicas` (explicit only) runs N independent copies.
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

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# Algorithmic words (8 B) per grid cell per launch of each kernel group — SURVEY.md §8(d) table.
WORDS_PER_CELL = {
    "ssp_rk3_substep": 20, "store_initial_state": 10, "poisson_source_term": 4,
    "poisson_fft_forward": 4, "poisson_tridiagonal": 2, "poisson_fft_inverse": 4,
    "make_pressure_correction": 7, "compute_velocities": 6,
    "compute_auxiliary_thermodynamic_variables": 5,
    "x_momentum_tendency": 5, "y_momentum_tendency": 5, "z_momentum_tendency": 7,
    "potential_temperature_tendency": 6, "moisture_tendency": 5,
    "scalar_tendencies": 11, "momentum_tendencies": 17, "tendencies": 28,
    "ssp_rk3_substep+store_initial_state": 30, "project_and_diagnose": 18,
    # tendency kernels with the RK update folded in: tendency words + the RK update words of their fields
    "x_momentum_tendency+rk3": 9, "y_momentum_tendency+rk3": 9, "z_momentum_tendency+rk3": 11,
    "scalar_tendencies+rk3": 19,
    # lean whole-step seam (bz_tendency5_kernels.h): the diagnostics are no separate passes any more — each momentum kernel derives its
    # velocity component (compute_velocities: 2 of its 6 words), the scalar kernel does the theta, q^v, T diagnosis (5 words), the
    # projection of stages 1-2 is make_pressure_correction alone (7).  Per stage: 24 + 11 + 11 + 13 + 7 + Poisson 14 = 80, as before.
    "x_momentum_tendency+rk3+velocity": 11, "y_momentum_tendency+rk3+velocity": 11, "z_momentum_tendency+rk3+velocity": 13,
    "scalar_tendencies+rk3+thermo": 24, "project_momentum": 7,
    # chunked Poisson pipeline (level ranges go source -> x -> y transforms, and y -> x -> projection, back to back)
    "poisson_source_term+fft_forward": 8, "poisson_fft_inverse+project_momentum": 11,
    "poisson_fft_inverse+project_and_diagnose": 22,
    # hand-written x transforms (bz_xfft_kernels.h): source term inside the forward x pass (4 + 2), y passes and the vertical
    # solves on the transposed spectrum (2 each), momentum projection inside the inverse x pass (2 + 7)
    "poisson_source_term+fft_x": 6, "poisson_fft_y_forward": 2, "poisson_fft_y_inverse": 2,
    "poisson_fft_x+project_momentum": 9, "poisson_fft_x_inverse": 2,
}
A_STEP_WORDS = 250          # 3 stages x 80 + 10

# Compulsory words per cell per launch of the FUSED kernels of the lean whole-step seam: every distinct 3-D array the kernel has to
# read or write, once.  The contract figures above were written for the reference's unfused kernel list (a fused kernel inherits the
# sum of what it replaces: 24 words for the scalar-pair kernel), so a fused kernel that got faster can exceed the 8 TB/s roof in
# contract bytes (round 3: 1.02) — which says nothing.  `roofline.achieved` is therefore priced in compulsory bytes; the contract
# figure is reported beside it (`contract_frac`), and `traffic` is what the counters saw.
COMPULSORY_WORDS = {
    "scalar_tendencies+rk3+thermo": 10,        # R rho_u, rho_v, rho_w, rho_theta, rho_q; U0 x 2 (read, or written in stage 1); W rho_theta, rho_q, T
    "x_momentum_tendency+rk3+velocity": 5,     # R rho_u, rho_v, rho_w; U0; W predictor
    "y_momentum_tendency+rk3+velocity": 5,
    "z_momentum_tendency+rk3+velocity": 7,     # + R T, rho_q (buoyancy)
    "project_momentum": 7, "project_and_diagnose": 18,
    "poisson_source_term+fft_x": 4,            # R predictor rho_u, rho_v, rho_w; W half spectrum
    "poisson_fft_y_forward": 2, "poisson_tridiagonal": 2, "poisson_fft_y_inverse": 2, "poisson_fft_x_inverse": 2,
}
A_STEP_COMPULSORY_WORDS = 3 * (10 + 5 + 5 + 7 + 4 + 2 + 2 + 2 + 2) + 2 * 7 + 18      # = 149 words per cell and step (lean seam)
METRIC = "grid-cells advanced/sec (tendency+Poisson step), 512^3 anelastic"


def bubble(x, y, z):
    """theta_i = theta0 exp(N^2 z / g) + 10 max(0, 1 - r/2000), bubble centre (0, 0, 3000 m)."""
    r = np.sqrt(x ** 2 + y ** 2 + (z - 3000.0) ** 2)
    return 300.0 * np.exp(1e-6 * z / 9.81) + 10.0 * np.maximum(0.0, 1.0 - r / 2000.0)


EXTENT = ((-10e3, 10e3), (-10e3, 10e3), (0.0, 10e3))


def compressible_milestone(bz, device, steps=2, substep_float32=False):
    """Second milestone, reported beside the headline metric, never as `value`: the compressible
    split-explicit WS-RK3 step (acoustic substep loop) on a 512 x 512 x 256 bubble, Float64, dt = 1 s.
    substep_float32: the same model with substep_floattype = Float32 (acoustic_substepping.jl:199-235): the substepper's ten working
    fields stored as Float32, arithmetic and every model field Float64 — reported under `substep_floattype_float32`."""
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
    nsub = [m.stage_substeps(1.0, b)[0] for b in (1 / 3, 1 / 2, 1.0)]
    sub_ms = sum(prof[k][0] for k in prof if k.startswith("acoustic_horizontal") or k.startswith("acoustic_column")) / steps
    per_sub = sub_ms / sum(nsub)
    cells = Nx * Ny * Nz
    out = {"metric": "grid-cells advanced/sec, compressible split-explicit WS-RK3 step", "value": cells / (ms * 1e-3),
           "unit": "cells/s", "ms_per_step": ms, "grid": [Nx, Ny, Nz], "dt": 1.0, "substeps_per_stage": nsub,
           "acoustic_substep_ms": per_sub,
           "acoustic_substep_roofline": {"bound": "hbm", "algorithmic_bytes_per_cell_substep": 58 * 8,
                                         "achieved": cells * 58 * 8 / (per_sub * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                         "unit": "GB/s", "frac": cells * 58 * 8 / (per_sub * 1e-3) / 1e9 / HBM_PEAK_GBS},
           "finite": bool(torch.isfinite(m.velocities["w"].interior).all().item())}
    del m
    torch.cuda.empty_cache()
    if not substep_float32:
        try:
            r = compressible_milestone(bz, device, steps, substep_float32=True)
            out["substep_floattype_float32"] = {k: r[k] for k in ("value", "ms_per_step", "acoustic_substep_ms", "finite")}
            out["substep_floattype_float32"]["tolerance"] = "2e-6 of the field scale after three steps against the Float64 oracle (tests/test_gpu_compressible.py)"
        except Exception as exc:      # noqa: BLE001
            out["substep_floattype_float32"] = {"error": repr(exc)}
    return out


def float32_run(bz, device, N, steps=10, warmup=2):
    """The same workload with eltype(grid) = Float32 (lib/libbreeze_hip_f32.so) — what the reference's own GPU benchmarks run
    (benchmarking/src/convective_boundary_layer.jl:59,70).  Reported beside the Float64 headline, never as `value`
    ("Float32 run reported separately"); algorithmic bytes are 250 words x 4 B = 1000 B per cell and step."""
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
    for _ in range(steps):
        m.time_step(1.0)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    m.profile_enable(False)
    prof = {k: ms / steps for k, (ms, n) in m.profile().items() if n}
    cells = N ** 3
    rate = cells * steps / el
    out = {"dtype": "f32", "value": rate, "unit": "cells/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "grid": [N, N, N],
           "step_roofline": {"bound": "hbm", "achieved": rate * A_STEP_WORDS * 4 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": rate * A_STEP_WORDS * 4 / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_cell_step": A_STEP_WORDS * 4},
           "kernels_ms_per_step": prof, "finite": bool(torch.isfinite(m.momentum["ρw"].parent).all().item()),
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
    for _ in range(args.steps):
        m.time_step(dt)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # per-kernel times from a second, untimed pass: the HIP events around ~40 launches per step cost ~0.2 ms per step, 11 % of the step at
    # 256 x 256 x 128 (1.86 ms without them, 2.09 ms with them)
    psteps = min(args.steps, 20)
    m.profile_reset()
    m.profile_enable(True)
    for _ in range(psteps):
        m.time_step(dt)
    torch.cuda.synchronize()
    m.profile_enable(False)
    cells, word = Nx * Ny * Nz, 4 if f32 else 8
    kernels = {k: {"avg_ms": ms / n, "launches": n, "total_ms": ms} for k, (ms, n) in m.profile().items() if n}
    known = [k for k in kernels if k in WORDS_PER_CELL]
    roofline = None
    if known:
        dom = max(known, key=lambda k: kernels[k]["total_ms"])
        dom_bytes = WORDS_PER_CELL[dom] * word * cells
        achieved = dom_bytes / (kernels[dom]["avg_ms"] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": None, "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": kernels[dom]["avg_ms"]}
    rate = cells * args.steps / elapsed
    step_achieved = rate * A_STEP_WORDS * word / 1e9
    out = {"metric": "grid points per second (time_step!), convective boundary layer benchmark case",
           "value": rate, "unit": "cells/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32" if f32 else "f64", "data": "synthetic",
           "config": {"workload": f"BreezeBenchmarks convective_boundary_layer {Nx}x{Ny}x{Nz} (benchmarking/src/convective_boundary_layer.jl): "
                                  f"AnelasticDynamics, WENO{args.cbl_order}, halo 5, topology {args.cbl_topology}, FPlane + geostrophic forcing + u* drag + surface heat flux, "
                                  f"{'Float32' if f32 else 'Float64'}, dt={dt}s", "grid": [Nx, Ny, Nz], "dt": dt, "parallelism": "single GPU"},
           "roofline": roofline,
           "step_roofline": {"bound": "hbm", "achieved": step_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": step_achieved / HBM_PEAK_GBS,
                             "algorithmic_bytes_per_cell_step": A_STEP_WORDS * word,
                             "note": "the dry-bubble contract figure (250 words per cell and step); the forcing and flux kernels of this case move a few words more"},
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
    roofline = None
    if words and kernel in kernels:
        nbytes = words * word * cells
        achieved = nbytes / (kernels[kernel]["avg_ms"] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": None, "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": kernels[kernel]["avg_ms"]}
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
    gkw = {"float_type": np.float32} if f32 else {}
    if order != 5:
        gkw["halo"] = (5, 5, 5)
    G = bz.RectilinearGrid((Nx, Ny, Nz), x=(0.0, 168e3), y=(0.0, 168e3), z=(0.0, 20e3), **gkw)
    dyn = bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(), surface_pressure=1e5, reference_potential_temperature=300.0)
    mkw = dict(thermodynamic_constants=bz.ThermodynamicConstants(saturation_vapor_pressure=bz.TetensFormula()),
               microphysics=bz.DCMIP2016KesslerMicrophysics())
    slabs = world > 1 or args.slab
    transport = "rccl" if args.transport in ("auto", "rccl") else "torch"
    try:
        if slabs:
            m = bz.compressible.SlabCompressibleModel(G, rank, world, dyn, advection=bz.WENO(order=order), device=device,
                                                      transport=transport, **mkw)
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
               "kernels_ms_per_step": prof, "finite": finite,
               "comm_ms_per_step": sum(v for k, v in prof.items() if k.startswith("comm_")),
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
           "ms_per_step": None, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
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


def problem(args, world):
    """Global grid size, per-rank slab and the label of the workload."""
    N = args.size
    if args.workload == "config3":
        G = (1024, 1024, 512)
        label = ("dry thermal bubble 1024x1024x512 RectilinearGrid (Periodic,Periodic,Bounded), halo 3, AnelasticDynamics + WENO5 + "
                 "SSP-RK3, Float64, dt=1s, y-slabs (BASELINE.json configs[3])")
        scaling = "strong"
    elif args.scaling == "strong" or world == 1:
        G = (N, N, N)
        label = (f"dry thermal bubble {N}^3 RectilinearGrid (Periodic,Periodic,Bounded), halo 3, AnelasticDynamics + WENO5 + "
                 "SSP-RK3, Float64, dt=1s (BASELINE.json configs[1])")
        scaling = "strong" if world > 1 else args.scaling
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
        # launcher self-test (tests/test_bench_launcher.py): rendezvous, the slab communication pattern on CPU tensors
        # under gloo, result line — no model, no GPU
        from breeze_jl_amd.distributed import SlabDecomposition
        d = SlabDecomposition(20, 6, 4, 3, rank, world)
        try:
            comm_selfcheck(d, "cpu")
        except Exception as exc:   # noqa: BLE001
            fail(f"communication self-check failed: {exc!r}")
        ok = torch.ones(1, dtype=torch.int32)
        if dist is not None:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            dist.barrier()
        if rank == 0:
            print(json.dumps({"launcher_selftest": "ok", "n_gpus": world, "backend": "gloo"}), flush=True)
        if dist is not None:
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

        # Transport: "rccl" = the communicator inside the C library (bz_comm.hip: the whole step is one C call, halo exchange
        # overlapped with interior tiles); "torch" = the Python orchestration over torch.distributed.  "auto" tries the library
        # first — its first contact with more than one GPU is this very run — verifies a rank-coded halo exchange through it, and
        # only if that raises on some rank do ALL ranks switch to the torch transport.  The line says which one carried the run.
        def build(transport):
            m = SlabAtmosphereModel(ggrid, rank, world, advection=bz.WENO(order=5), surface_pressure=101325,
                                    potential_temperature=300, device=device, transport=transport)
            if transport == "rccl":
                library_halo_selfcheck(m)
            elif world > 1:
                comm_selfcheck(m.decomp, device)
            m.set(θ=bubbles)
            m.time_step(dt)
            torch.cuda.synchronize()
            return m

        transport, transport_note = ("rccl" if args.transport == "auto" else args.transport), None
        try:
            err = None
            try:
                model = build(transport)
            except Exception as exc:      # noqa: BLE001
                err = repr(exc)
                print(f"[bench rank {rank}] transport {transport}: {err}", file=sys.stderr, flush=True)
            if args.transport == "auto":
                bad = torch.tensor([1 if err else 0], dtype=torch.int32, device=device)
                if dist is not None:
                    dist.all_reduce(bad, op=dist.ReduceOp.MAX)
                if bad.item():
                    transport_note = f"library transport failed on a rank ({err or 'another rank'}); torch.distributed carried the run"
                    model = None
                    torch.cuda.empty_cache()
                    transport = "torch"
                    model = build(transport)
            elif err:
                raise RuntimeError(err)
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
        dom = max((k for k in kernels if k in WORDS_PER_CELL), key=lambda k: kernels[k]["total_ms"])
        contract_bytes = WORDS_PER_CELL[dom] * 8 * cells_rank
        dom_bytes = COMPULSORY_WORDS.get(dom, WORDS_PER_CELL[dom]) * 8 * cells_rank
        achieved = dom_bytes / (kernels[dom]["avg_ms"] * 1e-3) / 1e9
        contract_achieved = contract_bytes / (kernels[dom]["avg_ms"] * 1e-3) / 1e9
        # HBM-side bytes per launch of that kernel group measured by rocprofv3 PMC passes (committed under profiles/,
        # collected at 512^3 on one GPU with the same build: a reference figure, not a measurement of this very run)
        traffic, traffic_src = None, None
        if cells_rank == 512 ** 3:
            for fn in ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
                try:
                    with open(os.path.join(ROOT, "profiles", fn)) as fh:
                        traffic = json.load(fh)["per_kernel_group"][dom]["hbm_bytes_per_launch"]
                    traffic_src = "profiles/" + fn
                    break
                except (OSError, KeyError, ValueError):
                    continue
        roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": kernels[dom]["avg_ms"],
                    "algorithmic_words_per_cell": COMPULSORY_WORDS.get(dom, WORDS_PER_CELL[dom]),
                    "contract_words_per_cell": WORDS_PER_CELL[dom], "contract_achieved": contract_achieved,
                    "contract_frac": contract_achieved / HBM_PEAK_GBS,
                    "traffic_frac": (traffic / (kernels[dom]["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None}
        step_achieved = (cells_rank * args.steps / elapsed) * A_STEP_WORDS * 8 / 1e9
        kernel_ms = sum(v["total_ms"] for k, v in kernels.items() if not k.startswith("comm_")) / args.steps
        out = {
            "metric": METRIC,
            "value": value, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": label, "grid": list(G), "grid_per_gpu": [G[0], G[1] // world, G[2]], "dt": dt,
                       "parallelism": parallelism},
            "roofline": roofline,
            "step_roofline": {"bound": "hbm", "achieved": step_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": step_achieved / HBM_PEAK_GBS, "per": "GPU",
                              "algorithmic_bytes_per_cell_step": A_STEP_WORDS * 8,
                              "compulsory_bytes_per_cell_step": A_STEP_COMPULSORY_WORDS * 8,
                              "compulsory_frac": (cells_rank * args.steps / elapsed) * A_STEP_COMPULSORY_WORDS * 8 / 1e9 / HBM_PEAK_GBS},
            "kernels_ms_per_step": {k: v["total_ms"] / args.steps for k, v in sorted(kernels.items())},
            "kernel_launches_per_step": {k: v["launches"] / args.steps for k, v in sorted(kernels.items())},
            "finite": finite,
        }
        if use_slabs:
            out["comm_ms_per_step"] = comm_ms                       # inside point-to-point batches (slowest rank)
            out["compute_ms_per_step"] = kernel_ms                  # library kernels of rank 0 (HIP events)
            out["other_ms_per_step"] = max(0.0, ms_per_step - comm_ms - kernel_ms)   # packs, transforms' glue, launch gaps, waits
            out["comm_bytes_sent_per_step_per_gpu"] = comm_bytes
            out["transport"] = transport
            if transport_note:
                out["transport_note"] = transport_note
        if world == 1 and not args.no_compressible and not use_slabs and args.workload == "bubble":
            try:
                del model
                torch.cuda.empty_cache()
                out["second_milestone"] = compressible_milestone(bz, device)
            except Exception as exc:       # never let the side measurement take the headline line down
                out["second_milestone"] = {"error": repr(exc)}
        if world == 1 and not args.no_float32 and not use_slabs and args.workload == "bubble":
            try:
                out["float32"] = float32_run(bz, device, args.size)
            except Exception as exc:
                out["float32"] = {"error": repr(exc)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_size, args.cpu_budget)
            if args.workload == "bubble" and args.size == 512 and not args.no_cpu_full_size:
                try:
                    out["cpu_baseline_full_size"] = cpu_baseline_full_size(512)
                except Exception as exc:      # never let the side measurement take the headline line down
                    out["cpu_baseline_full_size"] = {"error": repr(exc)}
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
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-full-size", action="store_true", help="skip the 512^3 leg of the CPU baseline (runs only when the host has >= 96 GB free)")
    ap.add_argument("--replicas", action="store_true", help="N>1: independent replicas instead of the slab decomposition (explicit only)")
    ap.add_argument("--slab", action="store_true", help="N=1: run the slab driver (world 1) instead of the whole-step seam")
    ap.add_argument("--transport", choices=("auto", "rccl", "torch"), default="auto",
                    help="slab runs: the C library's RCCL communicator, torch.distributed, or try the first and fall back")
    ap.add_argument("--cpu-size", type=int, default=256)
    ap.add_argument("--cpu-budget", type=float, default=25.0, help="seconds of CPU work for the cpu_baseline leg")
    ap.add_argument("--no-float32", action="store_true", help="skip the Float32 run reported under `float32`")
    ap.add_argument("--config4-order", type=int, default=5, choices=(5, 7, 9), help="--workload config4: WENO order (the example uses 9)")
    ap.add_argument("--config4-float32", action="store_true", help="--workload config4 in Float32, the example's own precision (one GPU)")
    ap.add_argument("--no-compressible", action="store_true",
                    help="skip the short compressible split-explicit measurement reported under `second_milestone`")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="self-launched ranks are killed after this many seconds")
    ap.add_argument("--collective-timeout", type=float, default=300.0, help="process-group timeout (seconds)")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU/gloo: exercise the launcher and the slab communication pattern without a GPU")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_self(args, sys.argv[1:])
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
