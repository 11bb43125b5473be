#!/usr/bin/env python
"""bench.py — grid cells advanced per second by the anelastic WENO5 + Poisson SSP-RK3 step.

Metric and protocol mirror the reference's benchmark_time_stepping
(/root/reference/benchmarking/src/utils.jl:40-172: warm-up steps, device sync, timed steps, sync,
grid_points_per_second = Nx*Ny*Nz / time_per_step).  Workload: BASELINE.json configs[1], the dry
thermal bubble on a 512^3 RectilinearGrid (SURVEY.md §8d "C2"), Float64, fixed dt = 1 s,
deterministic synthetic initial state resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W [--size 512]

Prints ONE JSON line on rank 0.  For N > 1 it is launched by torch.distributed.run, one rank per GPU over RCCL:
weak scaling, every rank owns a 512 x 512 x 512 y-slab of a 512 x (512 N) x 512 periodic domain
(breeze.jl_amd/distributed.py: y-halo exchange + FFT transposes); `value` is the aggregate over ranks.
`--replicas` runs N independent copies of the N=1 workload instead (no collective in the data path).
"""
import argparse
import json
import os
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
}
A_STEP_WORDS = 250          # 3 stages x 80 + 10


def bubble(x, y, z):
    """theta_i = theta0 exp(N^2 z / g) + 10 max(0, 1 - r/2000), bubble centre (0, 0, 3000 m)."""
    r = np.sqrt(x ** 2 + y ** 2 + (z - 3000.0) ** 2)
    return 300.0 * np.exp(1e-6 * z / 9.81) + 10.0 * np.maximum(0.0, 1.0 - r / 2000.0)


EXTENT = ((-10e3, 10e3), (-10e3, 10e3), (0.0, 10e3))


def compressible_milestone(bz, device, steps=2):
    """Second milestone, reported beside the headline metric, never as `value`: the compressible
    split-explicit WS-RK3 step (acoustic substep loop) on a 512 x 512 x 256 bubble, Float64, dt = 1 s."""
    import torch
    Nx, Ny, Nz = 512, 512, 256
    grid = bz.RectilinearGrid((Nx, Ny, Nz), x=EXTENT[0], y=EXTENT[1], z=EXTENT[2])
    dyn = bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(), surface_pressure=1e5, reference_potential_temperature=300.0)
    m = bz.CompressibleAtmosphereModel(grid, dyn, advection=bz.WENO(order=5), device=device)
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
    return out


def cpu_baseline(n, steps):
    """The CPU oracle ("port": this repo's C/OpenMP restatement, not Breeze CPU()) timed on this
    box's host cores on a bounded sample of the same workload: the bubble at n^3."""
    from oracle import oracle as orc
    cores = len(os.sched_getaffinity(0))
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    g = orc.Grid((n, n, n), x=EXTENT[0], y=EXTENT[1], z=EXTENT[2])
    m = orc.OracleModel(g, potential_temperature=300.0)
    m.set(theta=bubble)
    m.time_step(1.0)                       # warm-up (also the first-step update_state)
    t0 = time.perf_counter()
    for _ in range(steps):
        m.time_step(1.0)
    dt = (time.perf_counter() - t0) / steps
    return {"value": n ** 3 / dt, "unit": "cells/s", "cores": cores, "kind": "port",
            "sample": f"dry thermal bubble {n}^3 Float64, {steps} steps after 1 warm-up, "
                      f"C/OpenMP oracle ({cores} threads) + numpy pocketfft"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--replicas", action="store_true", help="N>1: independent replicas instead of the slab decomposition")
    ap.add_argument("--slab", action="store_true", help="N=1: run the slab driver (world 1) instead of the whole-step seam")
    ap.add_argument("--cpu-size", type=int, default=128)
    ap.add_argument("--cpu-steps", type=int, default=8)
    ap.add_argument("--no-compressible", action="store_true",
                    help="skip the short compressible split-explicit measurement reported under `second_milestone`")
    args = ap.parse_args()

    import torch
    import breeze_jl_amd as bz

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    device = f"cuda:{local_rank}"

    N = args.size
    dt = 1.0
    use_slabs = (world > 1 and not args.replicas) or (world == 1 and args.slab)
    parallelism = "single GPU"
    if use_slabs:
        # weak scaling: the domain grows in y with the number of ranks, one bubble per 20 km of y
        from breeze_jl_amd.distributed import SlabAtmosphereModel
        Ly = (EXTENT[1][1] - EXTENT[1][0]) * world
        ggrid = bz.RectilinearGrid((N, N * world, N), x=EXTENT[0], y=(EXTENT[1][0], EXTENT[1][0] + Ly), z=EXTENT[2])

        def bubbles(x, y, z):
            yy = np.mod(y - EXTENT[1][0], EXTENT[1][1] - EXTENT[1][0]) + EXTENT[1][0]
            return bubble(x, yy, z)

        # the RCCL transport cannot be exercised from the 1-GPU build box: if constructing the slab model or its first step
        # raises on any rank, every rank falls back to independent replicas and the JSON line says so
        slab_error = None
        try:
            model = SlabAtmosphereModel(ggrid, rank, world, advection=bz.WENO(order=5), surface_pressure=101325,
                                        potential_temperature=300, device=device)
            model.set(θ=bubbles)
            model.time_step(dt)
            torch.cuda.synchronize()
        except Exception as exc:      # noqa: BLE001
            slab_error = repr(exc)
            print(f"[bench rank {rank}] slab driver failed: {slab_error}", file=sys.stderr, flush=True)
        if dist is not None:
            flag = torch.tensor([0 if slab_error else 1], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if flag.item() == 0:
                slab_error = slab_error or "another rank failed"
        if slab_error:
            use_slabs = False
            model = None
            torch.cuda.empty_cache()
        parallelism = f"{world} y-slabs of {N}x{N}x{N} (RCCL halo exchange + FFT transposes)"
    if not use_slabs:
        grid = bz.RectilinearGrid((N, N, N), x=EXTENT[0], y=EXTENT[1], z=EXTENT[2])
        ref = bz.ReferenceState(grid, surface_pressure=101325, potential_temperature=300)
        model = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5), device=device)
        model.set(θ=bubble)          # u = v = w = 0, dry
        if world > 1:
            parallelism = f"{world} independent replicas"
            if not args.replicas:
                parallelism += f" (fallback: slab driver failed: {slab_error})"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        model.time_step(dt)
    model.profile_reset()
    model.profile_enable(True)           # HIP events on the kernels' own stream, over the timed region
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.time_step(dt)
    barrier()
    elapsed = time.perf_counter() - t0
    model.profile_enable(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    cells = N ** 3
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * cells * args.steps / elapsed

    prof = model.profile()
    finite = bool(torch.isfinite(model.momentum["ρw"].parent).all().item())
    if rank == 0:
        kernels = {}
        for name, (ms, n) in prof.items():
            if n:
                kernels[name] = {"avg_ms": ms / n, "launches": n, "total_ms": ms}
        dom = max((k for k in kernels if k in WORDS_PER_CELL), key=lambda k: kernels[k]["total_ms"])
        dom_bytes = WORDS_PER_CELL[dom] * 8 * cells
        achieved = dom_bytes / (kernels[dom]["avg_ms"] * 1e-3) / 1e9
        # measured HBM-side bytes per launch of that kernel group (rocprofv3 PMC passes, committed under profiles/;
        # collected at the default 512^3 size only)
        traffic = None
        try:
            if N == 512:
                with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fh:
                    traffic = json.load(fh)["per_kernel_group"][dom]["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            traffic = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": kernels[dom]["avg_ms"]}
        step_achieved = (cells * args.steps / elapsed) * A_STEP_WORDS * 8 / 1e9
        out = {
            "metric": "grid-cells advanced/sec (tendency+Poisson step), 512^3 anelastic",
            "value": value, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"dry thermal bubble {N}^3 RectilinearGrid (Periodic,Periodic,Bounded), halo 3, "
                                   "AnelasticDynamics + WENO5 + SSP-RK3, Float64, dt=1s (BASELINE.json configs[1])",
                       "grid": [N, N * world, N] if use_slabs else [N, N, N], "grid_per_gpu": [N, N, N], "dt": dt,
                       "parallelism": parallelism},
            "roofline": roofline,
            "step_roofline": {"bound": "hbm", "achieved": step_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": step_achieved / HBM_PEAK_GBS,
                              "algorithmic_bytes_per_cell_step": A_STEP_WORDS * 8},
            "kernels_ms_per_step": {k: v["total_ms"] / args.steps for k, v in sorted(kernels.items())},
            "finite": finite,
        }
        if world == 1 and not args.no_compressible and not use_slabs:
            try:
                del model
                torch.cuda.empty_cache()
                out["second_milestone"] = compressible_milestone(bz, device)
            except Exception as exc:       # never let the side measurement take the headline line down
                out["second_milestone"] = {"error": repr(exc)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_size, args.cpu_steps)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
