"""bench.py's multi-rank launcher, driven end to end without a GPU: `python bench.py --gpus 2 --selftest-launcher` has no
rendezvous environment, so the script must start its own ranks (torch.distributed.run), the ranks rendezvous under gloo, run
the slab communication pattern of breeze.jl_amd/distributed.py on CPU tensors (y-halo exchange + both spectral transposes,
verified against what the neighbours must have sent) and rank 0 prints exactly one JSON line that the parent relays."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, env=env,
                       timeout=timeout, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    return p, lines


@pytest.mark.parametrize("world", [2, 4])
def test_self_launch_runs_the_slab_exchange_under_gloo(world):
    p, lines = _run(["--gpus", str(world), "--selftest-launcher"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["launcher_selftest"] == "ok" and out["n_gpus"] == world and out["backend"] == "gloo"
    # the defaults of a multi-rank run: the curve BASELINE.md §4 names — 512^3 split N ways — over the library's
    # RCCL transport; weak scaling and config3 are explicit options
    assert out["scaling"] == "strong" and out["grid"] == [512, 512, 512] and out["grid_per_gpu"] == [512, 512 // world, 512]
    assert out["transport"] == "rccl" and out["preflight"]["status"] == "ok"


def test_explicit_scaling_options_and_preflight_mode():
    p, lines = _run(["--gpus", "2", "--selftest-launcher", "--scaling", "weak"])
    assert p.returncode == 0 and json.loads(lines[0])["scaling"] == "weak" and json.loads(lines[0])["grid"] == [512, 1024, 512]
    p, lines = _run(["--gpus", "2", "--selftest-launcher", "--workload", "config3"])
    assert p.returncode == 0 and json.loads(lines[0])["grid_per_gpu"] == [1024, 512, 512]
    p, lines = _run(["--gpus", "2", "--selftest-launcher", "--preflight"])
    assert p.returncode == 0 and json.loads(lines[0])["preflight_only"] is True


def test_failed_preflight_ends_the_run_with_an_error_line():
    """(5b) a preflight failure on ANY rank: one JSON line with an "error" field, non-zero exit code, no timed region"""
    p, lines = _run(["--gpus", "2", "--selftest-launcher", "--selftest-fail"])
    assert p.returncode != 0
    assert lines, (p.stdout, p.stderr[-1500:])
    out = json.loads(lines[-1])
    assert out["value"] is None and "error" in out and "launcher_selftest" not in out


def test_no_silent_transport_switch():
    """(5c) `auto` is gone: the transport is what the command line says (rccl by default, torch explicitly) or the run fails"""
    p, _ = _run(["--gpus", "2", "--selftest-launcher", "--transport", "auto"])
    assert p.returncode == 2 and "invalid choice" in p.stderr
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "transport_note" not in src and "carried the run" not in src


def test_failed_multi_gpu_run_prints_an_error_line_not_a_fallback():
    """No GPU in this container: the slab driver cannot start.  The launcher must say so in a JSON line with an "error" field
    and a non-zero exit code — not hang, not measure replicas instead."""
    p, lines = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--size", "16", "--launch-timeout", "240",
                     "--collective-timeout", "60"])
    if p.returncode == 0:
        pytest.skip("a multi-GPU box: the slab run succeeded")
    assert lines, (p.stdout, p.stderr[-2000:])
    out = json.loads(lines[-1])
    assert out["value"] is None and "error" in out and out["n_gpus"] == 2
    assert "replica" not in out["error"]


def test_workload_shapes():
    sys.path.insert(0, ROOT)
    import argparse
    import bench
    a = argparse.Namespace(size=512, scaling=None, workload="bubble")
    assert bench.problem(a, 1)[0] == (512, 512, 512) and bench.problem(a, 1)[2] == "weak"      # one GPU: the contract's label
    assert bench.problem(a, 8)[0] == (512, 512, 512) and bench.problem(a, 8)[2] == "strong"    # default for N > 1: BASELINE.md §4
    a.scaling = "weak"
    assert bench.problem(a, 1)[0] == (512, 512, 512)
    assert bench.problem(a, 8)[0] == (512, 4096, 512) and bench.problem(a, 8)[2] == "weak"
    a.scaling = "strong"
    assert bench.problem(a, 8)[0] == (512, 512, 512) and bench.problem(a, 8)[2] == "strong"
    a.workload = "config3"
    G, label, scaling = bench.problem(a, 8)
    assert G == (1024, 1024, 512) and "configs[3]" in label and scaling == "strong"
    with pytest.raises(ValueError):
        bench.problem(argparse.Namespace(size=500, scaling="strong", workload="bubble"), 8)
