"""The kernel instantiations the 512^3 bench times, compared with the oracle (VERDICT r02 "What's weak" 2 / "Next round" 1).

The oracle tests of test_gpu_parity.py use Nz <= 40, so only `k_tridiag_coop<16>` / `<8>` / the sequential Thomas kernel, partial
x-transform chunks and short z-marches of the tendency kernels met the oracle there.  Here the shapes select
  * `k_tridiag_coop<64>` (128 <= Nz <= 512; csrc/bz_poisson.hip: tridiag_coop_segs),
  * full 16-level `k_x_forward` / `k_x_inverse` blocks at Nx = 512 and Nx = 1024 (csrc/bz_xfft_kernels.h: xf_chunk),
  * XCD-remapped launches of the lean tendency kernels with >= 128-level z-marches (csrc/bz_tendency5.hip: pick_chunk5),
  * the slab path and the compressible column kernels with Nz = 128.
Reference: /root/reference/src/AnelasticEquations/anelastic_pressure_solver.jl:84-105 (source term + solve!),
/root/reference/src/TimeSteppers/ssp_runge_kutta_3.jl:209-278 (the step).  Oracle parity status as everywhere: WENO / halo / Thomas
arithmetic restated from Oceananigans, unpinned."""
import numpy as np
import pytest

from helpers import PROG, bubble_theta, make_pair, push_state, randomize, relerr

pytestmark = pytest.mark.gpu


def _interior(om, name):
    return om.grid.interior(getattr(om, name), zface=(name in ("rw", "w")))


@pytest.mark.parametrize("size", [(512, 8, 128), (1024, 8, 128), (64, 64, 512), (128, 16, 256), (64, 8, 200), (32, 24, 130)])
def test_pressure_correction_on_long_columns_matches_oracle(oracle, bz, size):
    """per-operator Poisson solve + projection, 1e-11: cooperative tridiagonal kernel with 64 segments, full x-transform chunks.
    Nz = 128 / 256 / 512 take the exact variants `k_tridiag_coop<64, 2 | 4 | 8, true>` (every segment holds Nz / 64 rows, straight-line
    prefetching loop), Nz = 200 and 130 the general one with ragged segments; several column groups per workgroup in every case."""
    om, hm = make_pair(oracle, bz, size)
    randomize(om, seed=3)
    push_state(om, hm, names=("ru", "rv", "rw"))
    dt = 0.7
    om.compute_pressure_correction(dt)
    bz.compute_pressure_correction_(hm, dt)
    hm.synchronize()
    assert relerr(hm.dynamics.pressure_anomaly.cpu(), om.phi) < 1e-11
    om.make_pressure_correction(dt)
    bz.make_pressure_correction_(hm, dt)
    hm.synchronize()
    for n in ("ru", "rv", "rw"):
        f = {"ru": hm.momentum["ρu"], "rv": hm.momentum["ρv"], "rw": hm.momentum["ρw"]}[n]
        assert relerr(f.interior_cpu(), _interior(om, n)) < 1e-11, n
    g = om.grid
    scale = np.max(np.abs(_interior(om, "ru"))) / min(g.dx, g.dy, float(np.min(np.diff(g.zf))))      # the finest spacing sets the size of a difference quotient
    assert hm.max_abs_divergence() < 1e-12 * scale


@pytest.mark.parametrize("size", [(128, 128, 128), (64, 64, 256)])
def test_three_steps_on_long_columns_match_oracle(oracle, bz, size):
    """whole-step lean seam, three steps, 1e-9: >= 128-level marches of k5_scalar_pair / k6_u / k6_v / k6_w in XCD-remapped launches,
    source term inside 16-level x-transform blocks, k_tridiag_coop<64>, both projection kernels"""
    om, hm = make_pair(oracle, bz, size)
    th = bubble_theta(300.0, om.constants.g)
    om.set(theta=th, u=3.0, v=-2.0)
    hm.set(θ=th, u=3.0, v=-2.0)
    for _ in range(3):
        om.time_step(2.0)
        hm.time_step(2.0)
    hm.synchronize()
    mom = max(np.max(np.abs(_interior(om, n))) for n in ("ru", "rv", "rw"))
    for n, k in PROG.items():
        got, want = hm.prognostic_fields()[k].interior_cpu(), _interior(om, n)
        scale = mom if n in ("ru", "rv", "rw") else max(np.max(np.abs(want)), 1e-3)
        assert np.max(np.abs(got - want)) / scale < 1e-9, n
    for n, f in (("u", hm.velocities["u"]), ("theta", hm.potential_temperature), ("T", hm.temperature)):
        assert relerr(f.cpu(), getattr(om, n)) < 1e-9, n


@pytest.mark.parametrize("size", [(128, 64, 128)])
def test_tendencies_on_long_columns_match_oracle(oracle, bz, size):
    """per-operator tendencies on random fields with 128 levels (every buffer of the wall cascade, long z chunks).  Tolerance 5e-12:
    the maximum over 10^6 cells of the FMA-contraction difference of a WENO flux of rough data (DESIGN.md "Strict-parity library";
    the 10^4-cell cases of test_gpu_parity.py stay below 1e-12, this one measured 2.5e-12 on rho theta)"""
    om, hm = make_pair(oracle, bz, size)
    randomize(om, seed=11)
    om.compute_tendencies()
    push_state(om, hm, names=("ru", "rv", "rw", "rtheta", "rq", "u", "v", "w", "theta", "q", "T"))
    bz.compute_tendencies_(hm)
    hm.synchronize()
    for n, k in PROG.items():
        zf = n == "rw"
        want, got = om.grid.interior(om.G[n], zface=zf), hm.G[k].interior_cpu()
        if zf:
            want, got = want[1:-1], got[1:-1]
        assert relerr(got, want) < 5e-12, n


def test_two_rank_slab_step_with_128_levels_matches_oracle(oracle, bz):
    """library-owned distributed step on two ranks sharing the GPU, Nz = 128 (cooperative tridiagonal kernel on the kx-slab,
    x transforms writing / reading the all-to-all messages in 16-level blocks), against the oracle on the whole domain, 1e-9"""
    import test_comm as tc
    size, steps, dt = (64, 32, 128), 2, 2.0
    models = tc.run_slabs(bz, size, 2, steps, dt, True)
    og = oracle.Grid(size, x=tc.EXTENT[0], y=tc.EXTENT[1], z=tc.EXTENT[2])
    om = oracle.OracleModel(og, potential_temperature=300.0)
    om.set(theta=tc.theta_ic, u=3.0, v=-2.0, qt=tc.q_ic)
    for _ in range(steps):
        om.time_step(dt)
    mom = max(np.abs(og.interior(getattr(om, n), zface=(n == "rw"))).max() for n in ("ru", "rv", "rw"))
    for name, get in tc.FIELDS.items():
        got = np.concatenate([get(m).interior_cpu() for m in models], axis=1)
        want = og.interior(getattr(om, name), zface=(name == "rw"))
        scale = mom if name in ("ru", "rv", "rw") else max(np.max(np.abs(want)), 1e-3)
        assert np.max(np.abs(got - want)) / scale < 1e-9, name


def test_compressible_steps_with_128_levels_match_oracle(oracle, bz):
    """split-explicit WS-RK3, two steps on 32 x 16 x 128: the acoustic column kernels' 128-level sweeps against the oracle"""
    import test_gpu_compressible as tg
    from oracle import oracle_compressible as oc
    om, hm = tg.make_pair(oracle, oc, bz, size=(32, 16, 128), substeps=6)
    g = om.grid

    def theta(x, y, z):
        r = np.sqrt(x ** 2 + y ** 2 + (z - 3000.0) ** 2)
        return 300.0 + 2.0 * np.maximum(0.0, 1.0 - r / 2000.0)

    def qv(x, y, z):
        return 5e-3 * np.exp(-z / 2e3) * (1 + 0.2 * np.sin(2 * np.pi * x / 8e3)) + 0 * y

    rho = om.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
    om.set(rho=rho, theta=theta, u=lambda x, y, z: 3.0 + 0 * x + 0 * y + 0 * z, v=0.0, w=0.0, qv=qv)
    hm.set(ρ=rho, θ=theta, u=lambda x, y, z: 3.0 + 0 * x + 0 * y + 0 * z, v=0.0, w=0.0, qᵗ=qv)
    for _ in range(2):
        om.time_step(0.5)
        hm.time_step(0.5)
    worst = tg.cmp_interior(om, hm, ("rho_d", "rtheta", "rq", "ru", "rv", "rw", "T", "p"), 5e-9)
    print("compressible Nz=128:", {k: f"{v:.1e}" for k, v in worst.items()})
