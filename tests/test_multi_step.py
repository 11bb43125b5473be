"""Multi-step seam, bz_time_steps_anelastic: the loop of the reference's benchmark driver, many_time_steps!
(/root/reference/benchmarking/src/timestepping.jl:11-16) — n steps with nothing reading the model in between.  On the lean tier every
step but the last ends with the momentum-only projection; with diagnose_last the state on return must carry the very bits n single
steps leave (halos included: whole parent arrays are compared), without it the prognostic state must and update_state! must
rebuild the diagnostics."""
import numpy as np
import pytest

from helpers import bubble_theta

pytestmark = pytest.mark.gpu


def _model(bz, size=(64, 16, 16), **kw):
    grid = bz.RectilinearGrid(size, x=(-10e3, 10e3), y=(-10e3, 10e3), z=(0, 10e3))
    m = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, potential_temperature=300)), advection=bz.WENO(), **kw)
    m.set(θ=bubble_theta(300.0, 9.81), u=1.5, v=-0.5)
    return m


def _all_fields(m):
    f = dict(m.prognostic_fields())
    f.update({"u": m.velocities["u"], "v": m.velocities["v"], "w": m.velocities["w"], "θ": m.potential_temperature,
              "q": m.specific_moisture, "T": m.temperature, "ϕ": m.dynamics.pressure_anomaly})
    return f


def _assert_identical(a, b, names=None):
    fa, fb = _all_fields(a), _all_fields(b)
    for k in (names or fa):
        assert np.array_equal(fa[k].cpu(), fb[k].cpu()), k


@pytest.mark.parametrize("n", [1, 2, 3, 4])
def test_many_steps_with_last_diagnosis_are_bitwise_n_single_steps(bz, n):
    a, b = _model(bz), _model(bz)
    for _ in range(n):
        a.time_step(2.0)
    b.time_steps(2.0, n, diagnose_last=True)
    a.synchronize(); b.synchronize()
    assert b.clock.iteration == n and b.clock.time == a.clock.time
    assert not bz.diagnostics_stale(b)
    _assert_identical(a, b)


@pytest.mark.parametrize("n", [1, 2, 3])
def test_undiagnosed_steps_then_update_state(bz, n):
    """diagnose_last = False leaves the prognostic state current (possibly parked in the G slots) and the diagnostics stale;
    update_state! settles it: prognostic fields bit for bit, diagnostics as the per-operator kernels write them (1e-13: the fused
    diagnosis and the per-operator kernels are separate compilations)."""
    a, b = _model(bz), _model(bz)
    for _ in range(n):
        a.time_step(2.0)
    b.time_steps(2.0, n, diagnose_last=False)
    assert bz.diagnostics_stale(b)
    bz.update_state_(b, compute_tendencies=False)
    assert not bz.diagnostics_stale(b)
    a.synchronize(); b.synchronize()
    prog = list(a.prognostic_fields())
    for k in prog:      # interiors bit for bit (the halos are refilled by update_state!, same values)
        assert np.array_equal(a.prognostic_fields()[k].cpu(), b.prognostic_fields()[k].cpu()), k
    fa, fb = _all_fields(a), _all_fields(b)
    for k in ("u", "v", "w", "θ", "q", "T"):
        x, y = fa[k].cpu(), fb[k].cpu()
        assert np.max(np.abs(x - y)) <= 1e-13 * max(np.max(np.abs(x)), 1e-30), k


def test_stepping_goes_on_after_undiagnosed_steps(bz):
    """3 undiagnosed + 2 diagnosed steps == 5 single steps, bit for bit; so do 1 + (single step) + 2 undiagnosed + 1 diagnosed."""
    a, b, c = _model(bz), _model(bz), _model(bz)
    for _ in range(5):
        a.time_step(2.0)
    b.time_steps(2.0, 3, diagnose_last=False)
    b.time_steps(2.0, 2, diagnose_last=True)
    c.time_steps(2.0, 1, diagnose_last=False)
    c.time_step(2.0)
    c.time_steps(2.0, 2, diagnose_last=False)
    c.time_steps(2.0, 1, diagnose_last=True)
    for m in (a, b, c):
        m.synchronize()
    _assert_identical(a, b)
    _assert_identical(a, c)


def test_many_steps_with_the_cbl_forcing_stack(bz):
    """the reference benchmark's own case rides the lean tier (FPlane + geostrophic forcing + bottom fluxes): same equality."""
    from breeze_jl_amd import benchmarks
    kw = dict(size=(64, 32, 16), float_type=np.float64, advection=bz.WENO(order=5))
    a, b = benchmarks.convective_boundary_layer(**kw), benchmarks.convective_boundary_layer(**kw)
    for _ in range(3):
        a.time_step(0.05)
    b.time_steps(0.05, 3, diagnose_last=True)
    a.synchronize(); b.synchronize()
    _assert_identical(a, b)


def test_other_tiers_diagnose_every_step(bz):
    """a model outside the lean tier (saturation adjustment) takes the same call and simply diagnoses every step"""
    grid = bz.RectilinearGrid((32, 16, 16), x=(-10e3, 10e3), y=(-10e3, 10e3), z=(0, 10e3))

    def make():
        m = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, potential_temperature=300)),
                               advection=bz.WENO(), microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium()))
        m.set(θ=bubble_theta(300.0, 9.81), qᵗ=0.012, u=1.0)
        return m

    a, b = make(), make()
    for _ in range(2):
        a.time_step(2.0)
    b.time_steps(2.0, 2, diagnose_last=False)
    a.synchronize(); b.synchronize()
    assert not bz.diagnostics_stale(b)
    _assert_identical(a, b)


def _run_slab_ranks(bz, size, world, plan, dt):
    """`plan`: list of (n, diagnose_last) calls; () entries mean one single time_step.  Ranks are host threads sharing cuda:0."""
    import threading
    import uuid
    import torch
    from breeze_jl_amd import distributed as bz_dist
    from test_comm import EXTENT, q_ic, theta_ic
    G = bz.RectilinearGrid(size, x=EXTENT[0], y=EXTENT[1], z=EXTENT[2])
    group = "local:" + uuid.uuid4().hex
    models, errors = [None] * world, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                m = bz_dist.SlabAtmosphereModel(G, rank, world, advection=bz.WENO(), potential_temperature=300, device="cuda:0", transport=group)
                m.set(θ=theta_ic, u=3.0, v=-2.0, qᵗ=q_ic)
                for call in plan:
                    if call:
                        m.time_steps(dt, call[0], diagnose_last=call[1])
                    else:
                        m.time_step(dt)
                m.synchronize()
            models[rank] = m
        except Exception as e:      # noqa: BLE001
            import traceback
            errors.append((rank, repr(e), traceback.format_exc()))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    return models


@pytest.mark.parametrize("world", [1, 2])
def test_slab_ranks_multi_step_is_bitwise_the_single_step_sequence(bz, world):
    """the distributed lean step takes the same flag: undiagnosed steps exchange five halo fields (asynchronously, like stages 1-2)
    instead of ten; the final state — y halos included — carries the bits of five single distributed steps"""
    from test_comm import FIELDS
    size = (32, 32, 16)
    a = _run_slab_ranks(bz, size, world, [(), (), (), (), ()], 2.0)
    b = _run_slab_ranks(bz, size, world, [(3, False), (2, True)], 2.0)
    c = _run_slab_ranks(bz, size, world, [(2, False), (), (2, True)], 2.0)
    for name, get in FIELDS.items():
        for ma, mb, mc in zip(a, b, c):
            assert np.array_equal(get(ma).cpu(), get(mb).cpu()), name
            assert np.array_equal(get(ma).cpu(), get(mc).cpu()), name


def test_consumers_of_stale_diagnostics_rebuild_or_refuse(bz):
    """after time_steps(..., diagnose_last=False) the velocities / θ / qᵛ / T are older than the prognostic state.  The C entry
    points that read them rebuild them when they are handed the state (bz_compute_tendencies) or refuse (bz_cell_advection_timescale takes
    bare pointers: BZ_ERR_INVALID + message); the host wrappers (cell_advection_timescale, Field.cpu()) rebuild first."""
    import ctypes as C
    a, b, c = _model(bz), _model(bz), _model(bz)
    for _ in range(2):
        a.time_step(2.0)
    a.synchronize()
    want_tau = bz.cell_advection_timescale(a)
    # (i) the C entry point refuses to answer from stale velocities
    b.time_steps(2.0, 2, diagnose_last=False)
    assert bz.diagnostics_stale(b)
    out = b._T.real()
    rc = b._lib.bz_cell_advection_timescale(b._ctx, C.c_void_p(b.velocities["u"].ptr()), C.c_void_p(b.velocities["v"].ptr()),
                                            C.c_void_p(b.velocities["w"].ptr()), C.byref(out))
    assert rc != 0 and b"stale" in b._lib.bz_last_error(b._ctx)
    # (ii) the host wrapper rebuilds, then answers what the diagnosed run answers (1e-13: separate compilations of the diagnosis)
    tau = bz.cell_advection_timescale(b)
    assert not bz.diagnostics_stale(b)
    assert abs(tau - want_tau) <= 1e-12 * want_tau
    # (iii) a host read of a diagnostic field rebuilds as well
    c.time_steps(2.0, 2, diagnose_last=False)
    assert bz.diagnostics_stale(c)
    w = c.velocities["w"].cpu()
    assert not bz.diagnostics_stale(c)
    ref = a.velocities["w"].cpu()
    assert np.max(np.abs(w - ref)) <= 1e-13 * max(np.max(np.abs(ref)), 1e-30)
    # (iv) bz_compute_tendencies with the state at hand rebuilds before it reads u, v, w, θ, qᵛ, T
    d, e = _model(bz), _model(bz)
    d.time_steps(2.0, 2, diagnose_last=True)
    e.time_steps(2.0, 2, diagnose_last=False)
    bz.compute_tendencies_(d)
    bz.compute_tendencies_(e)
    assert not bz.diagnostics_stale(e)
    d.synchronize(); e.synchronize()
    for k in d.G:
        x, y = d.G[k].cpu(), e.G[k].cpu()
        assert np.max(np.abs(x - y)) <= 1e-12 * max(np.max(np.abs(x)), 1e-30), k
