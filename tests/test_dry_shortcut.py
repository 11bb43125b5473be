"""Moisture scan of the lean seam (csrc/bz_step.hip: bzi_scan_moisture; round 4).  The reference advects rho q^v in every model, dry ones
included (/root/reference/src/AtmosphereModels/update_atmosphere_model_state.jl:333-343); where the scan that opens every step call finds
the field identically zero, the scalar-pair and z-momentum kernels skip every access to it.  The tests demand what makes that legitimate:
the same bits as the general path, and a moisture field that appears later — through set! or behind the library's back — is seen."""
import numpy as np
import pytest

from helpers import bubble_theta

pytestmark = pytest.mark.gpu


def _model(bz, q=None, size=(64, 16, 24)):
    grid = bz.RectilinearGrid(size, x=(-10e3, 10e3), y=(-10e3, 10e3), z=(0, 10e3))
    m = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, potential_temperature=300)), advection=bz.WENO())
    kw = dict(θ=bubble_theta(300.0, 9.81), u=2.0, v=-1.0)
    if q is not None:
        kw["qᵗ"] = q
    m.set(**kw)
    return m


def _fields(m):
    f = dict(m.prognostic_fields())
    f.update({"T": m.temperature, "q": m.specific_moisture, "w": m.velocities["w"]})
    return {k: v.cpu() for k, v in f.items()}


def _equal(a, b):
    fa, fb = _fields(a), _fields(b)
    for k in fa:
        assert np.array_equal(fa[k], fb[k]), k


def test_dry_run_with_and_without_the_shortcut_is_bitwise_equal(bz, monkeypatch):
    a = _model(bz)
    a.time_steps(2.0, 4)
    monkeypatch.setenv("BZ_NO_DRY_SHORTCUT", "1")
    b = _model(bz)
    b.time_steps(2.0, 4)
    a.synchronize(); b.synchronize()
    _equal(a, b)
    assert float(np.abs(a.moisture_density.cpu()).max()) == 0.0


def test_moist_run_is_untouched_by_the_scan(bz, monkeypatch):
    q = lambda x, y, z: 5e-3 * np.exp(-z / 2500.0) * (1 + 0.2 * np.sin(2 * np.pi * x / 20e3)) + 0 * y      # noqa: E731
    a = _model(bz, q)
    a.time_steps(2.0, 3)
    monkeypatch.setenv("BZ_NO_DRY_SHORTCUT", "1")
    b = _model(bz, q)
    b.time_steps(2.0, 3)
    a.synchronize(); b.synchronize()
    _equal(a, b)


def test_moisture_that_appears_later_is_seen(bz, monkeypatch):
    """two dry steps, then moisture (i) through set! and (ii) written straight into the field's array without telling the library —
    a dry model is scanned at every call — each against the same procedure on a model that never takes the shortcut"""
    import torch
    q = lambda x, y, z: 4e-3 * np.exp(-z / 3000.0) * (1 + 0.3 * np.cos(2 * np.pi * y / 20e3)) + 0 * x      # noqa: E731

    def procedure(how):
        m = _model(bz)
        m.time_steps(2.0, 2)
        if how == "set":
            m.set(qᵗ=q)                                     # set! ends with update_state!: the scan state is reset
        else:                                               # behind the library's back: interior written, halos filled, no update_state!
            x, y, z = m.grid.nodes(m.moisture_density.loc)
            Hz, Nz = m.grid.Hz, m.grid.Nz
            rho = torch.from_numpy(m.dynamics.reference_state.density[Hz:Hz + Nz].copy()).to(m.device)[:, None, None]
            qv = torch.from_numpy(np.broadcast_to(q(x, y, z), tuple(m.moisture_density.interior.shape)).copy()).to(m.device)
            m.moisture_density.interior.copy_(rho * qv)
            bz.fill_halo_regions_(m, m.moisture_density)
        m.time_steps(2.0, 2)
        m.synchronize()
        return m

    for how in ("set", "backdoor"):
        monkeypatch.delenv("BZ_NO_DRY_SHORTCUT", raising=False)
        a = procedure(how)
        monkeypatch.setenv("BZ_NO_DRY_SHORTCUT", "1")
        b = procedure(how)
        _equal(a, b)
        assert float(np.abs(a.moisture_density.interior_cpu()).max()) > 1e-3      # the moisture is there and has been advected


def test_contexts_with_a_moisture_source_never_take_the_shortcut(bz):
    """the CBL stack has no moisture flux (shortcut on: a moisture_scan group shows in the profile); the BOMEX stack has one (no scan)"""
    m = bz.benchmarks.convective_boundary_layer((64, 32, 16), float_type=np.float64, advection=bz.WENO(order=5), halo=(3, 3, 3))
    m.profile_enable(True)
    m.time_steps(0.05, 2)
    m.synchronize()
    assert "moisture_scan" in m.profile()


def _slab_run(bz, world, q, steps, size=(32, 32, 16)):
    """`steps` steps in ONE bz_time_steps_anelastic call on `world` y-slab ranks (host threads sharing cuda:0, in-process transport)"""
    import threading
    import uuid
    import torch
    from breeze_jl_amd import distributed as bz_dist
    G = bz.RectilinearGrid(size, x=(-10e3, 10e3), y=(-10e3, 10e3), z=(0, 10e3))
    group = "local:" + uuid.uuid4().hex
    models, errors = [None] * world, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                m = bz_dist.SlabAtmosphereModel(G, rank, world, advection=bz.WENO(), potential_temperature=300, device="cuda:0", transport=group)
                m.set(θ=bubble_theta(300.0, 9.81), u=3.0, v=40.0, qᵗ=q)
                m.time_steps(2.0, steps, diagnose_last=True)
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


def test_moisture_localised_to_one_slab_crosses_the_slab_edge(bz, monkeypatch):
    """the scan's verdict has to be the DOMAIN's.  A vapour blob sits wholly inside rank 0's slab, four rows from rank 1
    (whose slab and halo rows are exactly zero when the call starts) and is blown across the edge during a four-step call.  With a
    rank-local verdict rank 1 would take the dry path for the whole call and drop what arrives through its halo rows.  Demanded: the same
    bits as the run that never takes the shortcut, moisture inside rank 1 at the end, and the single-GPU fields to 1e-11."""
    def q(x, y, z):      # cone of radius 3 km around y = -5 km: rows j <= 12 of 32 (y < -1.9 km); rank 1 owns rows 16 .. 31, its halo rows are 13 .. 15
        return 5e-3 * np.maximum(0.0, 1.0 - np.sqrt((y + 5000.0) ** 2 + (z - 3000.0) ** 2) / 3000.0) + 0 * x
    size, steps = (32, 32, 16), 4
    a = _slab_run(bz, 2, q, steps, size)
    monkeypatch.setenv("BZ_NO_DRY_SHORTCUT", "1")
    b = _slab_run(bz, 2, q, steps, size)
    monkeypatch.delenv("BZ_NO_DRY_SHORTCUT")
    for ma, mb in zip(a, b):
        fa, fb = _fields(ma), _fields(mb)
        for k in fa:
            assert np.array_equal(fa[k], fb[k]), k
    assert float(np.abs(a[1].moisture_density.interior_cpu()).max()) > 0.0      # the blob did reach rank 1
    G = bz.RectilinearGrid(size, x=(-10e3, 10e3), y=(-10e3, 10e3), z=(0, 10e3))
    ref = bz.AtmosphereModel(G, dynamics=bz.AnelasticDynamics(bz.ReferenceState(G, potential_temperature=300)), advection=bz.WENO())
    ref.set(θ=bubble_theta(300.0, 9.81), u=3.0, v=40.0, qᵗ=q)
    ref.time_steps(2.0, steps)
    ref.synchronize()
    want = ref.moisture_density.interior_cpu()
    got = np.concatenate([m.moisture_density.interior_cpu() for m in a], axis=1)
    assert np.max(np.abs(got - want)) <= 1e-11 * np.max(np.abs(want))


def test_dry_slab_ranks_keep_the_shortcut_and_its_bits(bz, monkeypatch):
    """a dry domain on two ranks: the all-reduced verdict is "dry" on both, and the result carries the bits of the no-shortcut run"""
    a = _slab_run(bz, 2, 0.0, 3)
    monkeypatch.setenv("BZ_NO_DRY_SHORTCUT", "1")
    b = _slab_run(bz, 2, 0.0, 3)
    for ma, mb in zip(a, b):
        fa, fb = _fields(ma), _fields(mb)
        for k in fa:
            assert np.array_equal(fa[k], fb[k]), k
        assert float(np.abs(ma.moisture_density.cpu()).max()) == 0.0


def test_moisture_written_behind_a_slab_rank_between_calls(bz, monkeypatch):
    """two ranks step a dry domain (verdict "dry" on both), then rank 0 alone gets vapour written straight into its array and both step on:
    the next call's all-reduced scan must turn BOTH ranks moist at the same call (a rank that kept scanning while its peer had stopped
    would hang in the all-reduce), and the result carries the bits of the run that never takes the shortcut"""
    import threading
    import uuid
    import torch
    from breeze_jl_amd import distributed as bz_dist
    size = (32, 32, 16)

    def run_all():
        G = bz.RectilinearGrid(size, x=(-10e3, 10e3), y=(-10e3, 10e3), z=(0, 10e3))
        group = "local:" + uuid.uuid4().hex
        models, errors = [None, None], []

        def run(rank):
            try:
                torch.cuda.set_device(0)
                with torch.cuda.stream(torch.cuda.Stream()):
                    m = bz_dist.SlabAtmosphereModel(G, rank, 2, advection=bz.WENO(), potential_temperature=300, device="cuda:0", transport=group)
                    m.set(θ=bubble_theta(300.0, 9.81), u=3.0, v=40.0)
                    m.time_steps(2.0, 2)
                    m.synchronize()
                    if rank == 0:      # behind the library's back: interior rows 4 .. 9 of rank 0's slab
                        m.moisture_density.interior[:, 4:10, :] = 4e-3
                    m.time_steps(2.0, 2)
                    m.time_steps(2.0, 1)
                    m.synchronize()
                models[rank] = m
            except Exception as e:      # noqa: BLE001
                import traceback
                errors.append((rank, repr(e), traceback.format_exc()))

        threads = [threading.Thread(target=run, args=(r,)) for r in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=120)
        assert not any(t.is_alive() for t in threads), "a rank hung"
        assert not errors, errors
        return models

    a = run_all()
    monkeypatch.setenv("BZ_NO_DRY_SHORTCUT", "1")
    b = run_all()
    for ma, mb in zip(a, b):
        fa, fb = _fields(ma), _fields(mb)
        for k in fa:
            assert np.array_equal(fa[k], fb[k]), k
    assert float(np.abs(a[0].moisture_density.interior_cpu()).max()) > 0.0
