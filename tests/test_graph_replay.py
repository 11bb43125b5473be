"""hipGraph replay of whole steps (csrc/bz_graph.hip): a replayed step must leave exactly the bits the ordinary launch sequence
leaves — same kernels, same order, same arguments — for every tier of the step (lean seam, fused-RK tier with the BOMEX physics
list, per-operator 2-D model, compressible split-explicit with Kessler), and a changed dt must be a different recording."""
import numpy as np
import pytest

from helpers import bubble_theta

pytestmark = pytest.mark.gpu


def _run(make, steps, dts, graph):
    m = make()
    m.graph_enable(graph)
    for i in range(steps):
        m.time_step(dts[i % len(dts)])
    m.synchronize()
    fields = {k: f.interior_cpu().copy() for k, f in m.prognostic_fields().items()}
    return fields, m.graph_info()


def _check(make, steps=6, dts=(2.0,), captures=1):
    a, info_a = _run(make, steps, dts, True)
    b, info_b = _run(make, steps, dts, False)
    assert info_a[0] and info_a[1] == captures, info_a
    assert info_a[2] >= steps - 2 * captures - (len(dts) - 1), info_a      # first sighting runs directly, second records
    assert info_b == (False, 0, 0)      # replay is opt-in
    for k in a:
        assert np.isfinite(a[k]).all()
        assert np.array_equal(a[k], b[k]), (k, np.abs(a[k] - b[k]).max())


def test_lean_seam_replay_is_bit_identical(bz):
    def make():
        g = bz.RectilinearGrid((32, 24, 16), x=(-4e3, 4e3), y=(-3e3, 3e3), z=(0.0, 8e3))
        m = bz.AtmosphereModel(g, dynamics=bz.AnelasticDynamics(bz.ReferenceState(g, potential_temperature=300.0)), advection=bz.WENO(order=5))
        m.set(θ=bubble_theta(300.0, 9.81, r0=2e3, zc=2500.0), u=3.0)
        return m
    _check(make)


def test_two_time_step_sizes_are_two_recordings(bz):
    def make():
        g = bz.RectilinearGrid((32, 16, 16), x=(-4e3, 4e3), y=(-2e3, 2e3), z=(0.0, 8e3))
        m = bz.AtmosphereModel(g, dynamics=bz.AnelasticDynamics(bz.ReferenceState(g, potential_temperature=300.0)), advection=bz.WENO(order=5))
        m.set(θ=bubble_theta(300.0, 9.81, r0=2e3, zc=2500.0))
        return m
    a, info = _run(make, 8, (2.0, 1.0), True)
    b, _ = _run(make, 8, (2.0, 1.0), False)
    assert info[1] == 2 and info[2] == 4, info
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_two_dimensional_model_replay_is_bit_identical(bz):
    def make():
        g = bz.RectilinearGrid((64, 48), x=(-10e3, 10e3), z=(0.0, 10e3), topology=(bz.Periodic, bz.Flat, bz.Bounded))
        m = bz.AtmosphereModel(g, dynamics=bz.AnelasticDynamics(bz.ReferenceState(g, potential_temperature=300.0)), advection=bz.WENO(order=5))
        th = bubble_theta(300.0, 9.81, r0=2e3, zc=3e3)
        m.set(θ=lambda x, z: th(x, 0 * x, z))
        return m
    _check(make)


def test_bomex_physics_replay_is_bit_identical(bz):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_forcings import EXTENT as FEXT, _hip_forcing_kwargs

    def make():
        g = bz.RectilinearGrid((32, 32, 16), x=FEXT[0], y=FEXT[1], z=FEXT[2])
        ref = bz.ReferenceState(g, surface_pressure=101500.0, potential_temperature=299.1)
        m = bz.AtmosphereModel(g, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5), closure=bz.SmagorinskyLilly(),
                               microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium()), **_hip_forcing_kwargs(bz))
        rng = np.random.default_rng(2)
        m.set(θ=299.1 + 0.3 * rng.standard_normal((16, 32, 32)), qᵗ=0.016 + 1e-3 * rng.standard_normal((16, 32, 32)), u=-6.0)
        return m
    _check(make, dts=(3.0,))


def test_compressible_kessler_replay_is_bit_identical(bz):
    def make():
        size, extent = (24, 16, 20), dict(x=(0.0, 16e3), y=(0.0, 12e3), z=(0.0, 8e3))
        thb = lambda z: 300.0 + 0.0035 * z
        qvb = lambda z: float(0.013 * np.exp(-z / 2800.0))
        grid = bz.RectilinearGrid(size, **extent)
        tc = bz.ThermodynamicConstants(saturation_vapor_pressure=bz.TetensFormula())
        dyn = bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(substeps=6), surface_pressure=1e5,
                                      reference_potential_temperature=thb, reference_vapor_mass_fraction=qvb)
        m = bz.CompressibleAtmosphereModel(grid, dyn, advection=bz.WENO(order=5), thermodynamic_constants=tc,
                                           microphysics=bz.DCMIP2016KesslerMicrophysics())
        bub = lambda x, y, z: np.maximum(0.0, 1.0 - np.sqrt(((x - 8e3) / 4e3) ** 2 + ((y - 6e3) / 4e3) ** 2 + ((z - 1500.0) / 1500.0) ** 2))
        ref = m.dynamics.reference_state
        rho = np.asarray(ref.density)[grid.Hz:grid.Hz + grid.Nz][:, None, None] + np.zeros((size[2], size[1], size[0]))
        m.set(ρ=rho, θ=lambda x, y, z: thb(z) + 2.0 * bub(x, y, z), u=5.0, v=0.0, w=0.0,
              qᵗ=lambda x, y, z: np.vectorize(qvb)(z) + 0.003 * bub(x, y, z) + 0 * x + 0 * y)
        return m
    _check(make, steps=5)


def test_replay_on_a_non_default_stream(bz):
    """The context follows PyTorch's current stream (bz_set_stream); recording happens on the library's own stream and the replay is
    enqueued on the caller's, also when that is not the legacy default stream."""
    import torch
    out = []
    for graph in (True, False):
        with torch.cuda.stream(torch.cuda.Stream()):
            g = bz.RectilinearGrid((32, 16, 16), x=(-4e3, 4e3), y=(-2e3, 2e3), z=(0.0, 8e3))
            m = bz.AtmosphereModel(g, dynamics=bz.AnelasticDynamics(bz.ReferenceState(g, potential_temperature=300.0)), advection=bz.WENO(order=5))
            m.set(θ=bubble_theta(300.0, 9.81, r0=2e3, zc=2500.0), u=2.0)
            m.graph_enable(graph)
            for _ in range(5):
                m.time_step(2.0)
            m.synchronize()
            torch.cuda.current_stream().synchronize()
            out.append(({k: f.interior_cpu().copy() for k, f in m.prognostic_fields().items()}, m.graph_info()))
    assert out[0][1][1] == 1 and out[0][1][2] == 3, out[0][1]
    for k in out[0][0]:
        assert np.array_equal(out[0][0][k], out[1][0][k]), k
