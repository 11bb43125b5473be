"""User tracers (`tracers = (:a, :b)`): the reference's own test (test/tracer_dynamics.jl:7-25) is a smoke test —
a and b = sin / cos(2 pi x / Lx) on a (16, 8, 8) grid with u = 1, one step runs.  Restated with numbers: oracle properties on the
CPU, HIP-vs-oracle parity on the GPU."""
import numpy as np
import pytest

from helpers import bubble_theta

SIZE, EXT = (16, 8, 8), dict(x=(0, 1000.0), y=(0, 500.0), z=(0, 500.0))
A = lambda x, y, z: np.sin(2 * np.pi * x / 1000.0) + 0 * y + 0 * z
B = lambda x, y, z: np.cos(2 * np.pi * x / 1000.0) + 0 * y + 0 * z


def _oracle(oracle, n=2):
    og = oracle.Grid(SIZE, **EXT)
    return oracle.OracleModel(og, tracers=n)


def test_reference_tracer_smoke_and_conservation(oracle):
    m = _oracle(oracle)
    m.set(u=1.0, rc0=A, rc1=B)
    g = m.grid
    s0 = [g.interior(m.rc0).sum(), g.interior(m.rc1).sum()]
    a0 = g.interior(m.rc0).copy()
    m.time_step(1.0)
    assert np.isfinite(g.interior(m.rc0)).all()
    assert np.abs(g.interior(m.rc0) - a0).max() > 1e-4          # it moved
    for s, n in zip(s0, ("rc0", "rc1")):                          # flux form on a periodic / walled box
        assert abs(g.interior(getattr(m, n)).sum() - s) < 1e-12 * np.abs(g.interior(getattr(m, n))).sum()


def test_uniform_mixing_ratio_is_preserved_by_the_projected_flow(oracle):
    """c = 1 (rho c = rho_r): -div(rho u c) = -div(rho u) = 0 after the projection, so the tracer stays rho_r to round-off
    while a buoyant bubble stirs the box."""
    og = oracle.Grid((16, 12, 16), x=(-4e3, 4e3), y=(-3e3, 3e3), z=(0.0, 8e3))
    m = oracle.OracleModel(og, potential_temperature=300.0, tracers=1)
    rho = m.ref.density[og.Hz:og.Hz + og.Nz][:, None, None] + np.zeros((16, 12, 16))
    m.set(theta=bubble_theta(300.0, 9.81, r0=2e3, zc=2500.0), rc0=rho)
    for _ in range(3):
        m.time_step(2.0)
    assert np.abs(og.interior(m.w, True)).max() > 1e-2
    assert np.abs(og.interior(m.rc0) - rho).max() < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("whole_step", [True, False])
def test_tracers_match_oracle(oracle, bz, whole_step):
    og = oracle.Grid((32, 20, 16), x=(-4e3, 4e3), y=(-3e3, 3e3), z=(0.0, 8e3))
    om = oracle.OracleModel(og, potential_temperature=300.0, tracers=2)
    grid = bz.RectilinearGrid((32, 20, 16), x=(-4e3, 4e3), y=(-3e3, 3e3), z=(0.0, 8e3))
    ref = bz.ReferenceState(grid, potential_temperature=300.0)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5), tracers=(":a", "b"))
    assert list(hm.tracers) == ["a", "b"] and list(hm.prognostic_fields())[-2:] == ["a", "b"]
    th = bubble_theta(300.0, 9.81, r0=2e3, zc=2500.0)
    a = lambda x, y, z: np.sin(2 * np.pi * x / 8e3) * np.exp(-z / 4e3) + 0 * y
    b = lambda x, y, z: 1.0 + 0.5 * np.cos(2 * np.pi * y / 6e3) * (z / 8e3) + 0 * x
    om.set(theta=th, u=3.0, rc0=a, rc1=b)
    hm.tracers["a"].set_interior(a)
    hm.tracers["b"].set_interior(b)
    hm.set(θ=th, u=3.0)
    for _ in range(3):
        om.time_step(2.0)
        bz.time_step_(hm, 2.0, whole_step=whole_step)
    hm.synchronize()
    for n, k in (("rc0", "a"), ("rc1", "b")):
        want = og.interior(getattr(om, n))
        got = hm.tracers[k].interior_cpu()
        assert np.abs(got - want).max() < 1e-9 * np.abs(want).max(), n
        want_c = og.interior(getattr(om, "c" + n[2:]))
        assert np.abs(hm.specific_tracers[k].interior_cpu() - want_c).max() < 1e-9 * np.abs(want_c).max()
    want = og.interior(om.rtheta)
    assert np.abs(hm.potential_temperature_density.interior_cpu() - want).max() < 1e-10 * np.abs(want).max()
