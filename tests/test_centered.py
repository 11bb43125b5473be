"""advection = Centered(order = 2), the AtmosphereModel constructor's default in the reference (every reference test that builds
`AtmosphereModel(grid)` without an advection keyword runs it).  Same kernels and same oracle sources, compiled with the
reconstructions collapsed to 2-point symmetric means (libbreeze_hip_centered2.so / libbreeze_oracle_centered2.so).  The Oceananigans
scheme is not vendored: parity unpinned beyond the closed forms checked here."""
import numpy as np
import pytest

from helpers import PROG, bubble_theta, push_state, randomize, relerr


def test_centered_scalar_flux_is_the_two_point_mean_times_mass_flux(oracle):
    """Uniform wind U, c = sin(kx): G = -rho U (c[i+1] - c[i-1]) / (2 dx) exactly; a quadratic-in-z profile advected by a
    uniform w would need walls, so the vertical direction is checked through conservation below."""
    g = oracle.Grid((16, 12, 10), x=(0, 1600.0), y=(0, 1200.0), z=(0, 1000.0))
    m = oracle.OracleModel(g, advection="Centered2")
    k = 2 * np.pi / 1600.0
    m.set(theta=lambda x, y, z: 288 + np.sin(k * x) + 0 * y + 0 * z, u=3.0, enforce_mass_conservation=False)
    m.update_state()
    rho = m.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
    th = g.interior(m.theta)
    want = -rho * 3.0 * (np.roll(th, -1, 2) - np.roll(th, 1, 2)) / (2 * g.dx)
    assert np.abs(g.interior(m.G["rtheta"]) - want).max() < 1e-13


def test_centered_bubble_conserves_and_projects(oracle):
    g = oracle.Grid((16, 12, 16), x=(-4e3, 4e3), y=(-3e3, 3e3), z=(0.0, 8e3))
    m = oracle.OracleModel(g, potential_temperature=300.0, advection="Centered2")
    m.set(theta=bubble_theta(300.0, 9.81, r0=2e3, zc=2500.0))
    s0 = g.interior(m.rtheta).sum()
    for _ in range(3):
        m.time_step(2.0)
    assert abs(g.interior(m.rtheta).sum() - s0) < 1e-12 * s0
    assert np.abs(m.divergence()).max() < 1e-10
    assert np.abs(g.interior(m.w, True)).max() > 1e-2


@pytest.mark.gpu
def test_centered_tendencies_and_steps_match_oracle(oracle, bz):
    size, ext = (32, 20, 16), ((-4e3, 4e3), (-3e3, 3e3), (0.0, 8e3))
    og = oracle.Grid(size, x=ext[0], y=ext[1], z=ext[2])
    om = oracle.OracleModel(og, potential_temperature=300.0, advection="Centered2")
    grid = bz.RectilinearGrid(size, x=ext[0], y=ext[1], z=ext[2])
    ref = bz.ReferenceState(grid, potential_temperature=300.0)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref))          # advection defaults to Centered(order = 2)
    assert isinstance(hm.advection, bz.Centered) and "centered2" in hm._lib._name
    randomize(om, seed=4)
    om.compute_tendencies()
    push_state(om, hm, names=("ru", "rv", "rw", "rtheta", "rq"))
    bz.update_state_(hm, compute_tendencies=True)
    hm.synchronize()
    for n, k in PROG.items():
        zf = n == "rw"
        want, got = og.interior(om.G[n], zface=zf), hm.G[k].interior_cpu()
        if zf:
            want, got = want[1:-1], got[1:-1]
        assert relerr(got, want) < 1e-12, n
    th = bubble_theta(300.0, 9.81, r0=2e3, zc=2500.0)
    for whole in (True, False):
        om2 = oracle.OracleModel(og, potential_temperature=300.0, advection="Centered2")
        hm2 = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.Centered(order=2))
        om2.set(theta=th, u=2.0)
        hm2.set(θ=th, u=2.0)
        for _ in range(3):
            om2.time_step(2.0)
            bz.time_step_(hm2, 2.0, whole_step=whole)
        hm2.synchronize()
        mom = max(np.abs(og.interior(getattr(om2, n), zface=(n == "rw"))).max() for n in ("ru", "rv", "rw"))
        for n, k in PROG.items():
            want = og.interior(getattr(om2, n), zface=(n == "rw"))
            got = hm2.prognostic_fields()[k].interior_cpu()
            scale = mom if n in ("ru", "rv", "rw") else max(np.abs(want).max(), 1e-3)
            assert np.abs(got - want).max() / scale < 1e-10, (n, whole)
    # a WENO-5 model in the same process keeps its own library
    hm3 = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5))
    assert "centered2" not in hm3._lib._name


@pytest.mark.gpu
def test_centered_compressible_steps_match_oracle(oracle, bz):
    """CompressibleDynamics with the default advection (Centered(order = 2)): slow tendencies and the moisture transport on the
    collapsed stencils; two split-explicit steps."""
    from oracle import oracle_compressible as oc
    size, ext = (16, 12, 12), dict(x=(0.0, 8e3), y=(0.0, 6e3), z=(0.0, 6e3))
    og = oracle.Grid(size, **ext)
    om = oc.CompressibleOracleModel(og, time_discretization=oc.SplitExplicit(substeps=6), surface_pressure=1e5,
                                    reference_potential_temperature=300.0, advection="Centered2")
    grid = bz.RectilinearGrid(size, **ext)
    dyn = bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(substeps=6), surface_pressure=1e5,
                                  reference_potential_temperature=300.0)
    hm = bz.CompressibleAtmosphereModel(grid, dyn)
    assert isinstance(hm.advection, bz.Centered)
    th = lambda x, y, z: 300.0 + 2.0 * np.maximum(0.0, 1.0 - np.sqrt((x - 4e3) ** 2 + (y - 3e3) ** 2 + (z - 2e3) ** 2) / 1.5e3)
    qv = lambda x, y, z: 4e-3 * np.exp(-z / 2e3) + 0 * x + 0 * y
    rho = om.ref.density[og.Hz:og.Hz + og.Nz][:, None, None]
    om.set(rho=rho, theta=th, u=3.0, v=-1.0, w=0.0, qv=qv)
    hm.set(ρ=rho, θ=th, u=3.0, v=-1.0, w=0.0, qᵗ=qv)
    for _ in range(2):
        om.time_step(2.0)
        hm.time_step(2.0)
    hm.synchronize()
    getters = {"rho_d": hm.dynamics.dry_density, "ru": hm.momentum["ρu"], "rv": hm.momentum["ρv"], "rw": hm.momentum["ρw"],
               "rtheta": hm.potential_temperature_density, "rq": hm.moisture_density, "T": hm.temperature, "p": hm.dynamics.pressure}
    mom = max(np.abs(og.interior(getattr(om, n), n == "rw")).max() for n in ("ru", "rv", "rw"))
    for n, f in getters.items():
        want = og.interior(getattr(om, n), n == "rw")
        scale = mom if n in ("ru", "rv", "rw") else np.abs(want).max()
        assert np.abs(f.interior_cpu() - want).max() / scale < 5e-9, n
