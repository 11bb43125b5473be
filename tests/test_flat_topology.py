"""(Periodic, Flat, Bounded): BASELINE configs[0], the reference's 2-D x-z dry thermal bubble (README.md:67-75,
examples/dry_thermal_bubble.jl), on the device as a genuinely two-dimensional problem (Ny = 1, no y halos): the per-operator
WENO-5 kernels drop their y terms, the pressure solve transforms rows only.  Checked against the oracle's own Flat implementation
(parity status as for the 3-D path: WENO / halos recalled from Oceananigans, unpinned) and against the y-invariant 3-D run."""
import numpy as np
import pytest

from helpers import PROG, relerr

pytestmark = pytest.mark.gpu
EXT = dict(x=(-10e3, 10e3), z=(0.0, 10e3))


def theta2d(x, z):      # README.md:71: 300 + 2 cos^2(pi/2 min(1, r / 2000)), r from (0, 2000 m)
    r = np.sqrt(x ** 2 + (z - 2000.0) ** 2)
    return 300.0 + 2.0 * np.cos(np.pi / 2 * np.minimum(1.0, r / 2000.0)) ** 2


def _pair(oracle, bz, size):
    og = oracle.Grid(size, x=EXT["x"], z=EXT["z"], topology=("Periodic", "Flat", "Bounded"))
    om = oracle.OracleModel(og, surface_pressure=101325.0, potential_temperature=300.0)
    grid = bz.RectilinearGrid(size, x=EXT["x"], z=EXT["z"], topology=(bz.Periodic, bz.Flat, bz.Bounded))
    ref = bz.ReferenceState(grid, surface_pressure=101325.0, potential_temperature=300.0)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5))
    return om, hm


def test_two_dimensional_tendencies_match_oracle(oracle, bz):
    import torch
    om, hm = _pair(oracle, bz, (48, 40))
    g = om.grid
    rng = np.random.default_rng(3)
    rho = om.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
    sh = (g.Nz, 1, g.Nx)
    g.interior(om.ru)[...] = rho * 4.0 * rng.standard_normal(sh)
    g.interior(om.rv)[...] = rho * 3.0 * rng.standard_normal(sh)        # v is advected as a passive component in 2-D
    wi = np.zeros((g.Nz + 1, 1, g.Nx))
    wi[1:-1] = 2.0 * rng.standard_normal((g.Nz - 1, 1, g.Nx))
    g.interior(om.rw, True)[...] = wi
    g.interior(om.rtheta)[...] = rho * (300.0 + 2.0 * rng.standard_normal(sh))
    g.interior(om.rq)[...] = rho * np.abs(5e-3 * rng.standard_normal(sh))
    om.update_state(compute_tendencies=True)
    from helpers import ORACLE_TO_HIP
    for n in ("ru", "rv", "rw", "rtheta", "rq"):
        ORACLE_TO_HIP[n](hm).parent.copy_(torch.from_numpy(getattr(om, n)))
    bz.update_state_(hm, compute_tendencies=True)
    hm.synchronize()
    for n, f in (("u", hm.velocities["u"]), ("w", hm.velocities["w"]), ("T", hm.temperature)):
        assert relerr(f.cpu(), getattr(om, n)) < 1e-14, n
    for n, k in PROG.items():
        zf = n == "rw"
        want, got = g.interior(om.G[n], zface=zf), hm.G[k].interior_cpu()
        if zf:
            want, got = want[1:-1], got[1:-1]
        assert relerr(got, want) < 1e-12, (n, relerr(got, want))


def test_config0_two_dimensional_bubble_steps_match_oracle(oracle, bz):
    """README.md:67-75 at reduced resolution (64 x 64 instead of 256 x 256 so that the CPU side stays in seconds), dt = 2 s"""
    om, hm = _pair(oracle, bz, (64, 64))
    om.set(theta=lambda x, y, z: theta2d(x, z) + 0 * y)
    hm.set(θ=theta2d)                                  # f(x, z), as on the reference's Flat grids
    for _ in range(3):
        om.time_step(2.0)
        hm.time_step(2.0)
    hm.synchronize()
    g = om.grid
    for n, k in PROG.items():
        got = hm.prognostic_fields()[k].interior_cpu()
        want = g.interior(getattr(om, n), zface=(n == "rw"))
        scale = max(np.max(np.abs(want)), 1e-3)
        assert np.max(np.abs(got - want)) / scale < 1e-9, (n, np.max(np.abs(got - want)) / scale)
    assert np.abs(hm.momentum["ρv"].interior_cpu()).max() == 0.0
    scale = np.max(np.abs(g.interior(om.ru))) / g.dx
    assert hm.max_abs_divergence() < 1e-12 * max(scale, 1e-6)


def test_two_dimensional_run_equals_the_y_invariant_three_dimensional_run(bz):
    size2, Ny = (64, 48), 8
    g2 = bz.RectilinearGrid(size2, x=EXT["x"], z=EXT["z"], topology=(bz.Periodic, bz.Flat, bz.Bounded))
    g3 = bz.RectilinearGrid((size2[0], Ny, size2[1]), x=EXT["x"], y=(0.0, 1.0 * Ny), z=EXT["z"])
    out = []
    for g in (g2, g3):
        m = bz.AtmosphereModel(g, dynamics=bz.AnelasticDynamics(bz.ReferenceState(g, potential_temperature=300.0)), advection=bz.WENO(order=5))
        m.set(θ=(theta2d if g is g2 else (lambda x, y, z: theta2d(x, z) + 0 * y)), u=2.0)
        for _ in range(3):
            m.time_step(2.0)
        m.synchronize()
        out.append({k: f.interior_cpu()[:, 0, :] for k, f in m.prognostic_fields().items()})
    for k in out[0]:
        scale = max(np.abs(out[1][k]).max(), 1e-3)
        assert np.abs(out[0][k] - out[1][k]).max() / scale < 1e-10, k
