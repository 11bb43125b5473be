"""(Periodic, Bounded, Bounded): walls in y — the topology the reference's benchmark driver exposes as PBB
(/root/reference/benchmarking/run_benchmarks.jl:130, benchmarking/src/convective_boundary_layer.jl:42-43).

What a Bounded y changes (Oceananigans semantics, recalled — the package is not vendored: PARITY UNPINNED, like the Bounded z the model
always had): rho v and v are y-face fields with impenetrable wall faces j = 0 and j = Ny; centre-in-y fields take a no-flux halo row;
WENO and the Centered advecting-flux interpolation lose order next to the walls through the same buffer cascade as in z; the
Fourier-tridiagonal solver uses the cosine transform along y with the eigenvalues (2 sin(pi j / (2 Ny)) / dy)^2; the projection never
touches the wall faces.  The reference holds no known-answer test for this topology; the oracle is checked through the properties that
define it (exact discrete projection, closed walls, conservation), the HIP path against the oracle."""
import numpy as np
import pytest

from helpers import relerr

SIZE = (16, 12, 10)
EXT = dict(x=(0.0, 1600.0), y=(0.0, 1200.0), z=(0.0, 1000.0))
TOPO = ("Periodic", "Bounded", "Bounded")


def theta0(x, y, z):
    return 300.0 + 2.0 * np.exp(-((x - 800.0) ** 2 + (y - 300.0) ** 2 + (z - 400.0) ** 2) / 200.0 ** 2)


def u0(x, y, z):
    return 1.0 + 0.5 * np.sin(2 * np.pi * y / 1200.0) + 0 * x + 0 * z


def v0(x, y, z):
    return np.sin(np.pi * y / 1200.0) * np.cos(2 * np.pi * x / 1600.0) + 0 * z


def _oracle(oracle, size=SIZE, **kw):
    g = oracle.Grid(size, topology=TOPO, **EXT)
    return g, oracle.OracleModel(g, potential_temperature=300.0, **kw)


def test_cosine_eigenvalues_and_exact_discrete_projection(oracle):
    """solve!(phi, FourierTridiagonalPoissonSolver) with a Bounded y: after the projection the discrete divergence of a random momentum
    field vanishes to round-off — the cosine modes diagonalise the staggered second difference with zero normal gradient at the walls."""
    g, m = _oracle(oracle)
    lam = oracle.poisson_eigenvalues(g.Ny, g.dy, oracle.BOUNDED)
    assert lam[0] == 0.0 and np.all(np.diff(lam) > 0) and abs(lam[-1] - (2 * np.sin((g.Ny - 1) * np.pi / (2 * g.Ny)) / g.dy) ** 2) < 1e-18
    rng = np.random.default_rng(2)
    g.interior(m.ru)[...] = rng.standard_normal(g.interior(m.ru).shape)
    g.interior(m.rv)[...] = rng.standard_normal(g.interior(m.rv).shape)
    g.interior(m.rw, True)[1:-1] = rng.standard_normal(g.interior(m.rw, True)[1:-1].shape)
    m.fill_momentum_halos()
    assert np.all(g.interior(m.rv)[:, 0, :] == 0.0) and np.all(m.rv[:, g.Hy + g.Ny, :] == 0.0)      # both wall faces
    assert np.abs(m.divergence()).max() > 1e-3
    m.compute_pressure_correction(1.0)
    m.make_pressure_correction(1.0)
    m.fill_momentum_halos()
    assert np.abs(m.divergence()).max() < 1e-13
    assert np.all(g.interior(m.rv)[:, 0, :] == 0.0)


def test_walls_stay_closed_and_scalars_are_conserved(oracle):
    g, m = _oracle(oracle)
    m.set(theta=theta0, u=u0, v=v0)
    s0 = g.interior(m.rtheta).sum()
    for _ in range(3):
        m.time_step(2.0)
    assert np.all(g.interior(m.rv)[:, 0, :] == 0.0) and np.all(g.interior(m.v)[:, 0, :] == 0.0)
    assert np.abs(m.divergence()).max() < 1e-13
    assert abs(g.interior(m.rtheta).sum() - s0) < 1e-13 * s0          # flux form, no flux through any wall
    assert np.abs(g.interior(m.w, True)).max() > 1e-2                   # the bubble next to the south wall moves


def test_order_drops_next_to_the_walls_as_in_z(oracle):
    """the y buffers are the z buffers: face j uses WENO5 for 3 <= j <= Ny - 3, WENO3 at j = 2 and Ny - 2, first-order upwind at
    j = 1 and Ny - 1 (og_buffer_at), so a field linear in y is reconstructed exactly wherever B >= 2"""
    L = oracle.lib()
    assert [L.og_buffer_at(j, 12, 1, 1) for j in range(1, 12)] == [1, 2, 3, 3, 3, 3, 3, 3, 3, 2, 1]
    assert [L.og_buffer_at(j, 12, 1, 0) for j in range(0, 12)] == [1, 2, 3, 3, 3, 3, 3, 3, 3, 3, 2, 1]


def test_options_outside_the_walled_scope_raise(bz):
    """closure, microphysics, tracers, other orders and formulations are not built for a Bounded y: the host says so before touching
    the device (the library's bz_set_* entry points return BZ_ERR_UNSUPPORTED for the same list)"""
    grid = bz.RectilinearGrid((16, 16, 8), topology=(bz.Periodic, bz.Bounded, bz.Bounded), **EXT)
    dyn = lambda: bz.AnelasticDynamics(bz.ReferenceState(grid, potential_temperature=300.0))
    for kw in (dict(closure=bz.SmagorinskyLilly()), dict(tracers=("a",)), dict(advection=bz.Centered(order=2)),
               dict(formulation="StaticEnergy"), dict(microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium()))):
        kw.setdefault("advection", bz.WENO(order=5))
        with pytest.raises(NotImplementedError):
            bz.AtmosphereModel(grid, dynamics=dyn(), **kw)
    with pytest.raises(NotImplementedError):
        bz.RectilinearGrid((16, 16, 8), topology=(bz.Bounded, bz.Periodic, bz.Bounded), **EXT) and \
            bz.AtmosphereModel(bz.RectilinearGrid((16, 16, 8), topology=(bz.Bounded, bz.Periodic, bz.Bounded), **EXT), advection=bz.WENO(order=5))


@pytest.mark.gpu
@pytest.mark.parametrize("order", [5, 7, 9])
def test_bounded_y_tendencies_match_oracle(oracle, bz, order):
    """Ny = 16, Nz = 12: every buffer of the cascade 9 -> 7 -> 5 -> 3 -> 1 occurs next to the y walls and next to the z walls"""
    from helpers import PROG, push_state, randomize
    g = oracle.Grid((32, 16, 12), topology=TOPO, halo=(5, 5, 5), **EXT)
    om = oracle.OracleModel(g, potential_temperature=300.0, advection=f"WENO{order}")
    grid = bz.RectilinearGrid((32, 16, 12), topology=(bz.Periodic, bz.Bounded, bz.Bounded), halo=(5, 5, 5), **EXT)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, potential_temperature=300.0)), advection=bz.WENO(order=order))
    randomize(om, seed=4)
    om.update_state(compute_tendencies=True)
    push_state(om, hm, names=("ru", "rv", "rw", "rtheta", "rq", "u", "v", "w", "theta", "q", "T"))
    for f in hm.G.values():
        f.parent.zero_()
    bz.compute_tendencies_(hm)
    hm.synchronize()
    for n, k in PROG.items():
        zf = n == "rw"
        want, got = g.interior(om.G[n], zface=zf), hm.G[k].interior_cpu()
        if zf:
            want, got = want[1:-1], got[1:-1]
        if n == "rv":      # the wall face j = 0 is never written
            want, got = want[:, 1:], got[:, 1:]
        assert relerr(got, want) < (1e-12 if order == 5 else 1e-11), (n, relerr(got, want))      # orders 7, 9: tests/test_weno_orders.py


@pytest.mark.gpu
def test_bounded_y_order_nine_steps_match_oracle(oracle, bz):
    g = oracle.Grid((32, 24, 12), topology=TOPO, halo=(5, 5, 5), **EXT)
    om = oracle.OracleModel(g, potential_temperature=300.0, advection="WENO9")
    grid = bz.RectilinearGrid((32, 24, 12), topology=(bz.Periodic, bz.Bounded, bz.Bounded), halo=(5, 5, 5), **EXT)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, potential_temperature=300.0)), advection=bz.WENO(order=9))
    om.set(theta=theta0, u=u0, v=v0)
    hm.set(θ=theta0, u=u0, v=v0)
    for _ in range(3):
        om.time_step(2.0)
        hm.time_step(2.0)
    hm.synchronize()
    pairs = {"ru": hm.momentum["ρu"], "rv": hm.momentum["ρv"], "rw": hm.momentum["ρw"], "rtheta": hm.potential_temperature_density}
    mom = max(np.abs(g.interior(getattr(om, n), n == "rw")).max() for n in ("ru", "rv", "rw"))
    for n, f in pairs.items():
        want, got = g.interior(getattr(om, n), n == "rw"), f.interior_cpu()
        scale = mom if n in ("ru", "rv", "rw") else np.abs(want).max()
        assert np.abs(got - want).max() < 2e-8 * scale, n          # the multi-step tolerance of tests/test_weno_orders.py
    assert float(hm.momentum["ρv"].interior[:, 0, :].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(32, 16, 12), (96, 40, 10), (16, 128, 8)])
def test_bounded_y_pressure_solve_matches_oracle(oracle, bz, size):
    """solve_for_anelastic_pressure! with the cosine transform along y (k_dct_line around the contiguous y plan; Nx = 96 takes the
    radix-3 x stage): phi against the oracle's scipy DCT + complex Thomas solve, then the projected momentum."""
    g, om = _oracle(oracle, size=size)
    grid = bz.RectilinearGrid(size, topology=(bz.Periodic, bz.Bounded, bz.Bounded), **EXT)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, potential_temperature=300.0)), advection=bz.WENO(order=5))
    rng = np.random.default_rng(7)
    g.interior(om.ru)[...] = rng.standard_normal(g.interior(om.ru).shape)
    g.interior(om.rv)[...] = rng.standard_normal(g.interior(om.rv).shape)
    g.interior(om.rw, True)[1:-1] = rng.standard_normal(g.interior(om.rw, True)[1:-1].shape)
    om.fill_momentum_halos()
    for n, k in (("ru", "ρu"), ("rv", "ρv"), ("rw", "ρw")):
        hm.momentum[k].parent.copy_(__import__("torch").from_numpy(getattr(om, n)))
    om.compute_pressure_correction(0.7)
    bz.compute_pressure_correction_(hm, 0.7)
    hm.synchronize()
    want = g.interior(om.phi)
    assert relerr(hm.dynamics.pressure_anomaly.interior_cpu(), want) < 1e-11
    om.make_pressure_correction(0.7)
    bz.make_pressure_correction_(hm, 0.7)
    for n, k in (("ru", "ρu"), ("rv", "ρv"), ("rw", "ρw")):
        assert relerr(hm.momentum[k].interior_cpu(), g.interior(getattr(om, n), n == "rw")) < 1e-11, n
    assert hm.max_abs_divergence() < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("size,stretched", [((32, 16, 12), False), ((64, 24, 16), True)])
def test_bounded_y_steps_match_oracle(oracle, bz, size, stretched):
    z = 1000.0 * np.linspace(0.0, 1.0, size[2] + 1) ** 1.3 if stretched else EXT["z"]
    ext = dict(EXT, z=z)
    g = oracle.Grid(size, topology=TOPO, **ext)
    om = oracle.OracleModel(g, potential_temperature=300.0)
    grid = bz.RectilinearGrid(size, topology=(bz.Periodic, bz.Bounded, bz.Bounded), **ext)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, potential_temperature=300.0)), advection=bz.WENO(order=5))
    om.set(theta=theta0, u=u0, v=v0)
    hm.set(θ=theta0, u=u0, v=v0)
    for _ in range(3):
        om.time_step(2.0)
        hm.time_step(2.0)
    hm.synchronize()
    pairs = {"ru": hm.momentum["ρu"], "rv": hm.momentum["ρv"], "rw": hm.momentum["ρw"], "rtheta": hm.potential_temperature_density}
    mom = max(np.abs(g.interior(getattr(om, n), n == "rw")).max() for n in ("ru", "rv", "rw"))
    for n, f in pairs.items():
        want, got = g.interior(getattr(om, n), n == "rw"), f.interior_cpu()
        scale = mom if n in ("ru", "rv", "rw") else np.abs(want).max()
        assert np.abs(got - want).max() < 1e-9 * scale, n
    assert float(hm.momentum["ρv"].interior[:, 0, :].abs().max()) == 0.0          # south wall face
    assert hm.max_abs_divergence() < 1e-11
