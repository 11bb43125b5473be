"""Lateral boundaries of the acoustic substep loop (`_zero_{x,y}_wall_face!`, `_relax_open_boundary_{x,y}!`,
/root/reference/src/CompressibleEquations/acoustic_substepping.jl:1300-1395) on grids with a Bounded x and / or y.

CPU part: the oracle's restatement against the properties the reference's own tests check (test/acoustic_substepping_open_boundaries.jl:
the keyword's validation; no-op without open sides; the outermost cell is pulled towards the prescribed wall value) and mass conservation
in a closed box.  GPU part: bz_acoustic_substep_loop through the C-ABI against the oracle on the same seeded inputs."""
import numpy as np
import pytest

TOPOLOGIES = [("Bounded", "Periodic", "Bounded"), ("Periodic", "Bounded", "Bounded"), ("Bounded", "Bounded", "Bounded")]
EXTENT = dict(x=(0.0, 12e3), y=(-4e3, 4e3), z=(0.0, 8e3))


def oracle_model(oracle, oc, topo, size=(20, 12, 16), **td):
    og = oracle.Grid(size, x=EXTENT["x"], y=EXTENT["y"], z=EXTENT["z"], topology=topo)
    return oc.CompressibleOracleModel(og, time_discretization=oc.SplitExplicit(**td), reference_potential_temperature=300.0)


def fill_model_halos(om, value_bc=None):
    """The model's own boundary conditions on the Bounded sides, one halo cell deep (all the loop reads): zero gradient for the centre
    fields (the default), or — value_bc = (rho_wall factor) on every Bounded side — ValueBoundaryCondition(v): c[halo] = 2 v - c[cell]."""
    g = om.grid
    Hx, Hy, Nx, Ny = g.Hx, g.Hy, g.Nx, g.Ny
    bx, by = g.topo[0] == 1, g.topo[1] == 1
    for name in ("rho_d", "rho", "rtheta", "rq", "p", "ru", "rv", "rw", "T", "theta", "q"):
        f = getattr(om, name)
        vb = value_bc is not None and name in ("rho_d", "rtheta")
        v = None
        if vb:
            col = om.ref.density if name == "rho_d" else om.ref.density * 300.0
            v = (value_bc * col[:f.shape[0]])[:, None]
        if bx and name != "ru":
            f[:, :, Hx - 1] = (2 * v - f[:, :, Hx]) if vb else f[:, :, Hx]
            f[:, :, Hx + Nx] = (2 * v - f[:, :, Hx + Nx - 1]) if vb else f[:, :, Hx + Nx - 1]
        if by and name != "rv":
            f[:, Hy - 1, :] = (2 * v - f[:, Hy, :]) if vb else f[:, Hy, :]
            f[:, Hy + Ny, :] = (2 * v - f[:, Hy + Ny - 1, :]) if vb else f[:, Hy + Ny - 1, :]


def seeded_wall_state(om, seed, open_sides=(), value_bc=None):
    """A seeded stage state U^L, an outer-step state U0 a little away from it, slow tendencies and the linearisation, with the model's
    halos on the Bounded sides filled by its boundary conditions; wall-normal momentum zero on impenetrable wall faces."""
    g = om.grid
    rng = np.random.default_rng(seed)
    x, y, z = g.nodes("ccc")
    sh = (g.Nz, g.Ny, g.Nx)
    Lx, Ly, Lz = g.Nx * g.dx, g.Ny * g.dy, g.zf[-1] - g.zf[0]

    def field(amp):
        smooth = np.sin(2 * np.pi * x / Lx + 0.3) * np.cos(2 * np.pi * y / Ly - 0.2) * np.sin(np.pi * (z - g.zf[0]) / Lz)
        return amp * (np.broadcast_to(smooth, sh) * 0.7 + 0.3 * rng.standard_normal(sh))

    I = g.interior
    rho_c = om.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
    I(om.rho_d)[...] = rho_c * (1 + field(0.01))
    I(om.rq)[...] = I(om.rho_d) * np.abs(field(4e-3))
    I(om.rtheta)[...] = I(om.rho_d) * (300.0 + 0.004 * z + field(3.0))
    I(om.ru)[...] = rho_c * field(4.0)
    I(om.rv)[...] = rho_c * field(4.0)
    I(om.rw, True)[1:-1] = (rho_c * field(2.0))[1:]
    bx, by = g.topo[0] == 1, g.topo[1] == 1
    if bx and "west" not in open_sides:
        I(om.ru)[:, :, 0] = 0.0
    if by and "south" not in open_sides:
        I(om.rv)[:, 0, :] = 0.0
    om.update_state(compute_tendencies=False)      # total density, theta, q, T, p of the interior (its halo fills cover the periodic directions)
    fill_model_halos(om, value_bc)
    for n in om.PROGNOSTIC:
        om.U0[n][...] = getattr(om, n)
    for n in ("rho_d", "rtheta", "ru", "rv"):       # the stage state away from U0 the way an earlier stage leaves it
        I(getattr(om, n))[...] *= 1 + 1e-3 * rng.standard_normal(sh)
    I(om.rw, True)[1:-1] *= 1 + 1e-3 * rng.standard_normal((g.Nz - 1, g.Ny, g.Nx))
    if bx and "west" not in open_sides:
        I(om.ru)[:, :, 0] = 0.0
    if by and "south" not in open_sides:
        I(om.rv)[:, 0, :] = 0.0
    om.update_state(compute_tendencies=False)
    fill_model_halos(om, value_bc)
    om.refresh_linearization()
    # slow tendencies: any smooth + rough field does (the loop only reads them); the wall-normal ones matter at open faces only
    for n, amp in (("rho_d", 1e-4), ("rtheta", 3e-2), ("ru", 2e-2), ("rv", 2e-2)):
        I(om.G[n])[...] = field(amp)
    I(om.G["rw"], True)[1:-1] = field(2e-2)[1:]
    om.lateral_open = {k: (k in open_sides) for k in ("west", "east", "south", "north")}


def total_mass(om):
    g = om.grid
    return float((g.interior(om.rho_d) * np.asarray(g.dzc[g.Hz:g.Hz + g.Nz])[:, None, None]).sum())


# ---- CPU: the oracle's restatement -----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("topo", TOPOLOGIES)
def test_closed_box_conserves_mass_and_keeps_wall_faces_at_zero(oracle, oc, topo):
    """enforce_wall_impenetrability!: with impenetrable walls on every Bounded side and no slow density tendency the loop moves mass
    only between cells (the predictor's divergence telescopes to the wall fluxes, which are zero): total dry mass is conserved to
    rounding, and the wall faces of the momentum perturbations are exact zeros after the loop."""
    om = oracle_model(oracle, oc, topo, substeps=6)
    seeded_wall_state(om, 11)
    om.G["rho_d"][...] = 0.0
    g = om.grid
    m0 = total_mass(om) + float((g.interior(om.U0["rho_d"] - om.rho_d) * np.asarray(g.dzc[g.Hz:g.Hz + g.Nz])[:, None, None]).sum())
    om.acoustic_substep_loop(2.0, 1.0)
    assert abs(total_mass(om) - m0) <= 1e-12 * abs(m0)
    if g.topo[0] == 1:
        assert not g.interior(om.rup)[:, :, 0].any() and not om.rup[g.Hz:g.Hz + g.Nz, g.Hy:g.Hy + g.Ny, g.Hx + g.Nx].any()
        assert np.abs(g.interior(om.rup)[:, :, 1]).max() > 0
    if g.topo[1] == 1:
        assert not g.interior(om.rvp)[:, 0, :].any() and not om.rvp[g.Hz:g.Hz + g.Nz, g.Hy + g.Ny, g.Hx:g.Hx + g.Nx].any()
        assert np.abs(g.interior(om.rvp)[:, 1, :]).max() > 0


def test_relaxation_is_a_no_op_without_open_sides(oracle, oc):
    """test/acoustic_substepping_open_boundaries.jl:128-157: no side carries an active open condition -> apply_open_boundary_relaxation!
    changes nothing, whatever the factor."""
    out = []
    for alpha in (0.5, 0.9):
        om = oracle_model(oracle, oc, TOPOLOGIES[2], substeps=4)
        seeded_wall_state(om, 12)
        om.open_boundary_relaxation = alpha
        om.acoustic_substep_loop(2.0, 0.5)
        out.append((om.rho_d.copy(), om.rtheta.copy()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_relaxation_pulls_the_outermost_cell_towards_the_wall_value(oracle, oc):
    """test/acoustic_substepping_open_boundaries.jl:159-260 at the level of one loop: the four sides open with ValueBoundaryCondition(1.05
    rho_ref) on rho and rho theta — after N_tau substeps the outermost cells of rho^L + rho' sit much closer to the wall value than the
    bulk does (the cumulative pull 1 - (1 - alpha)^N_tau), on every side, for both fields."""
    om = oracle_model(oracle, oc, TOPOLOGIES[2], substeps=8)
    seeded_wall_state(om, 13, open_sides=("west", "east", "south", "north"), value_bc=1.05)
    g = om.grid
    for n in om.PROGNOSTIC:          # a quiet state: the pull is the only large signal
        om.G[n][...] = 0.0
    om.acoustic_substep_loop(1.0, 1.0)
    I = g.interior
    wall = 1.05 * om.ref.density[g.Hz:g.Hz + g.Nz][:, None]
    for name, w in (("rho_d", wall), ("rtheta", wall * 300.0)):
        f = I(getattr(om, name))
        bulk = np.abs(f[:, g.Ny // 2, g.Nx // 2][:, None] - w).mean()
        for edge in (f[:, :, 0], f[:, :, -1], f[:, 0, :], f[:, -1, :]):
            assert np.abs(edge - w).mean() < 0.5 * bulk, name


def test_open_boundary_relaxation_keyword_is_validated():
    """test/acoustic_substepping_open_boundaries.jl:105-126: default 0.5, custom values kept, (0, 1] enforced."""
    import breeze_jl_amd
    c = breeze_jl_amd.compressible
    assert c.SplitExplicitTimeDiscretization().open_boundary_relaxation == 0.5
    assert c.SplitExplicitTimeDiscretization(open_boundary_relaxation=0.25).open_boundary_relaxation == 0.25
    for bad in (0, 1.5, -0.1):
        with pytest.raises(ValueError):
            c.SplitExplicitTimeDiscretization(open_boundary_relaxation=bad)
    assert c.is_active_open_bc(c.NormalFlowBoundaryCondition(6.0)) and not c.is_active_open_bc(c.NormalFlowBoundaryCondition())
    assert not c.is_active_open_bc(None)


# ---- GPU: the library against the oracle -------------------------------------------------------------------------------------------
WALL_CASES = [
    dict(substeps=6),
    dict(substeps=6, damping_coefficient=None, forward_weight=0.55),
    dict(substeps=4, damping_coefficient=0.05, damp_vertical=True),
    dict(substeps=1),
    dict(substeps=6, direct_damping=True),
    dict(substeps=5, sponge=(0.2, 3000.0, "cubic"), apply_first_substep_pressure_gradient=True),
]


def hip_model(bz, om, topo, open_sides=(), substep_floattype=None, alpha=0.5, float_type=None):
    otd = om.td
    g = om.grid
    gkw = {} if float_type is None else dict(float_type=float_type)
    grid = bz.RectilinearGrid((g.Nx, g.Ny, g.Nz), x=EXTENT["x"], y=EXTENT["y"], z=EXTENT["z"], topology=topo, **gkw)
    damping = (bz.NoDivergenceDamping() if otd.damping_coefficient is None
               else bz.DirectDivergenceDamping(coefficient=otd.damping_coefficient) if otd.direct_damping
               else bz.ThermalDivergenceDamping(coefficient=otd.damping_coefficient, damp_vertical=otd.damp_vertical,
                                                length_scale=otd.damping_length_scale))
    sponge = None
    if otd.sponge is not None:
        ramp = {"linear": bz.LinearRamp, "cubic": bz.CubicRamp, "sin2": bz.Sin2Ramp}[otd.sponge[2]]()
        sponge = bz.UpperSponge(damping_rate=otd.sponge[0], depth=otd.sponge[1], ramp=ramp)
    btd = bz.SplitExplicitTimeDiscretization(substeps=otd.substeps, acoustic_cfl=otd.acoustic_cfl, forward_weight=otd.forward_weight,
                                             damping=damping, sponge=sponge, apply_first_substep_pressure_gradient=otd.apply_first,
                                             open_boundary_relaxation=alpha)
    dyn = bz.CompressibleDynamics(btd, reference_potential_temperature=300.0, reference_state="auto")
    flow = bz.NormalFlowBoundaryCondition
    bcs = {"ρu": bz.FieldBoundaryConditions(west=flow(1.0) if "west" in open_sides else None, east=flow(1.0) if "east" in open_sides else None),
           "ρv": bz.FieldBoundaryConditions(south=flow(1.0) if "south" in open_sides else None, north=flow(1.0) if "north" in open_sides else None)}
    return bz.CompressibleAtmosphereModel(grid, dyn, advection=bz.WENO(order=5), substep_floattype=substep_floattype, boundary_conditions=bcs)


def run_pair(oracle, oc, bz, topo, td, beta, open_sides=(), value_bc=None, seed=21, size=(20, 12, 16), substep_floattype=None, alpha=0.5):
    from tests.test_gpu_compressible import push
    om = oracle_model(oracle, oc, topo, size=size, **td)
    seeded_wall_state(om, seed, open_sides=open_sides, value_bc=value_bc)
    om.open_boundary_relaxation = alpha
    hm = hip_model(bz, om, topo, open_sides, substep_floattype, alpha)
    push(om, hm, substepper=substep_floattype is None)
    bz.compressible.refresh_linearization_(hm)
    dt = 2.0
    om.acoustic_substep_loop(dt, beta)
    bz.compressible.acoustic_rk3_substep_loop_(hm, dt, beta)
    return om, hm


def compare(om, hm, tol=2e-11, state_tol=1e-12):
    from tests.test_gpu_compressible import SUB, cmp_interior
    g = om.grid
    sub = hm.timestepper.substepper
    for n, k in SUB.items():
        if n in ("Pi", "thL", "gR"):
            continue
        zf = n in ("rwp", "aw", "Gs")
        a, b = getattr(sub, k).interior_cpu(), g.interior(getattr(om, n), zf)
        scale = {"rp": np.abs(g.interior(om.rho_d)).max() * 1e-3, "rthp": np.abs(g.interior(om.rtheta)).max() * 1e-3}.get(n)
        err = np.abs(a - b).max() / (scale or max(np.abs(b).max(), 1e-300))
        assert err <= tol, (n, err)
    cmp_interior(om, hm, ("rho_d", "rtheta", "ru", "rv", "rw"), state_tol)


@pytest.mark.gpu
@pytest.mark.parametrize("topo", TOPOLOGIES)
@pytest.mark.parametrize("td", WALL_CASES)
def test_wall_loop_matches_oracle(oracle, oc, bz, topo, td):
    """bz_acoustic_substep_loop on a closed box (impenetrable walls on every Bounded side): perturbations, time-averaged velocities and
    the recovered state against the oracle; the wall faces of the stored momentum perturbations are exact zeros."""
    om, hm = run_pair(oracle, oc, bz, topo, td, 0.5)
    compare(om, hm)
    g = om.grid
    sub = hm.timestepper.substepper
    if g.topo[0] == 1:
        assert not sub.momentum_perturbation_u.interior_cpu()[:, :, 0].any()
    if g.topo[1] == 1:
        assert not sub.momentum_perturbation_v.interior_cpu()[:, 0, :].any()


@pytest.mark.gpu
@pytest.mark.parametrize("topo,open_sides", [(TOPOLOGIES[0], ("west", "east")), (TOPOLOGIES[1], ("south",)), (TOPOLOGIES[1], ("north",)),
                                             (TOPOLOGIES[2], ("west", "east", "south", "north")), (TOPOLOGIES[2], ("east", "south"))])
@pytest.mark.parametrize("td", [dict(substeps=6), dict(substeps=6, direct_damping=True), dict(substeps=4, damping_coefficient=None)])
@pytest.mark.parametrize("beta", [1.0 / 3.0, 1.0])
def test_open_boundary_loop_matches_oracle(oracle, oc, bz, topo, open_sides, td, beta):
    """Open sides (NormalFlowBoundaryCondition on the wall-normal momentum, ValueBoundaryCondition halos on rho and rho theta):
    _relax_open_boundary_{x,y}! every substep, west / south faces advanced by the substep, the other sides impenetrable."""
    om, hm = run_pair(oracle, oc, bz, topo, td, beta, open_sides=open_sides, value_bc=1.03, alpha=0.4)
    compare(om, hm)


@pytest.mark.gpu
def test_wall_loop_in_float32_storage_and_unfused_kernels(oracle, oc, bz, monkeypatch):
    """substep_floattype = Float32 on walls (the fills and the relaxation in the storage type): 2e-5 of the oracle; BZ_NO_AC_FUSE=1 (the
    per-kernel sequence of the reference) at the tolerance of the fused one."""
    om, hm = run_pair(oracle, oc, bz, TOPOLOGIES[2], dict(substeps=6), 0.5, open_sides=("west", "north"), value_bc=1.03, substep_floattype=np.float32)
    from tests.test_gpu_compressible import cmp_interior
    cmp_interior(om, hm, ("rho_d", "rtheta", "ru", "rv", "rw"), 2e-5)
    monkeypatch.setenv("BZ_NO_AC_FUSE", "1")
    om, hm = run_pair(oracle, oc, bz, TOPOLOGIES[2], dict(substeps=6), 0.5, open_sides=("west", "north"), value_bc=1.03)
    compare(om, hm)


@pytest.mark.gpu
def test_wall_contexts_run_the_acoustic_loop_only(oracle, oc, bz):
    """Everything else of the compressible model on lateral walls fails loudly (BZ_ERR_UNSUPPORTED), in the library and in the host mirror."""
    om = oracle_model(oracle, oc, TOPOLOGIES[0])
    hm = hip_model(bz, om, TOPOLOGIES[0])
    with pytest.raises(NotImplementedError):
        hm.time_step(1.0)
    with pytest.raises(NotImplementedError):
        hm.set(θ=300.0)
    for fn in (bz.compressible.update_state_, bz.compressible.compute_slow_tendencies_):
        with pytest.raises(Exception, match="Bounded x or y"):
            fn(hm)


@pytest.mark.gpu
def test_lateral_boundary_setter_validates_its_arguments(oracle, oc, bz):
    """bz_set_acoustic_lateral_boundaries: the factor must lie in (0, 1] (time_discretizations.jl:573-574) and an open side needs a Bounded
    topology in its direction; a periodic context refuses any open side."""
    om = oracle_model(oracle, oc, TOPOLOGIES[1])
    hm = hip_model(bz, om, TOPOLOGIES[1])
    f = hm._lib.bz_set_acoustic_lateral_boundaries
    assert f(hm._ctx, 0, 0, 1, 1, 0.5) == 0
    for alpha in (0.0, -0.1, 1.5):
        assert f(hm._ctx, 0, 0, 1, 0, alpha) != 0
    assert f(hm._ctx, 1, 0, 0, 0, 0.5) != 0          # west of a Periodic x
    from tests.test_gpu_compressible import make_pair
    _, pm = make_pair(oracle, oc, bz, size=(16, 8, 8))
    assert pm._lib.bz_set_acoustic_lateral_boundaries(pm._ctx, 0, 0, 0, 1, 0.5) != 0
    assert pm._lib.bz_set_acoustic_lateral_boundaries(pm._ctx, 0, 0, 0, 0, 0.5) == 0


@pytest.mark.gpu
def test_wall_loop_on_a_float32_grid(oracle, oc, bz):
    """eltype(grid) = Float32 (libbreeze_hip_f32.so, the generated twin): the loop on (Bounded, Bounded, Bounded) with two open sides against the
    Float64 oracle — perturbation fields relative to their own scale (what Float32 resolves worst), the recovered state at Float32 round-off."""
    import torch
    from tests.test_gpu_compressible import O2H, PROG, SUB
    topo, open_sides = TOPOLOGIES[2], ("west", "north")
    om = oracle_model(oracle, oc, topo, substeps=6)
    seeded_wall_state(om, 31, open_sides=open_sides, value_bc=1.03)
    om.open_boundary_relaxation = 0.4
    hm = hip_model(bz, om, topo, open_sides, alpha=0.4, float_type=np.float32)
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))      # noqa: E731
    for n, f in O2H.items():
        f(hm).parent.copy_(f32(getattr(om, n)))
    for n, k in PROG.items():
        hm.G[k].parent.copy_(f32(om.G[n]))
        hm.U0[k].parent.copy_(f32(om.U0[n]))
    for n, k in SUB.items():
        getattr(hm.timestepper.substepper, k).parent.copy_(f32(getattr(om, n)))
    bz.compressible.refresh_linearization_(hm)
    om.acoustic_substep_loop(2.0, 0.5)
    bz.compressible.acoustic_rk3_substep_loop_(hm, 2.0, 0.5)
    g = om.grid
    sub = hm.timestepper.substepper
    for n, k, tol in (("rup", "momentum_perturbation_u", 2e-3), ("rvp", "momentum_perturbation_v", 2e-3), ("rwp", "momentum_perturbation_w", 2e-3)):
        a, b = getattr(sub, k).interior_cpu().astype(np.float64), g.interior(getattr(om, n), n == "rwp")
        assert np.abs(a - b).max() <= tol * np.abs(b).max(), (n, np.abs(a - b).max() / np.abs(b).max())
    for n in ("rho_d", "rtheta", "ru", "rv", "rw"):
        a, b = O2H[n](hm).interior_cpu().astype(np.float64), g.interior(getattr(om, n), n == "rw")
        scale = max(np.abs(g.interior(getattr(om, c), c == "rw")).max() for c in ("ru", "rv", "rw")) if n in ("ru", "rv", "rw") else np.abs(b).max()
        # (the momentum carries the Float32 rounding of six substeps of pressure-gradient increments; the densities one recovery)
        assert np.abs(a - b).max() <= (1e-5 if n in ("ru", "rv", "rw") else 2e-6) * scale, (n, np.abs(a - b).max() / scale)
    assert not sub.momentum_perturbation_v.interior_cpu()[:, 0, :].any()      # the impenetrable south wall
