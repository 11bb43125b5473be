"""GPU parity of the compressible split-explicit path: every entry point of the C ABI
against the CPU oracle on the same seeded inputs.  Float64; tolerances are relative to the field's max-abs and
written next to each comparison."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oc(oracle):
    from oracle import oracle_compressible
    return oracle_compressible


EXTENT = dict(x=(-4e3, 4e3), y=(-3e3, 3e3), z=(0.0, 8e3))


def make_pair(oracle, oc, bz, size=(24, 16, 20), theta_ref=300.0, reference=True, z_faces=None, substep_floattype=None, **td):
    z = z_faces if z_faces is not None else EXTENT["z"]
    og = oracle.Grid(size, x=EXTENT["x"], y=EXTENT["y"], z=z)
    otd = oc.SplitExplicit(**td)
    om = oc.CompressibleOracleModel(og, time_discretization=otd, reference_potential_temperature=theta_ref,
                                    reference_state=reference)
    grid = bz.RectilinearGrid(size, x=EXTENT["x"], y=EXTENT["y"], z=z)
    damping = (bz.NoDivergenceDamping() if otd.damping_coefficient is None
               else bz.DirectDivergenceDamping(coefficient=otd.damping_coefficient) if otd.direct_damping
               else bz.ThermalDivergenceDamping(coefficient=otd.damping_coefficient, damp_vertical=otd.damp_vertical,
                                                length_scale=otd.damping_length_scale))
    sponge = None
    if otd.sponge is not None:
        ramp = {"linear": bz.LinearRamp, "cubic": bz.CubicRamp, "sin2": bz.Sin2Ramp}[otd.sponge[2]]()
        sponge = bz.UpperSponge(damping_rate=otd.sponge[0], depth=otd.sponge[1], ramp=ramp)
    btd = bz.SplitExplicitTimeDiscretization(substeps=otd.substeps, acoustic_cfl=otd.acoustic_cfl,
                                             forward_weight=otd.forward_weight, damping=damping, sponge=sponge,
                                             apply_first_substep_pressure_gradient=otd.apply_first,
                                             substep_distribution={"proportional": bz.ProportionalSubsteps, "constant": bz.ConstantSubstepSize,
                                                                   "monolithic_first_stage": bz.MonolithicFirstStage}[otd.substep_distribution]())
    dyn = bz.CompressibleDynamics(btd, reference_potential_temperature=theta_ref if reference else None,
                                  reference_state="auto" if reference else None)
    hm = bz.CompressibleAtmosphereModel(grid, dyn, advection=bz.WENO(order=5), substep_floattype=substep_floattype)
    return om, hm


def seeded_state(om, seed, amp_u=4.0, amp_theta=3.0, amp_q=4e-3, amp_rho=0.01):
    """Smooth + rough seeded perturbation of a hydrostatic column; consistent halos / diagnostics via update_state."""
    g = om.grid
    rng = np.random.default_rng(seed)
    x, y, z = g.nodes("ccc")
    Lx, Ly, Lz = g.Nx * g.dx, g.Ny * g.dy, g.zf[-1] - g.zf[0]
    sh = (g.Nz, g.Ny, g.Nx)

    def field(amp):
        smooth = np.sin(2 * np.pi * x / Lx + 0.3) * np.cos(2 * np.pi * y / Ly - 0.2) * np.sin(np.pi * (z - g.zf[0]) / Lz)
        return amp * (np.broadcast_to(smooth, sh) * 0.7 + 0.3 * rng.standard_normal(sh))

    if om.ref is not None:
        rho_c = om.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
    else:
        rho_c = 1.2 * np.exp(-g.zc / 8e3)[:, None, None]
    I = g.interior
    I(om.rho_d)[...] = rho_c * (1 + field(amp_rho))
    I(om.rq)[...] = I(om.rho_d) * np.abs(field(amp_q))
    I(om.rtheta)[...] = I(om.rho_d) * (300.0 + 0.004 * z + field(amp_theta))
    I(om.ru)[...] = rho_c * field(amp_u)
    I(om.rv)[...] = rho_c * field(amp_u)
    I(om.rw, True)[1:-1] = (rho_c * field(0.5 * amp_u))[1:]
    om.seed_time_averaged_velocities()
    om.update_state(compute_tendencies=False)
    om.seed_time_averaged_velocities()
    om.update_state(compute_tendencies=True)


O2H = {
    "rho_d": lambda m: m.dynamics.dry_density, "rho": lambda m: m.dynamics.total_density, "p": lambda m: m.dynamics.pressure,
    "ru": lambda m: m.momentum["ρu"], "rv": lambda m: m.momentum["ρv"], "rw": lambda m: m.momentum["ρw"],
    "rtheta": lambda m: m.potential_temperature_density, "rq": lambda m: m.moisture_density,
    "u": lambda m: m.velocities["u"], "v": lambda m: m.velocities["v"], "w": lambda m: m.velocities["w"],
    "theta": lambda m: m.potential_temperature, "q": lambda m: m.specific_moisture, "T": lambda m: m.temperature,
}
SUB = {"Pi": "exner", "thL": "potential_temperature", "gR": "gamma_R_mixture", "rp": "density_perturbation",
       "rthp": "density_potential_temperature_perturbation", "rup": "momentum_perturbation_u",
       "rvp": "momentum_perturbation_v", "rwp": "momentum_perturbation_w", "au": "time_averaged_u",
       "av": "time_averaged_v", "aw": "time_averaged_w", "Gs": "slow_vertical_momentum_tendency"}
PROG = {"rho_d": "ρᵈ", "ru": "ρu", "rv": "ρv", "rw": "ρw", "rtheta": "ρθ", "rq": "ρq"}


def push(om, hm, substepper=True):
    import torch
    for n, f in O2H.items():
        f(hm).parent.copy_(torch.from_numpy(getattr(om, n)))
    for n, k in PROG.items():
        hm.G[k].parent.copy_(torch.from_numpy(om.G[n]))
        hm.U0[k].parent.copy_(torch.from_numpy(om.U0[n]))
    if substepper:
        for n, k in SUB.items():
            getattr(hm.timestepper.substepper, k).parent.copy_(torch.from_numpy(getattr(om, n)))


def rel(a, b):
    scale = max(np.abs(b).max(), 1e-300)
    return np.abs(a - b).max() / scale


VECTOR_GROUPS = (("ru", "rv", "rw"), ("u", "v", "w"))


def cmp_interior(om, hm, names, tol, zface_names=("rw", "w")):
    """max-abs error relative to the field's max-abs; components of a vector share the vector's scale."""
    g = om.grid
    worst = {}
    for n in names:
        zf = n in zface_names
        a = O2H[n](hm).interior_cpu()
        b = g.interior(getattr(om, n), zf)
        scale = np.abs(b).max()
        for grp in VECTOR_GROUPS:
            if n in grp:
                scale = max(np.abs(g.interior(getattr(om, c), c in zface_names)).max() for c in grp)
        worst[n] = np.abs(a - b).max() / max(scale, 1e-300)
    bad = {k: v for k, v in worst.items() if not v <= tol}
    assert not bad, f"mismatch beyond {tol}: {bad} (all: {worst})"
    return worst


def test_update_state_matches_oracle(oracle, oc, bz):
    """update_state!: total density, velocities, theta, q, T (Newton), p and every halo the tendencies read."""
    om, hm = make_pair(oracle, oc, bz)
    seeded_state(om, 1)
    push(om, hm)
    for n in ("rho", "u", "v", "w", "theta", "q", "T", "p"):      # outputs must come from the kernel
        O2H[n](hm).parent.zero_()
    bz.compressible.update_state_(hm, compute_tendencies=False)
    cmp_interior(om, hm, ("rho", "u", "v", "w", "theta", "q", "T", "p"), 1e-14)
    # halos: x/y periodic images on interior levels, first z-halo cell of no-flux fields, walls of w
    g = om.grid
    Hz, Nz = g.Hz, g.Nz
    for n in ("rho_d", "rho", "ru", "rv", "rtheta", "rq", "u", "v", "theta", "q", "T", "p"):
        a, b = O2H[n](hm).cpu(), getattr(om, n)
        assert rel(a[Hz - 1:Hz + Nz + 1], b[Hz - 1:Hz + Nz + 1]) <= 1e-14, n
    for n in ("rw", "w"):
        a, b = O2H[n](hm).cpu(), getattr(om, n)
        assert rel(a[Hz:Hz + Nz + 1], b[Hz:Hz + Nz + 1]) <= 1e-14, n


def test_moisture_tendency_and_linearization(oracle, oc, bz):
    om, hm = make_pair(oracle, oc, bz)
    seeded_state(om, 2)
    om.refresh_linearization()
    push(om, hm)
    hm.G["ρq"].parent.zero_()
    for k in ("exner", "potential_temperature", "gamma_R_mixture"):
        getattr(hm.timestepper.substepper, k).parent.zero_()
    bz.compressible.update_state_(hm, compute_tendencies=True)
    bz.compressible.refresh_linearization_(hm)
    g = om.grid
    assert rel(hm.G["ρq"].interior_cpu(), g.interior(om.G["rq"])) <= 1e-12
    sub = hm.timestepper.substepper
    assert rel(sub.exner.interior_cpu(), g.interior(om.Pi)) <= 1e-15
    assert rel(sub.potential_temperature.interior_cpu(), g.interior(om.thL)) <= 1e-15
    assert rel(sub.gamma_R_mixture.interior_cpu(), g.interior(om.gR)) <= 1e-15


@pytest.mark.parametrize("stretched", [False, True])
def test_slow_tendencies_match_oracle(oracle, oc, bz, stretched):
    """compute_slow_momentum_tendencies! + compute_slow_scalar_tendencies!: WENO-5 advection only; 1e-12 of max-abs."""
    zf = None
    if stretched:
        s = np.linspace(0, 1, 21)
        zf = 8e3 * (0.6 * s + 0.4 * s ** 2)
    om, hm = make_pair(oracle, oc, bz, z_faces=zf)
    seeded_state(om, 3)
    om.compute_slow_tendencies()
    push(om, hm)
    for k in hm.G:
        if k != "ρq":
            hm.G[k].parent.zero_()
    bz.compressible.compute_slow_tendencies_(hm)
    g = om.grid
    for n, k in PROG.items():
        if n == "rq":
            continue
        a = hm.G[k].interior_cpu()
        b = g.interior(om.G[n], n == "rw")
        if n == "rw":
            a, b = a[1:-1], b[1:-1]
        assert rel(a, b) <= 1e-12, (n, rel(a, b))


@pytest.mark.parametrize("size", [(64, 8, 20), (128, 8, 70)])
def test_slow_tendencies_on_rows_of_64_cells_take_the_exchange_kernel(oracle, oc, bz, size, monkeypatch):
    """Rows of a multiple of 64 cells with Ny a multiple of 4: the rho theta (+ rho_d) and moisture tendencies run
    k_scalar_tendency_rho3d_x (csrc/bz_compressible.hip: every face flux evaluated once, exchanged by lane shuffle / LDS rows in groups of
    four levels, short level chunks on small grids) — against the oracle at the tolerance of the test above, and against the kernel it
    replaces (BZ_NO_RHO3D_EXCHANGE=1; an ulp of a flux apart).  128 x 8 x 70: two tiles in x, two tile rows (outside rows across the
    periodic boundary), level chunks of 8 with partial groups."""
    zf = 8e3 * (0.6 * np.linspace(0, 1, size[2] + 1) + 0.4 * np.linspace(0, 1, size[2] + 1) ** 2)
    got = {}
    for exchange in (True, False):
        if exchange:
            monkeypatch.delenv("BZ_NO_RHO3D_EXCHANGE", raising=False)
        else:
            monkeypatch.setenv("BZ_NO_RHO3D_EXCHANGE", "1")
        om, hm = make_pair(oracle, oc, bz, size=size, z_faces=zf)
        seeded_state(om, 5)
        om.update_state(compute_tendencies=True)          # the moisture tendency (total-density carrier, time-averaged velocities)
        om.compute_slow_tendencies()
        push(om, hm)
        for k in hm.G:
            hm.G[k].parent.zero_()
        bz.compressible.update_state_(hm, compute_tendencies=True)          # the same kernel with the averaged velocities in the velocity slots
        bz.compressible.compute_slow_tendencies_(hm)
        got[exchange] = {k: hm.G[k].interior_cpu().copy() for k in hm.G}
    g = om.grid
    for n, k in PROG.items():
        b = g.interior(om.G[n], n == "rw")
        for exchange in (True, False):
            a = got[exchange][k]
            if n == "rw":
                assert rel(a[1:-1], b[1:-1]) <= 1e-12, (n, exchange)
            else:
                assert rel(a, b) <= 1e-12, (n, exchange, rel(a, b))
        assert rel(got[True][k], got[False][k]) <= 1e-13, n
    assert np.abs(got[True]["ρθ"]).max() > 0 and np.abs(got[True]["ρq"]).max() > 0


CASES = [
    dict(substeps=6),                                            # default damping, N_tau = 2, 3, 6
    dict(substeps=6, damping_coefficient=None, forward_weight=0.55),
    dict(substeps=4, damping_coefficient=0.05, damp_vertical=True),
    dict(substeps=1),                                            # degenerate one-substep stages (gate always on)
    dict(substeps=6, apply_first_substep_pressure_gradient=True),
    dict(),                                                      # adaptive substep count from the acoustic CFL
    dict(substeps=6, direct_damping=True),                       # DirectDivergenceDamping (acoustic_substepping.jl:1146-1188)
    dict(substeps=6, damping_length_scale=180.0, damp_vertical=True),   # ThermalDivergenceDamping(length_scale = l): fixed alpha l^2 / dtau (:1085-1092)
    dict(substeps=1, direct_damping=True, damping_coefficient=0.15),
    dict(substeps=6, sponge=(0.2, 3000.0, "cubic")),             # UpperSponge (acoustic_substepping.jl:584-602), each ramp shape
    dict(substeps=4, sponge=(0.5, 2000.0, "linear"), damp_vertical=True),
    dict(sponge=(0.1, 4000.0, "sin2")),
]


@pytest.mark.parametrize("td", CASES)
@pytest.mark.parametrize("beta", [1.0 / 3.0, 0.5, 1.0])
def test_acoustic_substep_loop_matches_oracle(oracle, oc, bz, td, beta):
    """acoustic_rk3_substep_loop!(model, substepper, dt, beta, U0): perturbations, time-averaged velocities, recovered
    state and velocities after the loop, from a stage state that differs from U0 (non-zero rewind)."""
    reference = td.get("substeps", 0) != 1
    om, hm = make_pair(oracle, oc, bz, reference=reference, **td)
    seeded_state(om, 4)
    for n in om.PROGNOSTIC:
        om.U0[n][...] = getattr(om, n)
    # move the stage state away from U0 the way an earlier stage would
    g = om.grid
    rng = np.random.default_rng(5)
    for n in ("rho_d", "rtheta", "ru", "rv"):
        g.interior(getattr(om, n))[...] *= 1 + 1e-3 * rng.standard_normal((g.Nz, g.Ny, g.Nx))
    g.interior(om.rw, True)[1:-1] *= 1 + 1e-3 * rng.standard_normal((g.Nz - 1, g.Ny, g.Nx))
    om.update_state(compute_tendencies=True)
    om.refresh_linearization()
    om.compute_slow_tendencies()
    push(om, hm)
    bz.compressible.refresh_linearization_(hm)     # fills the context's gamma R Pi scratch
    dt = 2.0
    om.acoustic_substep_loop(dt, beta)
    bz.compressible.acoustic_rk3_substep_loop_(hm, dt, beta)
    n_tau, _ = hm.stage_substeps(dt, beta)
    assert n_tau == om.last_substeps[-1]
    sub = hm.timestepper.substepper
    tol = 2e-11
    for n, k in SUB.items():
        if n in ("Pi", "thL", "gR"):
            continue
        zf = n in ("rwp", "aw", "Gs")
        a, b = getattr(sub, k).interior_cpu(), g.interior(getattr(om, n), zf)
        scale = {"rp": np.abs(g.interior(om.rho_d)).max() * 1e-3, "rthp": np.abs(g.interior(om.rtheta)).max() * 1e-3}.get(n)
        err = np.abs(a - b).max() / (scale or max(np.abs(b).max(), 1e-300))
        assert err <= tol, (n, err, n_tau)
    cmp_interior(om, hm, ("rho_d", "rtheta", "ru", "rv", "rw", "u", "v", "w"), 1e-12)


@pytest.mark.parametrize("td", [dict(substeps=6), dict(), dict(substeps=4, damping_coefficient=0.05, damp_vertical=True),
                                dict(substeps=6, direct_damping=True), dict(substeps=6, sponge=(0.2, 3000.0, "cubic"))])
def test_time_steps_match_oracle(oracle, oc, bz, td):
    """Three full WS-RK3 steps of a warm bubble with a moist tracer, whole-step seam: 1e-9 of max-abs."""
    om, hm = make_pair(oracle, oc, bz, size=(24, 16, 24), **td)
    g = om.grid

    def theta(x, y, z):
        r = np.sqrt(x ** 2 + y ** 2 + (z - 3000.0) ** 2)
        return 300.0 + 2.0 * np.maximum(0.0, 1.0 - r / 2000.0)

    def qv(x, y, z):
        return 5e-3 * np.exp(-z / 2e3) * (1 + 0.2 * np.sin(2 * np.pi * x / 8e3)) + 0 * y

    rho = om.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
    om.set(rho=rho, theta=theta, u=lambda x, y, z: 3.0 + 0 * x + 0 * y + 0 * z, v=0.0, w=0.0, qv=qv)
    hm.set(ρ=rho, θ=theta, u=lambda x, y, z: 3.0 + 0 * x + 0 * y + 0 * z, v=0.0, w=0.0, qᵗ=qv)
    cmp_interior(om, hm, ("rho_d", "rho", "rtheta", "rq", "ru", "T", "p"), 1e-14)
    for _ in range(3):
        om.time_step(2.0)
        hm.time_step(2.0)
    # rounding-level differences (hipcc contracts a*b+c into FMAs, device pow vs libm; the refdiv library, which keeps
    # the oracle's WENO operation order, shows the same figure) grow through 3 steps x 3 stages x up to 18 acoustic
    # substeps to ~1e-9 of the velocity scale
    worst = cmp_interior(om, hm, ("rho_d", "rtheta", "rq", "ru", "rv", "rw", "u", "v", "w", "theta", "q", "T", "p"), 5e-9)
    print("3-step parity:", {k: f"{v:.1e}" for k, v in worst.items()})
    sub = hm.timestepper.substepper
    scale = max(np.abs(g.interior(getattr(om, n), n == "aw")).max() for n in ("au", "av", "aw"))
    for n, k in (("au", "time_averaged_u"), ("av", "time_averaged_v"), ("aw", "time_averaged_w")):
        err = np.abs(getattr(sub, k).interior_cpu() - g.interior(getattr(om, n), n == "aw")).max() / scale
        assert err <= 5e-9, (n, err)


def test_time_steps_on_rows_of_64_cells_match_oracle(oracle, oc, bz):
    """the three-step case above on 64 x 8 x 24: the scalar tendencies of every stage run the exchange kernel (k_scalar_tendency_rho3d_x);
    dt = 0.5 s: the 64-cell rows are 125 m wide, a quarter of the cells above"""
    om, hm = make_pair(oracle, oc, bz, size=(64, 8, 24), substeps=6)
    g = om.grid

    def theta(x, y, z):
        r = np.sqrt(x ** 2 + y ** 2 + (z - 3000.0) ** 2)
        return 300.0 + 2.0 * np.maximum(0.0, 1.0 - r / 2000.0)

    def qv(x, y, z):
        return 5e-3 * np.exp(-z / 2e3) * (1 + 0.2 * np.sin(2 * np.pi * x / 8e3)) + 0 * y

    rho = om.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
    om.set(rho=rho, theta=theta, u=lambda x, y, z: 3.0 + 0 * x + 0 * y + 0 * z, v=0.0, w=0.0, qv=qv)
    hm.set(ρ=rho, θ=theta, u=lambda x, y, z: 3.0 + 0 * x + 0 * y + 0 * z, v=0.0, w=0.0, qᵗ=qv)
    for _ in range(3):
        om.time_step(0.5)
        hm.time_step(0.5)
    cmp_interior(om, hm, ("rho_d", "rtheta", "rq", "ru", "rv", "rw", "u", "v", "w", "theta", "q", "T", "p"), 5e-9)


@pytest.mark.parametrize("size", [(128, 32, 12), (64, 64, 10)])
def test_forward_sweep_with_xcd_bands_matches_oracle(oracle, oc, bz, size, monkeypatch):
    """Round 5: where the forward sweep's grid has a multiple of 8 tile rows (Ny a multiple of 32) every XCD owns a band of tile rows and walks
    it x fastest (csrc/bz_compressible.hip: k_ac_column_forward, AcParams::xcd; in launch order an XCD owns a tile COLUMN and every x
    neighbour sits behind another L2).  Two tile columns x one row per band, and one column x two rows per band: three steps against the
    oracle and bit for bit against the launch-order run (BZ_AC_XCD=0)."""
    def run(xcd):
        monkeypatch.setenv("BZ_AC_XCD", "1" if xcd else "0")
        om, hm = make_pair(oracle, oc, bz, size=size, substeps=6)
        g = om.grid

        def theta(x, y, z):
            r = np.sqrt(x ** 2 + (y - 300.0) ** 2 + (z - 3000.0) ** 2)
            return 300.0 + 2.0 * np.maximum(0.0, 1.0 - r / 2000.0)

        rho = om.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
        u = lambda x, y, z: 3.0 + 0 * x + 0 * y + 0 * z      # noqa: E731
        v = lambda x, y, z: -2.0 + 0 * x + 0 * y + 0 * z     # noqa: E731
        om.set(rho=rho, theta=theta, u=u, v=v, w=0.0)
        hm.set(ρ=rho, θ=theta, u=u, v=v, w=0.0)
        for _ in range(3):
            hm.time_step(0.5)
        hm.synchronize()
        return om, hm

    om, a = run(True)
    for _ in range(3):
        om.time_step(0.5)
    cmp_interior(om, a, ("rho_d", "rtheta", "ru", "rv", "rw", "u", "v", "w", "theta", "T", "p"), 5e-9)
    _, b = run(False)
    fa, fb = a.prognostic_fields(), b.prognostic_fields()
    for k in fa:
        assert np.array_equal(fa[k].interior_cpu(), fb[k].interior_cpu()), k



@pytest.mark.parametrize("size", [(128, 32, 12), (64, 64, 10), (40, 12, 9)])
@pytest.mark.parametrize("td", [dict(substeps=6), dict(substeps=4, damping_coefficient=0.05, damp_vertical=True), dict(substeps=6, damping_coefficient=None, sponge=(0.2, 3000.0, "cubic"))])
def test_forward_sweep_round6_kernel(oracle, oc, bz, size, td, monkeypatch):
    """Round 6: k_ac_forward2 (every load of a level before its first store, 32-bit offsets from uniform bases, x neighbours from the
    neighbouring lanes) carries the arithmetic text of k_ac_column_forward: without the fold of the p^L gradient it reproduces the round-5
    kernel to 1e-13 (BZ_AC_FWD2=0) and itself bit for bit in every block shape / barrier / pipelining / lane-shift variant (BZ_AC_CFG, BZ_AC_BX); with the fold (Gp_ru = G_ru - dx p^L once per stage, the shipped default)
    three steps stay within 5e-9 of the oracle and 1e-12 of the unfolded run.  Rows of two tiles, one tile and a ragged 40-cell row."""
    def run(fwd2, pfold, cfg=29, bx=128):
        monkeypatch.setenv("BZ_AC_FWD2", str(fwd2))
        monkeypatch.setenv("BZ_AC_PFOLD", str(pfold))
        monkeypatch.setenv("BZ_AC_CFG", str(cfg))
        monkeypatch.setenv("BZ_AC_BX", str(bx))
        om, hm = make_pair(oracle, oc, bz, size=size, **td)
        g = om.grid

        def theta(x, y, z):
            r = np.sqrt(x ** 2 + (y - 300.0) ** 2 + (z - 3000.0) ** 2)
            return 300.0 + 2.0 * np.maximum(0.0, 1.0 - r / 2000.0)

        rho = om.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
        u = lambda x, y, z: 3.0 + 0 * x + 0 * y + 0 * z      # noqa: E731
        v = lambda x, y, z: -2.0 + 0 * x + 0 * y + 0 * z     # noqa: E731
        om.set(rho=rho, theta=theta, u=u, v=v, w=0.0)
        hm.set(ρ=rho, θ=theta, u=u, v=v, w=0.0)
        for _ in range(3):
            hm.time_step(0.5)
        hm.synchronize()
        return om, hm

    om, a = run(1, 2)      # 2: the fold in every stage (the default folds stages of >= 5 substeps only)
    for _ in range(3):
        om.time_step(0.5)
    cmp_interior(om, a, ("rho_d", "rtheta", "ru", "rv", "rw", "u", "v", "w", "theta", "T", "p"), 5e-9)
    _, old = run(0, 0)
    fo = {k: f.interior_cpu() for k, f in old.prognostic_fields().items()}
    # same arithmetic text as the round-5 kernel; hipcc contracts a few products differently in the restructured loop, and again in the
    # two-level trip of the pipelined variants: bit for bit within each group, 1e-13 between the groups and against the round-5 kernel
    groups = (((6, 64), (0, 64), (2, 128), (4, 256), (5, 64), (5, 128), (1, 256), (22, 64)),
              ((29, 128), (8, 64), (12, 128), (13, 64), (13, 256), (28, 128), (29, 512)))
    for grp in groups:
        _, ref = run(1, 0, *grp[0])
        fr = {k: f.interior_cpu() for k, f in ref.prognostic_fields().items()}
        for k in fr:
            assert rel(fr[k], fo[k]) <= 1e-13, (k, grp[0])
        for cfg, bx in grp[1:]:
            _, b = run(1, 0, cfg, bx)
            for k, f in b.prognostic_fields().items():
                assert np.array_equal(f.interior_cpu(), fr[k]), (k, cfg, bx)
    for k, f in a.prognostic_fields().items():
        assert rel(f.interior_cpu(), fo[k]) <= 1e-12, k


@pytest.mark.parametrize("size", [(64, 16, 12), (40, 12, 9)])
def test_dry_steps_skip_the_unread_time_averages_bitwise(oracle, oc, bz, size, monkeypatch):
    """Round 6: inside bz_time_step_compressible the time-averaged velocities of stages 1 and 2 feed only the moisture tendency of the next
    stage; a model whose rho q the opening scan found identically zero skips that tendency (exact zeros), so its substep kernels of stages
    1 and 2 neither read nor write the three accumulators and the stage epilogue forms no averages (AcParams::skip_avg_if_dry; device-side
    word, no host decision).  Stage 3 accumulates as ever.  Dry: every prognostic field AND the substepper's time-averaged velocities after
    the step carry the bits of the run that never skips (BZ_NO_DRY_SHORTCUT=1); then vapour is set and two more steps stay equal as well
    (the scan turns the word to "moist": nothing is skipped any more); the oracle agrees at the usual 5e-9."""
    def run(no_shortcut):
        if no_shortcut:
            monkeypatch.setenv("BZ_NO_DRY_SHORTCUT", "1")
        else:
            monkeypatch.delenv("BZ_NO_DRY_SHORTCUT", raising=False)
        om, hm = make_pair(oracle, oc, bz, size=size, substeps=6)
        g = om.grid

        def theta(x, y, z):
            r = np.sqrt(x ** 2 + (y - 300.0) ** 2 + (z - 3000.0) ** 2)
            return 300.0 + 2.0 * np.maximum(0.0, 1.0 - r / 2000.0)

        rho = om.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
        u = lambda x, y, z: 3.0 + 0 * x + 0 * y + 0 * z      # noqa: E731
        om.set(rho=rho, theta=theta, u=u, v=-2.0, w=0.0)
        hm.set(ρ=rho, θ=theta, u=u, v=-2.0, w=0.0)
        out = []
        for _ in range(3):
            hm.time_step(0.5)
        hm.synchronize()
        sub = hm.timestepper.substepper
        snap = lambda: {**{k: f.interior_cpu().copy() for k, f in hm.prognostic_fields().items()},      # noqa: E731
                        **{n: getattr(sub, n).interior_cpu().copy() for n in ("time_averaged_u", "time_averaged_v", "time_averaged_w")}}
        out.append(snap())
        qv = lambda x, y, z: 4e-3 * np.exp(-z / 2500.0) * (1.0 + 0.2 * np.sin(2 * np.pi * x / 20e3)) + 0 * y      # noqa: E731
        hm.set(qᵗ=qv)
        for _ in range(2):
            hm.time_step(0.5)
        hm.synchronize()
        out.append(snap())
        return om, hm, out

    om, hm, a = run(False)
    _, _, b = run(True)
    for x, y in zip(a, b):
        for k in x:
            assert np.array_equal(x[k], y[k]), k
    assert np.abs(a[1]["ρq"]).max() > 0 and np.abs(a[0]["ρq"]).max() == 0
    for _ in range(3):
        om.time_step(0.5)
    ref_fields = a[0]
    g = om.grid
    want = g.interior(om.ru, zface=False)
    got = ref_fields["ρu"]
    assert np.abs(got - want).max() / np.abs(want).max() < 5e-9


@pytest.mark.parametrize("kind", ["dry", "vapour", "kessler", "saturation_adjustment"])
def test_whole_step_rotates_its_buffers_instead_of_storing_the_initial_state(oracle, oc, bz, kind, monkeypatch):
    """Round 6: bz_time_step_compressible copies nothing into U0 (store_initial_state!, 12 words per cell and step).  The state arrays stay
    intact as "U0" while stage 1 writes its recovered state into the U0 ARRAYS (out of place, rho_d with the other fields: no separate
    density pass), stages 2 - 3 run there, and stage 3's epilogue writes the final state back (csrc/bz_compressible.hip:
    compressible_step_body).  Same arithmetic on the same values: every prognostic field, every diagnostic, the substepper's fields and the
    Kessler species after three steps carry the bits of the run that copies (BZ_AC_ROTATE=0) — halos included."""
    out = {}
    for rot in (1, 0):
        monkeypatch.setenv("BZ_AC_ROTATE", str(rot))
        kw = {}
        if kind == "kessler":
            kw = dict(thermodynamic_constants=bz.ThermodynamicConstants(saturation_vapor_pressure=bz.TetensFormula()),
                      microphysics=bz.DCMIP2016KesslerMicrophysics())
        if kind == "saturation_adjustment":
            kw = dict(microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium()))
        grid = bz.RectilinearGrid((64, 16, 24), x=EXTENT["x"], y=EXTENT["y"], z=EXTENT["z"])
        dyn = bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(substeps=6), reference_potential_temperature=300.0)
        hm = bz.CompressibleAtmosphereModel(grid, dyn, advection=bz.WENO(order=5), **kw)

        def theta(x, y, z):
            r = np.sqrt(x ** 2 + y ** 2 + (z - 3000.0) ** 2)
            return 300.0 + 2.0 * np.maximum(0.0, 1.0 - r / 2000.0)

        qv = lambda x, y, z: (1.4e-2 if kind in ("kessler", "saturation_adjustment") else 5e-3) * np.exp(-z / 2e3) * (1 + 0.2 * np.sin(2 * np.pi * x / 8e3)) + 0 * y      # noqa: E731
        Hz, Nz = grid.Hz, grid.Nz
        rho = hm.dynamics.reference_state.density[Hz:Hz + Nz][:, None, None]
        sets = dict(ρ=rho, θ=theta, u=lambda x, y, z: 3.0 + 0 * x + 0 * y + 0 * z, v=-1.0, w=0.0)
        if kind != "dry":
            sets["qᵗ"] = qv
        hm.set(**sets)
        for _ in range(3):
            hm.time_step(0.5)
        hm.synchronize()
        sub = hm.timestepper.substepper
        snap = {k: f.cpu().copy() for k, f in hm.prognostic_fields().items()}
        snap.update({k: f.cpu().copy() for k, f in hm.velocities.items()})
        snap.update(theta=hm.potential_temperature.cpu().copy(), T=hm.temperature.cpu().copy(), p=hm.dynamics.pressure.cpu().copy(),
                    rho=hm.dynamics.total_density.cpu().copy(), q=hm.specific_moisture.cpu().copy())
        snap.update({n: getattr(sub, n).cpu().copy() for n in ("time_averaged_u", "time_averaged_v", "time_averaged_w", "density_perturbation",
                                                               "momentum_perturbation_w", "potential_temperature")})
        out[rot] = snap
    for k in out[1]:
        assert np.array_equal(out[1][k], out[0][k]), k
    assert all(np.isfinite(v).all() for v in out[1].values())
    if kind != "dry":
        assert np.abs(out[1]["ρq"]).max() > 0


@pytest.mark.parametrize("substeps", [6, 5, 2])
def test_time_averages_accumulated_in_pairs_of_substeps(oracle, oc, bz, substeps, monkeypatch):
    """Round 6 (AcParams::acc_mode): the forward sweep adds <u>, <v> two substeps at a time — a += (u'_{n-1} + u'_n) in the second substep of a
    pair, the first one leaves the accumulators alone — with the pairs counted from the stage's last substep (N_tau = 2, 3, 6 / 2, 3, 5 / 1, 1, 2:
    an odd substep out, stages that are one pair, one-substep stages).  Against the substep-by-substep accumulation (BZ_AC_PAIR_AVG=0): the
    averages differ by roundings of the accumulator (1e-14 of max-abs), a moist run's state after three steps by what the moisture tendency
    makes of that (1e-13); the oracle comparison of the default build is test_time_steps_match_oracle."""
    out = {}
    for pair in (1, 0):
        monkeypatch.setenv("BZ_AC_PAIR_AVG", str(pair))
        om, hm = make_pair(oracle, oc, bz, size=(64, 16, 24), substeps=substeps)
        g = om.grid

        def theta(x, y, z):
            r = np.sqrt(x ** 2 + y ** 2 + (z - 3000.0) ** 2)
            return 300.0 + 2.0 * np.maximum(0.0, 1.0 - r / 2000.0)

        qv = lambda x, y, z: 5e-3 * np.exp(-z / 2e3) * (1 + 0.2 * np.sin(2 * np.pi * x / 8e3)) + 0 * y      # noqa: E731
        rho = om.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
        hm.set(ρ=rho, θ=theta, u=lambda x, y, z: 3.0 + 0 * x + 0 * y + 0 * z, v=-1.0, w=0.0, qᵗ=qv)
        for _ in range(3):
            hm.time_step(0.5)      # 64-cell rows are 125 m wide
        hm.synchronize()
        sub = hm.timestepper.substepper
        out[pair] = {**{k: f.interior_cpu().copy() for k, f in hm.prognostic_fields().items()},
                     **{n: getattr(sub, n).interior_cpu().copy() for n in ("time_averaged_u", "time_averaged_v", "time_averaged_w")}}
    a, b = out[1], out[0]
    assert np.array_equal(a["time_averaged_w"], b["time_averaged_w"]) or rel(a["time_averaged_w"], b["time_averaged_w"]) <= 1e-13
    for k in ("time_averaged_u", "time_averaged_v"):
        assert rel(a[k], b[k]) <= 1e-14, (k, rel(a[k], b[k]))
    for k in hm.prognostic_fields():
        assert rel(a[k], b[k]) <= 1e-13, (k, rel(a[k], b[k]))
    assert np.abs(a["ρq"]).max() > 0 and all(np.isfinite(v).all() for v in a.values())



@pytest.mark.parametrize("storage", [None, "float32"])
@pytest.mark.parametrize("size,td", [((64, 16, 12), dict(substeps=6)), ((40, 12, 9), dict(substeps=4, damping_coefficient=0.05, damp_vertical=True)),
                                     ((128, 32, 10), dict(substeps=1))])
def test_first_sweeps_form_the_initial_perturbations_bitwise(oracle, oc, bz, size, td, storage, monkeypatch):
    """Round 6: k_ac_stage_init no longer stores the stage's initial perturbations U0 - U (five words per cell) for the first forward sweep
    to read back: k_ac_forward2<.., INIT> and k_ac_column_backward<.., INIT> form them from U0 and U themselves (stage 1 of a whole step:
    exact zeros), rounded through the storage type as the stored ones were, and store only the initial (rho theta)' the next substep's
    damping reads.  Three moist steps (every stage accumulates, the moisture tendency is evaluated) carry the bits of the run with the
    stored perturbations (BZ_AC_INIT_FOLD=0): Float64 and Float32 substep storage, damped and undamped, one substep per stage (the
    first sweep is also the last)."""
    def run(fold):
        monkeypatch.setenv("BZ_AC_INIT_FOLD", "1" if fold else "0")
        om, hm = make_pair(oracle, oc, bz, size=size, substep_floattype=np.float32 if storage else None, **td)
        g = om.grid

        def theta(x, y, z):
            r = np.sqrt(x ** 2 + (y - 300.0) ** 2 + (z - 3000.0) ** 2)
            return 300.0 + 2.0 * np.maximum(0.0, 1.0 - r / 2000.0)

        rho = om.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
        qv = lambda x, y, z: 4e-3 * np.exp(-z / 2500.0) * (1.0 + 0.2 * np.sin(2 * np.pi * x / 20e3)) + 0 * y      # noqa: E731
        hm.set(ρ=rho, θ=theta, u=3.0, v=-2.0, w=0.0, qᵗ=qv)
        for _ in range(3):
            hm.time_step(0.5)
        hm.synchronize()
        sub = hm.timestepper.substepper
        return {**{k: f.interior_cpu().copy() for k, f in hm.prognostic_fields().items()},
                **{n: getattr(sub, n).interior_cpu().copy() for n in ("time_averaged_u", "time_averaged_v", "time_averaged_w")}}

    a, b = run(True), run(False)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert np.isfinite(a["ρw"]).all() and np.abs(a["ρw"]).max() > 0


@pytest.mark.parametrize("size,kessler", [((64, 16, 12), False), ((128, 24, 70), False), ((64, 8, 9), True)])
def test_lds_tiled_scalar_tendency_matches_the_l1_kernel_and_the_oracle(oracle, oc, bz, size, kessler, monkeypatch):
    """Round 6: k_scalar_rho3d_lds (c and rho staged in LDS tiles with their frames, stencils by ds_read, the structure of k6_u) evaluates the
    fluxes of k_scalar_tendency_rho3d_x with the same expressions: three moist steps (rho theta + G_rho, the moisture tendency with the
    time-averaged velocities; with Kessler the two species as well) agree with the L1 kernel's run (BZ_SCALAR_LDS=0) to 1e-13 and with the
    oracle at the usual tolerance; one and two tile columns, 9 to 70 levels (two 64-level chunks), one tile row of eight."""
    def run(lds):
        monkeypatch.setenv("BZ_SCALAR_LDS", "1" if lds else "0")
        if kessler:
            og = oracle.Grid(size, x=EXTENT["x"], y=EXTENT["y"], z=EXTENT["z"])
            om = oc.CompressibleOracleModel(og, time_discretization=oc.SplitExplicit(substeps=6), reference_potential_temperature=300.0, microphysics="Kessler")
            grid = bz.RectilinearGrid(size, x=EXTENT["x"], y=EXTENT["y"], z=EXTENT["z"])
            dyn = bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(substeps=6), reference_potential_temperature=300.0)
            hm = bz.CompressibleAtmosphereModel(grid, dyn, advection=bz.WENO(order=5),
                                                thermodynamic_constants=bz.ThermodynamicConstants(saturation_vapor_pressure=bz.TetensFormula()),
                                                microphysics=bz.DCMIP2016KesslerMicrophysics())
        else:
            om, hm = make_pair(oracle, oc, bz, size=size, substeps=6)
        g = om.grid

        def theta(x, y, z):
            r = np.sqrt(x ** 2 + (y - 300.0) ** 2 + (z - 3000.0) ** 2)
            return 300.0 + 2.0 * np.maximum(0.0, 1.0 - r / 2000.0)

        rho = om.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
        qv = lambda x, y, z: 4e-3 * np.exp(-z / 2500.0) * (1.0 + 0.2 * np.sin(2 * np.pi * x / 20e3)) + 0 * y      # noqa: E731
        kw = dict(qcl=lambda x, y, z: 2e-4 * np.exp(-((z - 2000.0) / 800.0) ** 2) + 0 * x + 0 * y,
                  qr=lambda x, y, z: 1e-4 * np.exp(-((z - 1500.0) / 700.0) ** 2) + 0 * x + 0 * y) if kessler else {}
        om.set(rho=rho, theta=theta, u=3.0, v=-2.0, w=0.0, qv=qv, **kw)
        hm.set(ρ=rho, θ=theta, u=3.0, v=-2.0, w=0.0, qᵗ=qv, **kw)
        for _ in range(3):
            hm.time_step(0.5)
        hm.synchronize()
        return om, hm

    om, a = run(True)
    _, b = run(False)
    fa, fb = a.prognostic_fields(), b.prognostic_fields()
    for k in fa:
        assert rel(fa[k].interior_cpu(), fb[k].interior_cpu()) <= 1e-13, k
    for _ in range(3):
        om.time_step(0.5)
    cmp_interior(om, a, ("rho_d", "rtheta", "ru", "rv", "rw", "rq"), 1e-8 if kessler else 5e-9)


def test_whole_step_matches_operator_sequence(oracle, oc, bz):
    """bz_time_step_compressible (fused linearisation, no redundant velocity pass) == the reference's operator
    sequence issued call by call."""
    om, a = make_pair(oracle, oc, bz, substeps=6)
    _, b = make_pair(oracle, oc, bz, substeps=6)
    seeded_state(om, 7)
    for m in (a, b):
        push(om, m)
        m.clock.iteration = 1        # state already prepared by the oracle's update_state
    bz.compressible.time_step_(a, 1.5, whole_step=True)
    bz.compressible.time_step_(b, 1.5, whole_step=False)
    for n, f in O2H.items():
        x, y = f(a).interior_cpu(), f(b).interior_cpu()
        assert rel(x, y) <= 1e-14, n


def test_rest_state_stays_quiet_and_conserves_mass(oracle, oc, bz):
    """test/substepper_structural.jl S4/S5 and acoustic_substepping_stability.jl:329-353 on the device."""
    om, hm = make_pair(oracle, oc, bz, size=(16, 16, 16), theta_ref=lambda z: 250.0 * np.exp(9.80665 * z / (1005.0 * 250.0)))
    ref = hm.dynamics.reference_state
    g = hm.grid
    Hz, Nz = g.Hz, g.Nz
    Rd = 8.314462618 / 0.02897
    import torch
    rho = torch.from_numpy(ref.density.copy()).to(hm.device)[:, None, None]
    rth = torch.from_numpy(ref.pressure / (Rd * np.where(ref.exner_function == 0, 1, ref.exner_function))).to(hm.device)[:, None, None]
    hm.dynamics.dry_density.parent.copy_(rho.expand_as(hm.dynamics.dry_density.parent))
    hm.potential_temperature_density.parent.copy_(rth.expand_as(hm.potential_temperature_density.parent))
    bz.compressible.update_state_(hm, compute_tendencies=False)
    M0 = hm.dynamics.dry_density.interior.sum().item()
    for _ in range(5):
        hm.time_step(6.0)
    M1 = hm.dynamics.dry_density.interior.sum().item()
    assert abs(M1 - M0) / M0 <= 1e-12
    w = hm.velocities["w"].interior_cpu()
    assert np.isfinite(w).all()
    assert np.abs(w).max() < np.sqrt(np.finfo(float).eps)
    assert np.abs(w[-1]).max() == 0.0 and np.abs(w[0]).max() == 0.0
    assert np.abs(hm.velocities["u"].interior_cpu()).max() < np.sqrt(np.finfo(float).eps)


def test_compressible_kessler_model_matches_oracle(oracle, oc, bz):
    """CompressibleDynamics + DCMIP2016KesslerMicrophysics: condensate-loaded total density, Newton EOS with the latent term,
    gamma R_m with liquid, species transported with the acoustic-mean velocities, column update closing the step."""
    size = (16, 12, 16)
    extent = dict(x=(0.0, 4e3), y=(0.0, 3e3), z=(0.0, 4e3))
    og = oracle.Grid(size, **extent)
    om = oc.CompressibleOracleModel(og, time_discretization=oc.SplitExplicit(substeps=6), surface_pressure=1e5,
                                    reference_potential_temperature=300.0, microphysics="Kessler")
    grid = bz.RectilinearGrid(size, **extent)
    tc = bz.ThermodynamicConstants(saturation_vapor_pressure=bz.TetensFormula())
    dyn = bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(substeps=6), surface_pressure=1e5, reference_potential_temperature=300.0)
    hm = bz.CompressibleAtmosphereModel(grid, dyn, advection=bz.WENO(order=5), thermodynamic_constants=tc,
                                        microphysics=bz.DCMIP2016KesslerMicrophysics())
    bub = lambda x, y, z: np.maximum(0.0, 1.0 - np.sqrt((x - 2e3) ** 2 + (y - 1.5e3) ** 2 + (z - 1500.0) ** 2) / 1200.0)
    th = lambda x, y, z: 300.0 + 0.004 * z + 1.0 * bub(x, y, z)
    qv = lambda x, y, z: 0.014 * np.exp(-z / 3000.0) + 0.004 * bub(x, y, z)
    qcl = lambda x, y, z: 0.003 * bub(x, y, z)
    qr = lambda x, y, z: 0.001 * bub(x, y, z)
    rho = om.ref.density[og.Hz:og.Hz + og.Nz][:, None, None]
    om.set(rho=rho, theta=th, u=2.0, v=0.0, w=0.0, qv=qv, qcl=qcl, qr=qr)
    hm.set(ρ=rho, θ=th, u=2.0, v=0.0, w=0.0, qᵗ=qv, qcl=qcl, qr=qr)
    μ = hm.microphysical_fields
    g = om.grid
    cmp_interior(om, hm, ("rho_d", "rho", "rtheta", "rq", "T", "p"), 1e-13)
    assert rel(μ["ρqᶜˡ"].interior_cpu(), g.interior(om.rqcl)) <= 1e-15
    for _ in range(2):
        om.time_step(2.0)
        hm.time_step(2.0)
    worst = cmp_interior(om, hm, ("rho_d", "rtheta", "rq", "ru", "rw", "T", "p"), 1e-8)
    for n, k in (("rqcl", "ρqᶜˡ"), ("rqr", "ρqʳ"), ("W", "𝕎ʳ"), ("qcl", "qᶜˡ")):
        want = g.interior(getattr(om, n))
        assert np.abs(μ[k].interior_cpu() - want).max() <= 1e-8 * max(np.abs(want).max(), 1e-9), n
    assert g.interior(om.W).max() > 0.5
    print("compressible kessler parity:", {k: f"{v:.1e}" for k, v in worst.items()})


def test_compressible_kessler_on_moist_reference_matches_oracle(oracle, oc, bz):
    """The configuration shape of BASELINE configs[4]: z-dependent reference theta and reference vapour (moist
    ExnerReferenceState), pressure-balanced bubble, Kessler; two steps against the oracle."""
    size = (16, 12, 16)
    extent = dict(x=(0.0, 16e3), y=(0.0, 12e3), z=(0.0, 8e3))
    thb = lambda z: 300.0 + 0.0035 * z
    qvb = lambda z: float(0.013 * np.exp(-z / 2800.0))
    og = oracle.Grid(size, **extent)
    om = oc.CompressibleOracleModel(og, time_discretization=oc.SplitExplicit(substeps=6), surface_pressure=1e5,
                                    reference_potential_temperature=thb, reference_vapor_mass_fraction=qvb, microphysics="Kessler")
    grid = bz.RectilinearGrid(size, **extent)
    tc = bz.ThermodynamicConstants(saturation_vapor_pressure=bz.TetensFormula())
    dyn = bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(substeps=6), surface_pressure=1e5,
                                  reference_potential_temperature=thb, reference_vapor_mass_fraction=qvb)
    hm = bz.CompressibleAtmosphereModel(grid, dyn, advection=bz.WENO(order=5), thermodynamic_constants=tc,
                                        microphysics=bz.DCMIP2016KesslerMicrophysics())
    bub = lambda x, y, z: np.maximum(0.0, 1.0 - np.sqrt(((x - 8e3) / 4e3) ** 2 + ((y - 6e3) / 4e3) ** 2 + ((z - 1500.0) / 1500.0) ** 2))
    th = lambda x, y, z: thb(z) + 2.0 * bub(x, y, z)
    qv = lambda x, y, z: np.vectorize(qvb)(z) + 0.003 * bub(x, y, z) + 0 * x + 0 * y
    rho_ref = om.ref.density[og.Hz:og.Hz + og.Nz][:, None, None]
    x, y, z = og.nodes("ccc")
    rho = rho_ref * thb(z) / th(x, y, z)            # pressure_balanced_density
    om.set(rho=rho, theta=th, u=5.0, v=0.0, w=0.0, qv=qv)
    hm.set(ρ=rho, θ=th, u=5.0, v=0.0, w=0.0, qᵗ=qv)
    cmp_interior(om, hm, ("rho_d", "rho", "rtheta", "rq", "T", "p"), 1e-13)
    for _ in range(2):
        om.time_step(2.0)
        hm.time_step(2.0)
    cmp_interior(om, hm, ("rho_d", "rtheta", "rq", "ru", "rw", "T", "p"), 1e-8)
    assert np.abs(og.interior(om.rw, True)).max() > 1e-3


def test_compressible_saturation_adjustment_matches_oracle(oracle, oc, bz):
    """CompressibleDynamics + SaturationAdjustment(WarmPhaseEquilibrium): the density-based adjustment (q^v on the saturation
    curve at the cell's own total density), one Newton temperature for dynamics and microphysics, gamma R_m with the liquid
    fraction; the reference's integration scenario (test/compressible_saturation_adjustment.jl:114-148) plus two full steps."""
    from oracle import thermo as th_mod
    size = (16, 12, 12)
    extent = dict(x=(0.0, 4e3), y=(0.0, 3e3), z=(0.0, 3e3))
    thref = lambda z: 300.0 * np.exp(9.80616 * z / (1005 * 300.0))
    og = oracle.Grid(size, **extent)
    om = oc.CompressibleOracleModel(og, time_discretization=oc.SplitExplicit(substeps=6), surface_pressure=1e5,
                                    reference_potential_temperature=thref, microphysics="SaturationAdjustment")
    grid = bz.RectilinearGrid(size, **extent)
    dyn = bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(substeps=6), surface_pressure=1e5, reference_potential_temperature=thref)
    hm = bz.CompressibleAtmosphereModel(grid, dyn, advection=bz.WENO(order=5),
                                        microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium()))
    bub = lambda x, y, z: np.maximum(0.0, 1.0 - np.sqrt((x - 2e3) ** 2 + (y - 1.5e3) ** 2 + (z - 1200.0) ** 2) / 900.0)
    th = lambda x, y, z: 300.0 + 0.5 * bub(x, y, z) + 0 * z
    qt = lambda x, y, z: 0.012 + 0.014 * bub(x, y, z) + 0.004 * (z / 3e3)
    rho = om.ref.density[og.Hz:og.Hz + og.Nz][:, None, None]
    om.set(rho=rho, theta=th, u=2.0, v=0.0, w=0.0, qv=qt)
    hm.set(ρ=rho, θ=th, u=2.0, v=0.0, w=0.0, qᵗ=qt)
    μ = hm.microphysical_fields
    g = om.grid
    ql = g.interior(om.ql)
    assert 0.02 < (ql > 0).mean() < 0.98           # cloudy and clear cells
    cmp_interior(om, hm, ("rho_d", "rho", "rtheta", "rq", "T", "p"), 1e-12)
    assert np.abs(μ["qˡ"].interior_cpu() - ql).max() < 1e-13
    assert np.abs(μ["qᵛ"].interior_cpu() - g.interior(om.qv)).max() < 1e-13
    # the reference's own check on the device fields: q^v on the saturation curve at the cell's density in saturated cells
    tc = th_mod.ThermoConstants()
    T, r, qv = hm.temperature.interior_cpu(), hm.dynamics.total_density.interior_cpu(), μ["qᵛ"].interior_cpu()
    for idx in zip(*np.nonzero(μ["qˡ"].interior_cpu() > 1e-6)):
        assert abs(qv[idx] - th_mod.saturation_specific_humidity(T[idx], r[idx], tc, "liquid")) < 1e-4
    for _ in range(2):
        om.time_step(1.0)
        hm.time_step(1.0)
    cmp_interior(om, hm, ("rho_d", "rtheta", "rq", "ru", "rw", "T", "p"), 1e-8)
    assert np.abs(μ["qˡ"].interior_cpu() - g.interior(om.ql)).max() < 1e-9


@pytest.mark.parametrize("td", [dict(substeps=8, substep_distribution="constant"), dict(substep_distribution="constant"),
                                dict(substeps=8, substep_distribution="monolithic_first_stage")])
def test_substep_distributions_match_oracle(oracle, oc, bz, td):
    """ConstantSubstepSize / MonolithicFirstStage (acoustic_substepping.jl:468-508): the stage counts and sizes the library computes
    are the oracle's, and two full steps agree as for the default distribution"""
    om, hm = make_pair(oracle, oc, bz, size=(24, 16, 20), **td)
    g, c = om.grid, om.constants
    for dt in (0.7, 2.0, 5.0):
        for beta in (1 / 3, 1 / 2, 1.0):
            n, dtau = hm.stage_substeps(dt, beta)
            want = oc.stage_substep_count_and_size(om.td.substeps, beta, dt, g, c, om.td.acoustic_cfl, om.td.substep_distribution)
            assert n == want[0] and abs(dtau - want[1]) <= 1e-15 * dt, (dt, beta, n, dtau, want)

    def theta(x, y, z):
        r = np.sqrt(x ** 2 + y ** 2 + (z - 3000.0) ** 2)
        return 300.0 + 2.0 * np.maximum(0.0, 1.0 - r / 2000.0)

    rho = om.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
    om.set(rho=rho, theta=theta, u=lambda x, y, z: 3.0 + 0 * x + 0 * y + 0 * z, v=0.0, w=0.0, qv=0.0)
    hm.set(ρ=rho, θ=theta, u=lambda x, y, z: 3.0 + 0 * x + 0 * y + 0 * z, v=0.0, w=0.0, qᵗ=0.0)
    for _ in range(2):
        om.time_step(2.0)
        hm.time_step(2.0)
    cmp_interior(om, hm, ("rho_d", "rtheta", "ru", "rv", "rw", "T", "p"), 5e-9)
    assert om.last_substeps == [hm.stage_substeps(2.0, b)[0] for b in (1 / 3, 1 / 2, 1.0)]


@pytest.mark.parametrize("td", [dict(substeps=6), dict(), dict(substeps=4, damping_coefficient=0.05, damp_vertical=True),
                                dict(substeps=6, damping_coefficient=None), dict(substeps=6, sponge=(0.2, 3000.0, "cubic")),
                                dict(substeps=8, substep_distribution="constant")])
def test_float32_substep_storage_in_a_float64_model(oracle, oc, bz, td):
    """substep_floattype = Float32 (acoustic_substepping.jl:199-235): the ten working fields of the substepper are float arrays, the
    kernels read them, compute in Float64 and store Float32; everything else stays Float64.  Three steps of the six time-discretisation
    variants against the Float64 oracle.  Tolerance 2e-6 of the field scale: a stored perturbation carries 6e-8 of its own size, the
    linearisation arrays theta^L, gamma R^m Pi 6e-8 of theirs, accumulated over 3 steps x 3 stages x up to 18 substeps (measured: a
    few 1e-7; the Float64-storage run of the same cases agrees to 5e-9).  And the path is exercised: the result differs from the
    Float64-storage run by more than rounding."""
    import torch
    om, hm = make_pair(oracle, oc, bz, size=(24, 16, 24), substep_floattype=np.float32, **td)
    _, h64 = make_pair(oracle, oc, bz, size=(24, 16, 24), **td)
    sub = hm.timestepper.substepper
    for name, _ in sub.FIELDS:
        want = torch.float32 if name in sub.WORKING else torch.float64
        assert getattr(sub, name).parent.dtype == want, name
    assert hm.momentum["ρu"].parent.dtype == torch.float64
    g = om.grid

    def theta(x, y, z):
        r = np.sqrt(x ** 2 + y ** 2 + (z - 3000.0) ** 2)
        return 300.0 + 2.0 * np.maximum(0.0, 1.0 - r / 2000.0)

    def qv(x, y, z):
        return 5e-3 * np.exp(-z / 2e3) * (1 + 0.2 * np.sin(2 * np.pi * x / 8e3)) + 0 * y

    rho = om.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
    u0 = lambda x, y, z: 3.0 + 0 * x + 0 * y + 0 * z
    om.set(rho=rho, theta=theta, u=u0, v=0.0, w=0.0, qv=qv)
    for m in (hm, h64):
        m.set(ρ=rho, θ=theta, u=u0, v=0.0, w=0.0, qᵗ=qv)
    for _ in range(3):
        om.time_step(2.0)
        hm.time_step(2.0)
        h64.time_step(2.0)
    worst = cmp_interior(om, hm, ("rho_d", "rtheta", "rq", "ru", "rv", "rw", "T", "p"), 2e-6)
    print("float32 substep storage vs float64 oracle:", {k: f"{v:.1e}" for k, v in worst.items()})
    diff = np.abs(hm.momentum["ρw"].interior_cpu() - h64.momentum["ρw"].interior_cpu()).max()
    assert diff > 1e-12 * np.abs(h64.momentum["ρw"].interior_cpu()).max()


def test_float32_substep_storage_is_rejected_where_it_is_not_built(bz):
    grid = bz.RectilinearGrid((16, 16, 8), x=(0, 4e3), y=(0, 4e3), z=(0, 4e3))
    dyn = lambda **kw: bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(substeps=6, **kw), reference_potential_temperature=300.0)
    with pytest.raises(Exception):      # DirectDivergenceDamping reads the working fields as the grid's real
        bz.CompressibleAtmosphereModel(grid, dyn(damping=bz.DirectDivergenceDamping(coefficient=0.1)), advection=bz.WENO(), substep_floattype=np.float32)
    with pytest.raises(NotImplementedError):
        bz.CompressibleAtmosphereModel(bz.RectilinearGrid((16, 16, 8), x=(0, 4e3), y=(0, 4e3), z=(0, 4e3), float_type=np.float32), dyn(),
                                       advection=bz.WENO(), substep_floattype=np.float64)


def _sin2_mask(z, zlo=5e3, zhi=8e3):
    """the sponge mask of examples/tropical_cyclone_with_rainband.jl:460-463: sin^2(pi xi / 2) above z-"""
    xi = (z - zlo) / (zhi - zlo)
    return np.sin(np.pi * xi / 2) ** 2 * (xi > 0)


def _rainband_heating(x, y, z):
    return 1e-3 * np.exp(-((x - 1e3) ** 2 + (y + 500.0) ** 2) / 1500.0 ** 2) * np.sin(np.pi * np.clip((z - 1e3) / 4e3, 0.0, 1.0)) ** 2


def _cyclone_pair(oracle, oc, bz, size=(24, 16, 20)):
    """CompressibleDynamics(SplitExplicitTimeDiscretization()) + FPlane + sponges on rho u, rho v, rho w (to zero) and rho theta (to the
    reference profile) + the prescribed heating keyed theta: the forcing list of examples/tropical_cyclone_with_rainband.jl:419-514"""
    f, rate = 5e-4, 1.0 / 333.0
    og = oracle.Grid(size, x=EXTENT["x"], y=EXTENT["y"], z=EXTENT["z"])
    om = oc.CompressibleOracleModel(og, time_discretization=oc.SplitExplicit(substeps=6), reference_potential_temperature=300.0, coriolis_f=f)
    Hz, Nz = og.Hz, og.Nz
    rth_bg = om.ref.density[Hz:Hz + Nz] * 300.0
    om.relaxation = {"ru": (rate * _sin2_mask(og.zc), np.zeros(Nz)), "rv": (rate * _sin2_mask(og.zc), np.zeros(Nz)),
                     "rw": (rate * _sin2_mask(og.zf), np.zeros(Nz + 1)), "rtheta": (rate * _sin2_mask(og.zc), rth_bg)}
    x, y, z = og.nodes("ccc")
    om.field_forcing = (np.broadcast_to(_rainband_heating(x, y, z), (og.Nz, og.Ny, og.Nx)).copy(), True)
    grid = bz.RectilinearGrid(size, x=EXTENT["x"], y=EXTENT["y"], z=EXTENT["z"])
    dyn = bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(substeps=6), reference_potential_temperature=300.0, reference_state="auto")
    sponge = lambda target=0.0: bz.Relaxation(rate=rate, mask=_sin2_mask, target=target)
    hm = bz.CompressibleAtmosphereModel(grid, dyn, advection=bz.WENO(order=5), coriolis=bz.FPlane(f=f),
                                        forcing={"ρu": sponge(), "ρv": sponge(), "ρw": sponge(), "ρθ": sponge(rth_bg), "θ": bz.Forcing(_rainband_heating)})
    return om, hm


def test_compressible_coriolis_and_sponge_slow_tendencies(oracle, oc, bz):
    om, hm = _cyclone_pair(oracle, oc, bz)
    seeded_state(om, 7)
    om.compute_slow_tendencies()
    plain = oc.CompressibleOracleModel(om.grid, time_discretization=oc.SplitExplicit(substeps=6), reference_potential_temperature=300.0)
    seeded_state(plain, 7)
    plain.compute_slow_tendencies()
    push(om, hm)
    for k in hm.G:
        if k != "ρq":
            hm.G[k].parent.zero_()
    bz.compressible.compute_slow_tendencies_(hm)
    g = om.grid
    for n, k in PROG.items():
        if n == "rq":
            continue
        a, b, c = hm.G[k].interior_cpu(), g.interior(om.G[n], n == "rw"), g.interior(plain.G[n], n == "rw")
        if n == "rw":
            a, b, c = a[1:-1], b[1:-1], c[1:-1]
        assert rel(a, b) <= 1e-12, (n, rel(a, b))
        if n != "rho_d":
            assert np.abs(b - c).max() > 0, n                                        # the terms are there ...
            assert np.abs((a - c) - (b - c)).max() <= 1e-9 * np.abs(b - c).max(), n      # ... and they are what the device added


def test_compressible_cyclone_forcing_list_steps_match_oracle(oracle, oc, bz):
    om, hm = _cyclone_pair(oracle, oc, bz, size=(24, 16, 24))
    g = om.grid

    def theta(x, y, z):
        r = np.sqrt(x ** 2 + y ** 2 + (z - 3000.0) ** 2)
        return 300.0 + 2.0 * np.maximum(0.0, 1.0 - r / 2000.0)

    rho = om.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
    vortex = lambda x, y, z: 5.0 * np.sin(2 * np.pi * y / 8e3) + 0 * x + 0 * z
    om.set(rho=rho, theta=theta, u=vortex, v=lambda x, y, z: -5.0 * np.sin(2 * np.pi * x / 8e3) + 0 * y + 0 * z, w=0.0)
    hm.set(ρ=rho, θ=theta, u=vortex, v=lambda x, y, z: -5.0 * np.sin(2 * np.pi * x / 8e3) + 0 * y + 0 * z, w=0.0)
    for _ in range(3):
        om.time_step(2.0)
        hm.time_step(2.0)
    worst = cmp_interior(om, hm, ("rho_d", "rtheta", "ru", "rv", "rw", "u", "v", "w", "theta", "T", "p"), 5e-9)
    print("cyclone forcing list, 3 steps:", {k: f"{v:.1e}" for k, v in worst.items()})


def test_compressible_forcings_outside_the_built_list_raise(bz):
    grid = bz.RectilinearGrid((16, 16, 8), x=EXTENT["x"], y=EXTENT["y"], z=EXTENT["z"])
    dyn = lambda: bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(substeps=6), reference_potential_temperature=300.0, reference_state="auto")
    with pytest.raises(NotImplementedError):
        bz.CompressibleAtmosphereModel(grid, dyn(), advection=bz.WENO(order=5), forcing={"w": bz.Relaxation(rate=0.1)})
    with pytest.raises(NotImplementedError):
        bz.CompressibleAtmosphereModel(grid, dyn(), advection=bz.WENO(order=5), forcing={"θ": bz.Forcing(lambda z: 1e-3)})
    with pytest.raises(NotImplementedError):
        bz.CompressibleAtmosphereModel(grid, dyn(), advection=bz.WENO(order=5), forcing={"u": bz.Forcing(lambda x, y, z: 1e-3 + 0 * x)})
