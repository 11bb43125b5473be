"""The reference's example scripts, by configuration: each model is constructed with the example's own keyword list (topology, halo,
formulation, advection scheme, microphysics, closure, precision) at a reduced grid and stepped; the run must stay finite and do the
physically expected thing.  Parity of every ingredient is established elsewhere (file named per case); this file checks that the
combinations a user of the reference would type are accepted as written."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _finite(m):
    return all(np.isfinite(f.interior_cpu()).all() for f in m.prognostic_fields().values())


def test_dry_thermal_bubble_jl(bz):
    """examples/dry_thermal_bubble.jl:15-25: 2-D (Periodic, Flat, Bounded), halo (5, 5), StaticEnergy, WENO(order = 9)
    (parity: tests/test_flat_topology.py)"""
    grid = bz.RectilinearGrid((64, 64), halo=(5, 5), x=(-10e3, 10e3), z=(0.0, 10e3), topology=(bz.Periodic, bz.Flat, bz.Bounded))
    model = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid)), formulation=":StaticEnergy", advection=bz.WENO(order=9))
    θ = lambda x, z: 288.0 * np.exp(1e-6 * z / 9.81) + 10.0 * np.where(np.hypot(x, z - 3e3) < 2e3, np.cos(np.pi / 2 * np.hypot(x, z - 3e3) / 2e3) ** 2, 0.0)
    model.set(θ=θ)
    for _ in range(20):
        model.time_step(2.0)
    model.synchronize()
    assert _finite(model) and model.velocities["w"].interior_cpu().max() > 0.5      # the bubble rises


@pytest.mark.parametrize("float_type", [np.float64, np.float32])
def test_cloudy_thermal_bubble_jl(bz, float_type):
    """examples/cloudy_thermal_bubble.jl:20-75,96-140: (Bounded, Flat, Bounded), 128 x 128, halo (5, 5), surface_pressure 1e5, WENO(order = 9);
    first dry, then with SaturationAdjustment(equilibrium = WarmPhaseEquilibrium()) (parity: tests/test_bounded_x.py); in both precisions"""
    grid = bz.RectilinearGrid((128, 128), halo=(5, 5), x=(-10e3, 10e3), z=(0.0, 10e3), topology=(bz.Bounded, bz.Flat, bz.Bounded), float_type=float_type)
    tc = bz.ThermodynamicConstants()
    for micro in (None, bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium())):
        ref = bz.ReferenceState(grid, tc, surface_pressure=1e5, potential_temperature=300.0)
        model = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), thermodynamic_constants=tc, advection=bz.WENO(order=9), microphysics=micro)
        θ = lambda x, z: 300.0 + 2.0 * np.cos(np.pi * np.minimum(1.0, np.sqrt((x / 2e3) ** 2 + ((z - 2e3) / 2e3) ** 2)) / 2) ** 2
        if micro is None:
            model.set(θ=θ)
        else:
            model.set(θ=θ, qᵗ=lambda x, z: 0.020 * np.exp(-z / 2000.0) + 0 * x)
        for _ in range(20):
            model.time_step(2.0)
        model.synchronize()
        assert _finite(model) and model.velocities["w"].interior_cpu().max() > 0.1      # the bubble rises
        assert float(model.momentum["ρu"].interior[:, :, 0].abs().max()) == 0.0          # the west wall stays closed
        assert model.max_abs_divergence() < (1e-10 if float_type is np.float64 else 1e-3)


def test_bomex_jl(bz):
    """examples/bomex.jl:40-210: Float32, WENO(order = 9), SaturationAdjustment, SmagorinskyLilly, Coriolis + geostrophic + subsidence +
    drying / cooling forcings, bottom fluxes (parity: tests/test_closure.py, test_forcings.py, test_float32.py)"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_forcings import EXTENT, _hip_forcing_kwargs
    grid = bz.RectilinearGrid((32, 32, 24), halo=(5, 5, 5), x=EXTENT[0], y=EXTENT[1], z=EXTENT[2], float_type=np.float32)
    ref = bz.ReferenceState(grid, surface_pressure=101500.0, potential_temperature=299.1)
    model = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=9), closure=bz.SmagorinskyLilly(),
                               microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium()), **_hip_forcing_kwargs(bz, full=True))
    rng = np.random.default_rng(0)
    sh = (24, 32, 32)
    model.set(θ=lambda x, y, z: 298.7 + 0.004 * np.maximum(z - 520.0, 0.0) + 0.1 * rng.standard_normal(sh),
              qᵗ=lambda x, y, z: 0.017 * np.exp(-z / 2500.0) * (1 + 0.01 * rng.standard_normal(sh)), u=-8.75)
    for _ in range(10):
        model.time_step(2.0)
    model.synchronize()
    assert _finite(model) and model.closure_fields["νₑ"].interior_cpu().max() > 0.0


def test_splitting_supercell_jl(bz):
    """examples/splitting_supercell.jl:86-290: Float32, CompressibleDynamics with the split-explicit discretisation, DCMIP2016 Kessler with
    TetensFormula constants, WENO(order = 9) (parity: tests/test_gpu_compressible.py, test_weno_orders.py, test_float32.py)"""
    grid = bz.RectilinearGrid((32, 32, 20), halo=(5, 5, 5), x=(0.0, 48e3), y=(0.0, 48e3), z=(0.0, 20e3), float_type=np.float32)
    θb = lambda z: 300.0 + 0.004 * z
    dyn = bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(), surface_pressure=1e5, reference_potential_temperature=θb)
    model = bz.CompressibleAtmosphereModel(grid, dyn, advection=bz.WENO(order=9), microphysics=bz.DCMIP2016KesslerMicrophysics(),
                                           thermodynamic_constants=bz.ThermodynamicConstants(saturation_vapor_pressure=bz.TetensFormula()))
    col = np.asarray(dyn.reference_state.density)[grid.Hz:grid.Hz + grid.Nz][:, None, None]
    bub = lambda x, y, z: np.maximum(0.0, 1.0 - np.sqrt(((x - 24e3) / 10e3) ** 2 + ((y - 24e3) / 10e3) ** 2 + ((z - 1500.0) / 1500.0) ** 2))
    θ = lambda x, y, z: θb(z) + 3.0 * bub(x, y, z)
    model.set(ρ=lambda x, y, z: col * θb(z) / θ(x, y, z), θ=θ, u=lambda x, y, z: 0.0015 * np.minimum(z, 5e3) + 0 * x + 0 * y, v=0.0, w=0.0,
              qᵗ=lambda x, y, z: 0.014 * np.exp(-z / 2500.0) + 0 * x + 0 * y)
    for _ in range(5):
        model.time_step(2.0)
    model.synchronize()
    assert _finite(model) and model.velocities["w"].interior_cpu().max() > 0.05


def test_inertia_gravity_wave_jl(bz):
    """examples/inertia_gravity_wave.jl:63-140: 2-D (Periodic, Flat, Bounded) with halo (5, 5), anelastic and split-explicit compressible
    models side by side (parity: tests/test_flat_topology.py)"""
    Lx, Lz = 300e3, 10e3
    θbg = lambda z: 300.0 * np.exp(1e-4 * z / 9.80665)
    θi = lambda x, z: θbg(z) + 0.01 * np.sin(np.pi * z / Lz) / (1 + (x - Lx / 3) ** 2 / 5000.0 ** 2)
    grid = bz.RectilinearGrid((96, 10), halo=(5, 5), x=(0.0, Lx), z=(0.0, Lz), topology=(bz.Periodic, bz.Flat, bz.Bounded))
    an = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, surface_pressure=1e5, potential_temperature=300.0)),
                            advection=bz.WENO())
    an.set(θ=θi, u=20.0)
    dyn = bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(), surface_pressure=1e5, reference_potential_temperature=θbg)
    cm = bz.CompressibleAtmosphereModel(grid, dyn, advection=bz.WENO())
    col = np.asarray(dyn.reference_state.density)[grid.Hz:grid.Hz + grid.Nz][:, None, None] + np.zeros((10, 1, 96))
    cm.set(ρ=col, θ=θi, u=20.0, v=0.0, w=0.0)
    for _ in range(10):
        an.time_step(12.0)
        cm.time_step(12.0)
    an.synchronize()
    cm.synchronize()
    wa, wc = an.velocities["w"].interior_cpu(), cm.velocities["w"].interior_cpu()
    assert _finite(an) and _finite(cm) and np.abs(wa).max() > 1e-5
    # the two dynamical cores produce the same gravity wave to a few per cent at this amplitude
    assert np.abs(wa - wc).max() < 0.2 * np.abs(wa).max()


def test_rico_jl(bz):
    """examples/rico.jl:40-190: Float32, per-field advection with bounds-preserving WENO for the moisture density, FPlane + geostrophic +
    subsidence + large-scale drying / cooling profiles, the w sponge (Relaxation with a GaussianMask), bulk drag / sensible heat (keyed rho e) /
    vapour fluxes, SmagorinskyLilly (parity: tests/test_bounded_weno.py, test_forcings.py, test_relaxation.py, test_float32.py); the example's
    one-moment microphysics is outside this build — warm-phase saturation adjustment stands in."""
    Lz = 4e3
    grid = bz.RectilinearGrid((32, 32, 20), x=(0.0, 12.8e3), y=(0.0, 12.8e3), z=(0.0, Lz), float_type=np.float32)
    T0 = 299.8
    ws = lambda z: -0.005 * min(z, 2260.0) / 2260.0
    geo = bz.geostrophic_forcings(lambda z: -9.9 + 2e-3 * z, lambda z: -3.8)
    sub = bz.SubsidenceForcing(ws)
    forcing = {"u": (sub, geo.u), "v": (sub, geo.v), "w": bz.Relaxation(rate=1 / 8, mask=bz.GaussianMask(center=3500.0, width=500.0)),
               "qᵉ": (sub, bz.Forcing(lambda z: -1.0e-8 + (1.3456e-8) * min(z, 2980.0) / 2980.0)), "θ": (sub, bz.Forcing(lambda z: -2.5 / 86400.0))}
    bcs = {"ρe": bz.FieldBoundaryConditions(bottom=bz.BulkSensibleHeatFlux(coefficient=1.094e-3, surface_temperature=T0)),
           "ρqᵉ": bz.FieldBoundaryConditions(bottom=bz.BulkVaporFlux(coefficient=1.133e-3, surface_temperature=T0)),
           "ρu": bz.FieldBoundaryConditions(bottom=bz.BulkDrag(coefficient=1.229e-3)), "ρv": bz.FieldBoundaryConditions(bottom=bz.BulkDrag(coefficient=1.229e-3))}
    model = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, surface_pressure=101540.0, potential_temperature=297.9)),
                               advection={"momentum": bz.WENO(), "ρθ": bz.WENO(), "ρqᵉ": bz.WENO(bounds=(0, 1))}, coriolis=bz.FPlane(f=4.5e-5),
                               microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium()), closure=bz.SmagorinskyLilly(),
                               forcing=forcing, boundary_conditions=bcs)
    rng = np.random.default_rng(1)
    model.set(θ=lambda x, y, z: 297.9 + 0.0035 * z + 0.1 * rng.standard_normal((20, 32, 32)), qᵗ=lambda x, y, z: 0.016 * np.exp(-z / 2000.0) + 0 * x + 0 * y, u=-9.0, v=-3.8)
    for _ in range(10):
        model.time_step(2.0)
    model.synchronize()
    q = model.specific_moisture.interior_cpu()
    assert _finite(model) and q.min() >= -1e-7 and q.max() <= 1.0


def test_cloudy_kelvin_helmholtz_jl(bz, oracle):
    """examples/cloudy_kelvin_helmholtz.jl:33-40: 2-D (Periodic, Flat, Bounded), default dynamics, WENO(order = 5), SaturationAdjustment;
    a sheared moist layer — here also against the oracle's Flat implementation (two steps, 1e-9)."""
    size, ext = (96, 32), dict(x=(0.0, 10e3), z=(0.0, 3e3))
    grid = bz.RectilinearGrid(size, topology=(bz.Periodic, bz.Flat, bz.Bounded), **ext)
    model = bz.AtmosphereModel(grid, advection=bz.WENO(order=5), microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium()))
    θ0 = 288.0
    θ = lambda x, z: θ0 * np.exp(1e-4 * z / 9.80665) + 0.01 * np.sin(2 * np.pi * x / 10e3) * np.exp(-((z - 1.5e3) / 200.0) ** 2)
    u = lambda x, z: 5.0 * np.tanh((z - 1.5e3) / 150.0)
    q = lambda x, z: 0.012 * np.exp(-z / 2500.0) * (1.0 + 0.5 * np.exp(-((z - 1.5e3) / 300.0) ** 2))
    model.set(θ=θ, u=u, qᵗ=q)
    og = oracle.Grid(size, topology=("Periodic", "Flat", "Bounded"), **ext)
    om = oracle.OracleModel(og, microphysics="SaturationAdjustment")
    om.set(theta=lambda x, y, z: θ(x, z) + 0 * y, u=lambda x, y, z: u(x, z) + 0 * y, qt=lambda x, y, z: q(x, z) + 0 * y)
    for _ in range(2):
        model.time_step(1.0)
        om.time_step(1.0)
    model.synchronize()
    assert _finite(model) and model.microphysical_fields["qˡ"].interior_cpu().max() > 0.0      # the moist layer is cloudy
    for n, f in (("ru", model.momentum["ρu"]), ("rtheta", model.potential_temperature_density), ("rq", model.moisture_density), ("T", model.temperature)):
        want = og.interior(getattr(om, n))
        assert np.abs(f.interior_cpu() - want).max() < 1e-9 * np.abs(want).max(), n


def test_neutral_atmospheric_boundary_layer_jl(bz):
    """examples/neutral_atmospheric_boundary_layer.jl:36-146: 96^3 at halo (5, 5, 5), WENO(order = 9), SmagorinskyLilly, FPlane + geostrophic
    forcing, friction-velocity drag on rho u / rho v, Gaussian sponges on rho w (to zero) and rho theta (to the capping-inversion profile);
    parity of the list: tests/test_relaxation.py::test_neutral_boundary_layer_physics_list"""
    N = 48                                     # the example's 96 halved; same extents
    grid = bz.RectilinearGrid((N, N, N), halo=(5, 5, 5), x=(0.0, 3000.0), y=(0.0, 3000.0), z=(0.0, 1000.0))
    p0, th0 = 1e5, 300.0
    ref = bz.ReferenceState(grid, surface_pressure=p0, potential_temperature=th0)
    dz = 1000.0 / N
    zi1 = 468.0
    zi2, Gi, Gtop = zi1 + 6 * dz, 8 / (6 * dz), 0.003
    thr = lambda z: np.where(z < zi1, th0, np.where(z < zi2, th0 + Gi * (z - zi1), th0 + Gi * (zi2 - zi1) + Gtop * (z - zi2)))
    rho0 = p0 / (287.0 * th0)
    drag = bz.FluxBoundaryCondition(bz.FrictionVelocityDrag(rho0, 0.5, epsilon=1e-12))
    mask = bz.GaussianMask(center=1000.0, width=200.0)
    zc = np.asarray(grid.zᶜ)
    rho = ref.density[grid.Hz:grid.Hz + N]                       # the example's ρθᵣ = reference_state.density * θᵣ(z)
    geo = bz.geostrophic_forcings(lambda z: 15.0, lambda z: 0.0)
    forcing = {"u": geo.u, "v": geo.v, "ρw": bz.Relaxation(rate=0.01, mask=mask), "ρθ": bz.Relaxation(rate=0.01, mask=mask, target=rho * thr(zc))}
    model = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), coriolis=bz.FPlane(f=1e-4), advection=bz.WENO(order=9), forcing=forcing,
                               closure=bz.SmagorinskyLilly(), boundary_conditions={"ρu": bz.FieldBoundaryConditions(bottom=drag),
                                                                                  "ρv": bz.FieldBoundaryConditions(bottom=drag)})
    rng = np.random.default_rng(1)
    model.set(θ=lambda x, y, z: thr(z) + 0.1 * rng.standard_normal((N, N, N)) * (z < zi1), u=15.0)
    th_top0 = model.potential_temperature_density.interior_cpu()[-2].mean()
    for _ in range(10):
        model.time_step(0.5)
    model.synchronize()
    assert _finite(model) and model.closure_fields["νₑ"].interior_cpu().max() > 0.0
    assert abs(model.potential_temperature_density.interior_cpu()[-2].mean() - th_top0) < 1e-3 * th_top0      # the sponge holds the inversion profile


def test_tropical_cyclone_with_rainband_jl(bz):
    """examples/tropical_cyclone_with_rainband.jl:153-175,419-514: (Periodic, Periodic, Bounded) at 5 km x 333 m, halo (5, 5, 5),
    CompressibleDynamics(SplitExplicitTimeDiscretization()), FPlane, WENO(order = 5), sin^2 sponges above 20 km on rho u, rho v, rho w and
    rho theta, the prescribed rainband heating keyed theta (parity: tests/test_gpu_compressible.py::test_compressible_cyclone_*)"""
    N, Nz, L, Lz = 32, 25, 160e3, 25e3
    grid = bz.RectilinearGrid((N, N, Nz), halo=(5, 5, 5), x=(-L / 2, L / 2), y=(-L / 2, L / 2), z=(0.0, Lz))
    θb = lambda z: 300.0 + 0.004 * z
    dyn = bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(), surface_pressure=101500.0, reference_potential_temperature=θb)
    ref = bz.ExnerReferenceState(grid, surface_pressure=101500.0, potential_temperature=θb)      # the example's ρθᵣ comes from its reference state
    Hz = grid.Hz
    mask = lambda z: np.sin(np.pi * np.clip((z - 20e3) / 5e3, 0.0, None) / 2) ** 2 * (z > 20e3)
    rate = 1.0 / 333.0
    sponge = lambda target=0.0: bz.Relaxation(rate=rate, mask=mask, target=target)
    heating = lambda x, y, z: (4.24 / 3600.0) * np.exp(-((np.sqrt(x ** 2 + y ** 2) - 40e3) / 10e3) ** 2) * np.sin(np.pi * np.clip((z - 4e3) / 8e3, 0.0, 1.0)) ** 2
    model = bz.CompressibleAtmosphereModel(grid, dyn, coriolis=bz.FPlane(f=5e-5), advection=bz.WENO(order=5),
                                           forcing={"ρu": sponge(), "ρv": sponge(), "ρw": sponge(), "θ": bz.Forcing(heating),
                                                    "ρθ": sponge(np.asarray(ref.density)[Hz:Hz + Nz] * (300.0 + 0.004 * np.asarray(grid.zᶜ)))})
    col = np.asarray(ref.density)[Hz:Hz + Nz][:, None, None]
    vmax, rm = 20.0, 30e3
    vt = lambda r: vmax * (r / rm) * np.exp(0.5 * (1 - (r / rm) ** 2))
    r = lambda x, y: np.sqrt(x ** 2 + y ** 2) + 1e-9
    model.set(ρ=col, θ=lambda x, y, z: 300.0 + 0.004 * z + 0 * x + 0 * y, u=lambda x, y, z: -vt(r(x, y)) * y / r(x, y) * np.exp(-z / 8e3),
              v=lambda x, y, z: vt(r(x, y)) * x / r(x, y) * np.exp(-z / 8e3), w=0.0, qᵗ=0.0)
    th0 = model.potential_temperature_density.interior_cpu().sum()
    for _ in range(5):
        model.time_step(10.0)
    model.synchronize()
    assert _finite(model)
    assert model.potential_temperature_density.interior_cpu().sum() > th0      # the rainband heats


def test_tropical_cyclone_world_jl(bz):
    """examples/tropical_cyclone_world.jl:60-173: halo (5, 5, 5), FPlane, momentum WENO(order = 9) with rho theta WENO(order = 5) and rho q^e
    WENO(order = 5, bounds = (0, 1)), warm-phase saturation adjustment, BulkDrag on rho u / rho v, BulkSensibleHeatFlux keyed rho e,
    BulkVaporFlux on rho q^e, a Gaussian sponge on rho w (the radiative-cooling forcing of the example is a discrete-form function of T
    and stays with the extension).  Parity of the pieces: tests/test_mixed_orders.py, test_forcings.py (bulk fluxes), test_relaxation.py"""
    N, Nz, L, H = 32, 24, 96e3, 28e3
    grid = bz.RectilinearGrid((N, N, Nz), halo=(5, 5, 5), x=(0.0, L), y=(0.0, L), z=(0.0, H))
    T0 = 300.0
    ref = bz.ReferenceState(grid, surface_pressure=101325.0, potential_temperature=T0)
    CD = CT = 1.5e-3
    bcs = {"ρu": bz.FieldBoundaryConditions(bottom=bz.BulkDrag(coefficient=CD, gustiness=1.0)),
           "ρv": bz.FieldBoundaryConditions(bottom=bz.BulkDrag(coefficient=CD, gustiness=1.0)),
           "ρe": bz.FieldBoundaryConditions(bottom=bz.BulkSensibleHeatFlux(coefficient=CT, gustiness=1.0, surface_temperature=T0)),
           "ρqᵉ": bz.FieldBoundaryConditions(bottom=bz.BulkVaporFlux(coefficient=0.8 * CT, gustiness=1.0, surface_temperature=T0))}
    sponge = bz.Relaxation(rate=1 / 30, mask=bz.GaussianMask(center=26e3, width=2e3))
    model = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), coriolis=bz.FPlane(f=3e-4), momentum_advection=bz.WENO(order=9),
                               scalar_advection={"ρθ": bz.WENO(order=5), "ρqᵉ": bz.WENO(order=5, bounds=(0, 1))},
                               microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium()), forcing={"ρw": sponge},
                               boundary_conditions=bcs)
    rng = np.random.default_rng(2)
    model.set(θ=lambda x, y, z: T0 + 0.004 * z + 0.2 * rng.standard_normal((Nz, N, N)) * (z < 1e3),
              qᵗ=lambda x, y, z: 0.015 * np.exp(-z / 3e3) + 0 * x + 0 * y, u=lambda x, y, z: 3.0 * np.sin(2 * np.pi * y / L) + 0 * x + 0 * z)
    q0 = model.moisture_density.interior_cpu().sum()
    for _ in range(8):
        model.time_step(5.0)
    model.synchronize()
    assert _finite(model)
    assert model.moisture_density.interior_cpu().sum() > q0          # the sea surface moistens the lowest level
    q = model.specific_moisture.interior_cpu()
    assert q.min() > -1e-6 and q.max() < 1.0
