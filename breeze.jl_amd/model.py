"""Host-side mirror of Breeze's AtmosphereModel API for the anelastic hot path.

Julia is not available in the build image, so this module plays the role of the Julia host code
above the C ABI (include/breeze_hip.h): same names, argument meaning and call order as
  AtmosphereModel(grid; dynamics, advection, ...)   src/AtmosphereModels/atmosphere_model.jl:114-314
  set!(model; θ=..., u=..., ...)                    src/AtmosphereModels/set_atmosphere_model.jl:198-362
  update_state!(model; compute_tendencies)          src/AtmosphereModels/update_atmosphere_model_state.jl:41-68
  time_step!(model, Δt)                             src/TimeSteppers/ssp_runge_kutta_3.jl:209-278
(paths relative to /root/reference).  Julia's `f!` is spelled `f_` here.

All field memory lives on the GPU (torch tensors are only the allocator/stream plumbing); every
numerical operation of the path is a HIP kernel behind libbreeze_hip.so.  There is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _lib
from .grids import Bounded, Center, Face, Flat, Periodic, RectilinearGrid
from .thermodynamics import (ReferenceState, ThermodynamicConstants, dry_air_gas_constant,
                             vapor_gas_constant)


class Centered:
    """Centered(order = 2): the AtmosphereModel constructor's default advection in the reference."""

    def __init__(self, order=2):
        if order != 2:
            raise NotImplementedError("Centered(order = 2) is implemented")
        self.order = 2


class WENO:
    """WENO(order=5; bounds=nothing).  order = 5 is the tuned path; orders 7 and 9 (the examples' choice) run through generic
    kernels (csrc/bz_tendency_generic.hip) on single-GPU anelastic models (θ or StaticEnergy, 2-D Flat grids too) and on compressible
    models (single GPU or y-slabs, with Kessler) whose grid carries halos of at least (order + 1) / 2 cells, as Oceananigans requires.  `bounds = (lo, hi)` (order 5) makes it the
    bounds-preserving scheme (Oceananigans' BoundsPreservingWENO) that the reference's moist examples give their moisture
    densities (examples/rico.jl:184-190, examples/tropical_cyclone_world.jl:169); it is a per-scalar scheme:
    `advection = {"momentum": WENO(), "ρθ": WENO(), "ρqᵉ": WENO(bounds=(0, 1))}`."""

    def __init__(self, order=5, bounds=None, ft2_hypothesis=0):
        """ft2_hypothesis (0, 1, 2): which reading of Oceananigans' second float type FT2 = Float32 the reconstruction follows (SURVEY
        Appendix D.1; csrc/bz_weno.h: BZ_WENO_FT2) — 0 everything in the grid's type (shipped), 1 newton_div quotients, 2 weights in Float32.
        1 and 2 run from lib/libbreeze_hip_ft2_<level>.so (Float64 grids): parity instrumentation for the day reference goldens exist."""
        if order not in (5, 7, 9):
            raise NotImplementedError("WENO orders 5, 7 and 9 are implemented in the HIP path")
        if ft2_hypothesis not in (0, 1, 2):
            raise ValueError("ft2_hypothesis is 0, 1 or 2")
        self.ft2_hypothesis = ft2_hypothesis
        if order != 5 and bounds is not None:
            raise NotImplementedError("bounds-preserving WENO is implemented for order 5")
        self.order = order
        if bounds is not None:
            bounds = (float(bounds[0]), float(bounds[1]))
            if not bounds[1] > bounds[0]:
                raise ValueError("bounds must be (lower, upper) with upper > lower")
        self.bounds = bounds


_MOISTURE_KEYS = ("ρq", "ρqᵛ", "ρqᵉ", "ρqᵗ")
_SPECIES_KEYS = ("ρqᶜˡ", "ρqʳ")


def _merge_advection(advection, momentum_advection, scalar_advection):
    """atmosphere_model.jl:146-152: a single `advection` serves momentum and scalars; otherwise `momentum_advection` and `scalar_advection`
    (one scheme, or a NamedTuple keyed by scalar name) each default to Centered(order = 2).  Returns `advection` in the form
    `_split_advection` takes: one scheme, or a dict keyed by `momentum` and the scalar names (`scalars` = every scalar not named)."""
    if advection is not None:
        if momentum_advection is not None or scalar_advection is not None:
            raise ValueError("pass either `advection` or `momentum_advection` / `scalar_advection`")
        return advection
    if momentum_advection is None and scalar_advection is None:
        return None
    momentum_advection = momentum_advection or Centered(order=2)
    scalar_advection = scalar_advection or Centered(order=2)
    merged = {"momentum": momentum_advection}
    if isinstance(scalar_advection, dict):
        merged.update({str(k).lstrip(":"): v for k, v in scalar_advection.items()})
        merged.setdefault("scalars", Centered(order=2))      # validate_tracer_advection: scalars a NamedTuple does not name take the default
    else:
        merged["scalars"] = scalar_advection
    return merged


def _split_advection(advection, tracer_names):
    """The reference accepts one scheme or a NamedTuple of schemes keyed by `momentum` and the scalar names
    (atmosphere_model.jl advection keyword; examples/rico.jl:186-190).  Returns (momentum scheme, bounds-preserving request or None,
    order of the scalars' schemes, whether scalars the NamedTuple does not name would take a scheme of another order)."""
    if not isinstance(advection, dict):
        if getattr(advection, "bounds", None) is not None:
            raise NotImplementedError("bounds-preserving WENO is a scalar scheme: pass advection = {'momentum': WENO(), ..., 'ρqᵉ': WENO(bounds=(0, 1))}")
        return advection, None, advection.order, False
    advection = {str(k).lstrip(":"): v for k, v in advection.items()}
    base = advection.get("momentum") or next(iter(advection.values()))
    named = {k: v for k, v in advection.items() if k not in ("momentum", "scalars")}
    rest = advection.get("scalars")          # scheme of the scalars not named (_merge_advection); absent: the momentum scheme's order
    orders = {getattr(v, "order", None) for v in named.values()}
    if len(orders) > 1:
        raise NotImplementedError("every scalar must be advected with a scheme of one order")
    scalar_order = orders.pop() if orders else (rest.order if rest is not None else base.order)
    unnamed_differ = bool(named) and rest is not None and rest.order != scalar_order
    if (base.order == 2) != (scalar_order == 2):
        raise NotImplementedError("Centered(order = 2) for momentum with WENO scalars (or the reverse) is not implemented")
    bounded, lo_hi = {"moisture": 0, "microphysical_species": 0, "tracers": 0}, None
    for key, scheme in named.items():
        b = getattr(scheme, "bounds", None)
        if b is None:
            continue
        if key in ("ρθ", "ρe"):
            raise NotImplementedError(f"bounds-preserving advection of {key} is not implemented (moisture, microphysical species, tracers)")
        if lo_hi is not None and b != lo_hi:
            raise NotImplementedError("one pair of bounds for all bounds-preserving scalars")
        lo_hi = b
        if key in _MOISTURE_KEYS:
            bounded["moisture"] = 1
        elif key in _SPECIES_KEYS:
            bounded["microphysical_species"] = 1
        elif key.lstrip("ρ") in tracer_names or key in tracer_names:
            bounded["tracers"] = 1
        else:
            raise ValueError(f"advection key {key!r} names no prognostic scalar of this model")
    if getattr(base, "bounds", None) is not None or getattr(rest, "bounds", None) is not None:
        raise NotImplementedError("bounds-preserving WENO is given per scalar: scalar_advection = {'ρθ': WENO(), 'ρqᵉ': WENO(bounds=(0, 1))}")
    if lo_hi is None:
        return base, None, scalar_order, unnamed_differ
    return base, dict(bounded, lower=lo_hi[0], upper=lo_hi[1]), scalar_order, unnamed_differ


class AnelasticDynamics:
    """AnelasticDynamics(reference_state)  (src/AnelasticEquations/anelastic_dynamics.jl:5-8)."""

    def __init__(self, reference_state):
        self.reference_state = reference_state
        self.pressure_anomaly = None   # materialised by AtmosphereModel


class Clock:
    def __init__(self):
        self.time, self.iteration, self.last_Δt = 0.0, 0, float("inf")


class Field:
    """A device-resident Oceananigans-style field: `.parent` is the halo-inclusive array
    ((z, y, x)-shaped, x fastest), `.interior` a view of the interior."""

    def __init__(self, grid, loc, device, float_type=None):
        import torch
        self.grid, self.loc = grid, loc
        self.zface = loc[2] is Face
        # float_type: CenterField(grid, FT) of another element type than eltype(grid) (the substepper's working fields)
        ft = getattr(grid, "ftype", 8) if float_type is None else np.dtype(float_type).itemsize
        self.dtype = torch.float32 if ft == 4 else torch.float64
        self.parent = torch.zeros(grid.parent_shape(self.zface), dtype=self.dtype, device=device)

    @property
    def interior(self):
        return self.parent[self.grid.interior_slices(self.zface)]

    def ptr(self):
        return self.parent.data_ptr()

    def set_interior(self, value):
        import torch
        g = self.grid
        if callable(value):
            x, y, z = g.nodes(self.loc)
            import inspect
            # functions of the non-Flat coordinates, as in Oceananigans: f(x, z) on a (Periodic, Flat, Bounded) grid
            two = g.topology[1] == "Flat" and len(inspect.signature(value).parameters) == 2
            value = value(x, z) if two else value(x, y, z)
        shape = tuple(self.interior.shape)
        arr = np.broadcast_to(np.asarray(value, dtype=np.float64), shape)
        self.interior.copy_(torch.from_numpy(np.array(arr, dtype=np.float64, order="C")).to(self.dtype))

    # diagnostic fields of a model (u, v, w, θ, qᵛ, T, pressure anomaly) know their owner: after time_steps(..., diagnose_last=False)
    # they are older than the prognostic state, and a host read rebuilds them first (update_state!) instead of returning stale values
    _owner = None

    def _fresh(self):
        m = self._owner() if self._owner is not None else None
        if m is not None and getattr(m, "_ctx", None) and diagnostics_stale(m):
            if getattr(m, "_collective_refresh", False):
                # slab models: rebuilding the diagnostics exchanges halos — a collective a single rank's read must not enter on its own
                # (a rank-0-only log line would wait for ranks that never come)
                raise RuntimeError("the diagnostic fields of this slab model are older than its prognostic state (time_steps(..., diagnose_last=False)): "
                                   "call model.refresh_diagnostics() on EVERY rank before reading them")
            m._refresh_diagnostics()

    def cpu(self):
        self._fresh()
        return self.parent.cpu().numpy()

    def interior_cpu(self):
        self._fresh()
        return self.interior.cpu().numpy()


_LOC = {"ccc": (Center, Center, Center), "fcc": (Face, Center, Center),
        "cfc": (Center, Face, Center), "ccf": (Center, Center, Face)}

# keyword spellings accepted by set_ (Julia names and ASCII transliterations)
_ALIASES = {"θ": "θ", "theta": "θ", "θˡⁱ": "θ", "ρθ": "ρθ", "rho_theta": "ρθ", "e": "e", "ρe": "ρe", "rho_e": "ρe",
            "u": "u", "v": "v", "w": "w", "ρu": "ρu", "ρv": "ρv", "ρw": "ρw",
            "rho_u": "ρu", "rho_v": "ρv", "rho_w": "ρw",
            "qᵗ": "q", "qt": "q", "qᵛ": "q", "qv": "q", "qᵉ": "q", "qe": "q", "ρqᵗ": "ρq", "ρqᵛ": "ρq", "ρqᵉ": "ρq", "ρqe": "ρq",
            "rho_q": "ρq", "qᶜˡ": "qcl", "qcl": "qcl", "qʳ": "qr", "qr": "qr", "T": "T", "ℋ": "ℋ", "H": "ℋ", "relative_humidity": "ℋ",      # Python NFKC-normalises the keyword ℋ to H
            # NFKC-normalised spellings (Python normalises identifiers used as keywords)
            "θli": "θ", "ρqt": "ρq", "ρqv": "ρq"}


class AtmosphereModel:
    """AtmosphereModel(grid; dynamics=AnelasticDynamics(ReferenceState(grid)), advection=WENO(order=5),
    formulation=:LiquidIcePotentialTemperature, thermodynamic_constants, timestepper=:SSPRungeKutta3).

    Dry anelastic dynamics without closure / Coriolis / forcing / microphysics — the configuration
    of BASELINE.json configs 1-2 and 4."""

    def __init__(self, grid, dynamics=None, advection=None, thermodynamic_constants=None,
                 formulation="LiquidIcePotentialTemperature", timestepper="SSPRungeKutta3",
                 closure=None, coriolis=None, microphysics=None, forcing=None, boundary_conditions=None,
                 tracers=(), device="cuda:0", momentum_advection=None, scalar_advection=None):
        import torch
        if not isinstance(grid, RectilinearGrid):
            raise TypeError("grid must be a RectilinearGrid")
        advection = _merge_advection(advection, momentum_advection, scalar_advection)
        bounded_x = grid.topology == (Bounded, Flat, Bounded)          # walls in x of a 2-D model: examples/cloudy_thermal_bubble.jl
        flat_y = grid.topology == (Periodic, Flat, Bounded) or bounded_x
        bounded_y = grid.topology == (Periodic, Bounded, Bounded)      # walls in y: the reference benchmark driver's PBB option
        if grid.topology != (Periodic, Periodic, Bounded) and not flat_y and not bounded_y:
            raise NotImplementedError("the HIP path implements topology (Periodic, Periodic, Bounded), (Periodic, Flat, Bounded), "
                                      "(Bounded, Flat, Bounded) and (Periodic, Bounded, Bounded)")
        if advection is None:
            advection = Centered(order=2)          # the reference's default (resolved before the Flat guard: ADVICE r02)
        _base = advection.get("momentum") or next(iter(advection.values())) if isinstance(advection, dict) else advection
        if flat_y and (not isinstance(_base, WENO) or _base.order not in (5, 7, 9)):
            # the reference's 2-D x-z cases (README.md:67-75, examples/dry_thermal_bubble.jl with WENO(order = 9)): the per-operator
            # kernels drop the y terms
            raise NotImplementedError("(Periodic, Flat, Bounded): WENO(order = 5 | 7 | 9) models are implemented")
        formulation = str(formulation).lstrip(":")
        _any_bounds = any(getattr(s, "bounds", None) is not None for s in (advection.values() if isinstance(advection, dict) else (advection,)))
        if bounded_x and (_any_bounds or grid.Nx % 2):
            raise NotImplementedError("(Bounded, Flat, Bounded): WENO(order = 5 | 7 | 9) models without bounds-preserving advection on an even "
                                      "number of columns are implemented")
        if bounded_y and (not isinstance(_base, WENO) or _base.order not in (5, 7, 9) or _any_bounds):
            # Coriolis, forcings, bottom flux boundary conditions and the closure reach their y neighbours through the halo rows (as on
            # y-slabs); microphysics, tracers and the StaticEnergy formulation are column- or cell-local
            raise NotImplementedError("(Periodic, Bounded, Bounded): WENO(order = 5 | 7 | 9) models without bounds-preserving advection "
                                      "are implemented")
        if formulation not in ("LiquidIcePotentialTemperature", "StaticEnergy"):
            raise NotImplementedError(f"formulation {formulation!r} is not implemented")
        self.formulation = formulation
        if timestepper not in ("SSPRungeKutta3", ":SSPRungeKutta3"):
            raise NotImplementedError("only SSPRungeKutta3 is implemented")
        from .forcings import SmagorinskyLilly
        if closure is not None and not isinstance(closure, SmagorinskyLilly):
            raise NotImplementedError("closure: SmagorinskyLilly() is implemented")
        self.closure = closure
        self.coriolis, self.forcing, self.boundary_conditions = coriolis, forcing, boundary_conditions
        from .microphysics import DCMIP2016KesslerMicrophysics, SaturationAdjustment, TetensFormula
        if microphysics is not None and not isinstance(microphysics, (SaturationAdjustment, DCMIP2016KesslerMicrophysics)):
            raise NotImplementedError("microphysics: SaturationAdjustment(equilibrium = WarmPhaseEquilibrium()) and "
                                      "DCMIP2016KesslerMicrophysics() are implemented")
        if microphysics is not None and formulation != "LiquidIcePotentialTemperature":
            raise NotImplementedError("microphysics is implemented for the potential-temperature formulation")
        self.microphysics = microphysics
        self._kessler = isinstance(microphysics, DCMIP2016KesslerMicrophysics)
        if self._kessler:      # validate_microphysics (dcmip2016_kessler.jl:196-207)
            tcs = thermodynamic_constants
            if tcs is None or not isinstance(getattr(tcs, "saturation_vapor_pressure", None), TetensFormula):
                raise ValueError("DCMIP2016KesslerMicrophysics requires `thermodynamic_constants` with a `TetensFormula` "
                                 "saturation vapor pressure formulation. Construct the model with, e.g., "
                                 "`thermodynamic_constants = ThermodynamicConstants(saturation_vapor_pressure = TetensFormula())`.")
        _named = set(str(k).lstrip(":") for k in advection) if isinstance(advection, dict) else set()
        _tracer_names = tuple(str(n).lstrip(":") for n in ((tracers,) if isinstance(tracers, str) else tracers))
        advection, self._bounded_advection, self._scalar_order, _unnamed_differ = _split_advection(advection, _tracer_names)
        if _unnamed_differ:
            # scalar_advection was a NamedTuple: every scalar of the model it does not name takes Centered(order = 2) in the reference
            missing = {"ρe" if formulation == "StaticEnergy" else "ρθ"} - _named
            if not _named & set(_MOISTURE_KEYS):
                missing.add("moisture")
            if self._kessler:
                missing |= set(_SPECIES_KEYS) - _named
            missing |= {t for t in _tracer_names if t not in _named and "ρ" + t not in _named}
            if missing:
                raise NotImplementedError("scalar_advection names only some scalars: the others would take Centered(order = 2), and one order "
                                          f"for all scalars is implemented (unnamed: {sorted(missing)})")
        if not torch.cuda.is_available():
            raise RuntimeError("AtmosphereModel needs a GPU: the HIP path has no CPU fallback")
        self.grid = grid
        self.advection = advection
        if isinstance(advection, WENO) and max(advection.order, self._scalar_order) != 5:
            order = max(advection.order, self._scalar_order)
            need = (order + 1) // 2
            if min(h for h, t in zip((grid.Hx, grid.Hy, grid.Hz), grid.topology) if t != Flat) < need:
                raise ValueError(f"WENO(order={order}) needs halos of at least {need} cells in every direction "
                                 f"(got {(grid.Hx, grid.Hy, grid.Hz)}): RectilinearGrid(..., halo=({need}, {need}, {need}))")
        if self._bounded_advection is not None and self._scalar_order != 5:
            raise NotImplementedError(f"WENO(order={self._scalar_order}): bounds-preserving advection is implemented for order 5")
        self.thermodynamic_constants = c = thermodynamic_constants or ThermodynamicConstants()
        if dynamics is None:
            dynamics = AnelasticDynamics(ReferenceState(grid, c))      # default_dynamics
        self.dynamics = dynamics
        self.clock = Clock()
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self._T = T = _lib.types(grid.ftype)
        if grid.ftype == 4:
            if not isinstance(advection, WENO):
                raise NotImplementedError("Float32 grids: WENO(order = 5 | 7 | 9) is wired up on the host side")
            self._lib = lib = _lib.load_f32()
        else:
            self._lib = lib = _lib.load(advection.order, getattr(advection, "ft2_hypothesis", 0))

        def fld(loc):
            return Field(grid, _LOC[loc], self.device)

        # materialize_momentum_and_velocities, formulation, moisture, diagnostics
        self.momentum = {"ρu": fld("fcc"), "ρv": fld("cfc"), "ρw": fld("ccf")}
        self.velocities = {"u": fld("fcc"), "v": fld("cfc"), "w": fld("ccf")}
        # :StaticEnergy keeps rho_e / e in the thermodynamic slots (energy_density, specific_energy)
        self.potential_temperature_density = fld("ccc")
        self.potential_temperature = fld("ccc")
        self.energy_density, self.specific_energy = self.potential_temperature_density, self.potential_temperature
        self.moisture_density = fld("ccc")
        self.specific_moisture = fld("ccc")
        self.temperature = fld("ccc")
        dynamics.pressure_anomaly = fld("ccc")
        import weakref
        for _f in (*self.velocities.values(), self.potential_temperature, self.specific_moisture, self.temperature, dynamics.pressure_anomaly):
            _f._owner = weakref.ref(self)
        self.microphysical_fields = {}
        if self._kessler:      # materialize_microphysical_fields(::DCMIP2016KM) (dcmip2016_kessler.jl:255-290)
            self.microphysical_fields = {k: fld("ccc") for k in ("ρqᶜˡ", "ρqʳ", "qᵛ", "qᶜˡ", "qʳ", "𝕎ʳ")}
            self.microphysical_fields["precipitation_rate"] = torch.zeros((grid.Ny + 2 * grid.Hy, grid.Nx + 2 * grid.Hx),
                                                                          dtype=self.momentum["ρu"].parent.dtype, device=self.device)
        # tracers = (:a, :b): prognostic density fields model.tracers[name]; the specific field sits beside it
        tracers = (tracers,) if isinstance(tracers, str) else tuple(tracers)
        self.tracers = {str(n).lstrip(":"): fld("ccc") for n in tracers}
        self.specific_tracers = {n: fld("ccc") for n in self.tracers}
        prog = self.prognostic_fields()
        self.U0 = {k: Field(grid, f.loc, self.device) for k, f in prog.items()}     # timestepper.U⁰
        self.G = {k: Field(grid, f.loc, self.device) for k, f in prog.items()}      # timestepper.Gⁿ

        # ---- context: pressure solver, column tables ----
        ref = dynamics.reference_state
        self._zf = np.ascontiguousarray(grid.zᶠ, dtype=T.np_real)
        bg = T.bz_grid()
        bg.Nx, bg.Ny, bg.Nz = grid.Nx, grid.Ny, grid.Nz
        bg.Hx, bg.Hy, bg.Hz = grid.Hx, grid.Hy, grid.Hz
        for d, t in enumerate(grid.topology_codes()):
            bg.topo[d] = t
        bg.ftype = grid.ftype
        bg.dx, bg.dy = grid.Δx, grid.Δy
        bg.zf = self._zf.ctypes.data_as(C.POINTER(T.real))
        bg.regular_z = 1 if grid.regular_z else 0
        bc = T.bz_constants(c.gravitational_acceleration, dry_air_gas_constant(c), vapor_gas_constant(c),
                            c.dry_air_heat_capacity, c.vapor_heat_capacity)
        self._ref_arrays = [np.ascontiguousarray(a, dtype=T.np_real)
                            for a in (ref.density, ref.pressure, ref.temperature)]
        br = T.bz_reference_state(ref.surface_pressure, ref.potential_temperature, ref.standard_pressure,
                                  *[a.ctypes.data_as(C.POINTER(T.real)) for a in self._ref_arrays])
        self._ctx = C.c_void_p()
        rc = self._create_context(lib, bg, bc, br, advection.order)
        if rc != 0:
            raise _lib.BreezeHIPError(f"bz_create failed with code {rc}")
        self._check(lib.bz_set_stream(self._ctx, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                    "bz_set_stream")
        if formulation == "StaticEnergy":
            self._check(lib.bz_set_formulation(self._ctx, 1), "bz_set_formulation")
        if self._scalar_order != advection.order:      # momentum_advection and scalar_advection of different orders
            self._check(lib.bz_set_scalar_advection_order(self._ctx, self._scalar_order), "bz_set_scalar_advection_order")
        # materialize_microphysical_fields(::WarmPhaseSaturationAdjustment): (q^v, q^l, q^e); q^e is the specific moisture slot
        if self._kessler:
            from .microphysics import kessler_parameter_struct
            μ = self.microphysical_fields
            P = kessler_parameter_struct(microphysics, c, ftype=grid.ftype)
            K = _lib.bz_kessler_model_fields()
            K.cloud_liquid_density, K.rain_density = μ["ρqᶜˡ"].ptr(), μ["ρqʳ"].ptr()
            K.U0_cloud_liquid_density, K.U0_rain_density = self.U0["ρqᶜˡ"].ptr(), self.U0["ρqʳ"].ptr()
            K.G_cloud_liquid_density, K.G_rain_density = self.G["ρqᶜˡ"].ptr(), self.G["ρqʳ"].ptr()
            K.vapor_mass_fraction, K.cloud_liquid_mass_fraction, K.rain_mass_fraction = μ["qᵛ"].ptr(), μ["qᶜˡ"].ptr(), μ["qʳ"].ptr()
            K.rain_terminal_velocity, K.precipitation_rate = μ["𝕎ʳ"].ptr(), μ["precipitation_rate"].data_ptr()
            self._check(lib.bz_set_kessler_microphysics(self._ctx, C.byref(P), C.byref(K), ref.standard_pressure),
                        "bz_set_kessler_microphysics")
        elif microphysics is not None:
            self.microphysical_fields = {"qᵛ": fld("ccc"), "qˡ": fld("ccc"), "qᵉ": self.specific_moisture}
            sa = T.bz_saturation_adjustment(c.liquid_reference_latent_heat, c.liquid_heat_capacity,
                                               c.energy_reference_temperature, c.triple_point_temperature,
                                               c.triple_point_pressure, microphysics.solver.abstol,
                                               microphysics.solver.maxiter, 0)
            self._check(lib.bz_set_saturation_adjustment(self._ctx, C.byref(sa),
                                                         C.c_void_p(self.microphysical_fields["qᵛ"].ptr()),
                                                         C.c_void_p(self.microphysical_fields["qˡ"].ptr())),
                        "bz_set_saturation_adjustment")
        if self.tracers:
            arr = (_lib.bz_tracer_fields * len(self.tracers))()
            for t, n in enumerate(self.tracers):
                arr[t].density, arr[t].specific = self.tracers[n].ptr(), self.specific_tracers[n].ptr()
                arr[t].U0, arr[t].G = self.U0[n].ptr(), self.G[n].ptr()
            self._check(lib.bz_set_tracers(self._ctx, len(self.tracers), arr), "bz_set_tracers")
        self.closure_fields = {}
        if self._bounded_advection is not None:
            ba = self._bounded_advection
            if ba["tracers"] and ba["tracers"] != 0 and len(self.tracers) == 0:
                raise ValueError("bounds-preserving tracer advection without tracers")
            bs = T.bz_bounds_preserving_advection(ba["lower"], ba["upper"], ba["moisture"], ba["microphysical_species"], ba["tracers"], 0)
            self._check(lib.bz_set_bounds_preserving_advection(self._ctx, C.byref(bs)), "bz_set_bounds_preserving_advection")
        if closure is not None:      # build_closure_fields: nu_e (atmosphere_model.jl:276)
            if self._kessler or formulation != "LiquidIcePotentialTemperature":
                raise NotImplementedError("SmagorinskyLilly is implemented for the potential-temperature formulation without Kessler")
            self.closure_fields = {"νₑ": fld("ccc")}
            cl = T.bz_smagorinsky_lilly(closure.C, closure.Cb, closure.Pr)
            self._check(lib.bz_set_closure(self._ctx, C.byref(cl), C.c_void_p(self.closure_fields["νₑ"].ptr())), "bz_set_closure")
        # coriolis / forcing / boundary_conditions of the BOMEX configuration -> one column-forcing stack (forcings.py)
        from .forcings import materialize_forcings, materialize_relaxation, split_relaxation
        forcing, _relax, _field = split_relaxation(forcing)      # Relaxation sponges and 3-D forcings: their own attachments
        Rx, self._relaxation_keepalive = materialize_relaxation(grid, _relax, formulation, T)
        if Rx is not None:
            self._check(lib.bz_set_relaxation(self._ctx, C.byref(Rx)), "bz_set_relaxation")
        from .forcings import materialize_field_forcing
        # Forcing(f(x, y, z)) on θ / e (or ρθ / ρe): a centre field the library reads at every tendency evaluation; refresh it with
        # model.thermodynamic_forcing_field.set_interior(...) when the forcing depends on time
        self.thermodynamic_forcing_field, _spec = materialize_field_forcing(grid, _field, formulation, self.device)
        if self.thermodynamic_forcing_field is not None:
            self._check(lib.bz_set_field_forcing(self._ctx, C.c_void_p(self.thermodynamic_forcing_field.ptr()), _spec), "bz_set_field_forcing")
        F, self._forcing_keepalive = materialize_forcings(grid, coriolis, forcing, boundary_conditions, T)
        if F is not None:
            if self._kessler or formulation != "LiquidIcePotentialTemperature":
                raise NotImplementedError("forcings are implemented for the potential-temperature formulation without Kessler")
            self._check(lib.bz_set_forcings(self._ctx, C.byref(F)), "bz_set_forcings")
        from .forcings import materialize_bulk_fluxes
        Bk = materialize_bulk_fluxes(boundary_conditions, ref, c, T)
        if Bk is not None:
            if self._kessler or formulation != "LiquidIcePotentialTemperature":
                raise NotImplementedError("bulk surface fluxes are implemented for the potential-temperature formulation without Kessler")
            self._check(lib.bz_set_bulk_surface_fluxes(self._ctx, C.byref(Bk)), "bz_set_bulk_surface_fluxes")
        self._state = self._make_state()
        self._U0 = self._make_prog(self.U0)
        self._G = self._make_prog(self.G)
        # initialize_model_thermodynamics!: θ = θ₀  (anelastic_time_stepping.jl:15-19)
        set_(self, θ=ref.potential_temperature)

    # -- plumbing ------------------------------------------------------------
    def _check(self, rc, what):
        _lib.check(self._lib, self._ctx, rc, what)

    def prognostic_fields(self):
        out = {"ρu": self.momentum["ρu"], "ρv": self.momentum["ρv"], "ρw": self.momentum["ρw"],
               "ρθ": self.potential_temperature_density, "ρq": self.moisture_density}
        if getattr(self, "_kessler", False):
            out["ρqᶜˡ"], out["ρqʳ"] = self.microphysical_fields["ρqᶜˡ"], self.microphysical_fields["ρqʳ"]
        out.update(getattr(self, "tracers", {}))
        return out

    def _make_state(self):
        s = _lib.bz_state()
        s.rho_u, s.rho_v, s.rho_w = (self.momentum[k].ptr() for k in ("ρu", "ρv", "ρw"))
        s.rho_theta, s.rho_q = self.potential_temperature_density.ptr(), self.moisture_density.ptr()
        s.u, s.v, s.w = (self.velocities[k].ptr() for k in ("u", "v", "w"))
        s.theta, s.q, s.T = self.potential_temperature.ptr(), self.specific_moisture.ptr(), self.temperature.ptr()
        s.phi = self.dynamics.pressure_anomaly.ptr()
        return s

    @staticmethod
    def _make_prog(d):
        p = _lib.bz_prognostic()
        p.rho_u, p.rho_v, p.rho_w, p.rho_theta, p.rho_q = (d[k].ptr() for k in ("ρu", "ρv", "ρw", "ρθ", "ρq"))
        return p

    def __del__(self):
        try:
            if getattr(self, "_ctx", None):
                self._lib.bz_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    # method spellings of the module-level functions
    def _create_context(self, lib, bg, bc, br, order):
        return lib.bz_create(C.byref(self._ctx), C.byref(bg), C.byref(bc), C.byref(br), order)

    def set(self, **kw):
        return set_(self, **kw)

    def time_step(self, Δt):
        return time_step_(self, Δt)

    def time_steps(self, Δt, n, diagnose_last=True):
        """n × time_step!(model, Δt) in one C call (bz_time_steps_anelastic): the reference's many_time_steps! loop
        (/root/reference/benchmarking/src/timestepping.jl:11-16) and run!'s stretch between two callback iterations.  With
        diagnose_last=False the velocities / θ / qᵛ / T / pressure anomaly stay stale until update_state_(model) (stepping may go on)."""
        return many_time_steps_(self, Δt, n, diagnose_last)

    def synchronize(self):
        self._check(self._lib.bz_sync(self._ctx), "bz_sync")

    def _refresh_diagnostics(self):
        """update_state!(model; compute_tendencies=false) after undiagnosed steps (slab models override it with the exchanging form)."""
        update_state_(self, compute_tendencies=False)

    def refresh_diagnostics(self):
        """Rebuild u, v, w, θ, qᵛ, T from the prognostic state if they are stale.  On slab models a collective: call it on every rank."""
        if diagnostics_stale(self):
            self._refresh_diagnostics()

    # -- profiling -----------------------------------------------------------
    def graph_enable(self, on=True):
        """hipGraph replay of whole steps (csrc/bz_graph.hip); opt-in, bit-identical to launched steps."""
        self._check(self._lib.bz_graph_enable(self._ctx, 1 if on else 0), "bz_graph_enable")

    def graph_info(self):
        """(enabled, steps recorded, steps replayed)"""
        en, cap, rep = C.c_int32(), C.c_int64(), C.c_int64()
        self._check(self._lib.bz_graph_info(self._ctx, C.byref(en), C.byref(cap), C.byref(rep)), "bz_graph_info")
        return bool(en.value), cap.value, rep.value

    def profile_enable(self, on=True):
        self._check(self._lib.bz_profile_enable(self._ctx, 1 if on else 0), "bz_profile_enable")

    def profile_reset(self):
        self._check(self._lib.bz_profile_reset(self._ctx), "bz_profile_reset")

    def profile(self):
        """{kernel group: (total_ms, launches)} accumulated since the last reset."""
        out = {}
        for i in range(self._lib.bz_profile_count(self._ctx)):
            name, ms, n = C.c_char_p(), self._T.real(), C.c_int64()
            self._check(self._lib.bz_profile_get(self._ctx, i, C.byref(name), C.byref(ms), C.byref(n)),
                        "bz_profile_get")
            out[name.value.decode()] = (ms.value, n.value)
        return out

    def max_abs_divergence(self):
        out = self._T.real()
        self._check(self._lib.bz_max_abs_divergence(self._ctx, C.byref(self._state), C.byref(out)),
                    "bz_max_abs_divergence")
        return out.value


# ---------------------------------------------------------------------------
# Julia-style generic functions on the model (f! -> f_)
# ---------------------------------------------------------------------------
def fill_halo_regions_(model, field, kind=None):
    if kind is None:
        kind = (1 if field.zface else 0) + (4 if field.loc[1] is Face else 0) + (8 if field.loc[0] is Face else 0)      # + 4 / + 8: y- / x-face field
    model._check(model._lib.bz_fill_halo_regions(model._ctx, C.c_void_p(field.ptr()), kind), "bz_fill_halo_regions")


def update_state_(model, compute_tendencies=True):
    model._check(model._lib.bz_update_state(model._ctx, C.byref(model._state), C.byref(model._G),
                                            1 if compute_tendencies else 0), "bz_update_state")


def compute_tendencies_(model):
    model._check(model._lib.bz_compute_tendencies(model._ctx, C.byref(model._state), C.byref(model._G)),
                 "bz_compute_tendencies")


def compute_scalar_tendency_(model, c, Gc):
    """Gc = -div_rhoUc(c) for one centre field c (halo-filled) with the model's velocities: compute_scalar_tendency!
    (update_atmosphere_model_state.jl:390-393), the launch the reference's scalar_tendency micro-benchmark times."""
    v = model.velocities
    model._check(model._lib.bz_compute_scalar_tendency(model._ctx, v["u"].ptr(), v["v"].ptr(), v["w"].ptr(), c.ptr(), Gc.ptr()),
                 "bz_compute_scalar_tendency")


def compute_closure_fields_(model):
    """compute_closure_fields!(model.closure_fields, model.closure, model) (update_atmosphere_model_state.jl:218)."""
    model._check(model._lib.bz_compute_closure_fields(model._ctx, C.byref(model._state)), "bz_compute_closure_fields")


def compute_flux_bc_tendencies_(model):
    """compute_flux_bc_tendencies!(model) (update_atmosphere_model_state.jl:418-434)."""
    model._check(model._lib.bz_compute_flux_bc_tendencies(model._ctx, C.byref(model._state), C.byref(model._G)),
                 "bz_compute_flux_bc_tendencies")


def compute_velocities_(model):
    model._check(model._lib.bz_compute_velocities(model._ctx, C.byref(model._state)), "bz_compute_velocities")


def compute_auxiliary_thermodynamic_variables_(model):
    model._check(model._lib.bz_compute_auxiliary_thermodynamic_variables(model._ctx, C.byref(model._state)),
                 "bz_compute_auxiliary_thermodynamic_variables")


def compute_pressure_correction_(model, Δt):
    model._check(model._lib.bz_compute_pressure_correction(model._ctx, C.byref(model._state), float(Δt)),
                 "bz_compute_pressure_correction")


def make_pressure_correction_(model, Δt):
    model._check(model._lib.bz_make_pressure_correction(model._ctx, C.byref(model._state), float(Δt)),
                 "bz_make_pressure_correction")


def store_initial_state_(model):
    model._check(model._lib.bz_store_initial_state(model._ctx, C.byref(model._state), C.byref(model._U0)),
                 "bz_store_initial_state")


def ssp_rk3_substep_(model, Δt, α):
    model._check(model._lib.bz_ssp_rk3_substep(model._ctx, C.byref(model._state), C.byref(model._U0),
                                               C.byref(model._G), float(Δt), float(α)), "bz_ssp_rk3_substep")


def enforce_mass_conservation_(model):
    """set_atmosphere_model.jl:121-128: one projection with Δt = 1."""
    compute_pressure_correction_(model, 1.0)
    make_pressure_correction_(model, 1.0)
    update_state_(model, compute_tendencies=False)


def set_(model, enforce_mass_conservation=True, **kw):
    """set!(model; kw...): θ / ρθ, u v w / ρu ρv ρw, qᵗ / ρqᵗ from numbers, arrays or f(x, y, z)."""
    import torch
    g = model.grid
    ref = model.dynamics.reference_state
    Hz, Nz = g.Hz, g.Nz
    ρc = torch.from_numpy(ref.density[Hz:Hz + Nz].copy()).to(model.device)[:, None, None]
    ρf_host = 0.5 * (ref.density[Hz - 1:Hz + Nz] + ref.density[Hz:Hz + Nz + 1])
    ρf = torch.from_numpy(ρf_host).to(model.device)[:, None, None]
    order = {"q": 0, "ρq": 0, "ℋ": 2}        # moisture first; ℋ after everything else (it needs the diagnosed temperature)
    for name, value in sorted(kw.items(), key=lambda kv: order.get(_ALIASES.get(kv[0]), 1)):
        key = _ALIASES.get(name)
        if key in ("e", "ρe") and model.formulation != "StaticEnergy":
            key = None
        if key in ("T", "ℋ") and (model.formulation != "LiquidIcePotentialTemperature" or getattr(model, "_kessler", False)):
            raise NotImplementedError(f"set!(model; {name}) is implemented for the potential-temperature formulation "
                                      "(microphysics nothing or SaturationAdjustment)")
        if key == "T":
            # set_thermodynamic_variable!(model, Val(:T), value) (potential_temperature_tendency.jl:202-250): theta^li from the
            # in-situ temperature with the current moisture fractions, theta = (T - L q^l / c_pm) / Pi
            c = model.thermodynamic_constants
            Rd, Rv = dry_air_gas_constant(c), vapor_gas_constant(c)
            model.temperature.set_interior(value)
            if model.microphysics is not None:
                qv, ql = model.microphysical_fields["qᵛ"].interior, model.microphysical_fields["qˡ"].interior
                cl, Ll = c.liquid_heat_capacity, c.liquid_reference_latent_heat
            else:
                qv, ql, cl, Ll = model.specific_moisture.interior, 0.0, 0.0, 0.0
            qd = 1.0 - (qv + ql)
            Rm, cpm = qd * Rd + qv * Rv, qd * c.dry_air_heat_capacity + qv * c.vapor_heat_capacity + ql * cl
            pr = torch.from_numpy(ref.pressure[Hz:Hz + Nz].copy()).to(model.device)[:, None, None]
            Π = (pr / ref.standard_pressure) ** (Rm / cpm)
            model.potential_temperature.interior.copy_((model.temperature.interior - Ll * ql / cpm) / Π)
            model.potential_temperature_density.interior.copy_(ρc * model.potential_temperature.interior)
            continue
        if key == "ℋ":
            # set!(model; ℋ) (set_atmosphere_model.jl:280-297): update_state!, q^v+ = SaturationSpecificHumidity(model, :equilibrium)
            # (vapor_saturation.jl:216-230) materialised before the moisture is overwritten, q = ℋ q^v+, rho q = rho_r q
            update_state_(model, compute_tendencies=False)
            c = model.thermodynamic_constants
            Rd, Rv = dry_air_gas_constant(c), vapor_gas_constant(c)
            T, qt = model.temperature.interior, model.specific_moisture.interior
            pr = torch.from_numpy(ref.pressure[Hz:Hz + Nz].copy()).to(model.device)[:, None, None]
            dc = c.vapor_heat_capacity - c.liquid_heat_capacity
            L0 = c.liquid_reference_latent_heat - dc * c.energy_reference_temperature
            ps = c.triple_point_pressure * (T / c.triple_point_temperature) ** (dc / Rv) * \
                torch.exp((1.0 / c.triple_point_temperature - 1.0 / T) * L0 / Rv)
            q1 = (Rd / Rv) * (1.0 - qt) * ps / (pr - ps)
            ρ = pr / ((Rd * (1.0 - qt) + Rv * qt) * T)
            q0 = ps / (ρ * Rv * T)
            qsat = torch.where(qt >= q0, q1, q0)
            model.specific_moisture.set_interior(value)
            model.specific_moisture.interior.mul_(qsat)
            model.moisture_density.interior.copy_(ρc * model.specific_moisture.interior)
            continue
        if key is None:
            raise ValueError(f"Cannot set! {name} in AtmosphereModel because {name} is neither a prognostic "
                             "variable, a settable thermodynamic variable, nor a settable diagnostic variable!")
        if key == "θ" and model.formulation == "StaticEnergy":
            # _energy_density_from_potential_temperature! (static_energy_tendency.jl:93-140): host set-up arithmetic
            c = model.thermodynamic_constants
            Rd, Rv = dry_air_gas_constant(c), vapor_gas_constant(c)
            scratch = model.temperature
            scratch.set_interior(value)
            q = model.specific_moisture.interior
            qd = 1.0 - q
            Rm, cpm = qd * Rd + q * Rv, qd * c.dry_air_heat_capacity + q * c.vapor_heat_capacity
            pr = torch.from_numpy(ref.pressure[Hz:Hz + Nz].copy()).to(model.device)[:, None, None]
            z = torch.from_numpy(np.ascontiguousarray(g.zᶜ)).to(model.device)[:, None, None]
            T = (pr / ref.standard_pressure) ** (Rm / cpm) * scratch.interior
            model.specific_energy.interior.copy_(cpm * T + c.gravitational_acceleration * z)
            model.energy_density.interior.copy_(ρc * model.specific_energy.interior)
        elif key == "θ" or key == "e":
            model.potential_temperature.set_interior(value)
            model.potential_temperature_density.interior.copy_(ρc * model.potential_temperature.interior)
        elif key == "ρθ" or key == "ρe":
            model.potential_temperature_density.set_interior(value)
        elif key == "q":
            model.specific_moisture.set_interior(value)
            model.moisture_density.interior.copy_(ρc * model.specific_moisture.interior)
        elif key == "ρq":
            model.moisture_density.set_interior(value)
        elif key in ("qcl", "qr"):          # settable specific microphysical names (set_atmosphere_model.jl:247-253)
            if not getattr(model, "_kessler", False):
                raise ValueError(f"Cannot set! {name}: the model has no Kessler microphysics")
            # only ρq is set (set!(ρμ, value); set!(ρμ, ρ * ρμ)); the diagnostic field is refreshed by update_state!
            dens = model.microphysical_fields["ρqᶜˡ" if key == "qcl" else "ρqʳ"]
            dens.set_interior(value)
            dens.interior.copy_(ρc * dens.interior)
        elif key in ("u", "v"):
            model.velocities[key].set_interior(value)
            model.momentum["ρ" + key].interior.copy_(ρc * model.velocities[key].interior)
        elif key == "w":
            model.velocities["w"].set_interior(value)
            model.momentum["ρw"].interior.copy_(ρf * model.velocities["w"].interior)
        else:   # ρu, ρv, ρw
            model.momentum[key].set_interior(value)
    if hasattr(model, "_finish_set"):      # y-slab models: update_state! + exchanges + the distributed projection, inside the library
        model._finish_set(enforce_mass_conservation)
        return
    update_state_(model, compute_tendencies=False)
    if enforce_mass_conservation:
        enforce_mass_conservation_(model)


def time_step_(model, Δt, whole_step=True):
    """time_step!(model, Δt) with SSP-RK3.  whole_step=True uses the single-call device-resident seam
    (bz_time_step_anelastic); False replays the reference's call sequence through the per-operator
    entry points (same kernels, used by the parity tests)."""
    if model.clock.iteration == 0:       # maybe_prepare_first_time_step!
        update_state_(model, compute_tendencies=True)
    if whole_step:
        model._check(model._lib.bz_time_step_anelastic(model._ctx, C.byref(model._state), C.byref(model._U0),
                                                       C.byref(model._G), float(Δt)), "bz_time_step_anelastic")
    else:
        store_initial_state_(model)
        for α in (1.0, 1.0 / 4.0, 2.0 / 3.0):
            compute_flux_bc_tendencies_(model)
            ssp_rk3_substep_(model, Δt, α)
            compute_pressure_correction_(model, α * Δt)
            make_pressure_correction_(model, α * Δt)
            update_state_(model, compute_tendencies=True)
        if getattr(model, "_kessler", False):
            from .microphysics import microphysics_model_update_
            microphysics_model_update_(model.microphysics, model, Δt=Δt)
    model.clock.time += Δt
    model.clock.last_Δt = float(Δt)
    model.clock.iteration += 1


def many_time_steps_(model, Δt, n, diagnose_last=True):
    """many_time_steps!(model, Δt, n) of the reference's benchmark driver through the multi-step seam."""
    n = int(n)
    if n <= 0:
        return
    if model.clock.iteration == 0:       # maybe_prepare_first_time_step!
        update_state_(model, compute_tendencies=True)
    model._check(model._lib.bz_time_steps_anelastic(model._ctx, C.byref(model._state), C.byref(model._U0), C.byref(model._G),
                                                    float(Δt), n, 1 if diagnose_last else 0), "bz_time_steps_anelastic")
    model.clock.time += n * Δt
    model.clock.last_Δt = float(Δt)
    model.clock.iteration += n


def diagnostics_stale(model):
    return bool(model._lib.bz_diagnostics_stale(model._ctx))


def cell_advection_timescale(model, formulation="ThreeDimensional"):
    """cell_advection_timescale(model) (src/AtmosphereModels/cell_advection_timescale.jl:47-66): the TimeStepWizard's
    advective timescale min 1/(|u|/Δx + |v|/Δy + |w|/Δz); formulation "Horizontal" drops the vertical term."""
    if diagnostics_stale(model):      # the velocities are diagnostics: rebuild them after undiagnosed steps (time_steps(..., diagnose_last=False))
        model._refresh_diagnostics()
    out = model._T.real()
    w = None if str(formulation).startswith("Horizontal") else C.c_void_p(model.velocities["w"].ptr())
    model._check(model._lib.bz_cell_advection_timescale(model._ctx, C.c_void_p(model.velocities["u"].ptr()),
                                                        C.c_void_p(model.velocities["v"].ptr()), w, C.byref(out)),
                 "bz_cell_advection_timescale")
    return out.value


def nan_checker(model):
    """default_nan_checker(model): NaN check of the first prognostic field (atmosphere_model.jl:561-572)."""
    name, field = next(iter(model.prognostic_fields().items()))
    out = C.c_int32()
    model._check(model._lib.bz_any_nan(model._ctx, C.c_void_p(field.ptr()), 1 if field.zface else 0, C.byref(out)), "bz_any_nan")
    return bool(out.value)
