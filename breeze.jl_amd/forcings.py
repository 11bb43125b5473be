"""Host-side mirror of the forcing / Coriolis / boundary-flux interface of the BOMEX configuration (BASELINE configs[2]).

The reference assembles these from Oceananigans' `Forcing`, `FluxBoundaryCondition`, `FPlane` and Breeze's own
`SubsidenceForcing` / `geostrophic_forcings` (examples/bomex.jl:80-207).  Every one of them is a horizontally uniform
column profile (times the reference density, src/Forcings/specific_forcing.jl:61-74), so `materialize_forcings` reduces the
model's `forcing` / `coriolis` / `boundary_conditions` keywords to one `bz_column_forcings` struct for `bz_set_forcings`.
Nothing here computes on the CPU: the profiles are evaluated once at construction (as `set!` of a column Field is in the
reference) and the tendencies are HIP kernels (csrc/bz_forcing.hip).
"""
import ctypes as C
import unicodedata
from collections import namedtuple

import numpy as np

from . import _lib


class SmagorinskyLilly:
    """SmagorinskyLilly(; C = 0.16, Cb = 1.0, Pr = 1.0) (Oceananigans.TurbulenceClosures, re-exported src/Breeze.jl:186,220)."""

    def __init__(self, C=0.16, Cb=1.0, Pr=1.0):
        self.C, self.Cb, self.Pr = float(C), float(Cb), float(Pr)


class FPlane:
    """FPlane(f=...) (Oceananigans.Coriolis)."""

    def __init__(self, f):
        self.f = float(f)


class SubsidenceForcing:
    """SubsidenceForcing(wˢ) (src/Forcings/subsidence_forcing.jl:14-58): wˢ a function of z or Nz+1 face values."""

    def __init__(self, subsidence_vertical_velocity):
        self.subsidence_vertical_velocity = subsidence_vertical_velocity


class GeostrophicForcing:
    """One component of geostrophic_forcings (src/Forcings/geostrophic_forcings.jl): direction "x": -f vᵍ, "y": +f uᵍ."""

    def __init__(self, geostrophic_velocity, direction):
        self.geostrophic_velocity, self.direction = geostrophic_velocity, direction


GeostrophicForcings = namedtuple("GeostrophicForcings", ("u", "v"))


def geostrophic_forcings(uᵍ, vᵍ):
    """geostrophic_forcings(uᵍ, vᵍ) -> (u = -f vᵍ, v = +f uᵍ); f comes from the model's coriolis at materialization."""
    return GeostrophicForcings(GeostrophicForcing(vᵍ, "x"), GeostrophicForcing(uᵍ, "y"))


class Forcing:
    """Forcing(profile): a column field / function of z added to the specific tendency (Oceananigans Forcing(field))."""

    def __init__(self, profile):
        self.profile = profile


class GaussianMask:
    """GaussianMask{:z}(center, width): exp(-(z - center)^2 / (2 width^2)) (Oceananigans; examples/rico.jl:104,
    examples/neutral_atmospheric_boundary_layer.jl:105)."""

    def __init__(self, center, width, direction="z"):
        if str(direction).lstrip(":") != "z":
            raise NotImplementedError("GaussianMask{:z}: sponges vary with height only")
        self.center, self.width = float(center), float(width)

    def __call__(self, z):
        return np.exp(-(z - self.center) ** 2 / (2 * self.width ** 2))


class Relaxation:
    """Relaxation(rate, mask, target): F = rate * mask(z) * (target(z) - field) (Oceananigans Relaxation; the sponge layers of
    examples/rico.jl:103-105,164, neutral_atmospheric_boundary_layer.jl:127, tropical_cyclone_world.jl).  mask: GaussianMask or a function
    of z (default 1); target: a number, a function of z or a column (default 0).  Keyed in `forcing` by a density name (ρu, ρv, ρw, ρθ / ρe,
    ρqᵉ ...: the density relaxes, as the reference's examples write it) or by u, v, w (specific forcing: ρᵣ F)."""

    def __init__(self, rate, mask=None, target=0.0):
        self.rate, self.mask, self.target = float(rate), mask, target


class FrictionVelocityDrag:
    """The example's bulk drag closure -ρ₀ u★² ρu / √(ρu² + ρv²) (examples/bomex.jl:95-99) as data."""

    def __init__(self, ρ0, ustar, epsilon=0.0):
        # epsilon: the ϵ under the square root of benchmarking/src/convective_boundary_layer.jl:142-146 (1e-10 there)
        self.ρ0, self.ustar, self.epsilon = float(ρ0), float(ustar), float(epsilon)


class BulkDrag:
    """BulkDrag(; coefficient = 1e-3, gustiness = 0, surface_temperature = nothing) (src/BoundaryConditions/bulk_drag.jl:60-84);
    without a surface temperature the anelastic default Π₀ θ₀ of the reference state is used
    (src/AnelasticEquations/anelastic_dynamics.jl:99-105)."""

    def __init__(self, coefficient=1e-3, gustiness=0.0, surface_temperature=None):
        self.coefficient, self.gustiness, self.surface_temperature = float(coefficient), float(gustiness), surface_temperature


class BulkSensibleHeatFlux:
    """BulkSensibleHeatFlux(; coefficient, gustiness = 0, surface_temperature) (bulk_scalar_fluxes.jl:53-56)."""

    def __init__(self, coefficient, surface_temperature, gustiness=0.0):
        self.coefficient, self.gustiness, self.surface_temperature = float(coefficient), float(gustiness), float(surface_temperature)


class BulkVaporFlux:
    """BulkVaporFlux(; coefficient, gustiness = 0, surface_temperature) (bulk_scalar_fluxes.jl:172-176)."""

    def __init__(self, coefficient, surface_temperature, gustiness=0.0):
        self.coefficient, self.gustiness, self.surface_temperature = float(coefficient), float(gustiness), float(surface_temperature)


class FluxBoundaryCondition:
    def __init__(self, condition):
        self.condition = condition


class FieldBoundaryConditions:
    def __init__(self, bottom=None, top=None, west=None, east=None, south=None, north=None):
        """bottom: the surface flux conditions of the anelastic model; west / east / south / north: NormalFlowBoundaryCondition on the
        wall-normal momentum of a compressible model with a Bounded x / y (the open lateral boundaries of the acoustic substep loop)."""
        if top is not None:
            raise NotImplementedError("only bottom flux boundary conditions are implemented")
        self.bottom = bottom
        self.west, self.east, self.south, self.north = west, east, south, north


def _key(name):
    return unicodedata.normalize("NFKC", str(name))


def _profile(value, z):
    if callable(value):
        return np.ascontiguousarray([float(value(zk)) for zk in z], dtype=np.float64)
    a = np.ascontiguousarray(value, dtype=np.float64)
    if a.shape != z.shape:
        raise ValueError(f"profile has shape {a.shape}, expected {z.shape}")
    return a


_RELAX_SLOTS = {"ρu": ("u", 0), "ρv": ("v", 0), "ρw": ("w", 0), "ρθ": ("theta", 0), "ρe": ("theta", 0), "ρqe": ("moisture", 0),
                "ρqv": ("moisture", 0), "ρqt": ("moisture", 0), "ρq": ("moisture", 0), "u": ("u", 1), "v": ("v", 2), "w": ("w", 4)}


def _is_field_forcing(item):
    """Forcing(f(x, y, z)) or Forcing(3-D array): a forcing that varies in the horizontal (the column stack takes functions of z)."""
    if not isinstance(item, Forcing):
        return False
    f = item.profile
    if callable(f):
        import inspect
        try:
            nargs = len(inspect.signature(f).parameters)
        except (TypeError, ValueError):
            return False
        if nargs == 4:      # Forcing(f(x, y, z, t)): nothing here re-evaluates a forcing field as the clock advances
            raise NotImplementedError("Forcing(f(x, y, z, t)): time-dependent 3-D forcings are not implemented — pass f(x, y, z) "
                                      "(evaluated once) or a 3-D array and refresh it yourself between steps")
        return nargs == 3
    return np.ndim(f) == 3


def split_relaxation(forcing):
    """-> (forcing without the entries below, {key: Relaxation}, {key: Forcing}): sponges go to bz_set_relaxation, 3-D forcings of the
    thermodynamic variable to bz_set_field_forcing, the rest to the column-forcing stack."""
    rest, relax, field = {}, {}, {}
    for name, entry in (forcing or {}).items():
        items = entry if isinstance(entry, (tuple, list)) else (entry,)
        keep = tuple(it for it in items if not isinstance(it, Relaxation) and not _is_field_forcing(it))
        mine = [it for it in items if isinstance(it, Relaxation)]
        fields = [it for it in items if _is_field_forcing(it)]
        if len(mine) > 1 or len(fields) > 1:
            raise NotImplementedError(f"one Relaxation and one 3-D Forcing per field ({name!r})")
        if mine:
            relax[_key(name)] = mine[0]
        if fields:
            if _key(name) not in ("θ", "ρθ", "e", "ρe"):
                raise NotImplementedError(f"3-D Forcing on {name!r}: the thermodynamic variable (θ / e, or ρθ / ρe) is implemented")
            if field:
                raise NotImplementedError("one 3-D Forcing of the thermodynamic variable")
            field[_key(name)] = fields[0]
        if keep:
            rest[name] = keep if len(keep) > 1 else keep[0]
    return (rest or None), relax, field


def materialize_field_forcing(grid, field, formulation, device):
    """{key: Forcing} -> (device Field holding F at the cell centres, specific flag) or (None, 0)."""
    if not field:
        return None, 0
    from .model import _LOC, Field
    (k, item), = field.items()
    if (k in ("e", "ρe")) != (formulation == "StaticEnergy"):
        raise ValueError(f"3-D Forcing on {k!r}: the thermodynamic variable of this formulation is {'e' if formulation == 'StaticEnergy' else 'θ'}")
    F = Field(grid, _LOC["ccc"], device)
    F.set_interior(item.profile)
    return F, 0 if k.startswith("ρ") else 1


def materialize_relaxation(grid, relax, formulation, T=None):
    """{key: Relaxation} -> (bz_column_relaxation, keepalive arrays) or (None, None)."""
    T = T or _lib.types(8)
    if not relax:
        return None, None
    zc, zf = np.asarray(grid.zᶜ, dtype=np.float64), np.asarray(grid.zᶠ, dtype=np.float64)
    S, keep, seen = T.bz_column_relaxation(), [], set()
    for k, r in relax.items():
        if k not in _RELAX_SLOTS:
            raise NotImplementedError(f"Relaxation on {k!r} is not implemented (ρu, ρv, ρw, ρθ / ρe, the moisture density; u, v, w)")
        if (k == "ρe") != (formulation == "StaticEnergy") and k in ("ρe", "ρθ"):
            raise ValueError(f"Relaxation on {k!r}: the thermodynamic density of this formulation is {'ρe' if formulation == 'StaticEnergy' else 'ρθ'}")
        slot, bit = _RELAX_SLOTS[k]
        if slot in seen:
            raise NotImplementedError(f"one Relaxation per field ({slot})")
        seen.add(slot)
        z = zf if slot == "w" else zc
        mask = np.ones_like(z) if r.mask is None else np.asarray([float(r.mask(zk)) for zk in z])
        rate = np.ascontiguousarray(r.rate * mask, dtype=T.np_real)
        target = np.ascontiguousarray(np.full_like(z, float(r.target)) if np.isscalar(r.target) else _profile(r.target, z), dtype=T.np_real)
        keep += [rate, target]
        setattr(S, "rate_" + slot, rate.ctypes.data_as(C.POINTER(T.real)))
        setattr(S, "target_" + slot, target.ctypes.data_as(C.POINTER(T.real)))
        S.specific_mask |= bit
    return S, keep


def materialize_forcings(grid, coriolis, forcing, boundary_conditions, T=None):
    """-> (bz_column_forcings, keepalive arrays) or (None, None) when nothing is attached.  T: _lib.types(grid.ftype) — the struct
    classes and scalar type of the grid's precision (profiles are evaluated in Float64 and stored in eltype(grid))."""
    T = T or _lib.types(8)
    if coriolis is None and not forcing and not boundary_conditions:
        return None, None
    if coriolis is not None and not isinstance(coriolis, FPlane):
        raise NotImplementedError("coriolis: FPlane is implemented")
    f = coriolis.f if coriolis is not None else 0.0
    zc, zf = np.asarray(grid.zᶜ, dtype=np.float64), np.asarray(grid.zᶠ, dtype=np.float64)
    slots = {"u": "u_forcing", "v": "v_forcing", "θ": "theta_forcing", "qe": "moisture_forcing", "qv": "moisture_forcing",
             "qt": "moisture_forcing", "e": "energy_forcing"}
    sub_flags = {"u": "subsidence_u", "v": "subsidence_v", "θ": "subsidence_theta", "qe": "subsidence_moisture",
                 "qv": "subsidence_moisture", "qt": "subsidence_moisture"}
    static, keep = {}, []
    S = T.bz_column_forcings()
    ws = None
    for name, entry in (forcing or {}).items():
        k = _key(name)
        if k.startswith("ρ"):
            raise ValueError("forcings are specific tendencies and must be supplied under the specific prognostic name "
                             "(e.g. `θ` instead of `ρθ`); the density factor is applied automatically")
        if k not in slots:
            raise NotImplementedError(f"forcing on {name!r} is not implemented")
        for item in (entry if isinstance(entry, (tuple, list)) else (entry,)):
            if isinstance(item, SubsidenceForcing):
                if k not in sub_flags:
                    raise NotImplementedError(f"SubsidenceForcing on {name!r} is not implemented")
                w = _profile(item.subsidence_vertical_velocity, zf)
                if ws is not None and not np.array_equal(ws, w):
                    raise NotImplementedError("one subsidence velocity profile per model")
                ws = w
                setattr(S, sub_flags[k], 1)
            elif isinstance(item, GeostrophicForcing):
                if coriolis is None:
                    raise ValueError("geostrophic forcings need the model's coriolis")
                if (item.direction, k) not in (("x", "u"), ("y", "v")):
                    raise ValueError("geostrophic.u belongs under `u`, geostrophic.v under `v`")
                prof = _profile(item.geostrophic_velocity, zc)
                prof = -f * prof if item.direction == "x" else f * prof
                static[slots[k]] = static.get(slots[k], 0.0) + prof
            elif isinstance(item, Forcing) or callable(item):
                prof = _profile(item.profile if isinstance(item, Forcing) else item, zc)
                static[slots[k]] = static.get(slots[k], 0.0) + prof
            else:
                raise NotImplementedError(f"forcing {item!r} is not implemented")
    for slot, prof in static.items():
        a = np.ascontiguousarray(prof, dtype=T.np_real)
        keep.append(a)
        setattr(S, slot, a.ctypes.data_as(C.POINTER(T.real)))
    if ws is not None:
        ws = np.ascontiguousarray(ws, dtype=T.np_real)
        keep.append(ws)
        S.subsidence_vertical_velocity = ws.ctypes.data_as(C.POINTER(T.real))
    S.coriolis_f = f
    drag = None
    for name, bcs in (boundary_conditions or {}).items():
        k = _key(name)
        bottom = bcs.bottom if isinstance(bcs, FieldBoundaryConditions) else bcs
        if bottom is None:
            continue
        cond = bottom.condition if isinstance(bottom, FluxBoundaryCondition) else bottom
        if isinstance(cond, (BulkDrag, BulkSensibleHeatFlux, BulkVaporFlux)):
            continue                      # collected by materialize_bulk_fluxes
        if k == "ρe":      # an energy flux keyed ρe in a potential-temperature model: Q / cᵖᵐ enters ρθ (BoundaryConditions.jl:218-227)
            if S.bottom_theta_flux:
                raise ValueError("Cannot specify boundary conditions on both ρθ and ρe")
            S.bottom_energy_flux = float(cond)
        elif k == "ρθ":
            if S.bottom_energy_flux:
                raise ValueError("Cannot specify boundary conditions on both ρθ and ρe")
            S.bottom_theta_flux = float(cond)
        elif k in ("ρqe", "ρqv", "ρqt"):
            S.bottom_moisture_flux = float(cond)
        elif k in ("ρu", "ρv"):
            if not isinstance(cond, FrictionVelocityDrag):
                raise NotImplementedError("momentum bottom flux: FrictionVelocityDrag(ρ₀, u★)")
            d = (cond.ρ0 * cond.ustar ** 2, cond.epsilon)
            if drag is not None and drag != d:
                raise NotImplementedError("ρu and ρv share one drag")
            drag = d
        else:
            raise NotImplementedError(f"boundary condition on {name!r} is not implemented")
    S.bottom_drag_rho0_ustar2, S.bottom_drag_epsilon = drag or (0.0, 0.0)
    if not static and ws is None and f == 0.0 and not (S.bottom_theta_flux or S.bottom_moisture_flux or S.bottom_drag_rho0_ustar2 or
                                                          S.bottom_energy_flux):
        return None, None                 # e.g. only bulk conditions were given
    return S, keep


def materialize_bulk_fluxes(boundary_conditions, reference_state, constants, T=None):
    """-> bz_bulk_surface_fluxes or None.  BulkDrag belongs on ρu / ρv (one coefficient for both), BulkSensibleHeatFlux on ρθ,
    BulkVaporFlux on the moisture density; anything else raises like the reference's regularization does."""
    B, found, drag = (T or _lib.types(8)).bz_bulk_surface_fluxes(), False, None
    for name, bcs in (boundary_conditions or {}).items():
        k = _key(name)
        bottom = bcs.bottom if isinstance(bcs, FieldBoundaryConditions) else bcs
        cond = bottom.condition if isinstance(bottom, FluxBoundaryCondition) else bottom
        if isinstance(cond, BulkDrag):
            if k not in ("ρu", "ρv"):
                raise ValueError("BulkDrag is a momentum boundary condition (ρu, ρv)")
            T0 = cond.surface_temperature
            if T0 is None:      # default_drag_surface_temperature(::AnelasticDynamics)
                from .thermodynamics import dry_air_gas_constant
                Π0 = (reference_state.surface_pressure / reference_state.standard_pressure) ** (
                    dry_air_gas_constant(constants) / constants.dry_air_heat_capacity)
                T0 = Π0 * reference_state.potential_temperature
            new = (cond.coefficient, cond.gustiness, float(T0))
            if drag is not None and drag != new:      # compared as given (the Float32 struct rounds what it stores)
                raise NotImplementedError("ρu and ρv share one BulkDrag")
            drag = new
            B.drag_coefficient, B.drag_gustiness, B.drag_surface_temperature = new
            found = True
        elif isinstance(cond, BulkSensibleHeatFlux):
            # keyed ρe in a potential-temperature model (examples/tropical_cyclone_world.jl:109-115) the condition is moved to ρθ with the
            # potential-temperature difference, unchanged (BoundaryConditions.jl:218-227, thermodynamic_variable_bcs.jl:283); bulk fluxes
            # are built for that formulation only (the model constructor says so for StaticEnergy)
            if k not in ("ρθ", "ρe"):
                raise ValueError("BulkSensibleHeatFlux belongs on ρθ (or ρe)")
            if B.heat_coefficient:
                raise ValueError("Cannot specify boundary conditions on both ρθ and ρe")
            B.heat_coefficient, B.heat_gustiness, B.heat_surface_temperature = cond.coefficient, cond.gustiness, cond.surface_temperature
            found = True
        elif isinstance(cond, BulkVaporFlux):
            if k not in ("ρqe", "ρqv", "ρqt"):
                raise ValueError("BulkVaporFlux belongs on the moisture density")
            B.vapor_coefficient, B.vapor_gustiness, B.vapor_surface_temperature = cond.coefficient, cond.gustiness, cond.surface_temperature
            found = True
    if not found:
        return None
    B.surface_pressure, B.standard_pressure = reference_state.surface_pressure, reference_state.standard_pressure
    c = constants
    B.liquid_latent_heat, B.liquid_heat_capacity = c.liquid_reference_latent_heat, c.liquid_heat_capacity
    B.energy_reference_temperature = c.energy_reference_temperature
    B.triple_point_temperature, B.triple_point_pressure = c.triple_point_temperature, c.triple_point_pressure
    return B
