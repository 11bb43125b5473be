"""Host-side mirror of Breeze's compressible split-explicit API above the C ABI (include/breeze_hip.h):

  CompressibleDynamics(time_discretization; ...)   src/CompressibleEquations/compressible_dynamics.jl:44-160
  SplitExplicitTimeDiscretization(; ...)           src/CompressibleEquations/time_discretizations.jl:535-598
  ThermalDivergenceDamping / NoDivergenceDamping   src/CompressibleEquations/time_discretizations.jl:142-262
  ExnerReferenceState(grid, constants; ...)        src/Thermodynamics/reference_states.jl:588-672,717-815
  AcousticSubstepper / AcousticRungeKutta3         src/CompressibleEquations/acoustic_substepping.jl:91-270,
                                                   src/TimeSteppers/acoustic_runge_kutta_3.jl:64-103
  set! / update_state! / time_step!                src/AtmosphereModels/set_atmosphere_model.jl:198-362,
                                                   src/TimeSteppers/acoustic_runge_kutta_3.jl:230-319
(paths relative to /root/reference; Julia's `f!` is spelled `f_`).  Every numerical operation of the time step is a
HIP kernel behind libbreeze_hip.so; the numpy code here only builds the 1-D reference columns at construction.
"""
import ctypes as C

import numpy as np

from . import _lib
from .grids import Bounded, Center, Face, Flat, Periodic, RectilinearGrid  # noqa: F401
from .model import WENO, Clock, Field, _LOC
from .thermodynamics import ThermodynamicConstants, dry_air_gas_constant, vapor_gas_constant


class NoDivergenceDamping:
    pass


class ThermalDivergenceDamping:
    def __init__(self, coefficient=0.1, length_scale=None, damp_vertical=False):
        if length_scale is not None and not float(length_scale) > 0:
            raise ValueError(f"`length_scale` must be positive (got {length_scale})")
        self.coefficient, self.damp_vertical = float(coefficient), bool(damp_vertical)
        self.length_scale = None if length_scale is None else float(length_scale)      # fixed diffusivity alpha l^2 / dtau


class DirectDivergenceDamping:
    """DirectDivergenceDamping(coefficient=0.1) (time_discretizations.jl:269-274): the horizontal theta-flux divergence of the
    perturbation momentum is differenced directly (acoustic_substepping.jl:1146-1188); horizontal only."""

    def __init__(self, coefficient=0.1):
        self.coefficient = float(coefficient)


class ProportionalSubsteps:
    """each stage covers beta dt with ceil(beta N) substeps sized to tile it (acoustic_substepping.jl:491-495; the default)"""
    code = 0


class ConstantSubstepSize:
    """one substep size dt / N for all stages, N rounded up to a multiple of 6 (acoustic_substepping.jl:497-501)"""
    code = 1


class MonolithicFirstStage:
    """stage 1 collapses to one substep of dt / 3, stages 2-3 as ConstantSubstepSize (acoustic_substepping.jl:503-508)"""
    code = 2


class LinearRamp:
    code = 1


class CubicRamp:
    code = 2


class Sin2Ramp:
    code = 3


class UpperSponge:
    """UpperSponge(damping_rate=0.2, depth=5e3, ramp=CubicRamp()) (time_discretizations.jl:435-512): implicit Rayleigh damping of
    (ρw)′ in a layer of thickness `depth` below z = grid.Lz, inside the column system of the acoustic substeps
    (acoustic_substepping.jl:584-602,639,948)."""

    def __init__(self, damping_rate=0.2, depth=5e3, ramp=None):
        ramp = CubicRamp() if ramp is None else ramp
        if not isinstance(ramp, (LinearRamp, CubicRamp, Sin2Ramp)):
            raise ValueError("`ramp` must be an `<:AbstractRamp` (e.g. `CubicRamp()`, `Sin2Ramp()`, `LinearRamp()`)")
        self.damping_rate, self.depth, self.ramp = float(damping_rate), float(depth), ramp


class SplitExplicitTimeDiscretization:
    def __init__(self, substeps=None, acoustic_cfl=0.5, forward_weight=0.65, thermodynamic_tendency_factor=1,
                 vertical_momentum_tendency_factor=1, apply_first_substep_pressure_gradient=False, damping=None,
                 sponge=None, substep_distribution=None, open_boundary_relaxation=0.5):
        damping = ThermalDivergenceDamping(coefficient=0.1) if damping is None else damping
        if not isinstance(damping, (ThermalDivergenceDamping, DirectDivergenceDamping, NoDivergenceDamping)):
            raise ValueError("`damping` must be an `AcousticDampingStrategy`")
        if sponge is not None and not isinstance(sponge, UpperSponge):
            raise ValueError("`sponge` must be an `UpperSponge` or None")
        if substep_distribution is not None and not isinstance(substep_distribution, (ProportionalSubsteps, ConstantSubstepSize, MonolithicFirstStage)):
            raise ValueError("`substep_distribution` must be ProportionalSubsteps(), ConstantSubstepSize() or MonolithicFirstStage()")
        if not acoustic_cfl > 0:
            raise ValueError(f"`acoustic_cfl` must be positive (got {acoustic_cfl})")
        if not 0 < open_boundary_relaxation <= 1:      # time_discretizations.jl:573-574
            raise ValueError(f"`open_boundary_relaxation` must be in (0, 1] (got {open_boundary_relaxation})")
        self.open_boundary_relaxation = float(open_boundary_relaxation)
        self.substeps = None if substeps is None else int(substeps)
        self.acoustic_cfl = float(acoustic_cfl)
        self.forward_weight = float(forward_weight)
        self.thermodynamic_tendency_factor = float(thermodynamic_tendency_factor)
        self.vertical_momentum_tendency_factor = float(vertical_momentum_tendency_factor)
        self.apply_first_substep_pressure_gradient = bool(apply_first_substep_pressure_gradient)
        self.damping = damping
        self.sponge = sponge
        self.substep_distribution = substep_distribution or ProportionalSubsteps()


class NormalFlowBoundaryCondition:
    """NormalFlowBoundaryCondition(value) on the wall-normal momentum of a Bounded side (the open boundary of
    test/acoustic_substepping_open_boundaries.jl:56-57): `condition is None` is the impenetrable default."""

    def __init__(self, condition=None):
        self.condition = condition


def is_active_open_bc(bc):
    """is_active_open_bc (acoustic_substepping.jl:1318)"""
    return isinstance(bc, NormalFlowBoundaryCondition) and bc.condition is not None


class NewtonSolver:
    """Breeze.Solvers.NewtonSolver (src/Solvers.jl:55-83): reltol = 0 only."""

    def __init__(self, abstol=1e-4, maxiter=8):
        self.reltol, self.abstol, self.maxiter = 0.0, float(abstol), int(maxiter)


def _z_metrics(grid):
    """Δzᶜ (Nz) and Δzᶠ (centre spacing at faces 0..Nz; entries 1..Nz-1 interior) as Oceananigans defines them."""
    Nz = grid.Nz
    if grid.regular_z:
        return np.full(Nz, grid.Δz), np.full(Nz + 1, grid.Δz)
    dzc = np.diff(grid.zᶠ)
    zc = grid.zᶜ
    dzf = np.empty(Nz + 1)
    dzf[1:Nz] = np.diff(zc)
    dzf[0] = 2 * (zc[0] - grid.zᶠ[0])
    dzf[Nz] = 2 * (grid.zᶠ[Nz] - zc[Nz - 1])
    return dzc, dzf


class ExnerReferenceState:
    """1-D, isentropic-mode ExnerReferenceState, dry or moist (`vapor_mass_fraction` a number or qᵛ(z)): the discrete
    hydrostatic balance (p[k]-p[k-1])/Δzᶠ + g (ρ[k]+ρ[k-1])/2 = 0 holds to rounding at every interior face
    (src/Thermodynamics/reference_states.jl:572-672,718-827)."""

    def __init__(self, grid, constants=None, surface_pressure=101325, potential_temperature=288, standard_pressure=1e5,
                 vapor_mass_fraction=None, reference_temperature=None):
        if reference_temperature is not None:
            raise NotImplementedError("the isothermal ExnerReferenceState mode is not implemented")
        c = constants or ThermodynamicConstants()
        self.constants = c
        Nz, Hz = grid.Nz, grid.Hz
        self.surface_pressure = p0 = float(surface_pressure)
        self.standard_pressure = pst = float(standard_pressure)
        θfun = potential_temperature if callable(potential_temperature) else (lambda z: float(potential_temperature) + 0.0 * z)
        self.surface_potential_temperature = float(θfun(np.float64(0.0)))
        Rd, Rv = dry_air_gas_constant(c), vapor_gas_constant(c)
        cpd, cpv, g = c.dry_air_heat_capacity, c.vapor_heat_capacity, c.gravitational_acceleration
        dzc, dzf = _z_metrics(grid)
        θ = np.asarray(θfun(grid.zᶜ), dtype=np.float64)
        if vapor_mass_fraction is None:
            qv = np.zeros(Nz)
        elif callable(vapor_mass_fraction):
            qv = np.array([float(vapor_mass_fraction(z)) for z in grid.zᶜ])
        else:
            qv = np.full(Nz, float(vapor_mass_fraction))
        self.vapor_mass_fraction = qv

        def moist(q):      # moist_reference_constants (reference_states.jl:572-577); exactly the dry constants for q = 0
            qd = 1 - q
            Rm, cpm = qd * Rd + q * Rv, qd * cpd + q * cpv
            return Rm, cpm, Rm / cpm

        π, p, ρ = np.zeros(Nz), np.zeros(Nz), np.zeros(Nz)
        Rm1, cpm1, κ1 = moist(qv[0])
        π_surface = (p0 / pst) ** κ1
        π[0] = π_surface - g * dzc[0] / (2 * cpm1 * θ[0])
        p[0] = pst * π[0] ** (1 / κ1)
        ρ[0] = p[0] / (Rm1 * θ[0] * π[0])
        for k in range(1, Nz):
            Rm, cpm, κ = moist(qv[k])
            θface = (θ[k] + θ[k - 1]) / 2
            pk = pst * (π[k - 1] - g * dzf[k] / (cpm * θface)) ** (1 / κ)
            A = g * pst ** κ / (2 * Rm * θ[k])
            Cc = p[k - 1] / dzf[k] - g * ρ[k - 1] / 2
            for _ in range(5):                                   # FixedIterations(5) Newton on the discrete balance
                ρp = pk ** (-κ)
                f = pk / dzf[k] + A * pk * ρp - Cc
                df = 1 / dzf[k] + A * (1 - κ) * ρp
                pk -= f / df
            p[k] = pk
            π[k] = (pk / pst) ** κ
            ρ[k] = pk / (Rm * θ[k] * π[k])
        self.surface_density = ρ0 = p0 / (Rm1 * self.surface_potential_temperature * π_surface)

        def with_halos(a, bottom_value=None):
            out = np.zeros(Nz + 2 * Hz)
            out[Hz:Hz + Nz] = a
            out[Hz - 1] = a[0] if bottom_value is None else 2 * bottom_value - a[0]
            out[Hz + Nz] = a[-1]
            return out

        self.pressure = with_halos(p, p0)
        self.density = with_halos(ρ, ρ0)
        self.exner_function = with_halos(π)
        self.potential_temperature = with_halos(θ)
        self.Nz, self.Hz = Nz, Hz


class CompressibleDynamics:
    def __init__(self, time_discretization=None, standard_pressure=1e5, surface_pressure=101325.0,
                 reference_potential_temperature=None, reference_vapor_mass_fraction=None, reference_state="auto"):
        if time_discretization is None or not isinstance(time_discretization, SplitExplicitTimeDiscretization):
            raise NotImplementedError("the HIP path implements CompressibleDynamics(SplitExplicitTimeDiscretization(...)); "
                                      "ExplicitTimeStepping is outside the hot-path scope")
        if reference_state not in ("auto", ":auto", None):
            raise ValueError(f"`reference_state` must be `:auto` or `nothing`; received {reference_state!r}.")
        if reference_state is None and reference_potential_temperature is not None:
            raise ValueError("`reference_state = nothing` disables the reference state and is mutually exclusive with an "
                             "explicit reference profile")
        self.time_discretization = time_discretization
        self.standard_pressure = float(standard_pressure)
        self.surface_pressure = float(surface_pressure)
        self._reference_vapor = reference_vapor_mass_fraction
        self._reference_spec = None if reference_state is None else (
            288.0 if reference_potential_temperature is None else reference_potential_temperature)
        # materialised by the model
        self.reference_state = None
        self.dry_density = self.total_density = self.pressure = None


class AcousticSubstepper:
    """Storage of the acoustic substepper (acoustic_substepping.jl:91-134, 207-231)."""

    FIELDS = (("exner", "ccc"), ("potential_temperature", "ccc"), ("gamma_R_mixture", "ccc"),
              ("density_perturbation", "ccc"), ("density_potential_temperature_perturbation", "ccc"),
              ("momentum_perturbation_u", "fcc"), ("momentum_perturbation_v", "cfc"), ("momentum_perturbation_w", "ccf"),
              ("density_predictor", "ccc"), ("density_potential_temperature_predictor", "ccc"),
              ("previous_density_potential_temperature_perturbation", "ccc"),
              ("time_averaged_u", "fcc"), ("time_averaged_v", "cfc"), ("time_averaged_w", "ccf"),
              ("slow_vertical_momentum_tendency", "ccf"), ("vertical_solver_source_term", "ccf"))

    # stored in substep_floattype (acoustic_substepping.jl:207-223); (rho w)', the solver's right-hand side, the slow vertical tendency
    # and the time-averaged velocities keep eltype(grid)
    WORKING = ("exner", "potential_temperature", "gamma_R_mixture", "density_perturbation", "density_potential_temperature_perturbation",
               "momentum_perturbation_u", "momentum_perturbation_v", "density_predictor", "density_potential_temperature_predictor",
               "previous_density_potential_temperature_perturbation")

    def __init__(self, grid, time_discretization, device, substep_floattype=None):
        td = time_discretization
        self.substeps, self.acoustic_cfl, self.forward_weight = td.substeps, td.acoustic_cfl, td.forward_weight
        self.damping = td.damping
        self.substep_floattype = np.dtype(grid.float_type if substep_floattype is None else substep_floattype).type
        if self.substep_floattype not in (np.float32, np.float64) or np.dtype(self.substep_floattype).itemsize > grid.ftype:
            raise NotImplementedError("substep_floattype: eltype(grid) or Float32 inside a Float64 model")
        for name, loc in self.FIELDS:
            setattr(self, name, Field(grid, _LOC[loc], device, float_type=self.substep_floattype if name in self.WORKING else None))
        self.linearization_exner = self.exner
        self.linearization_potential_temperature = self.potential_temperature
        self.linearization_gamma_R_mixture = self.gamma_R_mixture
        self.time_averaged_velocities = {"u": self.time_averaged_u, "v": self.time_averaged_v, "w": self.time_averaged_w}
        self.momentum_perturbation = {"u": self.momentum_perturbation_u, "v": self.momentum_perturbation_v,
                                      "w": self.momentum_perturbation_w}

    def struct(self):
        s = _lib.bz_acoustic_substepper()
        for name, _ in self.FIELDS:
            setattr(s, name, getattr(self, name).ptr())
        return s


class AcousticRungeKutta3:
    """Wicker-Skamarock RK3 with acoustic substepping: β = (1/3, 1/2, 1) (acoustic_runge_kutta_3.jl:64-103)."""

    def __init__(self, grid, prognostic_fields, dynamics, device, substep_floattype=None):
        self.β1, self.β2, self.β3 = 1.0 / 3.0, 1.0 / 2.0, 1.0
        self.U0 = {k: Field(grid, f.loc, device) for k, f in prognostic_fields.items()}
        self.Gn = {k: Field(grid, f.loc, device) for k, f in prognostic_fields.items()}
        self.substepper = AcousticSubstepper(grid, dynamics.time_discretization, device, substep_floattype=substep_floattype)


_ALIASES = {"θ": "θ", "theta": "θ", "θˡⁱ": "θ", "θli": "θ", "ρ": "ρ", "rho": "ρ", "u": "u", "v": "v", "w": "w",
            "qᵗ": "q", "qt": "q", "qᵛ": "q", "qv": "q", "qᶜˡ": "qcl", "qcl": "qcl", "qʳ": "qr", "qr": "qr"}


class CompressibleAtmosphereModel:
    """AtmosphereModel(grid; dynamics=CompressibleDynamics(SplitExplicitTimeDiscretization(...)), advection=WENO(order=5))
    — timestepper :AcousticRungeKutta3; coriolis = FPlane(f) and density-keyed Relaxation sponges on ρu, ρv, ρw, ρθ
    (examples/tropical_cyclone_with_rainband.jl:434-514) are the forcing terms built; closure = nothing."""

    def __init__(self, grid, dynamics, advection=None, thermodynamic_constants=None, temperature_solver=None,
                 closure=None, coriolis=None, microphysics=None, forcing=None, device="cuda:0", substep_floattype=None,
                 boundary_conditions=None):
        """substep_floattype: storage type of the acoustic substepper's working fields (the keyword of the reference's
        AcousticSubstepper constructor, acoustic_substepping.jl:181,199-235): None = eltype(grid); numpy.float32 inside a Float64
        model halves the bytes the substep kernels stream."""
        import torch
        if not isinstance(grid, RectilinearGrid):
            raise TypeError("grid must be a RectilinearGrid")
        # (Periodic, Flat, Bounded): the 2-D x-z cases of examples/acoustic_wave.jl:51 and inertia_gravity_wave.jl:70
        # a Bounded x and / or y (3-D): the acoustic substep loop with its lateral boundaries (refresh_linearization_, acoustic_rk3_substep_loop_);
        # set / time_step / update_state_ of such a model are not built (the library returns BZ_ERR_UNSUPPORTED)
        self.lateral_walls = Bounded in grid.topology[:2]
        if self.lateral_walls and (Flat in grid.topology or type(self) is not CompressibleAtmosphereModel):
            raise NotImplementedError("Bounded x / y of the compressible model: 3-D grids on one device")
        if not self.lateral_walls and grid.topology not in ((Periodic, Periodic, Bounded), (Periodic, Flat, Bounded)):
            raise NotImplementedError("the HIP path implements topology (Periodic, Periodic, Bounded) and (Periodic, Flat, Bounded)")
        self.boundary_conditions = boundary_conditions or {}
        if self.boundary_conditions and not self.lateral_walls:
            raise NotImplementedError("boundary_conditions: the open lateral boundaries of the acoustic substep loop on a Bounded x / y")
        if grid.topology[1] == Flat and not (isinstance(advection, WENO) and advection.bounds is None):
            raise NotImplementedError("(Periodic, Flat, Bounded): the WENO(order = 5 | 7 | 9) model is implemented")
        if isinstance(advection, WENO) and advection.order != 5:      # examples/splitting_supercell.jl:279 uses WENO(order = 9)
            need = (advection.order + 1) // 2
            if min(h for h, t in zip((grid.Hx, grid.Hy, grid.Hz), grid.topology) if t != Flat) < need:
                raise ValueError(f"WENO(order={advection.order}) needs halos of at least {need} cells in every direction")
            if advection.bounds is not None:
                raise NotImplementedError("bounds-preserving WENO is implemented for order 5")
        if not isinstance(dynamics, CompressibleDynamics):
            raise TypeError("dynamics must be CompressibleDynamics")
        if closure is not None:
            raise NotImplementedError("closure is outside the hot-path scope of this build")
        from .forcings import FPlane, split_relaxation
        if coriolis is not None and not isinstance(coriolis, FPlane):
            raise NotImplementedError("coriolis: FPlane is implemented")
        forcing_rest, self._relaxation, self._field_forcing = split_relaxation(forcing)
        if forcing_rest:
            raise NotImplementedError("CompressibleDynamics: Relaxation sponges keyed ρu, ρv, ρw, ρθ and a 3-D Forcing on θ / ρθ are the forcings implemented")
        if set(self._relaxation) - {"ρu", "ρv", "ρw", "ρθ"}:
            raise NotImplementedError("CompressibleDynamics: Relaxation sponges keyed ρu, ρv, ρw, ρθ")
        if (coriolis is not None or self._relaxation or self._field_forcing) and type(self) is not CompressibleAtmosphereModel:
            raise NotImplementedError("Coriolis and sponges of the compressible model: single-GPU contexts")
        self.coriolis, self.forcing = coriolis, forcing
        from .microphysics import DCMIP2016KesslerMicrophysics, SaturationAdjustment, TetensFormula
        if microphysics is not None and not isinstance(microphysics, (DCMIP2016KesslerMicrophysics, SaturationAdjustment)):
            raise NotImplementedError("compressible microphysics: DCMIP2016KesslerMicrophysics() and "
                                      "SaturationAdjustment(equilibrium = WarmPhaseEquilibrium()) are implemented")
        self.microphysics = microphysics
        self._kessler = isinstance(microphysics, DCMIP2016KesslerMicrophysics)
        self._sa = isinstance(microphysics, SaturationAdjustment)
        if self._kessler:      # validate_microphysics (dcmip2016_kessler.jl:196-207)
            tcs = thermodynamic_constants
            if tcs is None or not isinstance(getattr(tcs, "saturation_vapor_pressure", None), TetensFormula):
                raise ValueError("DCMIP2016KesslerMicrophysics requires `thermodynamic_constants` with a `TetensFormula` "
                                 "saturation vapor pressure formulation.")
        if advection is None:
            from .model import Centered
            advection = Centered(order=2)          # the reference's default
        if not torch.cuda.is_available():
            raise RuntimeError("CompressibleAtmosphereModel needs a GPU: the HIP path has no CPU fallback")
        self.grid, self.advection, self.dynamics = grid, advection, dynamics
        self.thermodynamic_constants = c = thermodynamic_constants or ThermodynamicConstants()
        self.temperature_solver = temperature_solver or NewtonSolver()
        self.clock = Clock()
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self._T = T = _lib.types(grid.ftype)
        if grid.ftype == 4:      # eltype(grid) = Float32 (examples/splitting_supercell.jl:86): the Float32 twin of the library
            if not isinstance(advection, WENO):
                raise NotImplementedError("Float32 grids: WENO(order = 5 | 7 | 9) is wired up")
            self._lib = lib = _lib.load_f32()
        else:
            self._lib = lib = _lib.load(advection.order, getattr(advection, "ft2_hypothesis", 0))

        def fld(loc):
            return Field(grid, _LOC[loc], self.device)

        dynamics.dry_density, dynamics.total_density, dynamics.pressure = fld("ccc"), fld("ccc"), fld("ccc")
        self.momentum = {"ρu": fld("fcc"), "ρv": fld("cfc"), "ρw": fld("ccf")}
        self.velocities = {"u": fld("fcc"), "v": fld("cfc"), "w": fld("ccf")}
        self.potential_temperature_density, self.potential_temperature = fld("ccc"), fld("ccc")
        self.moisture_density, self.specific_moisture, self.temperature = fld("ccc"), fld("ccc"), fld("ccc")
        self.microphysical_fields = {}
        if self._kessler:      # materialize_microphysical_fields(::DCMIP2016KM) (dcmip2016_kessler.jl:255-290)
            self.microphysical_fields = {k: fld("ccc") for k in ("ρqᶜˡ", "ρqʳ", "qᵛ", "qᶜˡ", "qʳ", "𝕎ʳ")}
            self.microphysical_fields["precipitation_rate"] = torch.zeros((grid.Ny + 2 * grid.Hy, grid.Nx + 2 * grid.Hx),
                                                                          dtype=self.potential_temperature.dtype, device=self.device)
        if dynamics._reference_spec is not None:
            dynamics.reference_state = ExnerReferenceState(grid, c, surface_pressure=dynamics.surface_pressure,
                                                           potential_temperature=dynamics._reference_spec,
                                                           vapor_mass_fraction=dynamics._reference_vapor,
                                                           standard_pressure=dynamics.standard_pressure)
        self.timestepper = AcousticRungeKutta3(grid, self.prognostic_fields(), dynamics, self.device, substep_floattype=substep_floattype)
        self.U0, self.G = self.timestepper.U0, self.timestepper.Gn

        # ---- context ----
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
        ref = dynamics.reference_state
        br = T.bz_exner_reference_state()
        br.standard_pressure = dynamics.standard_pressure
        self._ref_arrays = None
        if ref is not None:
            self._ref_arrays = [np.ascontiguousarray(a, dtype=T.np_real) for a in (ref.pressure, ref.density)]
            br.pressure, br.density = (a.ctypes.data_as(C.POINTER(T.real)) for a in self._ref_arrays)
        td = dynamics.time_discretization
        bt = T.bz_split_explicit()
        bt.substeps = 0 if td.substeps is None else td.substeps
        damp = td.damping
        bt.damp_vertical = int(isinstance(damp, ThermalDivergenceDamping) and damp.damp_vertical)
        bt.apply_first_substep_pressure_gradient = int(td.apply_first_substep_pressure_gradient)
        bt.newton_maxiter = self.temperature_solver.maxiter
        bt.acoustic_cfl, bt.forward_weight = td.acoustic_cfl, td.forward_weight
        bt.damping_coefficient = damp.coefficient if isinstance(damp, (ThermalDivergenceDamping, DirectDivergenceDamping)) else -1.0
        bt.direct_divergence_damping = int(isinstance(damp, DirectDivergenceDamping))
        bt.damping_length_scale = (damp.length_scale or 0.0) if isinstance(damp, ThermalDivergenceDamping) else 0.0
        bt.thermodynamic_tendency_factor = td.thermodynamic_tendency_factor
        bt.vertical_momentum_tendency_factor = td.vertical_momentum_tendency_factor
        bt.newton_abstol = self.temperature_solver.abstol
        bt.substep_distribution = td.substep_distribution.code
        sft = self.timestepper.substepper.substep_floattype
        bt.substep_float_bytes = 4 if (sft is np.float32 and grid.ftype == 8) else 0
        if td.sponge is not None:
            bt.sponge_ramp, bt.sponge_damping_rate, bt.sponge_depth = td.sponge.ramp.code, td.sponge.damping_rate, td.sponge.depth
        self._ctx = C.c_void_p()
        rc = self._create_context(lib, bg, bc, br, bt, advection.order)
        if rc != 0:
            raise _lib.BreezeHIPError(f"bz_create_compressible failed with code {rc}")
        self._check(lib.bz_set_stream(self._ctx, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "bz_set_stream")
        self._state = self._make_state()
        self._U0, self._G = self._make_prog(self.U0), self._make_prog(self.G)
        self._sub = self.timestepper.substepper.struct()
        if self._sa:      # materialize_microphysical_fields(::WarmPhaseSaturationAdjustment): (q^v, q^l, q^e); q^e is the moisture slot
            self.microphysical_fields = {"qᵛ": Field(grid, _LOC["ccc"], self.device), "qˡ": Field(grid, _LOC["ccc"], self.device),
                                         "qᵉ": self.specific_moisture}
            sa = T.bz_saturation_adjustment(c.liquid_reference_latent_heat, c.liquid_heat_capacity,
                                               c.energy_reference_temperature, c.triple_point_temperature,
                                               c.triple_point_pressure, microphysics.solver.abstol,
                                               microphysics.solver.maxiter, 0)
            self._check(lib.bz_set_saturation_adjustment(self._ctx, C.byref(sa),
                                                         C.c_void_p(self.microphysical_fields["qᵛ"].ptr()),
                                                         C.c_void_p(self.microphysical_fields["qˡ"].ptr())),
                        "bz_set_saturation_adjustment")
        if self._kessler:
            from .microphysics import kessler_parameter_struct
            μ = self.microphysical_fields
            P = kessler_parameter_struct(microphysics, c, ftype=grid.ftype)
            K = T.bz_kessler_model_fields()
            K.cloud_liquid_density, K.rain_density = μ["ρqᶜˡ"].ptr(), μ["ρqʳ"].ptr()
            K.U0_cloud_liquid_density, K.U0_rain_density = self.U0["ρqᶜˡ"].ptr(), self.U0["ρqʳ"].ptr()
            K.G_cloud_liquid_density, K.G_rain_density = self.G["ρqᶜˡ"].ptr(), self.G["ρqʳ"].ptr()
            K.vapor_mass_fraction, K.cloud_liquid_mass_fraction, K.rain_mass_fraction = μ["qᵛ"].ptr(), μ["qᶜˡ"].ptr(), μ["qʳ"].ptr()
            K.rain_terminal_velocity, K.precipitation_rate = μ["𝕎ʳ"].ptr(), μ["precipitation_rate"].data_ptr()
            self._check(lib.bz_set_kessler_microphysics(self._ctx, C.byref(P), C.byref(K), dynamics.standard_pressure),
                        "bz_set_kessler_microphysics")
        if coriolis is not None:      # the f-plane term of the slow momentum tendencies (bz_set_forcings with coriolis_f alone)
            Fc = T.bz_column_forcings()
            Fc.coriolis_f = coriolis.f
            self._check(lib.bz_set_forcings(self._ctx, C.byref(Fc)), "bz_set_forcings")
        if self._relaxation:
            from .forcings import materialize_relaxation
            Rx, self._relaxation_keepalive = materialize_relaxation(grid, self._relaxation, "LiquidIcePotentialTemperature", T)
            self._check(lib.bz_set_relaxation(self._ctx, C.byref(Rx)), "bz_set_relaxation")
        from .forcings import materialize_field_forcing
        self.thermodynamic_forcing_field, _spec = materialize_field_forcing(grid, self._field_forcing, "LiquidIcePotentialTemperature", self.device)
        if self.thermodynamic_forcing_field is not None:
            self._check(lib.bz_set_field_forcing(self._ctx, C.c_void_p(self.thermodynamic_forcing_field.ptr()), _spec), "bz_set_field_forcing")
        if self.lateral_walls:      # is_active_open_bc of the four sides (acoustic_substepping.jl:1339-1361)
            bu, bv = self.boundary_conditions.get("ρu"), self.boundary_conditions.get("ρv")
            sides = [is_active_open_bc(getattr(b, k, None)) and grid.topology[d] == Bounded
                     for b, k, d in ((bu, "west", 0), (bu, "east", 0), (bv, "south", 1), (bv, "north", 1))]
            self._check(lib.bz_set_acoustic_lateral_boundaries(self._ctx, *(int(x) for x in sides), td.open_boundary_relaxation),
                        "bz_set_acoustic_lateral_boundaries")
        # seed_pressure! (compressible_dynamics.jl:254-258)
        if ref is not None:
            Hz, Nz = grid.Hz, grid.Nz
            dynamics.pressure.set_interior(ref.pressure[Hz:Hz + Nz][:, None, None])
        else:
            dynamics.pressure.set_interior(dynamics.surface_pressure)

    def _create_context(self, lib, bg, bc, br, bt, order):
        return lib.bz_create_compressible(C.byref(self._ctx), C.byref(bg), C.byref(bc), C.byref(br), C.byref(bt), order)

    def _exchange(self, tensors):
        """y-halo exchange hook: nothing to do on a single GPU (the kernels wrap periodically)."""

    def _check(self, rc, what):
        _lib.check(self._lib, self._ctx, rc, what)

    def prognostic_fields(self):
        out = {"ρᵈ": self.dynamics.dry_density, "ρu": self.momentum["ρu"], "ρv": self.momentum["ρv"],
               "ρw": self.momentum["ρw"], "ρθ": self.potential_temperature_density, "ρq": self.moisture_density}
        if getattr(self, "_kessler", False):
            out["ρqᶜˡ"], out["ρqʳ"] = self.microphysical_fields["ρqᶜˡ"], self.microphysical_fields["ρqʳ"]
        return out

    def _make_state(self):
        s = _lib.bz_compressible_state()
        d = self.dynamics
        s.rho_d, s.rho, s.p = d.dry_density.ptr(), d.total_density.ptr(), d.pressure.ptr()
        s.rho_u, s.rho_v, s.rho_w = (self.momentum[k].ptr() for k in ("ρu", "ρv", "ρw"))
        s.rho_theta, s.rho_q = self.potential_temperature_density.ptr(), self.moisture_density.ptr()
        s.u, s.v, s.w = (self.velocities[k].ptr() for k in ("u", "v", "w"))
        s.theta, s.q, s.T = self.potential_temperature.ptr(), self.specific_moisture.ptr(), self.temperature.ptr()
        return s

    @staticmethod
    def _make_prog(d):
        p = _lib.bz_compressible_prognostic()
        p.rho_d, p.rho_u, p.rho_v, p.rho_w, p.rho_theta, p.rho_q = (d[k].ptr() for k in ("ρᵈ", "ρu", "ρv", "ρw", "ρθ", "ρq"))
        return p

    def __del__(self):
        try:
            if getattr(self, "_ctx", None):
                self._lib.bz_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    def set(self, **kw):
        return set_(self, **kw)

    def time_step(self, Δt):
        return time_step_(self, Δt)

    def synchronize(self):
        self._check(self._lib.bz_sync(self._ctx), "bz_sync")

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
        out = {}
        for i in range(self._lib.bz_profile_count(self._ctx)):
            name, ms, n = C.c_char_p(), self._T.real(), C.c_int64()
            self._check(self._lib.bz_profile_get(self._ctx, i, C.byref(name), C.byref(ms), C.byref(n)), "bz_profile_get")
            out[name.value.decode()] = (ms.value, n.value)
        return out

    def stage_substeps(self, Δt, β):
        n, dτ = C.c_int32(), self._T.real()
        self._check(self._lib.bz_stage_substeps(self._ctx, float(Δt), float(β), C.byref(n), C.byref(dτ)), "bz_stage_substeps")
        return n.value, dτ.value


# ---------------------------------------------------------------------------------------------------------------------
def update_state_(model, compute_tendencies=True):
    if getattr(model, "decomp", None) is not None:
        return model.update_state_slab(compute_tendencies)
    model._check(model._lib.bz_compressible_update_state(model._ctx, C.byref(model._state), C.byref(model._G),
                                                         C.byref(model._sub), 1 if compute_tendencies else 0),
                 "bz_compressible_update_state")


def refresh_linearization_(model):
    model._check(model._lib.bz_refresh_linearization(model._ctx, C.byref(model._state), C.byref(model._sub)),
                 "bz_refresh_linearization")


def seed_time_averaged_velocities_(model):
    model._check(model._lib.bz_seed_time_averaged_velocities(model._ctx, C.byref(model._state), C.byref(model._sub)),
                 "bz_seed_time_averaged_velocities")


def compute_slow_tendencies_(model):
    model._check(model._lib.bz_compute_slow_tendencies(model._ctx, C.byref(model._state), C.byref(model._G)),
                 "bz_compute_slow_tendencies")


def acoustic_rk3_substep_loop_(model, Δt, β):
    model._check(model._lib.bz_acoustic_substep_loop(model._ctx, C.byref(model._state), C.byref(model._U0), C.byref(model._G),
                                                     C.byref(model._sub), float(Δt), float(β)), "bz_acoustic_substep_loop")


def acoustic_rk3_substep_(model, Δt, β):
    model._check(model._lib.bz_acoustic_rk3_substep(model._ctx, C.byref(model._state), C.byref(model._U0), C.byref(model._G),
                                                    C.byref(model._sub), float(Δt), float(β)), "bz_acoustic_rk3_substep")


def store_initial_state_(model):
    for k, f in model.prognostic_fields().items():
        model.U0[k].parent.copy_(f.parent)


def set_(model, **kw):
    """set!(model; ρ, θ, u, v, w, qᵗ): total density first, moisture, then establish_densities!, θ and velocities
    (set_atmosphere_model.jl:198-362; compressible_time_stepping.jl:105-150)."""
    if getattr(model, "lateral_walls", False):
        raise NotImplementedError("set!: a compressible model on a Bounded x / y runs the acoustic substep loop only (fill its fields directly)")
    d = model.dynamics
    g = model.grid
    keys = {}
    for name, value in kw.items():
        key = _ALIASES.get(name)
        if key is None:
            raise ValueError(f"Cannot set! {name} in AtmosphereModel because {name} is neither a prognostic variable, a "
                             "settable thermodynamic variable, nor a settable diagnostic variable!")
        keys[key] = value
    ρd, ρ = d.dry_density, d.total_density
    if "ρ" in keys:
        ρd.set_interior(keys["ρ"])
    if "q" in keys:
        model.specific_moisture.set_interior(keys["q"])
        model.moisture_density.interior.copy_(ρd.interior * model.specific_moisture.interior)
    condensate = 0.0
    if getattr(model, "_kessler", False):        # settable specific microphysical names: only ρq is set
        for key, dens in (("qcl", "ρqᶜˡ"), ("qr", "ρqʳ")):
            if key in keys:
                f = model.microphysical_fields[dens]
                f.set_interior(keys[key])
                f.interior.copy_(ρd.interior * f.interior)
        condensate = model.microphysical_fields["ρqᶜˡ"].interior + (model.microphysical_fields["ρqʳ"].interior + 0.0)
    elif "qcl" in keys or "qr" in keys:
        raise ValueError("Cannot set! qᶜˡ / qʳ: the model has no Kessler microphysics")
    if "ρ" in keys:      # establish_densities!(total_density_given)
        ρ.interior.copy_(ρd.interior)
        ρd.interior.copy_(ρ.interior - (model.moisture_density.interior + condensate))
    from .model import fill_halo_regions_
    fill_halo_regions_(model, ρd, 0)
    model._exchange([ρd.parent])
    if "θ" in keys:
        model.potential_temperature.set_interior(keys["θ"])
        model.potential_temperature_density.interior.copy_(ρd.interior * model.potential_temperature.interior)
    P = ρd.parent
    Hx, Hy, Hz, Nx, Ny, Nz = g.Hx, g.Hy, g.Hz, g.Nx, g.Ny, g.Nz
    if "u" in keys:
        model.velocities["u"].set_interior(keys["u"])
        ρx = (ρd.interior + P[Hz:Hz + Nz, Hy:Hy + Ny, Hx - 1:Hx - 1 + Nx]) / 2
        model.momentum["ρu"].interior.copy_(ρx * model.velocities["u"].interior)
    if "v" in keys:
        model.velocities["v"].set_interior(keys["v"])
        # ℑy of a Flat direction is the identity
        ρy = ρd.interior if g.topology[1] == Flat else (ρd.interior + P[Hz:Hz + Nz, Hy - 1:Hy - 1 + Ny, Hx:Hx + Nx]) / 2
        model.momentum["ρv"].interior.copy_(ρy * model.velocities["v"].interior)
    if "w" in keys:
        model.velocities["w"].set_interior(keys["w"])
        ρz = (P[Hz:Hz + Nz + 1, Hy:Hy + Ny, Hx:Hx + Nx] + P[Hz - 1:Hz + Nz, Hy:Hy + Ny, Hx:Hx + Nx]) / 2
        model.momentum["ρw"].interior.copy_(ρz * model.velocities["w"].interior)
    update_state_(model, compute_tendencies=False)


def time_step_(model, Δt, whole_step=True):
    """time_step!(model::CompressibleAcousticModel, Δt) (acoustic_runge_kutta_3.jl:264-319) without callbacks.
    whole_step=True keeps the step behind one C call; False issues the operator sequence of the reference."""
    Δt = float(Δt)
    if getattr(model, "lateral_walls", False):
        raise NotImplementedError("time_step!: a compressible model on a Bounded x / y runs the acoustic substep loop only")
    if model.clock.iteration == 0:                         # maybe_prepare_first_time_step!
        seed_time_averaged_velocities_(model)
        update_state_(model, compute_tendencies=True)
    if getattr(model, "decomp", None) is not None:
        model.time_step_slab(Δt)
    elif whole_step:
        model._check(model._lib.bz_time_step_compressible(model._ctx, C.byref(model._state), C.byref(model._U0),
                                                          C.byref(model._G), C.byref(model._sub), Δt),
                     "bz_time_step_compressible")
    else:
        store_initial_state_(model)
        refresh_linearization_(model)                      # freeze_linearization_state!
        seed_time_averaged_velocities_(model)
        for β in (model.timestepper.β1, model.timestepper.β2, model.timestepper.β3):
            acoustic_rk3_substep_(model, Δt, β)
            update_state_(model, compute_tendencies=True)
        if getattr(model, "_kessler", False):
            model._check(model._lib.bz_compressible_kessler_update(model._ctx, C.byref(model._state), C.byref(model._G),
                                                                   C.byref(model._sub), Δt), "bz_compressible_kessler_update")
    model.clock.time += Δt
    model.clock.last_Δt = Δt
    model.clock.iteration += 1


# ---------------------------------------------------------------------------------------------------------------------
# y-slab decomposition: one process per GPU, halo exchanges only — the acoustic column solve is rank-local
# ---------------------------------------------------------------------------------------------------------------------
class SlabCompressibleModel(CompressibleAtmosphereModel):
    """CompressibleAtmosphereModel on one y-slab of a (Periodic, Periodic, Bounded) global grid.  Same kernels as the
    single-GPU model; inside a stage the two perturbation fields the next substep reads across the slab edge
    ((ρθ)′ and (ρv)′) are exchanged with the ring neighbours, everything else once per stage
    (breeze.jl_amd/distributed.py: SlabDecomposition, torch.distributed; backend "nccl" = RCCL on ROCm)."""

    def __init__(self, global_grid, rank, world, dynamics, advection=None, group=None, device=None, decomp=None,
                 transport="torch", **kw):
        """transport: "torch" — exchanges issued from Python through torch.distributed (`decomp` / `group`); "rccl" or
        "local:<name>" — the library owns the communicator (csrc/bz_comm.hip) and `time_step` is one C call per step."""
        import torch
        from .distributed import LibraryComm, SlabDecomposition, attach_library_transport
        if global_grid.topology != (Periodic, Periodic, Bounded):
            raise NotImplementedError("slab decomposition implements topology (Periodic, Periodic, Bounded)")
        if global_grid.Ny % world:
            raise ValueError(f"Ny={global_grid.Ny} is not divisible by {world} ranks")
        self.global_grid = G = global_grid
        self.rank, self.world = rank, world
        sft = kw.get("substep_floattype")
        if sft is not None and np.dtype(sft) == np.float32 and G.ftype == 8 and transport == "torch":
            # the library-owned step moves the Float32 rows of the working fields (csrc/bz_comm.hip: halo_exchange, half); the
            # Python-issued exchange assumes the grid's real
            raise NotImplementedError('substep_floattype = Float32 on y-slabs: transport "rccl" or "local:<name>"')
        Ny = G.Ny // world
        y0 = G.yᶠ[0] + rank * Ny * G.Δy
        z = (G.zᶠ[0], G.zᶠ[-1]) if G.regular_z else G.zᶠ
        grid = RectilinearGrid((G.Nx, Ny, G.Nz), x=(G.xᶠ[0], G.xᶠ[0] + G.Nx * G.Δx), y=(y0, y0 + Ny * G.Δy), z=z,
                               halo=(G.Hx, G.Hy, G.Hz), float_type=G.float_type)      # eltype(grid) travels with the slab
        grid.Δx, grid.Δy = G.Δx, G.Δy          # bit-identical spacings on every rank
        self.decomp = None                     # set after construction: the constructor's set-up is rank-local
        self.transport = transport
        library = transport != "torch"
        self._pending_decomp = LibraryComm(self) if library else (decomp or SlabDecomposition(G.Nx, Ny, G.Nz, G.Hy, rank, world, group))
        dev = device if device is not None else f"cuda:{torch.cuda.current_device()}"
        super().__init__(grid, dynamics, advection=advection, device=dev, **kw)
        if library:
            attach_library_transport(self, transport, group)
        self.decomp = self._pending_decomp
        if getattr(self.decomp, "rows", "absent") is None:      # one gather / scatter launch per direction for the halo rows
            self.decomp.rows = self._device_rows
        # second buffers of the (ρu)′, (ρv)′ ping-pong are owned here so that their halos can be exchanged (the library-owned
        # step exchanges the context's own scratch buffers)
        self._up2, self._vp2 = Field(grid, _LOC["fcc"], self.device), Field(grid, _LOC["cfc"], self.device)
        if not library:
            self._check(self._lib.bz_set_acoustic_scratch(self._ctx, C.c_void_p(self._up2.ptr()), C.c_void_p(self._vp2.ptr())),
                        "bz_set_acoustic_scratch")
        sub = self.timestepper.substepper
        self._th_buf = (sub.density_potential_temperature_perturbation.parent,
                        sub.previous_density_potential_temperature_perturbation.parent)
        self._v_buf = (sub.momentum_perturbation_v.parent, self._vp2.parent)
        self._u_buf = (sub.momentum_perturbation_u.parent, self._up2.parent)
        self._direct = isinstance(self.dynamics.time_discretization.damping, DirectDivergenceDamping)
        self._exchange([self.dynamics.pressure.parent])

    def comm_info(self):
        """(transport name, bytes this rank has sent, number of exchanges) of the library-owned communicator."""
        name, nbytes, nex = C.c_char_p(), C.c_int64(), C.c_int32()
        self._check(self._lib.bz_comm_info(self._ctx, C.byref(name), C.byref(nbytes), C.byref(nex)), "bz_comm_info")
        return name.value.decode(), nbytes.value, nex.value

    def _create_context(self, lib, bg, bc, br, bt, order):
        return lib.bz_create_compressible_slab(C.byref(self._ctx), C.byref(bg), C.byref(bc), C.byref(br), C.byref(bt), order,
                                               self.world, self.rank)

    def _exchange(self, tensors):
        d = self.decomp or self._pending_decomp
        d.exchange_y_halos(list(tensors))

    def _device_rows(self, fields, row0, nrows, buffer, unpack):
        n = len(fields)
        ptrs = (C.c_void_p * n)(*[f.data_ptr() for f in fields])
        levels = (C.c_int32 * n)(*[f.shape[0] for f in fields])
        self._check(self._lib.bz_pack_rows(self._ctx, ptrs, levels, n, int(row0), int(nrows), C.c_void_p(buffer.data_ptr()),
                                           1 if unpack else 0), "bz_pack_rows")

    # field groups ----------------------------------------------------------------------------------------------------
    def _prognostic_tensors(self):
        sub = self.timestepper.substepper
        return ([f.parent for f in self.prognostic_fields().values()] +
                [sub.time_averaged_u.parent, sub.time_averaged_v.parent, sub.time_averaged_w.parent])

    def _diagnostic_tensors(self):
        d = self.dynamics
        μ = [self.microphysical_fields[k].parent for k in ("qᵛ", "qᶜˡ", "qʳ")] if self._kessler else []
        if self._sa:       # the halo rows' linearisation reads the liquid fraction
            μ = [self.microphysical_fields[k].parent for k in ("qᵛ", "qˡ")]
        return ([d.total_density.parent, d.pressure.parent] + [self.velocities[k].parent for k in ("u", "v", "w")] +
                [self.potential_temperature.parent, self.specific_moisture.parent, self.temperature.parent] + μ)

    def update_state_slab(self, compute_tendencies=True):
        """update_state! with the neighbour exchanges of the slab decomposition.  The diagnosis needs ρᵈ of row -1 (face
        velocities); everything else is exchanged after it, when the x halos of the edge rows (written by the diagnosis
        kernel) are valid, so that the corner cells the WENO stencils touch arrive with the rows."""
        if self.transport != "torch":
            self._check(self._lib.bz_comm_compressible_update_state(self._ctx, C.byref(self._state), C.byref(self._G), C.byref(self._sub),
                                                                    1 if compute_tendencies else 0), "bz_comm_compressible_update_state")
            return
        self._exchange([self.dynamics.dry_density.parent])
        self._check(self._lib.bz_compressible_update_state(self._ctx, C.byref(self._state), C.byref(self._G), C.byref(self._sub), 0),
                    "bz_compressible_update_state")
        self._exchange(self._prognostic_tensors() + self._diagnostic_tensors())
        if compute_tendencies:
            self._check(self._lib.bz_compute_moisture_tendency(self._ctx, C.byref(self._state), C.byref(self._G), C.byref(self._sub)),
                        "bz_compute_moisture_tendency")

    def acoustic_rk3_substep_slab(self, Δt, β):
        """acoustic_rk3_substep!(model, Δt, β) on a slab (call order in include/breeze_hip.h, slab section)."""
        lib, ctx = self._lib, self._ctx
        st, U0, G, sub = C.byref(self._state), C.byref(self._U0), C.byref(self._G), C.byref(self._sub)
        refresh_linearization_(self)
        compute_slow_tendencies_(self)
        self._exchange([self.G["ρv"].parent])
        n, cur = C.c_int32(), C.c_int32()
        self._check(lib.bz_acoustic_stage_begin(ctx, st, U0, G, sub, float(Δt), float(β), C.byref(n), C.byref(cur)),
                    "bz_acoustic_stage_begin")
        for s in range(1, n.value + 1):
            self._exchange([self._th_buf[cur.value], self._v_buf[cur.value]])
            self._check(lib.bz_acoustic_substep(ctx, st, U0, G, sub, s, C.byref(cur)), "bz_acoustic_substep")
            if self._direct:      # apply_divergence_damping!(::DirectDivergenceDamping): needs the neighbours' freshly advanced rows
                self._exchange([self._u_buf[cur.value], self._v_buf[cur.value]])
                self._check(lib.bz_acoustic_direct_damping(ctx, st, U0, G, sub), "bz_acoustic_direct_damping")
        self._exchange([self._th_buf[cur.value]])
        self._check(lib.bz_acoustic_stage_end(ctx, st, U0, G, sub, float(Δt), float(β), 1), "bz_acoustic_stage_end")

    def time_step_slab(self, Δt):
        if self.transport != "torch":       # the whole distributed step behind one call (bz_comm.hip: bzi_dist_time_step_compressible)
            self._check(self._lib.bz_time_step_compressible(self._ctx, C.byref(self._state), C.byref(self._U0), C.byref(self._G),
                                                            C.byref(self._sub), float(Δt)), "bz_time_step_compressible")
            return
        store_initial_state_(self)
        for β in (self.timestepper.β1, self.timestepper.β2, self.timestepper.β3):
            self.acoustic_rk3_substep_slab(Δt, β)
            self.update_state_slab(compute_tendencies=True)
        if self._kessler:      # microphysics_model_update!: rank-local columns, then the exchanging update_state!
            self._check(self._lib.bz_compressible_kessler_update(self._ctx, C.byref(self._state), C.byref(self._G),
                                                                 C.byref(self._sub), float(Δt)), "bz_compressible_kessler_update")
            self.update_state_slab(compute_tendencies=True)
