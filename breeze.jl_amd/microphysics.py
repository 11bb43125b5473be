"""Host-side mirror of Breeze.Microphysics.SaturationAdjustment (src/Microphysics/saturation_adjustment.jl:20-60) and the
solver vocabulary of Breeze.Solvers (src/Solvers.jl:55-136): warm-phase equilibrium with the secant iteration."""


class WarmPhaseEquilibrium:
    pass


class MixedPhaseEquilibrium:
    def __init__(self, *a, **k):
        raise NotImplementedError("MixedPhaseEquilibrium is not implemented in the HIP path (warm phase only)")


class SecantSolver:
    def __init__(self, reltol=0, abstol=1e-4, maxiter=20):
        if reltol != 0:
            raise NotImplementedError("SecantSolver(reltol != 0) is not implemented in the HIP path")
        self.reltol, self.abstol, self.maxiter = 0.0, float(abstol), int(maxiter)


class SaturationAdjustment:
    """SaturationAdjustment(; solver = SecantSolver(abstol=1e-4, maxiter=20), equilibrium)."""

    def __init__(self, solver=None, equilibrium=None):
        if equilibrium is None or not isinstance(equilibrium, WarmPhaseEquilibrium):
            raise NotImplementedError("the HIP path implements SaturationAdjustment(equilibrium = WarmPhaseEquilibrium())")
        self.equilibrium = equilibrium
        self.solver = solver or SecantSolver()


# ---------------------------------------------------------------------------------------------------------------------
# DCMIP2016 Kessler warm-rain microphysics: the operator-split column update (src/Microphysics/dcmip2016_kessler.jl)
# ---------------------------------------------------------------------------------------------------------------------
class TetensFormula:
    """TetensFormula(; reference_saturation_vapor_pressure=610, reference_temperature=273.15, liquid_coefficient=17.27,
    liquid_temperature_offset=35.85, ...) (src/Thermodynamics/tetens_formula.jl:72-85); liquid surface only here."""

    def __init__(self, reference_saturation_vapor_pressure=610, reference_temperature=273.15, liquid_coefficient=17.27,
                 liquid_temperature_offset=35.85):
        self.reference_saturation_vapor_pressure = float(reference_saturation_vapor_pressure)
        self.reference_temperature = float(reference_temperature)
        self.liquid_coefficient = float(liquid_coefficient)
        self.liquid_temperature_offset = float(liquid_temperature_offset)


class DCMIP2016KesslerMicrophysics:
    """DCMIP2016KesslerMicrophysics(; ...) with the reference's defaults (dcmip2016_kessler.jl:154-169)."""

    DEFAULTS = dict(dcmip_temperature_scale=237.3, terminal_velocity_coefficient=36.34, density_scale=0.001,
                    terminal_velocity_exponent=0.1364, autoconversion_rate=0.001, autoconversion_threshold=0.001,
                    accretion_rate=2.2, accretion_exponent=0.875, evaporation_ventilation_coefficient_1=1.6,
                    evaporation_ventilation_coefficient_2=124.9, evaporation_ventilation_exponent_1=0.2046,
                    evaporation_ventilation_exponent_2=0.525, diffusivity_coefficient=2.55e8,
                    thermal_conductivity_coefficient=5.4e5, substep_cfl=0.8)

    def __init__(self, **kw):
        for k, v in self.DEFAULTS.items():
            setattr(self, k, float(kw.pop(k, v)))
        if kw:
            raise TypeError(f"unknown DCMIP2016KesslerMicrophysics parameters: {sorted(kw)}")


class KesslerMicrophysicalFields:
    """materialize_microphysical_fields(::DCMIP2016KesslerMicrophysics, grid, bcs) (dcmip2016_kessler.jl:255-290):
    prognostic rho q^cl, rho q^r and the diagnostic q^v, q^cl, q^r, W^r, precipitation_rate."""

    def __init__(self, model):
        import torch
        from .model import Field, _LOC
        g = model.grid
        # ASCII attribute names (Python normalises superscript identifiers): rho_qcl = ρqᶜˡ, rho_qr = ρqʳ, qv, qcl, qr, W = 𝕎ʳ
        for name in ("rho_qcl", "rho_qr", "qv", "qcl", "qr", "W"):
            setattr(self, name, Field(g, _LOC["ccc"], model.device))
        self.precipitation_rate = torch.zeros((g.Ny + 2 * g.Hy, g.Nx + 2 * g.Hx), dtype=self.qv.parent.dtype, device=model.device)      # eltype(grid): the Float32 library writes 4-byte reals


def kessler_parameter_struct(microphysics, constants, tetens=None, ftype=8):
    """bz_kessler_microphysics from DCMIP2016KesslerMicrophysics + the TetensFormula / liquid phase of the constants."""
    from . import _lib
    tf = tetens or getattr(constants, "saturation_vapor_pressure", None) or TetensFormula()
    P = _lib.types(ftype).bz_kessler_microphysics()
    for k in DCMIP2016KesslerMicrophysics.DEFAULTS:
        setattr(P, k, getattr(microphysics, k))
    P.tetens_reference_saturation_vapor_pressure = tf.reference_saturation_vapor_pressure
    P.tetens_reference_temperature = tf.reference_temperature
    P.tetens_liquid_coefficient = tf.liquid_coefficient
    P.tetens_liquid_temperature_offset = tf.liquid_temperature_offset
    P.liquid_latent_heat, P.liquid_heat_capacity = constants.liquid_reference_latent_heat, constants.liquid_heat_capacity
    return P


def microphysics_model_update_(microphysics, model, fields=None, Δt=None, tetens=None, density=None, pressure=None,
                               standard_pressure=None):
    """microphysics_model_update!(microphysics::DCMIP2016KesslerMicrophysics, model) (dcmip2016_kessler.jl:449-486) without
    the trailing update_state!: one launch of the column kernel.  `density` / `pressure`: 3-D Fields (compressible) or None
    for the anelastic reference columns of the model's context."""
    import ctypes as C
    from . import _lib
    if fields is None:          # the model owns the Kessler fields: microphysics_model_update!(microphysics, model)
        Δt = model.clock.last_Δt if Δt is None else Δt
        model._check(model._lib.bz_kessler_model_update(model._ctx, C.byref(model._state), C.byref(model._G), float(Δt)),
                     "bz_kessler_model_update")
        return
    P = kessler_parameter_struct(microphysics, model.thermodynamic_constants, tetens, ftype=getattr(model.grid, "ftype", 8))
    F = _lib.bz_kessler_fields()
    F.density = density.ptr() if density is not None else None
    F.pressure = pressure.ptr() if pressure is not None else None
    F.potential_temperature = model.potential_temperature.ptr()
    F.potential_temperature_density = model.potential_temperature_density.ptr()
    F.moisture_density = model.moisture_density.ptr()
    F.cloud_liquid_density, F.rain_density = fields.rho_qcl.ptr(), fields.rho_qr.ptr()
    F.vapor_mass_fraction, F.cloud_liquid_mass_fraction, F.rain_mass_fraction = fields.qv.ptr(), fields.qcl.ptr(), fields.qr.ptr()
    F.rain_terminal_velocity = fields.W.ptr()
    F.precipitation_rate = fields.precipitation_rate.data_ptr()
    if standard_pressure is None:
        ref = getattr(model.dynamics, "reference_state", None)
        standard_pressure = getattr(model.dynamics, "standard_pressure", None) or ref.standard_pressure
    model._check(model._lib.bz_kessler_microphysics_update(model._ctx, C.byref(P), C.byref(F), float(Δt), float(standard_pressure)),
                 "bz_kessler_microphysics_update")
