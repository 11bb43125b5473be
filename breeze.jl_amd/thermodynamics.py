"""Host-side mirror of the Breeze.Thermodynamics pieces the anelastic path needs at set-up time:
ThermodynamicConstants (src/Thermodynamics/thermodynamics_constants.jl:182-217) and the dry
adiabatic ReferenceState columns (src/Thermodynamics/reference_states.jl:84-123, 326-330, 402-445).
Pure numpy Float64; evaluated once per model."""
import numpy as np


class ThermodynamicConstants:
    def __init__(self, molar_gas_constant=8.314462618, gravitational_acceleration=9.81,
                 energy_reference_temperature=273.15, triple_point_temperature=273.16,
                 triple_point_pressure=611.657, dry_air_molar_mass=0.02897, dry_air_heat_capacity=1005,
                 vapor_molar_mass=0.018015, vapor_heat_capacity=1850,
                 liquid_reference_latent_heat=2500800, liquid_heat_capacity=4181,
                 ice_reference_latent_heat=2834000, ice_heat_capacity=2108, saturation_vapor_pressure=None):
        # saturation_vapor_pressure: None = ClausiusClapeyron (default) or a TetensFormula (breeze.jl_amd/microphysics.py)
        self.saturation_vapor_pressure = saturation_vapor_pressure
        self.molar_gas_constant = float(molar_gas_constant)
        self.gravitational_acceleration = float(gravitational_acceleration)
        self.energy_reference_temperature = float(energy_reference_temperature)
        self.triple_point_temperature = float(triple_point_temperature)
        self.triple_point_pressure = float(triple_point_pressure)
        self.dry_air_molar_mass = float(dry_air_molar_mass)
        self.dry_air_heat_capacity = float(dry_air_heat_capacity)
        self.vapor_molar_mass = float(vapor_molar_mass)
        self.vapor_heat_capacity = float(vapor_heat_capacity)
        # CondensedPhase liquid_water / water_ice (thermodynamics_constants.jl:87-93)
        self.liquid_reference_latent_heat = float(liquid_reference_latent_heat)
        self.liquid_heat_capacity = float(liquid_heat_capacity)
        self.ice_reference_latent_heat = float(ice_reference_latent_heat)
        self.ice_heat_capacity = float(ice_heat_capacity)


def dry_air_gas_constant(c):
    return c.molar_gas_constant / c.dry_air_molar_mass


def vapor_gas_constant(c):
    return c.molar_gas_constant / c.vapor_molar_mass


def surface_density(p0, θ0, pst, c):
    Rd, cpd = dry_air_gas_constant(c), c.dry_air_heat_capacity
    Π0 = (p0 / pst) ** (Rd / cpd)
    T0 = Π0 * θ0
    return p0 / (Rd * T0)


def adiabatic_hydrostatic_pressure(z, p0, θ0, pst, c):
    cpd, Rd, g = c.dry_air_heat_capacity, dry_air_gas_constant(c), c.gravitational_acceleration
    T0 = θ0 * (p0 / pst) ** (Rd / cpd)
    return p0 * (1 - g * z / (cpd * T0)) ** (cpd / Rd)


def adiabatic_hydrostatic_density(z, p0, θ0, pst, c):
    Rd, cpd = dry_air_gas_constant(c), c.dry_air_heat_capacity
    pr = adiabatic_hydrostatic_pressure(z, p0, θ0, pst, c)
    ρ0 = surface_density(p0, θ0, pst, c)
    return ρ0 * (pr / p0) ** (1 - Rd / cpd)


def hydrostatic_temperature(z, p0, θ0, pst, c):
    κ = dry_air_gas_constant(c) / c.dry_air_heat_capacity
    p = adiabatic_hydrostatic_pressure(z, p0, θ0, pst, c)
    return θ0 * (p / pst) ** κ


class ReferenceState:
    """ReferenceState(grid, constants; surface_pressure=101325, potential_temperature=288,
    standard_pressure=1e5): dry adiabatic hydrostatic columns with Oceananigans halos
    (bottom ValueBoundaryCondition for density/pressure, zero-gradient otherwise; first halo cell)."""

    def __init__(self, grid, constants=None, surface_pressure=101325, potential_temperature=288,
                 standard_pressure=1e5):
        if callable(potential_temperature):
            raise NotImplementedError("θᵣ(z) profiles (numerical hydrostatic integration) are not implemented")
        c = constants or ThermodynamicConstants()
        self.constants = c
        self.surface_pressure = p0 = float(surface_pressure)
        self.potential_temperature = θ0 = float(potential_temperature)
        self.standard_pressure = pst = float(standard_pressure)
        Nz, Hz = grid.Nz, grid.Hz
        self.Nz, self.Hz = Nz, Hz
        self.surface_density = surface_density(p0, θ0, pst, c)
        z = grid.zᶜ
        self.density = np.zeros(Nz + 2 * Hz)
        self.pressure = np.zeros(Nz + 2 * Hz)
        self.temperature = np.zeros(Nz + 2 * Hz)
        self.density[Hz:Hz + Nz] = adiabatic_hydrostatic_density(z, p0, θ0, pst, c)
        self.pressure[Hz:Hz + Nz] = adiabatic_hydrostatic_pressure(z, p0, θ0, pst, c)
        self.temperature[Hz:Hz + Nz] = hydrostatic_temperature(z, p0, θ0, pst, c)
        self.fill_halo_regions()

    def fill_halo_regions(self):
        Nz, Hz = self.Nz, self.Hz
        ρ, p, T = self.density, self.pressure, self.temperature
        ρ[Hz - 1] = 2 * self.surface_density - ρ[Hz]
        p[Hz - 1] = 2 * self.surface_pressure - p[Hz]
        T[Hz - 1] = T[Hz]
        ρ[Hz + Nz], p[Hz + Nz], T[Hz + Nz] = ρ[Hz + Nz - 1], p[Hz + Nz - 1], T[Hz + Nz - 1]
