"""Host-side mirror of the reference's benchmark cases (BreezeBenchmarks, /root/reference/benchmarking/src/).

`convective_boundary_layer` follows benchmarking/src/convective_boundary_layer.jl:59-185 keyword by keyword: the dry convective
boundary layer of Sauer & Munoz-Esparza (2020) section 4.2 on a 12 km x 12 km x 3 km box with 5-cell halos — AnelasticDynamics on a
309 K reference state, FPlane at 33.5 N, geostrophic forcing (9, 0) m/s, a 0.35 K m/s surface heat flux and the u* = 0.4 m/s
drag boundary conditions on rho u, rho v — in eltype Float32 by default, as every GPU benchmark of the reference runs
(.github/workflows/Benchmarks.yml:34-45: grids 256x256x128, 512x512x256, 768x768x256, WENO5 and WENO9, 5 warm-up + 150 steps).
The reference draws its +-0.25 K perturbation from `rand()`; here it comes from a seeded generator so that the device model and the
CPU oracle (tests/test_cbl.py) start from the same bits."""
import numpy as np

from .forcings import (FPlane, FieldBoundaryConditions, FluxBoundaryCondition, FrictionVelocityDrag, geostrophic_forcings)
from .grids import Bounded, Periodic, RectilinearGrid
from .model import AnelasticDynamics, AtmosphereModel, WENO
from .thermodynamics import ReferenceState, ThermodynamicConstants, dry_air_gas_constant

CBL = dict(Lx=12e3, Ly=12e3, Lz=3e3, p0=101325.0, theta0=309.0, latitude=33.5, Ug=9.0, Vg=0.0, heat_flux=0.35, ustar=0.4,
           drag_epsilon=1e-10, z_inversion=600.0, lapse_rate=0.004, perturbation=0.25, z_perturbation=400.0)


def cbl_coriolis_parameter():
    return 2 * 7.2921e-5 * np.sin(np.deg2rad(CBL["latitude"]))      # convective_boundary_layer.jl:114-116


def cbl_surface_density(constants=None):
    """rho0 = density(theta0, p0, q = 0) = p0 / (R^d theta0) (convective_boundary_layer.jl:125-127)"""
    c = constants or ThermodynamicConstants()
    return CBL["p0"] / (dry_air_gas_constant(c) * CBL["theta0"])


def cbl_initial_theta(size, seed=0):
    """(Nz, Ny, Nx) interior array of theta_i: theta0 below 600 m, +0.004 K/m above, +-0.25 K noise in the lowest 400 m"""
    Nx, Ny, Nz = size
    zc = (np.arange(Nz) + 0.5) * (CBL["Lz"] / Nz)
    noise = np.random.default_rng(seed).uniform(-1.0, 1.0, (Nz, Ny, Nx))
    base = CBL["theta0"] + np.maximum(0.0, zc - CBL["z_inversion"]) * CBL["lapse_rate"]
    return base[:, None, None] + CBL["perturbation"] * noise * (zc < CBL["z_perturbation"])[:, None, None]


def convective_boundary_layer(size=(64, 64, 64), float_type=np.float32, advection=None, closure=None, device="cuda:0", seed=0,
                              halo=(5, 5, 5), topology=(Periodic, Periodic, Bounded)):
    """AtmosphereModel of the CBL benchmark case with its initial condition set (simplified = false branch of the reference)."""
    Nx, Ny, Nz = size
    grid = RectilinearGrid((Nx, Ny, Nz), x=(0.0, CBL["Lx"]), y=(0.0, CBL["Ly"]), z=(0.0, CBL["Lz"]), halo=halo,
                           topology=topology, float_type=float_type)      # PBB = (Periodic, Bounded, Bounded): run_benchmarks.jl:130
    constants = ThermodynamicConstants()
    ref = ReferenceState(grid, constants, surface_pressure=CBL["p0"], potential_temperature=CBL["theta0"])
    rho0 = cbl_surface_density(constants)
    geo = geostrophic_forcings(lambda z: CBL["Ug"], lambda z: CBL["Vg"])
    drag = FieldBoundaryConditions(bottom=FluxBoundaryCondition(FrictionVelocityDrag(rho0, CBL["ustar"], epsilon=CBL["drag_epsilon"])))
    model = AtmosphereModel(grid, dynamics=AnelasticDynamics(ref), advection=advection or WENO(order=5), closure=closure,
                            coriolis=FPlane(f=cbl_coriolis_parameter()), forcing={"u": geo.u, "v": geo.v},
                            boundary_conditions={"ρθ": FieldBoundaryConditions(bottom=FluxBoundaryCondition(rho0 * CBL["heat_flux"])),
                                                 "ρu": drag, "ρv": drag},
                            thermodynamic_constants=constants, device=device)
    model.set(θ=cbl_initial_theta(size, seed), u=CBL["Ug"], v=CBL["Vg"])
    return model
