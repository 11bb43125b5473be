"""breeze.jl_amd — MI355X-native (gfx950) hot path for Breeze.jl's anelastic AtmosphereModel.

Contents: csrc/ (hand-written HIP kernels + the C ABI of include/breeze_hip.h), lib/ (the built
libbreeze_hip.so) and the host-side mirror of the reference's model API (grids, thermodynamics, model).

The directory name contains a dot, so it is loaded through `breeze_jl_amd` (see /breeze_jl_amd.py at the
repo root) rather than a plain `import`.
"""
from ._lib import BreezeHIPError, LIB_PATH, SYMBOLS, build, load  # noqa: F401
from .grids import Bounded, Center, Face, Flat, Periodic, RectilinearGrid  # noqa: F401
from .thermodynamics import ReferenceState, ThermodynamicConstants  # noqa: F401
from .model import (AnelasticDynamics, AtmosphereModel, Centered, Field, WENO, compute_auxiliary_thermodynamic_variables_,  # noqa: F401
                    compute_pressure_correction_, compute_scalar_tendency_, compute_tendencies_, compute_velocities_,
                    enforce_mass_conservation_, fill_halo_regions_, make_pressure_correction_, set_,
                    ssp_rk3_substep_, store_initial_state_, time_step_, update_state_)
from . import compressible  # noqa: F401,E402
from .compressible import (AcousticRungeKutta3, AcousticSubstepper, CompressibleAtmosphereModel, CompressibleDynamics,  # noqa: F401,E402
                           ExnerReferenceState, NewtonSolver, NoDivergenceDamping, ProportionalSubsteps, ConstantSubstepSize, MonolithicFirstStage,
                           SplitExplicitTimeDiscretization, ThermalDivergenceDamping, DirectDivergenceDamping, UpperSponge,
                           LinearRamp, CubicRamp, Sin2Ramp, NormalFlowBoundaryCondition)
from .microphysics import SaturationAdjustment, SecantSolver, WarmPhaseEquilibrium  # noqa: F401,E402
from .model import cell_advection_timescale, diagnostics_stale, many_time_steps_, nan_checker  # noqa: F401,E402
from .microphysics import (DCMIP2016KesslerMicrophysics, KesslerMicrophysicalFields, TetensFormula,  # noqa: F401,E402
                           microphysics_model_update_)
from .forcings import (BulkDrag, BulkSensibleHeatFlux, BulkVaporFlux, FPlane, FieldBoundaryConditions, FluxBoundaryCondition, Forcing, FrictionVelocityDrag,  # noqa: F401,E402
                       GaussianMask, GeostrophicForcing, Relaxation, SmagorinskyLilly, SubsidenceForcing, geostrophic_forcings)
from .model import compute_closure_fields_, compute_flux_bc_tendencies_  # noqa: F401,E402
from . import benchmarks  # noqa: F401,E402
