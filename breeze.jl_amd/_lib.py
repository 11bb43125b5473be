"""ctypes binding of libbreeze_hip.so (C ABI declared in include/breeze_hip.h).

The product path has no CPU fallback: if the HIP library is missing or fails to load, importing
this module's `load()` raises.  Nothing here imports the test oracle.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# BREEZE_HIP_LIB selects another build of the same ABI (e.g. lib/libbreeze_hip_refdiv.so, the variant that
# keeps the reference's three-quotient WENO weight formula, used for A/B timing and tight parity checks).
LIB_PATH = os.environ.get("BREEZE_HIP_LIB") or os.path.join(_HERE, "lib", "libbreeze_hip.so")
CSRC = os.path.join(_HERE, "csrc")

_dp = C.POINTER(C.c_double)
BZ_UNIQUE_ID_BYTES = 128


class bz_grid(C.Structure):
    _fields_ = [("Nx", C.c_int32), ("Ny", C.c_int32), ("Nz", C.c_int32),
                ("Hx", C.c_int32), ("Hy", C.c_int32), ("Hz", C.c_int32),
                ("topo", C.c_int32 * 3), ("ftype", C.c_int32),
                ("dx", C.c_double), ("dy", C.c_double),
                ("zf", _dp), ("regular_z", C.c_int32), ("reserved", C.c_int32)]


class bz_constants(C.Structure):
    _fields_ = [("gravitational_acceleration", C.c_double),
                ("dry_air_gas_constant", C.c_double),
                ("vapor_gas_constant", C.c_double),
                ("dry_air_heat_capacity", C.c_double),
                ("vapor_heat_capacity", C.c_double)]


class bz_reference_state(C.Structure):
    _fields_ = [("surface_pressure", C.c_double), ("potential_temperature", C.c_double),
                ("standard_pressure", C.c_double),
                ("density", _dp), ("pressure", _dp), ("temperature", _dp)]


_STATE_FIELDS = ("rho_u", "rho_v", "rho_w", "rho_theta", "rho_q", "u", "v", "w", "theta", "q", "T", "phi")
_PROG_FIELDS = ("rho_u", "rho_v", "rho_w", "rho_theta", "rho_q")


class bz_state(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _STATE_FIELDS]


class bz_prognostic(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _PROG_FIELDS]


class bz_saturation_adjustment(C.Structure):
    _fields_ = [("liquid_latent_heat", C.c_double), ("liquid_heat_capacity", C.c_double),
                ("energy_reference_temperature", C.c_double), ("triple_point_temperature", C.c_double),
                ("triple_point_pressure", C.c_double), ("abstol", C.c_double), ("maxiter", C.c_int32),
                ("reserved", C.c_int32)]


class bz_bulk_surface_fluxes(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "drag_coefficient", "drag_gustiness", "drag_surface_temperature",
        "heat_coefficient", "heat_gustiness", "heat_surface_temperature",
        "vapor_coefficient", "vapor_gustiness", "vapor_surface_temperature",
        "surface_pressure", "standard_pressure",
        "liquid_latent_heat", "liquid_heat_capacity", "energy_reference_temperature", "triple_point_temperature",
        "triple_point_pressure")]


class bz_tracer_fields(C.Structure):
    _fields_ = [("density", C.c_void_p), ("specific", C.c_void_p), ("U0", C.c_void_p), ("G", C.c_void_p)]


class bz_bounds_preserving_advection(C.Structure):
    _fields_ = [("lower", C.c_double), ("upper", C.c_double), ("moisture", C.c_int32), ("microphysical_species", C.c_int32),
                ("tracers", C.c_int32), ("reserved", C.c_int32)]


class bz_smagorinsky_lilly(C.Structure):
    _fields_ = [("smagorinsky_coefficient", C.c_double), ("reduction_factor", C.c_double), ("prandtl_number", C.c_double)]


class bz_column_forcings(C.Structure):
    _fields_ = [("u_forcing", _dp), ("v_forcing", _dp), ("theta_forcing", _dp), ("moisture_forcing", _dp),
                ("energy_forcing", _dp), ("subsidence_vertical_velocity", _dp),
                ("subsidence_u", C.c_int32), ("subsidence_v", C.c_int32), ("subsidence_theta", C.c_int32),
                ("subsidence_moisture", C.c_int32), ("coriolis_f", C.c_double),
                ("bottom_theta_flux", C.c_double), ("bottom_moisture_flux", C.c_double),
                ("bottom_drag_rho0_ustar2", C.c_double), ("bottom_drag_epsilon", C.c_double),
                ("bottom_energy_flux", C.c_double)]


class bz_column_relaxation(C.Structure):
    _fields_ = [(f"{a}_{b}", _dp) for b in ("u", "v", "w", "theta", "moisture") for a in ("rate", "target")] + [("specific_mask", C.c_int32)]


_KESSLER_PARAMS = ("dcmip_temperature_scale", "terminal_velocity_coefficient", "density_scale", "terminal_velocity_exponent",
                   "autoconversion_rate", "autoconversion_threshold", "accretion_rate", "accretion_exponent",
                   "evaporation_ventilation_coefficient_1", "evaporation_ventilation_coefficient_2",
                   "evaporation_ventilation_exponent_1", "evaporation_ventilation_exponent_2", "diffusivity_coefficient",
                   "thermal_conductivity_coefficient", "substep_cfl", "tetens_reference_saturation_vapor_pressure",
                   "tetens_reference_temperature", "tetens_liquid_coefficient", "tetens_liquid_temperature_offset",
                   "liquid_latent_heat", "liquid_heat_capacity")
_KESSLER_FIELDS = ("density", "pressure", "potential_temperature", "potential_temperature_density", "moisture_density",
                   "cloud_liquid_density", "rain_density", "vapor_mass_fraction", "cloud_liquid_mass_fraction",
                   "rain_mass_fraction", "rain_terminal_velocity", "precipitation_rate")


class bz_kessler_microphysics(C.Structure):
    _fields_ = [(n, C.c_double) for n in _KESSLER_PARAMS]


class bz_kessler_fields(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _KESSLER_FIELDS]


_KESSLER_MODEL_FIELDS = ("cloud_liquid_density", "rain_density", "U0_cloud_liquid_density", "U0_rain_density",
                         "G_cloud_liquid_density", "G_rain_density", "vapor_mass_fraction", "cloud_liquid_mass_fraction",
                         "rain_mass_fraction", "rain_terminal_velocity", "precipitation_rate")


class bz_kessler_model_fields(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _KESSLER_MODEL_FIELDS]


_CSTATE_FIELDS = ("rho_d", "rho", "rho_u", "rho_v", "rho_w", "rho_theta", "rho_q", "u", "v", "w", "theta", "q", "T", "p")
_CPROG_FIELDS = ("rho_d", "rho_u", "rho_v", "rho_w", "rho_theta", "rho_q")
_SUBSTEPPER_FIELDS = ("exner", "potential_temperature", "gamma_R_mixture", "density_perturbation",
                      "density_potential_temperature_perturbation", "momentum_perturbation_u", "momentum_perturbation_v",
                      "momentum_perturbation_w", "density_predictor", "density_potential_temperature_predictor",
                      "previous_density_potential_temperature_perturbation", "time_averaged_u", "time_averaged_v",
                      "time_averaged_w", "slow_vertical_momentum_tendency", "vertical_solver_source_term")


class bz_compressible_state(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _CSTATE_FIELDS]


class bz_compressible_prognostic(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _CPROG_FIELDS]


class bz_acoustic_substepper(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _SUBSTEPPER_FIELDS]


class bz_split_explicit(C.Structure):
    _fields_ = [("substeps", C.c_int32), ("damp_vertical", C.c_int32),
                ("apply_first_substep_pressure_gradient", C.c_int32), ("newton_maxiter", C.c_int32),
                ("acoustic_cfl", C.c_double), ("forward_weight", C.c_double), ("damping_coefficient", C.c_double),
                ("thermodynamic_tendency_factor", C.c_double), ("vertical_momentum_tendency_factor", C.c_double),
                ("newton_abstol", C.c_double), ("direct_divergence_damping", C.c_int32), ("sponge_ramp", C.c_int32),
                ("sponge_damping_rate", C.c_double), ("sponge_depth", C.c_double),
                ("substep_distribution", C.c_int32), ("substep_float_bytes", C.c_int32), ("damping_length_scale", C.c_double)]


class bz_exner_reference_state(C.Structure):
    _fields_ = [("standard_pressure", C.c_double), ("pressure", _dp), ("density", _dp)]


# every symbol include/breeze_hip.h declares: name -> (restype, argtypes)
_ctx = C.c_void_p
_sp, _pp = C.POINTER(bz_state), C.POINTER(bz_prognostic)
_csp, _cpp, _asp = C.POINTER(bz_compressible_state), C.POINTER(bz_compressible_prognostic), C.POINTER(bz_acoustic_substepper)
SYMBOLS = {
    "bz_create": (C.c_int, [C.POINTER(_ctx), C.POINTER(bz_grid), C.POINTER(bz_constants),
                            C.POINTER(bz_reference_state), C.c_int]),
    "bz_destroy": (None, [_ctx]),
    "bz_set_formulation": (C.c_int, [_ctx, C.c_int]),
    "bz_set_saturation_adjustment": (C.c_int, [_ctx, C.POINTER(bz_saturation_adjustment), C.c_void_p, C.c_void_p]),
    "bz_set_stream": (C.c_int, [_ctx, C.c_void_p]),
    "bz_graph_enable": (C.c_int, [_ctx, C.c_int]),
    "bz_graph_info": (C.c_int, [_ctx, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "bz_sync": (C.c_int, [_ctx]),
    "bz_last_error": (C.c_char_p, [_ctx]),
    "bz_fill_halo_regions": (C.c_int, [_ctx, C.c_void_p, C.c_int]),
    "bz_compute_velocities": (C.c_int, [_ctx, _sp]),
    "bz_compute_auxiliary_thermodynamic_variables": (C.c_int, [_ctx, _sp]),
    "bz_compute_tendencies": (C.c_int, [_ctx, _sp, _pp]),
    "bz_compute_scalar_tendency": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bz_set_scalar_advection_order": (C.c_int, [_ctx, C.c_int]),
    "bz_update_state": (C.c_int, [_ctx, _sp, _pp, C.c_int]),
    "bz_store_initial_state": (C.c_int, [_ctx, _sp, _pp]),
    "bz_ssp_rk3_substep": (C.c_int, [_ctx, _sp, _pp, _pp, C.c_double, C.c_double]),
    "bz_compute_pressure_correction": (C.c_int, [_ctx, _sp, C.c_double]),
    "bz_make_pressure_correction": (C.c_int, [_ctx, _sp, C.c_double]),
    "bz_time_step_anelastic": (C.c_int, [_ctx, _sp, _pp, _pp, C.c_double]),
    "bz_time_steps_anelastic": (C.c_int, [_ctx, _sp, _pp, _pp, C.c_double, C.c_int, C.c_int]),
    "bz_diagnostics_stale": (C.c_int, [_ctx]),
    "bz_create_slab": (C.c_int, [C.POINTER(_ctx), C.POINTER(bz_grid), C.POINTER(bz_constants),
                                 C.POINTER(bz_reference_state), C.c_int, C.c_int, C.c_int]),
    "bz_slab_info": (C.c_int, [_ctx] + [C.POINTER(C.c_int32)] * 5),
    "bz_ssp_rk3_substep_fused": (C.c_int, [_ctx, _sp, _pp, _pp, C.c_double, C.c_double, C.c_int]),
    "bz_poisson_source_term": (C.c_int, [_ctx, _sp, C.c_double, C.c_void_p]),
    "bz_spectral_tridiagonal_solve": (C.c_int, [_ctx, C.c_void_p, C.c_double]),
    "bz_project_and_diagnose": (C.c_int, [_ctx, _sp, C.c_void_p, C.c_void_p, C.c_double]),
    "bz_tendencies_fused_rk": (C.c_int, [_ctx, _sp, _pp, _pp, C.c_double, C.c_double, C.c_int]),
    "bz_poisson_source_term_from": (C.c_int, [_ctx, _sp, _pp, C.c_double, C.c_void_p]),
    "bz_project_and_diagnose_from": (C.c_int, [_ctx, _sp, _pp, C.c_void_p, C.c_void_p, C.c_double]),
    "bz_create_compressible": (C.c_int, [C.POINTER(_ctx), C.POINTER(bz_grid), C.POINTER(bz_constants),
                                         C.POINTER(bz_exner_reference_state), C.POINTER(bz_split_explicit), C.c_int]),
    "bz_compressible_update_state": (C.c_int, [_ctx, _csp, _cpp, _asp, C.c_int]),
    "bz_refresh_linearization": (C.c_int, [_ctx, _csp, _asp]),
    "bz_seed_time_averaged_velocities": (C.c_int, [_ctx, _csp, _asp]),
    "bz_compute_slow_tendencies": (C.c_int, [_ctx, _csp, _cpp]),
    "bz_stage_substeps": (C.c_int, [_ctx, C.c_double, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "bz_acoustic_substep_loop": (C.c_int, [_ctx, _csp, _cpp, _cpp, _asp, C.c_double, C.c_double]),
    "bz_acoustic_rk3_substep": (C.c_int, [_ctx, _csp, _cpp, _cpp, _asp, C.c_double, C.c_double]),
    "bz_time_step_compressible": (C.c_int, [_ctx, _csp, _cpp, _cpp, _asp, C.c_double]),
    "bz_create_compressible_slab": (C.c_int, [C.POINTER(_ctx), C.POINTER(bz_grid), C.POINTER(bz_constants),
                                              C.POINTER(bz_exner_reference_state), C.POINTER(bz_split_explicit), C.c_int,
                                              C.c_int, C.c_int]),
    "bz_acoustic_stage_begin": (C.c_int, [_ctx, _csp, _cpp, _cpp, _asp, C.c_double, C.c_double, C.POINTER(C.c_int32),
                                          C.POINTER(C.c_int32)]),
    "bz_acoustic_substep": (C.c_int, [_ctx, _csp, _cpp, _cpp, _asp, C.c_int32, C.POINTER(C.c_int32)]),
    "bz_acoustic_direct_damping": (C.c_int, [_ctx, _csp, _cpp, _cpp, _asp]),
    "bz_acoustic_stage_end": (C.c_int, [_ctx, _csp, _cpp, _cpp, _asp, C.c_double, C.c_double, C.c_int]),
    "bz_set_acoustic_scratch": (C.c_int, [_ctx, C.c_void_p, C.c_void_p]),
    "bz_set_acoustic_lateral_boundaries": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]),
    "bz_compute_moisture_tendency": (C.c_int, [_ctx, _csp, _cpp, _asp]),
    "bz_kessler_microphysics_update": (C.c_int, [_ctx, C.POINTER(bz_kessler_microphysics), C.POINTER(bz_kessler_fields),
                                                 C.c_double, C.c_double]),
    "bz_set_kessler_microphysics": (C.c_int, [_ctx, C.POINTER(bz_kessler_microphysics), C.POINTER(bz_kessler_model_fields),
                                              C.c_double]),
    "bz_kessler_model_update": (C.c_int, [_ctx, _sp, _pp, C.c_double]),
    "bz_compressible_kessler_update": (C.c_int, [_ctx, _csp, _cpp, _asp, C.c_double]),
    "bz_pack_transpose": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "bz_pack_rows": (C.c_int, [_ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]),
    "bz_slab_transform": (C.c_int, [_ctx, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]),
    "bz_comm_unique_id": (C.c_int, [C.c_void_p]),
    "bz_comm_init_rccl": (C.c_int, [_ctx, C.c_void_p]),
    "bz_comm_init_local": (C.c_int, [_ctx, C.c_char_p]),
    "bz_comm_destroy": (C.c_int, [_ctx]),
    "bz_comm_exchange_y_halos": (C.c_int, [_ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int32]),
    "bz_comm_update_state_and_project": (C.c_int, [_ctx, _sp, _pp, C.c_double, C.c_int]),
    "bz_comm_info": (C.c_int, [_ctx, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "bz_comm_compressible_update_state": (C.c_int, [_ctx, _csp, _cpp, _asp, C.c_int]),
    "bz_set_tracers": (C.c_int, [_ctx, C.c_int32, C.POINTER(bz_tracer_fields)]),
    "bz_set_closure": (C.c_int, [_ctx, C.POINTER(bz_smagorinsky_lilly), C.c_void_p]),
    "bz_set_bounds_preserving_advection": (C.c_int, [_ctx, C.POINTER(bz_bounds_preserving_advection)]),
    "bz_compute_closure_fields": (C.c_int, [_ctx, _sp]),
    "bz_set_bulk_surface_fluxes": (C.c_int, [_ctx, C.POINTER(bz_bulk_surface_fluxes)]),
    "bz_set_forcings": (C.c_int, [_ctx, C.POINTER(bz_column_forcings)]),
    "bz_set_relaxation": (C.c_int, [_ctx, C.POINTER(bz_column_relaxation)]),
    "bz_set_field_forcing": (C.c_int, [_ctx, C.c_void_p, C.c_int]),
    "bz_compute_forcings": (C.c_int, [_ctx, _sp]),
    "bz_compute_flux_bc_tendencies": (C.c_int, [_ctx, _sp, _pp]),
    "bz_cell_advection_timescale": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
    "bz_any_nan": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.POINTER(C.c_int32)]),
    "bz_profile_enable": (C.c_int, [_ctx, C.c_int]),
    "bz_profile_reset": (C.c_int, [_ctx]),
    "bz_profile_count": (C.c_int, [_ctx]),
    "bz_profile_get": (C.c_int, [_ctx, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                 C.POINTER(C.c_int64)]),
    "bz_max_abs_divergence": (C.c_int, [_ctx, _sp, C.POINTER(C.c_double)]),
}


def build(verbose=False):
    """Compile every HIP source for gfx950 into lib/libbreeze_hip.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if out.returncode != 0:
        raise RuntimeError("building libbreeze_hip.so failed:\n" + out.stdout)
    if verbose:
        print(out.stdout)
    return LIB_PATH


_lib = None


CENTERED2_LIB_PATH = os.path.join(_HERE, "lib", "libbreeze_hip_centered2.so")
_libs = {}


def load(advection_order=5, ft2_hypothesis=0):
    """Load libbreeze_hip.so (advection = WENO(order = 5)) or, for advection_order = 2, libbreeze_hip_centered2.so — the same
    sources built with the reconstructions collapsed to Centered(order = 2) — and bind every declared symbol.  ft2_hypothesis = 1 | 2:
    lib/libbreeze_hip_ft2_<level>.so, the WENO kernels built with BZ_WENO_FT2 (csrc/bz_weno.h).  Raises if the library is absent."""
    global _lib
    path = CENTERED2_LIB_PATH if advection_order == 2 else LIB_PATH      # WENO orders 5, 7, 9 live in the same library
    if ft2_hypothesis and advection_order != 2:
        path = os.path.join(_HERE, "lib", f"libbreeze_hip_ft2_{int(ft2_hypothesis)}.so")
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP extension is required (no CPU fallback). "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C breeze.jl_amd/csrc`.")
    # torch bundles its own HIP runtime; load it first so that libbreeze_hip.so binds to the same
    # libamdhip64 instance (two runtimes in one process cannot both see the device).
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _libs[path] = lib
    if advection_order != 2 and not ft2_hypothesis:
        _lib = lib
    return lib


# ---- Float32 twin (lib/libbreeze_hip_f32.so, generated by tools/gen_f32_sources.py: every double of the ABI is a float) --------
F32_LIB_PATH = os.environ.get("BREEZE_HIP_F32_LIB") or os.path.join(_HERE, "lib", "libbreeze_hip_f32.so")      # (BREEZE_HIP_F32_LIB: A/B builds of the Float32 twin)
_f32_structs = {}


def _f32_type(t):
    """The Float32 ABI's counterpart of a ctypes type of the Float64 ABI."""
    if t is C.c_double:
        return C.c_float
    if t is _dp:
        return C.POINTER(C.c_float)
    if isinstance(t, type) and issubclass(t, C.Structure):
        if t not in _f32_structs:
            fields = [(n, _f32_type(ft)) for n, ft in t._fields_]
            same = all(a[1] is b[1] for a, b in zip(fields, t._fields_))          # pointer-only structs are shared by both ABIs
            _f32_structs[t] = t if same else type(t.__name__, (C.Structure,), {"_fields_": fields})
        return _f32_structs[t]
    if isinstance(t, type) and issubclass(t, C._Pointer):
        inner = _f32_type(t._type_)
        return t if inner is t._type_ else C.POINTER(inner)
    if isinstance(t, type) and issubclass(t, C.Array):
        return t
    return t


class _Types:
    """Struct classes and the scalar type of one precision: types(8).bz_grid is bz_grid, types(4).bz_grid its Float32 mirror."""

    def __init__(self, ftype):
        self.ftype = ftype
        self.real = C.c_double if ftype == 8 else C.c_float
        self.np_real = "float64" if ftype == 8 else "float32"

    def __getattr__(self, name):
        cls = globals()[name]
        return cls if self.ftype == 8 else _f32_type(cls)


def types(ftype=8):
    if ftype not in (4, 8):
        raise ValueError("float type must be 4 (Float32) or 8 (Float64) bytes")
    return _Types(ftype)


def load_f32():
    """libbreeze_hip_f32.so with the Float32 signatures (WENO(order = 5) build)."""
    if F32_LIB_PATH in _libs:
        return _libs[F32_LIB_PATH]
    if not os.path.exists(F32_LIB_PATH):
        raise RuntimeError(f"{F32_LIB_PATH} not found: run `make -C breeze.jl_amd/csrc` (no CPU fallback)")
    import torch  # noqa: F401
    lib = C.CDLL(F32_LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = [_f32_type(a) for a in args]
    _libs[F32_LIB_PATH] = lib
    return lib


class BreezeHIPError(RuntimeError):
    pass


def check(lib, ctx, rc, what):
    if rc != 0:
        msg = lib.bz_last_error(ctx).decode() if ctx else ""
        raise BreezeHIPError(f"{what} failed with code {rc}: {msg}")
