/*
 * breeze_hip.h — C ABI of libbreeze_hip.so: the MI355X-native (gfx950) replacement for the
 * inner time-stepping hot path of Breeze.jl's anelastic AtmosphereModel.
 *
 * The reference has no FFI for this path: every operator is a Julia method chosen by multiple
 * dispatch and executed as KernelAbstractions kernels.  Each entry point below replaces one of
 * the Julia generic functions a package extension would specialise on a marker architecture
 * (precedent: /root/reference/ext/BreezeReactantExt/Timesteppers.jl:6-19) and forward with
 * `ccall((:bz_..., libbreeze_hip), ...)`; see INTEGRATION.md for the Julia stub.
 *
 * Conventions
 *  - Plain C types only.  All field pointers are DEVICE pointers to the *parent*
 *    (halo-inclusive) array of an Oceananigans Field, column-major, i fastest:
 *        element (i,j,k) [1-based interior] at p[(i-1+Hx) + Sx*((j-1+Hy) + Sy*(k-1+Hz))]
 *        Sx = Nx+2Hx, Sy = Ny+2Hy;  z-Face fields on Bounded z have Nz+1 levels.
 *    Column (reference-state) and grid arrays passed to bz_create are HOST pointers; they are
 *    copied at creation.
 *  - Every call returns 0 on success, a BZ_ERR_* code or a negated hipError_t / hipfftResult
 *    otherwise; nothing throws; bz_last_error() returns a message for the calling thread's ctx.
 *  - Calls are asynchronous and stream-ordered on the stream set by bz_set_stream (default: the
 *    HIP null stream).  bz_sync() blocks until the stream is idle.  No internal host threads.
 *  - The caller owns all field memory; the ctx owns FFT plans, column tables and scratch.
 *  - One ctx per device/stream.  Float64 only (ftype = 8) in this version.
 */
#ifndef BREEZE_HIP_H
#define BREEZE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BZ_PERIODIC 0
#define BZ_BOUNDED 1
#define BZ_FLAT 2

#define BZ_OK 0
#define BZ_ERR_INVALID 1      /* null pointer / inconsistent sizes */
#define BZ_ERR_UNSUPPORTED 2  /* topology, float type, halo or scheme not implemented */
#define BZ_ERR_ALLOC 3

/* Oceananigans RectilinearGrid (regular x, y; regular or stretched z). */
typedef struct bz_grid {
    int32_t Nx, Ny, Nz;
    int32_t Hx, Hy, Hz;   /* halo sizes (>= 3 for WENO order 5) */
    int32_t topo[3];      /* BZ_PERIODIC / BZ_BOUNDED / BZ_FLAT per direction */
    int32_t ftype;        /* sizeof(eltype(grid)): 8 */
    double dx, dy;        /* regular horizontal spacings */
    const double *zf;     /* HOST: Nz+1 z-face heights */
    int32_t regular_z;    /* 1: use dz = (zf[Nz]-zf[0])/Nz exactly, as Oceananigans regular grids do */
    int32_t reserved;
} bz_grid;

/* Breeze ThermodynamicConstants subset used on the dry/vapour path
 * (src/Thermodynamics/thermodynamics_constants.jl:182-217). */
typedef struct bz_constants {
    double gravitational_acceleration;
    double dry_air_gas_constant;      /* R / Md */
    double vapor_gas_constant;        /* R / Mv */
    double dry_air_heat_capacity;
    double vapor_heat_capacity;
} bz_constants;

/* Breeze ReferenceState columns (src/Thermodynamics/reference_states.jl:18-28, 402-445):
 * HOST arrays of length Nz+2Hz holding parent(field)[1,1,:] including filled halos. */
typedef struct bz_reference_state {
    double surface_pressure, potential_temperature, standard_pressure;
    const double *density;
    const double *pressure;
    const double *temperature;
} bz_reference_state;

/* model.momentum, formulation/moisture prognostics, velocities, diagnostics, pressure_anomaly
 * (src/AtmosphereModels/atmosphere_model.jl:37-61; src/AnelasticEquations/anelastic_dynamics.jl:5-8). */
typedef struct bz_state {
    double *rho_u, *rho_v, *rho_w;   /* model.momentum            (XFace, YFace, ZFace) */
    double *rho_theta;               /* formulation.potential_temperature_density        */
    double *rho_q;                   /* model.moisture_density                           */
    double *u, *v, *w;               /* model.velocities                                 */
    double *theta;                   /* formulation.potential_temperature                */
    double *q;                       /* specific_prognostic_moisture (qv)                */
    double *T;                       /* model.temperature                                */
    double *phi;                     /* dynamics.pressure_anomaly  (p'/rho_r)            */
} bz_state;

/* A NamedTuple of prognostic-shaped fields: timestepper.Gn or timestepper.U0
 * (src/TimeSteppers/ssp_runge_kutta_3.jl:53-60, 87-88). */
typedef struct bz_prognostic {
    double *rho_u, *rho_v, *rho_w, *rho_theta, *rho_q;
} bz_prognostic;

typedef struct bz_ctx bz_ctx;

/* dynamics_pressure_solver + materialize_advection: builds FFT plans, tridiagonal factors and
 * column tables.  weno_order: 5 (the tuned kernels), or 7 / 9 (WENO(order = 7 | 9), the examples' scheme: generic kernels, halos >=
 * (order + 1) / 2, anelastic or compressible model without bounds-preserving advection, operator-by-operator anelastic stepping; Centered(order = 2)
 * is weno_order = 2 of libbreeze_hip_centered2.so); topology (Periodic, Periodic, Bounded) or — single-GPU contexts —
 * (Periodic, Flat, Bounded) with Ny = 1, Hy = 0 (the reference's 2-D x-z cases, anelastic and compressible), (Bounded, Flat, Bounded) — the same
 * with walls in x (examples/cloudy_thermal_bubble.jl:20-24; anelastic, even Nx <= 4096, u / rho u with wall faces i = 0 and Nx, cosine transform
 * along x) — or (Periodic, Bounded, Bounded) —
 * walls in y, the reference benchmark driver's PBB option (benchmarking/run_benchmarks.jl:130): anelastic WENO(order = 5 | 7 | 9) contexts (no bounds-preserving
 * advection) stepped operator by operator, rho v / v with impenetrable wall faces j = 0 and Ny (face Ny in the first upper halo row), cosine transform along
 * y in the pressure solve (Nx a power of two in [16, 1024] or 3 * 2^m, Ny a multiple of 8 up to 4096); halos >= 3, Float64; every extent at least its
 * halo (Oceananigans' own N >= H rule — the reference's 4 x 4 x 4 smoke-test boxes work, odd extents too); anything else returns
 * BZ_ERR_UNSUPPORTED.  Grids with Nx < 2 Hx or Ny < 2 Hy run the per-operator kernels instead of the fused tiers.
 * Replaces: AtmosphereModels.dynamics_pressure_solver (src/AnelasticEquations/anelastic_pressure_solver.jl:11-24)
 *           and compute_main_diagonal!/compute_lower_diagonal! (:32-78). */
int bz_create(bz_ctx **ctx, const bz_grid *grid, const bz_constants *constants,
              const bz_reference_state *reference_state, int weno_order);
void bz_destroy(bz_ctx *ctx);
/* Thermodynamic formulation of the anelastic model (AtmosphereModel(...; formulation)): 0 = :LiquidIcePotentialTemperature
 * (default), 1 = :StaticEnergy (src/StaticEnergyFormulations/static_energy_formulation.jl:18-21,70-95,
 * static_energy_tendency.jl:39-72).  With 1 the `rho_theta` / `theta` slots of bz_state and bz_prognostic carry
 * energy_density (rho e) / specific_energy (e): T = (e - g z)/c_pm and G_rho_e = -div_rhoUc(e) - Iz(w * Iz(buoyancy)). */
int bz_set_formulation(bz_ctx *ctx, int formulation);
/* microphysics = SaturationAdjustment(equilibrium = WarmPhaseEquilibrium(), solver = SecantSolver(abstol, maxiter))
 * (src/Microphysics/saturation_adjustment.jl:20-60,82-86,168-235): the moisture prognostic is the equilibrium moisture
 * rho q^e (bz_state.rho_q / q), T comes from the secant iteration on the adjusted state, and q^v, q^l are diagnosed into
 * model.microphysical_fields.q^v / q^l (device parent arrays given here), which the buoyancy then reads
 * (grid_moisture_fractions, :127-131).  Constants: liquid CondensedPhase, energy reference temperature and triple point of
 * ThermodynamicConstants (src/Thermodynamics/thermodynamics_constants.jl:92,182-194).  params == NULL: microphysics = nothing.
 * On a CompressibleDynamics context (bz_create_compressible) the same call attaches the density-based adjustment
 * (adjust_thermodynamic_state(::LiquidIceDensityState, ::SaturationAdjustment), src/Microphysics/saturation_adjustment.jl:236-301):
 * saturation at the cell's own total density, one Newton temperature for dynamics and microphysics. */
typedef struct bz_saturation_adjustment {
    double liquid_latent_heat, liquid_heat_capacity;
    double energy_reference_temperature, triple_point_temperature, triple_point_pressure;
    double abstol;            /* SecantSolver abstol (default 1e-4), reltol = 0 */
    int32_t maxiter;          /* default 20 */
    int32_t reserved;
} bz_saturation_adjustment;
int bz_set_saturation_adjustment(bz_ctx *ctx, const bz_saturation_adjustment *params, double *q_vapor, double *q_liquid);
int bz_set_stream(bz_ctx *ctx, void *hip_stream);
int bz_sync(bz_ctx *ctx);
const char *bz_last_error(const bz_ctx *ctx);

/* fill_halo_regions! for one field (Oceananigans.BoundaryConditions; call sites
 * src/AtmosphereModels/update_atmosphere_model_state.jl:48,135-136,152,241-243).
 * kind: 0 = centre-in-z field, default (no-flux) z BCs;  1 = z-face field, impenetrable walls;
 *       2 = `nothing` z BCs (diagnostic velocities): periodic wrap only; 3 = as 2 for a z-face field.
 *       + 4 for a field on y faces (rho v, v): on a Bounded y its wall faces j = 0, Ny are set to 0, where a centre-in-y field takes
 *       its first halo row from the adjacent interior row (ignored on a Periodic y); + 8 likewise for a field on x faces (rho u, u). */
int bz_fill_halo_regions(bz_ctx *ctx, double *field, int kind);

/* AtmosphereModels.compute_velocities! (update_atmosphere_model_state.jl:122-155, kernel :248-254). */
int bz_compute_velocities(bz_ctx *ctx, const bz_state *s);
/* compute_auxiliary_thermodynamic_variables! (:225-246, kernel :256-292). */
int bz_compute_auxiliary_thermodynamic_variables(bz_ctx *ctx, const bz_state *s);
/* AtmosphereModels.compute_tendencies! (:294-387): G.rho_u, rho_v, rho_w, rho_theta, rho_q. */
int bz_compute_tendencies(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G);
/* scalar_tendency of ONE field: Gc = -div_rhoUc(c) from the velocities u, v, w (all halo-filled) — compute_scalar_tendency!
 * (update_atmosphere_model_state.jl:390-393, dynamics_kernel_functions.jl:132-159 without forcing), the kernel group the reference's
 * own micro-benchmark times (benchmarking/src/scalar_tendency.jl:16-25: Gc = -div_Uc; here with the reference density of the
 * anelastic model, rho_r = const on that benchmark's 1 m deep box).  WENO order of the context (5, 7 or 9). */
int bz_compute_scalar_tendency(bz_ctx *ctx, const double *u, const double *v, const double *w, const double *c, double *Gc);
/* AtmosphereModel(grid; momentum_advection, scalar_advection) with schemes of different orders (atmosphere_model.jl:80-82,126-127,148-158;
 * examples/tropical_cyclone_world.jl:167-169 and examples/prescribed_sea_surface_temperature.jl:72-73: momentum WENO(order = 9), every
 * scalar WENO(order = 5)): bz_create's weno_order is the momentum scheme, this sets the order (5, 7 or 9) of every scalar — rho theta /
 * rho e, moisture, microphysical species, tracers.  Halos must cover both schemes.  Orders that differ run operator by operator (one
 * kernel per field, the reference's own launch list) on anelastic contexts (y-slabs: the operator-by-operator distributed step);
 * bounds-preserving advection stays a WENO(order = 5)
 * scalar scheme.  Call before the first step. */
int bz_set_scalar_advection_order(bz_ctx *ctx, int order);
/* TimeSteppers.update_state!(model; compute_tendencies) (:41-68); G may be NULL iff compute_tendencies == 0. */
int bz_update_state(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G, int compute_tendencies);

/* store_initial_state! (src/TimeSteppers/ssp_runge_kutta_3.jl:180-186). */
int bz_store_initial_state(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0);
/* ssp_rk3_substep! (:114-173): u = (1-alpha) u0 + alpha (u + dt G) for the five prognostic fields. */
int bz_ssp_rk3_substep(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0,
                       const bz_prognostic *G, double dt, double alpha);

/* AtmosphereModels.compute_pressure_correction!(model, dt)
 * (src/AnelasticEquations/anelastic_time_stepping.jl:26-39 -> anelastic_pressure_solver.jl:84-105):
 * momentum halos, source term, Fourier-tridiagonal solve into s->phi, phi halos. */
int bz_compute_pressure_correction(bz_ctx *ctx, const bz_state *s, double dt);
/* AtmosphereModels.make_pressure_correction!(model, dt) (anelastic_time_stepping.jl:45-78). */
int bz_make_pressure_correction(bz_ctx *ctx, const bz_state *s, double dt);

/* OceananigansTimeSteppers.time_step!(model::AtmosphereModel{..SSPRungeKutta3}, dt)
 * (src/TimeSteppers/ssp_runge_kutta_3.jl:209-278) for dry anelastic dynamics without callbacks:
 * the whole step stays on the device (preferred seam).  The state must be consistent
 * (bz_update_state with compute_tendencies=1 called once after set!, as
 * maybe_prepare_first_time_step! does).
 * NOTE: this seam folds each ssp_rk3_substep! into the preceding tendency evaluation, so on return the G arrays hold
 * the last stage's predictor momentum instead of tendencies (they are scratch of the time stepper in the reference as
 * well); bz_compute_tendencies / bz_update_state rebuild them, and bz_ssp_rk3_substep does so automatically.
 * U0 CONTRACT: the U0 arrays are the time stepper's scratch and their contents are UNDEFINED after a step.  The reference fills them
 * with the step-start state (store_initial_state!) and nothing outside time_step! reads them; the lean tier never copies the state into
 * them (the state arrays stay intact as U0 until the last writer of the step) and parks stage-1/2 momentum and stage-2 scalars there,
 * the compressible step stores interior cells only.  No entry point of this library reads U0 (or its halos) across calls. */
int bz_time_step_anelastic(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0,
                           const bz_prognostic *G, double dt);

/* n calls of time_step!(model, dt) with nothing reading the model in between: the loop of the reference's benchmark driver,
 * many_time_steps! (benchmarking/src/timestepping.jl:11-16), and of run!(simulation) between two callback / output iterations
 * (an IterationInterval- / TimeInterval-aware host calls it with n = the distance to the next actuation).
 * On the lean whole-step tier (dry / vapour-only theta-formulation models, WENO order 5) the tendency kernels derive u, v, w, theta,
 * q^v, T from the prognostic fields, so every step but the last ends with the momentum-only projection and skips the projection +
 * diagnosis pass whose 13 output words per cell only host consumers read; the other tiers diagnose every step.
 *   diagnose_last != 0: on return every field and halo of `s` holds the bits n calls of bz_time_step_anelastic leave.
 *   diagnose_last == 0: the prognostic state is current but the diagnostics (and pressure_anomaly) are stale and rho_theta / rho_q may be
 *     parked in G->rho_theta / G->rho_q (the kernels' ping-pong partner).  Stepping may simply continue (either entry point);
 *     before anything else reads `s`, call bz_update_state(ctx, s, G, 0).  bz_diagnostics_stale(ctx) reports the flag.
 *     Entry points of this library that read the diagnostics and are handed the state rebuild them first (bz_compute_tendencies,
 *     bz_compute_closure_fields, bz_compute_forcings, bz_compute_flux_bc_tendencies, bz_kessler_model_update); bz_cell_advection_timescale,
 *     which takes bare velocity pointers, returns BZ_ERR_INVALID while the flag is set; bz_sync joins a pending halo exchange of a slab context. */
int bz_time_steps_anelastic(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0,
                            const bz_prognostic *G, double dt, int n, int diagnose_last);
int bz_diagnostics_stale(const bz_ctx *ctx);

/* ---- y-slab decomposition: one process per GPU --------------------------------------------------
 * The reference re-exports Oceananigans' Distributed architecture (src/Breeze.jl:172,183,209) and has no
 * distributed code of its own; these entry points are the rank-local kernels of this repo's decomposition.
 * `local_grid` describes this rank's slab: Ny = rows owned (global Ny = Ny * y_nranks), y halos are filled by the
 * caller's neighbour exchange instead of a periodic wrap.  The caller also performs the horizontal transforms of
 * the Poisson solve (x-FFT, all-to-all to kx slabs, y-FFT and back); this rank's spectral block is
 * [Nz][nkx][Ny_global] complex (ky fastest, so the y transform is contiguous) for kx in [kx0, kx0+nkx) of the
 * zero-padded half spectrum. */
int bz_create_slab(bz_ctx **ctx, const bz_grid *local_grid, const bz_constants *constants,
                   const bz_reference_state *reference_state, int weno_order, int y_nranks, int y_rank);
int bz_slab_info(bz_ctx *ctx, int32_t *y_nranks, int32_t *y_rank, int32_t *nkx, int32_t *kx0, int32_t *ny_global);
/* ssp_rk3_substep! with store_initial_state! folded into the first stage (first != 0 requires alpha == 1). */
int bz_ssp_rk3_substep_fused(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0, const bz_prognostic *G,
                             double dt, double alpha, int first);
/* compute_anelastic_source_term! into rhs (device, contiguous Nx*Ny*Nz); in slab mode row Ny of rho_v must hold
 * the neighbour's first row. */
int bz_poisson_source_term(bz_ctx *ctx, const bz_state *s, double dt, double *rhs);
/* Thomas solve along z of every horizontal wavenumber of this rank's spectral block, in place; `scale` multiplies
 * the right-hand side (inverse-transform normalisation); the (0,0) column gets its z-mean removed. */
int bz_spectral_tridiagonal_solve(bz_ctx *ctx, double *hat, double scale);
/* Transposing pack of the distributed transform (complex Float64, interleaved re/im): in is (Nz, A, ld), out (Nz, B, A),
 * out[k][b][a] = in[k][a][c0 + b], zero where c0 + b >= valid.  One call per destination rank builds its send block. */
int bz_pack_transpose(bz_ctx *ctx, const double *in, double *out, int32_t Nz, int32_t A, int32_t ld, int32_t c0, int32_t B,
                      int32_t valid);
/* The local transforms of the distributed Poisson solve (1-D batched rocFFT plans, unnormalised): which = 0 x forward
 * (real (Nz, Ny, Nx) -> complex (Nz, Ny, Nx/2+1)), 1 / 2 y forward / backward in place on this rank's (Nz, nkx, Ny_global) block,
 * 3 x backward (complex (Nz, Ny, ld), ld >= Nx/2+1 -> real (Nz, Ny, Nx); overwrites its input).  ld is read for which = 3 only. */
int bz_slab_transform(bz_ctx *ctx, int32_t which, double *in, double *out, int32_t ld);
/* y-halo exchange helper of the slab drivers: gathers (unpack = 0) or scatters (unpack != 0) parent rows [row0, row0 + nrows) of n
 * parent arrays (levels[m] z-levels each: Nz + 2 Hz for centre fields, Nz + 1 + 2 Hz for z-face fields; full x width) into / out of
 * one contiguous device buffer, field-major then (level, row, x).  n <= 24. */
int bz_pack_rows(bz_ctx *ctx, double *const *fields, const int32_t *levels, int32_t n, int32_t row0, int32_t nrows, double *buffer,
                 int32_t unpack);
/* make_pressure_correction! + compute_velocities! + thermodynamic diagnosis + x/z halo fills in one pass from the
 * contiguous solution phi_c (Nx*Ny*Nz); phi_below = phi of row j = -1, layout [k][i] (slab mode; else NULL). */
int bz_project_and_diagnose(bz_ctx *ctx, const bz_state *s, const double *phi_c, const double *phi_below, double dt);

/* compute_tendencies! with the next ssp_rk3_substep! folded into the kernels' store: predictor momentum goes to
 * G->rho_u/v/w (to be projected by bz_project_and_diagnose_from), rho_theta / rho_q advance in place, U0 is filled when
 * first != 0 (alpha must then be 1).  The *_from variants take the momentum predictor from `predictor` instead of s. */
int bz_tendencies_fused_rk(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0, const bz_prognostic *G,
                           double dt, double alpha, int first);
int bz_poisson_source_term_from(bz_ctx *ctx, const bz_state *s, const bz_prognostic *predictor, double dt, double *rhs);
int bz_project_and_diagnose_from(bz_ctx *ctx, const bz_state *s, const bz_prognostic *predictor, const double *phi_c,
                                 const double *phi_below, double dt);

/* ---- communication of the y-slab decomposition inside the library (BASELINE.json north_star: RCCL halo exchange and FFT
 * all-to-all behind the C ABI).  With a communicator attached to a slab context, bz_time_step_anelastic IS the distributed step:
 * the lean whole-step seam with, per stage, one row of predictor rho_v from the upper neighbour, the Poisson solve as local x
 * transform -> all-to-all of transposed blocks -> y transform + Thomas solve -> all-to-all back -> x transform, one row of phi
 * from the lower neighbour, and the Hy-row halo exchange of (rho_u, rho_v, rho_w, rho_theta, rho_q) on a side stream while the
 * next stage's interior tile rows compute.  Every exchange is a group of point-to-point messages (one per xGMI link).
 *   bz_comm_unique_id      rank 0: a 128-byte id (ncclGetUniqueId) that the host hands to every rank (MPI.Bcast, a file, ...)
 *   bz_comm_init_rccl      every rank, on its own device: ncclCommInitRank(y_nranks, id, y_rank); librccl is dlopen'ed
 *   bz_comm_init_local     the ranks are contexts of this process (one host thread each): peer copies ordered by events — the
 *                          transport the distributed step is tested with on a single GPU, and a single-process multi-GPU option
 *   bz_comm_exchange_y_halos            Hy halo rows of n parent arrays (levels[m] z levels each), both directions
 *   bz_comm_update_state_and_project    set!'s update_state!(compute_tendencies = false) + halo exchange and, if project != 0, the
 *                                       initial projection with dt (set_atmosphere_model.jl:121-128) and the exchange after it
 *   bz_comm_info           transport name, bytes sent by this rank so far, number of exchanges
 * Dry / vapour anelastic model (the configuration of BASELINE configs[1] and [3]); other physics return BZ_ERR_UNSUPPORTED. */
#define BZ_UNIQUE_ID_BYTES 128
int bz_comm_unique_id(void *id_out);
int bz_comm_init_rccl(bz_ctx *ctx, const void *id);
int bz_comm_init_local(bz_ctx *ctx, const char *group_name);
int bz_comm_destroy(bz_ctx *ctx);
int bz_comm_exchange_y_halos(bz_ctx *ctx, double *const *fields, const int32_t *levels, int32_t n);
int bz_comm_update_state_and_project(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G, double dt, int project);
int bz_comm_info(bz_ctx *ctx, const char **transport, int64_t *bytes_sent, int32_t *exchanges);
/* Compressible y-slab contexts (bz_create_compressible_slab) take the same communicators: bz_time_step_compressible then runs
 * the whole distributed WS-RK3 step in the library (per-substep exchange of (rho theta)' and (rho v)', per-stage exchange of
 * G_rho_v, prognostics, diagnostics and time-averaged velocities; the Kessler column update stays rank-local), and
 * bz_comm_compressible_update_state is update_state! with those exchanges (for set! and drivers stepping operator by operator). */
struct bz_compressible_state; struct bz_compressible_prognostic; struct bz_acoustic_substepper;
int bz_comm_compressible_update_state(bz_ctx *ctx, const struct bz_compressible_state *s, const struct bz_compressible_prognostic *G,
                                      const struct bz_acoustic_substepper *sub, int compute_tendencies);

/* ==== CompressibleDynamics + SplitExplicitTimeDiscretization ==================================
 * Fully compressible dynamics advanced by the Wicker-Skamarock RK3 outer loop with the linearised acoustic substep
 * loop (src/TimeSteppers/acoustic_runge_kutta_3.jl, src/CompressibleEquations/acoustic_substepping.jl).
 * Scope: (Periodic, Periodic, Bounded), microphysics = nothing (vapour is a passive mass fraction entering R_m, c_p,m),
 * ProportionalSubsteps, ThermalDivergenceDamping (optionally damp_vertical) or NoDivergenceDamping, no sponge,
 * LiquidIcePotentialTemperature formulation with the NewtonSolver temperature inversion. */

/* model.dynamics.{dry_density,total_density,pressure}, model.momentum, formulation / moisture prognostics and the
 * diagnostics (src/CompressibleEquations/compressible_dynamics.jl:44-55; atmosphere_model.jl:37-61). */
typedef struct bz_compressible_state {
    double *rho_d;                   /* dynamics.dry_density (prognostic)                */
    double *rho;                     /* dynamics.total_density = rho_d + rho q (diagnosed) */
    double *rho_u, *rho_v, *rho_w;   /* model.momentum                                   */
    double *rho_theta;               /* formulation.potential_temperature_density        */
    double *rho_q;                   /* model.moisture_density                           */
    double *u, *v, *w;               /* model.velocities (default boundary conditions)   */
    double *theta, *q, *T;           /* theta = rho_theta/rho_d, q = rho_q/rho, temperature */
    double *p;                       /* dynamics.pressure = rho R_m T                    */
} bz_compressible_state;

/* timestepper.Gn / timestepper.U0 of AcousticRungeKutta3 (acoustic_runge_kutta_3.jl:64-72): prognostic order
 * (rho_d, rho_u, rho_v, rho_w, rho_theta, rho_q). */
typedef struct bz_compressible_prognostic {
    double *rho_d, *rho_u, *rho_v, *rho_w, *rho_theta, *rho_q;
} bz_compressible_prognostic;

/* AcousticSubstepper fields (acoustic_substepping.jl:91-134); all caller-owned parent arrays. */
typedef struct bz_acoustic_substepper {
    double *exner, *potential_temperature, *gamma_R_mixture;     /* linearization_* (centre)                  */
    double *density_perturbation;                                /* rho'                                       */
    double *density_potential_temperature_perturbation;          /* (rho theta)'                               */
    double *momentum_perturbation_u, *momentum_perturbation_v;   /* (rho u)', (rho v)'                         */
    double *momentum_perturbation_w;                             /* (rho w)'  (z-face)                         */
    double *density_predictor, *density_potential_temperature_predictor;
    double *previous_density_potential_temperature_perturbation;
    double *time_averaged_u, *time_averaged_v, *time_averaged_w; /* time_averaged_velocities (w: z-face)       */
    double *slow_vertical_momentum_tendency;                     /* G^s_rho_w (z-face)                         */
    double *vertical_solver_source_term;                         /* tridiagonal right-hand side (z-face)       */
} bz_acoustic_substepper;

/* SplitExplicitTimeDiscretization (src/CompressibleEquations/time_discretizations.jl:535-562). */
typedef struct bz_split_explicit {
    int32_t substeps;                 /* acoustic substeps per dt; 0 = adaptive from acoustic_cfl (nothing)   */
    int32_t damp_vertical;            /* ThermalDivergenceDamping(damp_vertical)                              */
    int32_t apply_first_substep_pressure_gradient;
    int32_t newton_maxiter;           /* NewtonSolver maxiter (default 8)                                     */
    double acoustic_cfl;              /* default 0.5                                                          */
    double forward_weight;            /* omega, default 0.65                                                  */
    double damping_coefficient;       /* ThermalDivergenceDamping alpha (default 0.1); < 0: NoDivergenceDamping */
    double thermodynamic_tendency_factor, vertical_momentum_tendency_factor;   /* default 1                   */
    double newton_abstol;             /* NewtonSolver abstol (default 1e-4), reltol = 0                       */
    int32_t direct_divergence_damping; /* != 0: DirectDivergenceDamping(damping_coefficient) (time_discretizations.jl:269-274,
                                         acoustic_substepping.jl:1146-1188) instead of ThermalDivergenceDamping; horizontal only  */
    int32_t sponge_ramp;              /* UpperSponge (time_discretizations.jl:381-512; acoustic_substepping.jl:584-602,639,948):
                                         0 = sponge = nothing, 1 LinearRamp, 2 CubicRamp (the default ramp), 3 Sin2Ramp                 */
    double sponge_damping_rate;       /* peak rate at the lid, 1/s (default 0.2)                                                       */
    double sponge_depth;              /* layer thickness below z = grid.Lz, m (default 5e3)                                            */
    int32_t substep_distribution;     /* stage substep counts and sizes (acoustic_substepping.jl:468-508): 0 ProportionalSubsteps (default:
                                         ceil(beta N) substeps tiling beta dt), 1 ConstantSubstepSize (N rounded up to a multiple of 6,
                                         round(beta N) substeps of dt / N), 2 MonolithicFirstStage (stage 1: one substep of dt / 3)          */
    int32_t substep_float_bytes;      /* substep_floattype (acoustic_substepping.jl:199-235): storage type of the substepper's working fields
                                         — exner, potential_temperature, gamma_R_mixture, density_perturbation,
                                         density_potential_temperature_perturbation, momentum_perturbation_u / _v, density_predictor,
                                         density_potential_temperature_predictor, previous_density_potential_temperature_perturbation —
                                         0 = eltype(grid); 4 = Float32 (those ten pointers of bz_acoustic_substepper then address float arrays;
                                         (rho w)', the solver's right-hand side and the time-averaged velocities stay eltype(grid)).
                                         Thermal or no divergence damping; single-device contexts and y-slab contexts with a library-owned
                                         communicator (the per-substep halo messages then carry the Float32 rows). */
    double damping_length_scale;      /* ThermalDivergenceDamping(length_scale = l) (time_discretizations.jl:215-218,235-247): <= 0 = nothing,
                                         the local scale kappa = alpha min(dx, dy)^2 / dtau (acoustic_substepping.jl:1100-1110); > 0 the fixed
                                         diffusivity (alpha l^2) / dtau in both horizontal directions (:1085-1092).  The vertical part of
                                         damp_vertical keeps alpha dz^2 / dtau. */
} bz_split_explicit;

/* ExnerReferenceState columns (src/Thermodynamics/reference_states.jl:717-815): HOST arrays of length Nz+2Hz
 * (parent(field)[1,1,:], halos filled); both NULL for reference_state = nothing. */
typedef struct bz_exner_reference_state {
    double standard_pressure;
    const double *pressure;
    const double *density;
} bz_exner_reference_state;

/* materialize_dynamics(::CompressibleDynamics) + AcousticRungeKutta3/AcousticSubstepper construction
 * (compressible_dynamics.jl:205-252; acoustic_runge_kutta_3.jl:82-103; acoustic_substepping.jl:180-270). */
int bz_create_compressible(bz_ctx **ctx, const bz_grid *grid, const bz_constants *constants,
                           const bz_exner_reference_state *reference_state, const bz_split_explicit *time_discretization,
                           int weno_order);

/* update_state!(model; compute_tendencies) for CompressibleDynamics (update_atmosphere_model_state.jl:41-68 with
 * compressible_time_stepping.jl:83-103,191-242): total density, prognostic halos, velocities, theta, q, T (Newton), p,
 * and — when compute_tendencies != 0 — the moisture tendency G->rho_q built with the total density and the substepper's
 * time-averaged transport velocities (acoustic_runge_kutta_3.jl:352-358).  The momentum / rho_theta / rho_d tendencies
 * that the reference also computes here are overwritten by compute_slow_*_tendencies! before use and are not formed. */
int bz_compressible_update_state(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *G,
                                 const bz_acoustic_substepper *sub, int compute_tendencies);

/* refresh_linearization_basic_state! / prepare_acoustic_cache! (acoustic_substepping.jl:318-399,412-413). */
int bz_refresh_linearization(bz_ctx *ctx, const bz_compressible_state *s, const bz_acoustic_substepper *sub);
/* seed_time_averaged_velocities! (acoustic_substepping.jl:303-314). */
int bz_seed_time_averaged_velocities(bz_ctx *ctx, const bz_compressible_state *s, const bz_acoustic_substepper *sub);
/* compute_slow_momentum_tendencies! + compute_slow_scalar_tendencies! (acoustic_substep_helpers.jl:55-93,117-149):
 * G->rho_u, rho_v, rho_w (advection only, SlowTendencyMode), G->rho_d = -div(momentum), G->rho_theta. */
int bz_compute_slow_tendencies(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *G);
/* acoustic_substepping.jl:451-508 (ProportionalSubsteps): substep count and size of the stage covering beta*dt. */
int bz_stage_substeps(bz_ctx *ctx, double dt, double beta, int32_t *n_substeps, double *dtau);
/* acoustic_rk3_substep_loop!(model, substepper, dt, beta, U0) (acoustic_substepping.jl:1404-1590): slow rho_w assembly,
 * rewind initialisation, the substep loop, time-averaged velocities, recovery of the full state, halos, velocities. */
int bz_acoustic_substep_loop(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                             const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub, double dt, double beta);
/* acoustic_rk3_substep!(model, dt, beta) (acoustic_runge_kutta_3.jl:172-186): linearisation refresh, slow tendencies,
 * substep loop, WS-RK3 update of the moisture density. */
int bz_acoustic_rk3_substep(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                            const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub, double dt, double beta);
/* time_step!(model::CompressibleAcousticModel, dt) (acoustic_runge_kutta_3.jl:264-319) without callbacks; the state
 * must be consistent (bz_seed_time_averaged_velocities + bz_compressible_update_state(compute_tendencies=1) once
 * after set!, as maybe_prepare_first_time_step! does). */
int bz_time_step_compressible(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                              const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub, double dt);

/* ---- y-slab decomposition of the compressible path (halo exchanges only, the column solve is local) ----
 * `local_grid` is this rank's slab as in bz_create_slab.  The caller's neighbour exchange fills the y halos (Hy rows) at the
 * points listed below; x and z halos stay the library's business.  Sequence of one WS-RK3 stage:
 *   [state halos valid: rho_d, rho, momentum, rho_theta, rho_q, u, v, w, theta, q, T, p, time-averaged velocities; the U0 arrays are
 *    whole-array copies of the step-start state, halo rows included (the stage's first sweeps form U0 - U on the halo rows too)]
 *   bz_refresh_linearization            (also linearises one halo row on each side)
 *   bz_compute_slow_tendencies          -> exchange G->rho_v
 *   bz_acoustic_stage_begin             returns N_tau and which buffer pair is current (0: the substepper's own
 *                                       (rho theta)', (rho u)', (rho v)' fields, 1: previous_(rho theta)' / the scratch pair)
 *   for n = 1..N_tau:  exchange current (rho theta)' and (rho v)'  ->  bz_acoustic_substep(n)
 *                      [DirectDivergenceDamping: -> exchange the (rho u)', (rho v)' buffers bz_acoustic_substep made current
 *                       -> bz_acoustic_direct_damping]
 *   exchange current (rho theta)'  ->  bz_acoustic_stage_end
 *   exchange rho_d, momentum, rho_theta, rho_q, time-averaged velocities  ->  bz_compressible_update_state(compute_tendencies = 0)
 *   exchange rho, u, v, w, theta, q, T, p  ->  bz_compute_moisture_tendency */
int bz_create_compressible_slab(bz_ctx **ctx, const bz_grid *local_grid, const bz_constants *constants,
                                const bz_exner_reference_state *reference_state, const bz_split_explicit *time_discretization,
                                int weno_order, int y_nranks, int y_rank);
/* The three pieces of acoustic_rk3_substep_loop! (acoustic_substepping.jl:1404-1590): :1437-1441 | one iteration of
 * :1448-1555 | :1557-1590 without the trailing halo fills / compute_velocities! (done by the next update_state!). */
int bz_acoustic_stage_begin(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                            const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub, double dt, double beta,
                            int32_t *n_substeps, int32_t *current_buffer);
int bz_acoustic_substep(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                        const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub, int32_t substep,
                        int32_t *current_buffer);
int bz_acoustic_stage_end(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                          const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub, double dt, double beta,
                          int update_moisture);
/* apply_divergence_damping!(::DirectDivergenceDamping) (acoustic_substepping.jl:1158-1188) of the substep that just ran, on a y-slab
 * context: the divergence of the theta-weighted momentum perturbations reads (rho v)' of the neighbour's first row and the gradient
 * reads it one row below the slab, so the driver exchanges the current (rho u)', (rho v)' buffers between bz_acoustic_substep and this
 * call.  Single-GPU contexts damp inside bz_acoustic_substep; there, and without DirectDivergenceDamping, this is a no-op. */
int bz_acoustic_direct_damping(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                               const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub);
/* Lateral boundaries of the acoustic substep loop (acoustic_substepping.jl:1300-1395: apply_open_boundary_relaxation!,
 * enforce_wall_impenetrability!), compressible contexts on a grid with a Bounded x and / or y — (Bounded, Periodic, Bounded),
 * (Periodic, Bounded, Bounded), (Bounded, Bounded, Bounded); single device.  Such a context runs bz_refresh_linearization,
 * bz_acoustic_substep_loop and the three-piece stage interface; every other compressible entry point returns BZ_ERR_UNSUPPORTED on it.
 * The substepper's own fields take the zero-gradient halo of their default boundary conditions from the library; the model's fields
 * (rho_d, rho_theta, p, the slow tendencies) are read with the one halo row / column the caller's boundary conditions filled.
 * *_open != 0: that side carries an active open (normal-flow) boundary condition on the wall-normal momentum — the outermost cells of rho',
 * (rho theta)' are relaxed every substep towards (c^L[halo] - c^L[cell]) / 2 with factor open_boundary_relaxation in (0, 1]
 * (SplitExplicitTimeDiscretization(open_boundary_relaxation = 0.5)) and the west / south wall face of the momentum perturbation is advanced
 * by the substep like any other face; 0 (the default of every side): an impenetrable wall, whose face is held at zero.  The east / north
 * wall face (index N + 1 of the reference's face field) is written by no kernel of the reference's loop and is an exact zero here.
 * bz_acoustic_substep_loop ends with _recover_full_state! on such a context: the halo fills with the model's boundary conditions and
 * compute_velocities! (:1584-1587) are the caller's. */
int bz_set_acoustic_lateral_boundaries(bz_ctx *ctx, int west_open, int east_open, int south_open, int north_open,
                                       double open_boundary_relaxation);
/* Caller-owned second buffers of the (rho u)', (rho v)' ping-pong (XFace / YFace parent arrays), so that the slab driver can
 * exchange their halos; NULL, NULL returns to the context's own scratch. */
int bz_set_acoustic_scratch(bz_ctx *ctx, double *momentum_u_second_buffer, double *momentum_v_second_buffer);
/* The moisture part of compute_tendencies! alone (update_atmosphere_model_state.jl:330-343 with the time-averaged
 * transport velocities), for drivers that exchange halos between the diagnosis and the tendency. */
int bz_compute_moisture_tendency(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *G,
                                 const bz_acoustic_substepper *sub);

/* ---- DCMIP2016 Kessler warm-rain microphysics: the operator-split column kernel ----
 * microphysics_model_update!(::DCMIP2016KesslerMicrophysics, model) (src/Microphysics/dcmip2016_kessler.jl:449-486, kernel
 * :618-858): autoconversion, accretion, saturation adjustment (Tetens), rain evaporation and sedimentation with per-column
 * subcycling, in place on theta / rho_theta / rho_qv / rho_qcl / rho_qr; q^v, q^cl, q^r, W^r are diagnosed (they double
 * as the column workspace, as in the reference); the caller follows with update_state!.  density / pressure: 3-D parents
 * (CompressibleDynamics) or NULL for the context's reference columns (AnelasticDynamics).  precipitation_rate: the
 * horizontal parent (Sx x Sy) of mu.precipitation_rate. */
typedef struct bz_kessler_microphysics {      /* DCMIP2016KesslerMicrophysics (:40-70,154-169) */
    double dcmip_temperature_scale;
    double terminal_velocity_coefficient, density_scale, terminal_velocity_exponent;
    double autoconversion_rate, autoconversion_threshold;
    double accretion_rate, accretion_exponent;
    double evaporation_ventilation_coefficient_1, evaporation_ventilation_coefficient_2;
    double evaporation_ventilation_exponent_1, evaporation_ventilation_exponent_2;
    double diffusivity_coefficient, thermal_conductivity_coefficient;
    double substep_cfl;
    /* TetensFormula (src/Thermodynamics/tetens_formula.jl:72-85) and the liquid CondensedPhase of the constants */
    double tetens_reference_saturation_vapor_pressure, tetens_reference_temperature;
    double tetens_liquid_coefficient, tetens_liquid_temperature_offset;
    double liquid_latent_heat, liquid_heat_capacity;
} bz_kessler_microphysics;
typedef struct bz_kessler_fields {
    const double *density, *pressure;
    double *potential_temperature, *potential_temperature_density;
    double *moisture_density, *cloud_liquid_density, *rain_density;                     /* rho q^v, mu.rho q^cl, mu.rho q^r */
    double *vapor_mass_fraction, *cloud_liquid_mass_fraction, *rain_mass_fraction;      /* mu.q^v, mu.q^cl, mu.q^r          */
    double *rain_terminal_velocity;                                                     /* mu.W^r                          */
    double *precipitation_rate;
} bz_kessler_fields;
int bz_kessler_microphysics_update(bz_ctx *ctx, const bz_kessler_microphysics *params, const bz_kessler_fields *fields,
                                   double dt, double standard_pressure);
/* AtmosphereModel(...; microphysics = DCMIP2016KesslerMicrophysics()) for AnelasticDynamics: rho q^cl and rho q^r join the
 * prognostic fields (prognostic_field_names, dcmip2016_kessler.jl:216) with their U0 / Gn storage; the specific moisture slot
 * of bz_state is q^v; update_state! diagnoses q^cl, q^r, q^v and T = Pi(q) theta + L q^l/c_pm with q = (q^v, q^cl + q^r)
 * (:222-227,298-303,860-865); compute_tendencies! adds -div_rhoUc of both species; ssp_rk3_substep! / store_initial_state!
 * include them; bz_time_step_anelastic ends with the column update + update_state! (ssp_runge_kutta_3.jl:262-263).
 * params == NULL detaches.  bz_kessler_model_update is microphysics_model_update!(microphysics, model) for per-operator drivers. */
typedef struct bz_kessler_model_fields {
    double *cloud_liquid_density, *rain_density;                 /* mu.rho q^cl, mu.rho q^r (prognostic) */
    double *U0_cloud_liquid_density, *U0_rain_density;           /* timestepper.U0                        */
    double *G_cloud_liquid_density, *G_rain_density;             /* timestepper.Gn                        */
    double *vapor_mass_fraction, *cloud_liquid_mass_fraction, *rain_mass_fraction;     /* mu.q^v, q^cl, q^r */
    double *rain_terminal_velocity;                              /* mu.W^r                                */
    double *precipitation_rate;                                  /* horizontal parent (Sx x Sy)           */
} bz_kessler_model_fields;
int bz_set_kessler_microphysics(bz_ctx *ctx, const bz_kessler_microphysics *params, const bz_kessler_model_fields *fields,
                                double standard_pressure);
int bz_kessler_model_update(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G, double dt);
/* The same attachment works for a CompressibleDynamics context (bz_create_compressible): the total density then includes the
 * condensates (compressible_time_stepping.jl:83-103 with condensate_field_names), the Newton temperature inversion carries the
 * latent term, gamma R_m of the linearisation includes the liquid fraction, the species ride the time-averaged transport
 * velocities and the WS-RK3 scalar update, and bz_time_step_compressible ends with the column update (density = rho_d,
 * pressure = dynamics.pressure) + update_state! (acoustic_runge_kutta_3.jl:315-316).  On a y-slab context
 * (bz_create_compressible_slab) the call runs the rank-local column kernel only and the driver's exchanging update_state! follows. */
int bz_compressible_kessler_update(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *G,
                                   const bz_acoustic_substepper *sub, double dt);

/* ---- forcing, f-plane Coriolis and bottom-flux stack of the BOMEX configuration (BASELINE configs[2]) ----
 * Every forcing of examples/bomex.jl:104-196 is a horizontally uniform specific profile multiplied by the reference density
 * (SpecificForcing, src/Forcings/specific_forcing.jl:61-74), so the model's `forcing` NamedTuple reduces to columns:
 *   u_forcing / v_forcing        geostrophic_forcings: -f v_g(z), +f u_g(z) (src/Forcings/geostrophic_forcings.jl)
 *   theta_forcing, moisture_forcing   Forcing(field) profiles keyed under theta / q^e (q^v)
 *   energy_forcing               profile keyed under `e`; enters G_rho_theta as rho F_e / (c_pm Pi)
 *                                (src/PotentialTemperatureFormulations/potential_temperature_tendency.jl:86-104)
 *   subsidence_vertical_velocity SubsidenceForcing(w_s) on the flagged fields: F = -zb-average(w_s dz(horizontal average))
 *                                (src/Forcings/subsidence_forcing.jl:75-91), averages recomputed at every compute_forcings!
 *   coriolis_f                   FPlane(f): -x_f_cross_U, -y_f_cross_U (src/AtmosphereModels/dynamics_kernel_functions.jl:79,99)
 *   bottom_*_flux                FluxBoundaryCondition values on rho theta / rho q (examples/bomex.jl:80-87)
 *   bottom_drag_rho0_ustar2      rho0 u*^2 of the bulk drag FluxBoundaryCondition on rho u, rho v (examples/bomex.jl:95-101)
 *   bottom_drag_epsilon          the regulariser under the square root, J = -rho0 u*^2 rho_u / sqrt(rho_u^2 + rho_v^2 + eps)
 *                                (benchmarking/src/convective_boundary_layer.jl:142-146 uses 1e-10; the BOMEX example none)
 *   bottom_energy_flux           see the member
 * All pointers are HOST arrays (cell centres, length Nz; the subsidence velocity Nz+1 faces), copied by the call; NULL = absent.
 * With a stack attached bz_compute_tendencies adds the forcing + Coriolis terms, and bz_time_step_anelastic calls
 * bz_compute_flux_bc_tendencies before every RK substep (src/TimeSteppers/ssp_runge_kutta_3.jl:229,243,257).
 * Anelastic potential-temperature contexts (microphysics nothing or SaturationAdjustment; single device, or y-slabs through the library-owned
 * distributed step); a single-device CompressibleDynamics context accepts a stack whose only non-zero member is coriolis_f (the f-plane term
 * of its slow momentum tendencies, examples/tropical_cyclone_with_rainband.jl:508-514). */
typedef struct bz_column_forcings {
    const double *u_forcing;
    const double *v_forcing;
    const double *theta_forcing;
    const double *moisture_forcing;
    const double *energy_forcing;
    const double *subsidence_vertical_velocity;
    int32_t subsidence_u, subsidence_v, subsidence_theta, subsidence_moisture;
    double coriolis_f;
    double bottom_theta_flux, bottom_moisture_flux;
    double bottom_drag_rho0_ustar2;
    double bottom_drag_epsilon;
    double bottom_energy_flux;       /* a constant FluxBoundaryCondition keyed rho e in a potential-temperature model: the energy flux Q (W m^-2)
                                        enters rho theta as Q / c_pm with the mixture heat capacity of the lowest cell
                                        (EnergyFluxBoundaryCondition, src/BoundaryConditions/thermodynamic_variable_bcs.jl; BoundaryConditions.jl:218-227) */
} bz_column_forcings;
int bz_set_forcings(bz_ctx *ctx, const bz_column_forcings *forcings);       /* NULL detaches the stack */
/* ---- sponge layers: Relaxation(rate, mask, target) forcings with a horizontally uniform mask and target ----
 * Oceananigans' Relaxation — F = rate * mask(x, y, z) * (target(x, y, z, t) - field) — is how the reference's examples damp the top of
 * the domain: examples/rico.jl:103-105,164 (w = Relaxation(rate, GaussianMask{:z})), examples/neutral_atmospheric_boundary_layer.jl:103-136
 * (rho w relaxed to 0, rho theta to a reference profile), examples/tropical_cyclone_world.jl, tropical_cyclone_with_rainband.jl:434-490.
 * With mask and target functions of z alone the forcing is two columns per field:
 *   rate_*    rate * mask(z) at the field's vertical location (centres, length Nz; rho w: faces, length Nz + 1)
 *   target_*  target(z) there (NULL = 0)
 * for rho u, rho v, rho w, the thermodynamic density (rho theta or rho e) and the moisture density; NULL rate = no relaxation of that field.
 * Keyed by the density name (rho w = ...) the forcing enters the density tendency as is, G += rate (target - rho phi); keyed by the specific
 * name (w = ..., bit set in specific_mask: 1 u, 2 v, 4 w) it is a specific forcing (src/Forcings/specific_forcing.jl:61-74):
 * G += rho_r rate (target - phi) with the reference density at the field's location.  HOST arrays, copied by the call.  bz_compute_tendencies adds the terms
 * after the forcing stack; bz_time_step_anelastic then steps with the tendencies evaluated per operator (fused RK update, projection and
 * diagnosis; y-slab contexts: the operator-by-operator distributed step).  On a single-device CompressibleDynamics context (examples/tropical_cyclone_with_rainband.jl:434-514) the density-keyed
 * sponges of rho u, rho v, rho w, rho theta join the slow tendencies (bz_compute_slow_tendencies), and bz_set_forcings accepts coriolis_f alone. */
typedef struct bz_column_relaxation {
    const double *rate_u, *target_u;
    const double *rate_v, *target_v;
    const double *rate_w, *target_w;
    const double *rate_theta, *target_theta;
    const double *rate_moisture, *target_moisture;
    int32_t specific_mask;
} bz_column_relaxation;
int bz_set_relaxation(bz_ctx *ctx, const bz_column_relaxation *relaxation);      /* NULL detaches */
/* Forcing(f(x, y, z, t)) / Forcing(field) on the thermodynamic variable (examples/tropical_cyclone_with_rainband.jl:419-432,511-514: the
 * prescribed rainband heating keyed `theta`): a DEVICE array at cell centres in the parent layout of the fields, read at every tendency
 * evaluation — the caller owns it and refreshes its contents when the forcing depends on time.  specific = 1 (keyed theta / e): a specific
 * forcing, G_rho_theta += rho F with the coupling density (rho_r(z) of AnelasticDynamics, the prognostic dry density of CompressibleDynamics;
 * src/Forcings/specific_forcing.jl:61-74, compressible_dynamics.jl:385); specific = 0 (keyed rho theta / rho e): G += F.  NULL detaches.
 * On a y-slab the array is the slab's own parent array.  Compressible y-slabs: not built. */
int bz_set_field_forcing(bz_ctx *ctx, const double *thermodynamic_forcing, int specific);
/* compute_forcings!(model) (src/AtmosphereModels/update_atmosphere_model_state.jl:81-86) */
int bz_compute_forcings(bz_ctx *ctx, const bz_state *s);
/* compute_flux_bc_tendencies!(model) (src/AtmosphereModels/update_atmosphere_model_state.jl:418-434) */
int bz_compute_flux_bc_tendencies(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G);
/* Bulk aerodynamic bottom conditions with constant transfer coefficients on unfiltered fields:
 *   BulkDrag on rho u, rho v               J^u = -rho0 C^D U~ u          (src/BoundaryConditions/bulk_drag.jl:114-135)
 *   BulkSensibleHeatFlux on rho theta      J^th = -rho0 C^T U~ (theta - theta0)  (bulk_scalar_fluxes.jl:82-90,123-137)
 *   BulkVaporFlux on the moisture density  J^v = -rho0 C^v U~ (q^v - q^v+(T0, rho0))  (bulk_scalar_fluxes.jl:206-232)
 * with U~ = sqrt(U^2 + gustiness^2), U^2 interpolated to the flux location (BoundaryConditions.jl:64-85), rho0 = p0 / (R^d T0)
 * (src/Thermodynamics/reference_states.jl:73-76), theta0 = T0 / (p0/p_st)^(R^d/c_pd).  A coefficient <= 0 switches that
 * condition off.  Applied by bz_compute_flux_bc_tendencies (and inside bz_time_step_anelastic) next to the constant fluxes of
 * bz_column_forcings.  PolynomialCoefficient and FilteredSurfaceVelocities are not implemented. */
typedef struct bz_bulk_surface_fluxes {
    double drag_coefficient, drag_gustiness, drag_surface_temperature;
    double heat_coefficient, heat_gustiness, heat_surface_temperature;
    double vapor_coefficient, vapor_gustiness, vapor_surface_temperature;
    double surface_pressure, standard_pressure;
    /* liquid CondensedPhase + triple point of ThermodynamicConstants, for q^v+ of the vapour flux */
    double liquid_latent_heat, liquid_heat_capacity, energy_reference_temperature, triple_point_temperature, triple_point_pressure;
} bz_bulk_surface_fluxes;
int bz_set_bulk_surface_fluxes(bz_ctx *ctx, const bz_bulk_surface_fluxes *fluxes);      /* NULL detaches */

/* ---- user tracers of the anelastic model: AtmosphereModel(grid; tracers = (:a, :b)) ----
 * density = model.tracers.c (prognostic rho c), specific = c = rho c / rho_r (the reference converts in place around the tendency
 * evaluation, src/AtmosphereModels/update_atmosphere_model_state.jl:43,65,88-112; here it is a separate centre array), U0 / G the
 * timestepper mirrors.  Each tracer gets the WENO-5 scalar tendency -div_rhoUc(c) (:352-372) and the SSP-RK3 update.  At most
 * BZ_MAX_TRACERS; n = 0 detaches.  Single-device anelastic contexts; not combined with a closure in this build. */
#define BZ_MAX_TRACERS 8
typedef struct bz_tracer_fields {
    double *density;
    double *specific;
    double *U0;
    double *G;
} bz_tracer_fields;
int bz_set_tracers(bz_ctx *ctx, int32_t n, const bz_tracer_fields *tracers);

/* ---- bounds-preserving WENO for moisture-like scalars (second half) ----
 * advection = (; rho_q^e = WENO(order = 5, bounds = (0, 1)), ...) (examples/rico.jl:184-190, examples/tropical_cyclone_world.jl:169):
 * div_rhoUc(i, j, k, grid, ::BoundsPreservingWENO, rho, U, c) = V^-1 (bounded_tracer_flux_divergence_x + _y + _z)
 * (src/Advection.jl:42-47); the three divergences are Oceananigans' positivity-preserving limiter (not vendored; restated in
 * oracle/breeze_oracle.c: parity unpinned).  Flags select the scalars that use it: the moisture density (rho q^v / rho q^e), the
 * prognostic species of the attached microphysics (rho q^cl, rho q^r), the user tracers.  With any flag set bz_compute_tendencies
 * replaces those scalars' advective tendencies and bz_time_step_anelastic runs its operator-sequence tier.  Single-device anelastic
 * contexts of the WENO build; NULL detaches. */
typedef struct bz_bounds_preserving_advection {
    double lower, upper;
    int32_t moisture, microphysical_species, tracers, reserved;
} bz_bounds_preserving_advection;
int bz_set_bounds_preserving_advection(bz_ctx *ctx, const bz_bounds_preserving_advection *bounds);

/* ---- closure = SmagorinskyLilly() (BASELINE configs[2]) ----
 * Breeze's part — density-weighted stress and flux divergences (src/TurbulenceClosures/TurbulenceClosures.jl:44-101), their place
 * in the tendencies (src/AtmosphereModels/dynamics_kernel_functions.jl:80,100,128,157), N^2 = g dz(log theta_v)
 * (src/AtmosphereModels/atmosphere_model_buoyancy.jl:46-68) and compute_closure_fields! at the end of
 * compute_auxiliary_variables! (src/AtmosphereModels/update_atmosphere_model_state.jl:218) — is followed line by line; the eddy
 * viscosity and kinematic fluxes are Oceananigans' SmagorinskyLilly (0.110.14, not vendored), restated from its published form
 * (oracle/closure.py: parity unpinned).  eddy_viscosity is model.closure_fields.nu_e (centre field parent, caller-owned).
 * With a closure attached bz_compute_tendencies subtracts the divergences after the advective terms.  Single-device anelastic
 * potential-temperature contexts only. */
typedef struct bz_smagorinsky_lilly {
    double smagorinsky_coefficient;   /* C  = 0.16 */
    double reduction_factor;          /* Cb = 1.0: varsigma = sqrt(1 - min(1, Cb N^2+ / Sigma^2)) */
    double prandtl_number;            /* Pr = 1.0 (every scalar) */
} bz_smagorinsky_lilly;
int bz_set_closure(bz_ctx *ctx, const bz_smagorinsky_lilly *closure, double *eddy_viscosity);   /* NULL detaches */
int bz_compute_closure_fields(bz_ctx *ctx, const bz_state *s);

/* ---- the reductions of the run! loop around the step ---- */
/* cell_advection_timescale(model) (src/AtmosphereModels/cell_advection_timescale.jl:47-66): minimum over the interior of
 * 1 / (|u|/dx + |v|/dy + |w|/dz) into *out (host; +Inf for a fluid at rest); w == NULL gives the HorizontalFormulation.
 * Synchronises the stream.  u, v, w: velocity parent arrays (a slab rank reduces its own slab; combine with a min). */
int bz_cell_advection_timescale(bz_ctx *ctx, const double *u, const double *v, const double *w, double *out);
/* NaNChecker on one field (default_nan_checker, src/AtmosphereModels/atmosphere_model.jl:561-572): *out = 1 if any
 * interior value is NaN.  Synchronises. */
int bz_any_nan(bz_ctx *ctx, const double *field, int z_face, int32_t *out);

/* ---- instrumentation (not part of the reference interface) ---- */
/* When enabled, every kernel group is bracketed by hipEvents on the ctx stream. */
int bz_profile_enable(bz_ctx *ctx, int on);
int bz_profile_reset(bz_ctx *ctx);
/* number of instrumented kernel groups; name/total milliseconds/launch count of group idx
 * (synchronises the stream). */
int bz_profile_count(bz_ctx *ctx);
int bz_profile_get(bz_ctx *ctx, int idx, const char **name, double *total_ms, int64_t *launches);
/* divergence of momentum, max-abs over the interior, into *out (host); diagnostic for tests
 * (test/anelastic_pressure_solver_nonhydrostatic.jl:45-46). Synchronises. */
int bz_max_abs_divergence(bz_ctx *ctx, const bz_state *s, double *out);

/* hipGraph replay of whole time steps (csrc/bz_graph.hip).  bz_time_step_anelastic / bz_time_step_compressible are pure functions of
 * (argument structs, dt, configuration); on launch-bound grids the second call with the same arguments records the step with stream
 * capture and later calls replay it with one hipGraphLaunch.  Opt-in (bz_graph_enable, or BZ_GRAPH=1 in the environment): measured gain
 * on MI355X is 0-5 % on launch-bound grids and recording costs ~1 ms, so it pays only with a fixed dt.  Inactive while profiling is
 * enabled and on contexts with a communicator.  The arrays named by the structs must stay where they are, as
 * Oceananigans fields do.  No reference counterpart. */
int bz_graph_enable(bz_ctx *ctx, int on);
int bz_graph_info(bz_ctx *ctx, int32_t *enabled, int64_t *captures, int64_t *replays);

#ifdef __cplusplus
}
#endif
#endif /* BREEZE_HIP_H */
