# BreezeHIP.jl — Julia-side glue for libbreeze_hip.so (UNTESTED: Julia is not available in the build image).
#
# A package extension in the style of ext/BreezeReactantExt/Timesteppers.jl:6-19: the hot-path generic functions are
# specialised on a marker architecture and forwarded to the C ABI of include/breeze_hip.h.  Fields stay ordinary
# Oceananigans fields whose storage lives on the GPU (e.g. ROCArray parents); only raw device pointers cross the boundary.
module BreezeHIP

using Breeze
using Breeze.AtmosphereModels: AtmosphereModel, AtmosphereModels
using Breeze.TimeSteppers: SSPRungeKutta3
using Oceananigans
using Oceananigans.TimeSteppers: TimeSteppers as OceananigansTimeSteppers, update_state!, tick!
using Oceananigans.Grids: topology, Periodic, Bounded, Flat, znodes, Face
using Oceananigans.Architectures: AbstractArchitecture

const libbreeze_hip = get(ENV, "BREEZE_HIP_LIB", "libbreeze_hip.so")

"Marker architecture: fields are allocated by `child` (a GPU architecture), the hot path runs in libbreeze_hip."
struct HIPNative{A} <: AbstractArchitecture
    child :: A
end

struct BzGrid
    Nx::Int32; Ny::Int32; Nz::Int32
    Hx::Int32; Hy::Int32; Hz::Int32
    topo::NTuple{3, Int32}
    ftype::Int32
    dx::Float64; dy::Float64
    zf::Ptr{Float64}
    regular_z::Int32; reserved::Int32
end
struct BzConstants
    g::Float64; Rd::Float64; Rv::Float64; cpd::Float64; cpv::Float64
end
struct BzReferenceState
    p0::Float64; θ0::Float64; pst::Float64
    density::Ptr{Float64}; pressure::Ptr{Float64}; temperature::Ptr{Float64}
end
struct BzState
    ρu::Ptr{Float64}; ρv::Ptr{Float64}; ρw::Ptr{Float64}; ρθ::Ptr{Float64}; ρq::Ptr{Float64}
    u::Ptr{Float64}; v::Ptr{Float64}; w::Ptr{Float64}; θ::Ptr{Float64}; q::Ptr{Float64}; T::Ptr{Float64}
    ϕ::Ptr{Float64}
end
struct BzPrognostic
    ρu::Ptr{Float64}; ρv::Ptr{Float64}; ρw::Ptr{Float64}; ρθ::Ptr{Float64}; ρq::Ptr{Float64}
end

devptr(f) = Ptr{Float64}(UInt(pointer(parent(f))))     # device pointer of the halo-inclusive parent array
topocode(::Type{Periodic}) = Int32(0); topocode(::Type{Bounded}) = Int32(1); topocode(::Type{Flat}) = Int32(2)

check(rc, what, ctx) = rc == 0 || error("$what failed with code $rc: ",
                                        unsafe_string(ccall((:bz_last_error, libbreeze_hip), Cstring, (Ptr{Cvoid},), ctx)))

"Build the context once per model (dynamics_pressure_solver + materialize_advection time)."
function create_context(model::AtmosphereModel)
    grid = model.grid
    Nx, Ny, Nz = size(grid); Hx, Hy, Hz = Oceananigans.Grids.halo_size(grid)
    zf = collect(Float64, znodes(grid, Face()))
    ref = model.dynamics.reference_state
    c = model.thermodynamic_constants
    hostcol(f) = collect(Float64, vec(Array(parent(f))))          # Nz + 2Hz values, halos filled
    ρ, p, T = hostcol(ref.density), hostcol(ref.pressure), hostcol(ref.temperature)
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve zf ρ p T begin
        g = BzGrid(Nx, Ny, Nz, Hx, Hy, Hz, map(topocode, topology(grid)), 8, grid.Δxᶜᵃᵃ, grid.Δyᵃᶜᵃ, pointer(zf),
                   grid.z.Δᵃᵃᶜ isa Number ? 1 : 0, 0)
        k = BzConstants(c.gravitational_acceleration, c.molar_gas_constant / c.dry_air.molar_mass,
                        c.molar_gas_constant / c.vapor.molar_mass, c.dry_air.heat_capacity, c.vapor.heat_capacity)
        r = BzReferenceState(ref.surface_pressure, ref.potential_temperature, ref.standard_pressure,
                             pointer(ρ), pointer(p), pointer(T))
        rc = ccall((:bz_create, libbreeze_hip), Cint,
                   (Ref{Ptr{Cvoid}}, Ref{BzGrid}, Ref{BzConstants}, Ref{BzReferenceState}, Cint), ctx, g, k, r, weno_order(model.advection))
        rc == 0 || error("bz_create failed with code $rc (WENO orders 5, 7, 9 with halos >= (order + 1) / 2)")
    end
    so = scalar_weno_order(model.advection)      # model.advection = (; momentum, ρθ, ρqᵉ, ...) after the constructor merged the schemes
    so == weno_order(model.advection) || set_scalar_advection_order!(ctx[], so)
    return ctx[]
end

state(model) = BzState(devptr(model.momentum.ρu), devptr(model.momentum.ρv), devptr(model.momentum.ρw),
                       devptr(model.formulation.potential_temperature_density), devptr(model.moisture_density),
                       devptr(model.velocities.u), devptr(model.velocities.v), devptr(model.velocities.w),
                       devptr(model.formulation.potential_temperature), devptr(model.microphysical_fields.qᵛ),
                       devptr(model.temperature), devptr(model.dynamics.pressure_anomaly))
prognostic(nt) = BzPrognostic(devptr(nt.ρu), devptr(nt.ρv), devptr(nt.ρw), devptr(nt.ρθ), devptr(nt.ρqᵛ))

const CONTEXTS = IdDict{Any, Ptr{Cvoid}}()
context(model) = get!(() -> create_context(model), CONTEXTS, model)

const HIPModel = AtmosphereModel{<:Any, <:Any, <:HIPNative, <:SSPRungeKutta3}

# Whole-step seam (src/TimeSteppers/ssp_runge_kutta_3.jl:209-278)
function OceananigansTimeSteppers.time_step!(model::HIPModel, Δt; callbacks=[])
    ctx = context(model)
    model.clock.iteration == 0 && update_state!(model, callbacks; compute_tendencies=true)
    rc = ccall((:bz_time_step_anelastic, libbreeze_hip), Cint,
               (Ptr{Cvoid}, Ref{BzState}, Ref{BzPrognostic}, Ref{BzPrognostic}, Cdouble),
               ctx, state(model), prognostic(model.timestepper.U⁰), prognostic(model.timestepper.Gⁿ), Δt)
    check(rc, "bz_time_step_anelastic", ctx)
    tick!(model.clock, Δt)
    return nothing
end

# Multi-step seam: n steps with nothing reading the model in between — many_time_steps! of the benchmark driver
# (benchmarking/src/timestepping.jl:11-16) and the stretch of run!(simulation) up to the next callback / output-writer actuation.
# Every step but the last skips the projection + diagnosis pass (model.velocities, θ, qᵛ, T, pressure anomaly are host-side products);
# diagnose_last = true leaves exactly what n calls of time_step! leave.
function many_time_steps!(model::HIPModel, Δt, n::Integer; diagnose_last::Bool=true)
    n <= 0 && return nothing
    ctx = context(model)
    model.clock.iteration == 0 && update_state!(model; compute_tendencies=true)
    rc = ccall((:bz_time_steps_anelastic, libbreeze_hip), Cint,
               (Ptr{Cvoid}, Ref{BzState}, Ref{BzPrognostic}, Ref{BzPrognostic}, Cdouble, Cint, Cint),
               ctx, state(model), prognostic(model.timestepper.U⁰), prognostic(model.timestepper.Gⁿ), Δt, n, diagnose_last ? 1 : 0)
    check(rc, "bz_time_steps_anelastic", ctx)
    for _ in 1:n
        tick!(model.clock, Δt)
    end
    return nothing
end

# iterations until something scheduled on the simulation wants to look at the model (IterationInterval schedules; any other schedule: 1)
function iterations_to_next_actuation(sim)
    n = sim.stop_iteration - sim.model.clock.iteration
    for item in Iterators.flatten((values(sim.callbacks), values(sim.output_writers), values(sim.diagnostics)))
        sched = item.schedule
        n = sched isa IterationInterval ? min(n, sched.interval - mod(sim.model.clock.iteration - sched.offset, sched.interval)) : 1
    end
    return max(1, Int(min(n, typemax(Int32))))
end

# Simulation time_step! for the HIP model: fixed-Δt stretches between actuations go through the multi-step seam
function time_steps_to_next_actuation!(sim)
    n = iterations_to_next_actuation(sim)
    many_time_steps!(sim.model, sim.Δt, n; diagnose_last=true)
    return n
end

# Per-operator seams (src/AtmosphereModels/update_atmosphere_model_state.jl:41-68; src/AnelasticEquations/anelastic_time_stepping.jl:26-78)
function OceananigansTimeSteppers.update_state!(model::HIPModel, callbacks=[]; compute_tendencies=true)
    ctx = context(model)
    rc = ccall((:bz_update_state, libbreeze_hip), Cint, (Ptr{Cvoid}, Ref{BzState}, Ref{BzPrognostic}, Cint),
               ctx, state(model), prognostic(model.timestepper.Gⁿ), compute_tendencies ? 1 : 0)
    check(rc, "bz_update_state", ctx)
    return nothing
end
function AtmosphereModels.compute_pressure_correction!(model::HIPModel, Δt)
    ctx = context(model)
    check(ccall((:bz_compute_pressure_correction, libbreeze_hip), Cint, (Ptr{Cvoid}, Ref{BzState}, Cdouble),
                ctx, state(model), Δt), "bz_compute_pressure_correction", ctx)
end
function AtmosphereModels.make_pressure_correction!(model::HIPModel, Δt)
    ctx = context(model)
    check(ccall((:bz_make_pressure_correction, libbreeze_hip), Cint, (Ptr{Cvoid}, Ref{BzState}, Cdouble),
                ctx, state(model), Δt), "bz_make_pressure_correction", ctx)
end

# AtmosphereModel(grid; momentum_advection, scalar_advection) with schemes of different orders (atmosphere_model.jl:80-82,148-158):
# the context is created with the momentum order; this hands over the scalars' order (5, 7 or 9)
function set_scalar_advection_order!(ctx, order::Integer)
    check(ccall((:bz_set_scalar_advection_order, libbreeze_hip), Cint, (Ptr{Cvoid}, Cint), ctx, order), "bz_set_scalar_advection_order", ctx)
end

# Gc = -div_ρUc(c) of one centre field with the model's velocities: the launch of compute_scalar_tendency!
# (update_atmosphere_model_state.jl:390-393) and what benchmarking/src/scalar_tendency.jl:16-25 times
function compute_scalar_tendency!(Gc, model::HIPModel, c)
    ctx = context(model)
    U = model.velocities
    check(ccall((:bz_compute_scalar_tendency, libbreeze_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                ctx, pointer(parent(U.u)), pointer(parent(U.v)), pointer(parent(U.w)), pointer(parent(c)), pointer(parent(Gc))),
          "bz_compute_scalar_tendency", ctx)
end

##### ---------------------------------------------------------------------------------------------------------------
##### Physics attachments of the BOMEX / supercell configurations: the context builder calls these once, right after
##### bz_create, when the model carries the corresponding component (include/breeze_hip.h, same section names).
##### ---------------------------------------------------------------------------------------------------------------
struct BzSmagorinskyLilly
    smagorinsky_coefficient::Cdouble; reduction_factor::Cdouble; prandtl_number::Cdouble
end

struct BzColumnForcings
    u_forcing::Ptr{Cdouble}; v_forcing::Ptr{Cdouble}; theta_forcing::Ptr{Cdouble}; moisture_forcing::Ptr{Cdouble}
    energy_forcing::Ptr{Cdouble}; subsidence_vertical_velocity::Ptr{Cdouble}
    subsidence_u::Int32; subsidence_v::Int32; subsidence_theta::Int32; subsidence_moisture::Int32
    coriolis_f::Cdouble; bottom_theta_flux::Cdouble; bottom_moisture_flux::Cdouble; bottom_drag_rho0_ustar2::Cdouble; bottom_drag_epsilon::Cdouble
    bottom_energy_flux::Cdouble      # a constant flux keyed ρe in a θ model: Q / cᵖᵐ enters ρθ (EnergyFluxBoundaryCondition)
end

struct BzTracerFields
    density::Ptr{Cdouble}; specific::Ptr{Cdouble}; U0::Ptr{Cdouble}; G::Ptr{Cdouble}
end

"closure = SmagorinskyLilly(): νₑ is `model.closure_fields.νₑ` (src/AtmosphereModels/atmosphere_model.jl:272-276)."
function attach_closure!(ctx, closure, νₑ)
    c = closure.coefficient              # LillyCoefficient(smagorinsky, reduction_factor); Pr from closure.Pr
    cl = BzSmagorinskyLilly(c.smagorinsky, c.reduction_factor, first(values(closure.Pr)))
    check(ccall((:bz_set_closure, libbreeze_hip), Cint, (Ptr{Cvoid}, Ref{BzSmagorinskyLilly}, Ptr{Cdouble}),
                ctx, cl, pointer(parent(νₑ))), "bz_set_closure", ctx)
end

"""
Column forcings of examples/bomex.jl:104-207.  `profiles` holds host `Vector{Float64}`s evaluated once from the model's
materialized forcings (Array(interior(field, 1, 1, :))): geostrophic -f vᵍ / +f uᵍ, Forcing(field) profiles, the subsidence wˢ
on faces; `flags` says which specific fields carry a SubsidenceForcing; fluxes are the bottom FluxBoundaryCondition values.
"""
function attach_forcings!(ctx, profiles::NamedTuple, flags::NamedTuple, f, Jθ, Jq, ρ₀u★², ϵ = 0.0, 𝒬 = 0.0)
    p(name) = haskey(profiles, name) ? pointer(profiles[name]) : Ptr{Cdouble}(C_NULL)
    F = BzColumnForcings(p(:u), p(:v), p(:θ), p(:q), p(:e), p(:wˢ),
                         flags.u, flags.v, flags.θ, flags.q, f, Jθ, Jq, ρ₀u★², ϵ, 𝒬)      # ϵ: the drag's regulariser (convective_boundary_layer.jl:142-146)
    GC.@preserve profiles check(ccall((:bz_set_forcings, libbreeze_hip), Cint, (Ptr{Cvoid}, Ref{BzColumnForcings}), ctx, F),
                                "bz_set_forcings", ctx)
end

struct BzColumnRelaxation
    rate_u::Ptr{Cdouble}; target_u::Ptr{Cdouble}; rate_v::Ptr{Cdouble}; target_v::Ptr{Cdouble}; rate_w::Ptr{Cdouble}; target_w::Ptr{Cdouble}
    rate_theta::Ptr{Cdouble}; target_theta::Ptr{Cdouble}; rate_moisture::Ptr{Cdouble}; target_moisture::Ptr{Cdouble}
    specific_mask::Int32
end

"""
Sponge layers: every `Relaxation(rate, mask::GaussianMask{:z}, target)` of `model.forcing` (examples/rico.jl:103-105,164;
neutral_atmospheric_boundary_layer.jl:103-136) as two host columns, `rate .* mask.(z)` and `target.(z)` at the field's vertical nodes
(ρw / w: the Nz + 1 faces).  `columns = (; u = (rate, target), w = ..., θ = ..., q = ...)`; `specific` lists the entries keyed by the
specific names u, v, w (their forcing is ρᵣ F).
"""
function attach_relaxation!(ctx, columns::NamedTuple, specific = ())
    p(name, i) = haskey(columns, name) ? pointer(columns[name][i]) : Ptr{Cdouble}(C_NULL)
    mask = Int32((:u in specific) + 2 * (:v in specific) + 4 * (:w in specific))
    R = BzColumnRelaxation(p(:u, 1), p(:u, 2), p(:v, 1), p(:v, 2), p(:w, 1), p(:w, 2), p(:θ, 1), p(:θ, 2), p(:q, 1), p(:q, 2), mask)
    GC.@preserve columns check(ccall((:bz_set_relaxation, libbreeze_hip), Cint, (Ptr{Cvoid}, Ref{BzColumnRelaxation}), ctx, R),
                               "bz_set_relaxation", ctx)
end

"""
`Forcing(f(x, y, z, t))` / `Forcing(field)` on θ / e (specific = true) or ρθ / ρe: `F` is a CenterField the extension fills from the forcing
(and refreshes from a callback when it depends on time); the library reads it at every tendency evaluation.
"""
attach_field_forcing!(ctx, F, specific::Bool) =
    check(ccall((:bz_set_field_forcing, libbreeze_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint), ctx, pointer(parent(F)), specific),
          "bz_set_field_forcing", ctx)

"tracers = (:a, :b): density = model.tracers[n]; `specific` is an extra centre field owned by the extension."
function attach_tracers!(ctx, model, specific_fields)
    ts = model.timestepper
    t = [BzTracerFields(pointer(parent(model.tracers[n])), pointer(parent(specific_fields[n])),
                        pointer(parent(ts.U⁰[n])), pointer(parent(ts.Gⁿ[n]))) for n in keys(model.tracers)]
    check(ccall((:bz_set_tracers, libbreeze_hip), Cint, (Ptr{Cvoid}, Int32, Ptr{BzTracerFields}), ctx, length(t), t),
          "bz_set_tracers", ctx)
end

"compute_flux_bc_tendencies!(model) (src/AtmosphereModels/update_atmosphere_model_state.jl:418-434)"
function OceananigansTimeSteppers.compute_flux_bc_tendencies!(model::HIPModel)
    ctx = context(model)
    check(ccall((:bz_compute_flux_bc_tendencies, libbreeze_hip), Cint, (Ptr{Cvoid}, Ref{BzState}, Ref{BzPrognostic}),
                ctx, state(model), prognostic(model.timestepper.Gⁿ)), "bz_compute_flux_bc_tendencies", ctx)
    return nothing
end

##### ---------------------------------------------------------------------------------------------------------------
##### CompressibleDynamics + SplitExplicitTimeDiscretization (src/TimeSteppers/acoustic_runge_kutta_3.jl,
##### src/CompressibleEquations/acoustic_substepping.jl)
##### ---------------------------------------------------------------------------------------------------------------
using Breeze.TimeSteppers: AcousticRungeKutta3
using Breeze.CompressibleEquations: CompressibleDynamics, ThermalDivergenceDamping, ProportionalSubsteps, ConstantSubstepSize, MonolithicFirstStage
import Breeze.CompressibleEquations: acoustic_rk3_substep_loop!

struct BzCompressibleState
    ρᵈ::Ptr{Float64}; ρ::Ptr{Float64}; ρu::Ptr{Float64}; ρv::Ptr{Float64}; ρw::Ptr{Float64}; ρθ::Ptr{Float64}; ρq::Ptr{Float64}
    u::Ptr{Float64}; v::Ptr{Float64}; w::Ptr{Float64}; θ::Ptr{Float64}; q::Ptr{Float64}; T::Ptr{Float64}; p::Ptr{Float64}
end
struct BzCompressiblePrognostic
    ρᵈ::Ptr{Float64}; ρu::Ptr{Float64}; ρv::Ptr{Float64}; ρw::Ptr{Float64}; ρθ::Ptr{Float64}; ρq::Ptr{Float64}
end
struct BzAcousticSubstepper       # field order of include/breeze_hip.h: bz_acoustic_substepper
    Π::Ptr{Float64}; θ::Ptr{Float64}; γR::Ptr{Float64}
    ρ′::Ptr{Float64}; ρθ′::Ptr{Float64}; ρu′::Ptr{Float64}; ρv′::Ptr{Float64}; ρw′::Ptr{Float64}
    ρ′★::Ptr{Float64}; ρθ′★::Ptr{Float64}; ρθ′ˢ⁻::Ptr{Float64}
    ū::Ptr{Float64}; v̄::Ptr{Float64}; w̄::Ptr{Float64}
    Gˢρw::Ptr{Float64}; rhs::Ptr{Float64}
end
struct BzSplitExplicit
    substeps::Int32; damp_vertical::Int32; apply_first::Int32; newton_maxiter::Int32
    acoustic_cfl::Float64; forward_weight::Float64; damping_coefficient::Float64
    f_θ::Float64; f_w::Float64; newton_abstol::Float64
    direct_divergence_damping::Int32; sponge_ramp::Int32          # ramp: 0 none, 1 LinearRamp, 2 CubicRamp, 3 Sin2Ramp
    sponge_damping_rate::Float64; sponge_depth::Float64
    substep_distribution::Int32                                   # 0 ProportionalSubsteps, 1 ConstantSubstepSize, 2 MonolithicFirstStage
    substep_float_bytes::Int32                                    # 0: eltype(grid); 4: the substepper's working fields are Float32 arrays
    damping_length_scale::Float64                                 # ThermalDivergenceDamping(length_scale): 0 = nothing (local scale)
end
distcode(::ProportionalSubsteps) = Int32(0)
distcode(::ConstantSubstepSize) = Int32(1)
distcode(::MonolithicFirstStage) = Int32(2)
rampcode(::Nothing) = Int32(0)
rampcode(s::UpperSponge) = s.ramp isa LinearRamp ? Int32(1) : s.ramp isa CubicRamp ? Int32(2) : s.ramp isa Sin2Ramp ? Int32(3) :
                           error("BreezeHIP: custom sponge ramps are not supported")
struct BzExnerReference
    pst::Float64; pressure::Ptr{Float64}; density::Ptr{Float64}
end

const HIPAcousticModel = AtmosphereModel{<:CompressibleDynamics, <:Any, <:HIPNative, <:AcousticRungeKutta3}

cstate(m) = BzCompressibleState(devptr(m.dynamics.dry_density), devptr(m.dynamics.total_density),
                                devptr(m.momentum.ρu), devptr(m.momentum.ρv), devptr(m.momentum.ρw),
                                devptr(m.formulation.potential_temperature_density), devptr(m.moisture_density),
                                devptr(m.velocities.u), devptr(m.velocities.v), devptr(m.velocities.w),
                                devptr(m.formulation.potential_temperature), devptr(m.microphysical_fields.qᵛ),
                                devptr(m.temperature), devptr(m.dynamics.pressure))
cprognostic(nt) = BzCompressiblePrognostic(devptr(nt.ρᵈ), devptr(nt.ρu), devptr(nt.ρv), devptr(nt.ρw), devptr(nt.ρθ), devptr(nt.ρqᵛ))
substepper(a) = BzAcousticSubstepper(devptr(a.linearization_exner), devptr(a.linearization_potential_temperature),
                                     devptr(a.linearization_gamma_R_mixture), devptr(a.density_perturbation),
                                     devptr(a.density_potential_temperature_perturbation),
                                     devptr(a.momentum_perturbation.u), devptr(a.momentum_perturbation.v), devptr(a.momentum_perturbation.w),
                                     devptr(a.density_predictor), devptr(a.density_potential_temperature_predictor),
                                     devptr(a.previous_density_potential_temperature_perturbation),
                                     devptr(a.time_averaged_velocities.u), devptr(a.time_averaged_velocities.v), devptr(a.time_averaged_velocities.w),
                                     devptr(a.slow_vertical_momentum_tendency), devptr(a.vertical_solver_source_term))

function create_compressible_context(model)
    grid = model.grid
    Nx, Ny, Nz = size(grid); Hx, Hy, Hz = Oceananigans.Grids.halo_size(grid)
    zf = collect(Float64, znodes(grid, Face()))
    c = model.thermodynamic_constants
    a = model.timestepper.substepper
    ref = model.dynamics.reference_state
    hostcol(f) = collect(Float64, vec(Array(parent(f))))
    p, ρ = ref === nothing ? (Float64[], Float64[]) : (hostcol(ref.pressure), hostcol(ref.density))
    solver = model.formulation.temperature_solver                      # NewtonSolver(reltol = 0)
    damp = a.damping
    td = BzSplitExplicit(something(a.substeps, 0), damp isa ThermalDivergenceDamping && damp.damp_vertical,
                         a.apply_first_substep_pressure_gradient, solver.maxiter, a.acoustic_cfl, a.forward_weight,
                         damp isa Union{ThermalDivergenceDamping, DirectDivergenceDamping} ? damp.coefficient : -1.0,
                         a.thermodynamic_tendency_factor, a.vertical_momentum_tendency_factor, solver.abstol,
                         damp isa DirectDivergenceDamping, rampcode(a.sponge),
                         a.sponge === nothing ? 0.0 : a.sponge.damping_rate, a.sponge === nothing ? 0.0 : a.sponge.depth,
                         distcode(a.substep_distribution),
                         eltype(a.density_perturbation) === Float32 && eltype(grid) === Float64 ? Int32(4) : Int32(0),
                         damp isa ThermalDivergenceDamping && damp.length_scale !== nothing ? Float64(damp.length_scale) : 0.0)
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve zf p ρ begin
        g = BzGrid(Nx, Ny, Nz, Hx, Hy, Hz, map(topocode, topology(grid)), 8, grid.Δxᶜᵃᵃ, grid.Δyᵃᶜᵃ, pointer(zf),
                   grid.z.Δᵃᵃᶜ isa Number ? 1 : 0, 0)
        k = BzConstants(c.gravitational_acceleration, c.molar_gas_constant / c.dry_air.molar_mass,
                        c.molar_gas_constant / c.vapor.molar_mass, c.dry_air.heat_capacity, c.vapor.heat_capacity)
        r = BzExnerReference(model.dynamics.standard_pressure, ref === nothing ? C_NULL : pointer(p), ref === nothing ? C_NULL : pointer(ρ))
        rc = ccall((:bz_create_compressible, libbreeze_hip), Cint,
                   (Ref{Ptr{Cvoid}}, Ref{BzGrid}, Ref{BzConstants}, Ref{BzExnerReference}, Ref{BzSplitExplicit}, Cint), ctx, g, k, r, td, 5)
        rc == 0 || error("bz_create_compressible failed with code $rc")
    end
    # Bounded x / y: which sides carry an active open boundary condition on the wall-normal momentum (is_active_open_bc,
    # src/CompressibleEquations/acoustic_substepping.jl:1318) and the relaxation factor of the substepper; the loop seam below then enforces the
    # lateral boundaries itself (_zero_{x,y}_wall_face!, _relax_open_boundary_{x,y}!, :1323-1395).  On such a grid only the loop seam is forwarded:
    # the fills with the model's boundary conditions and compute_velocities! after _recover_full_state! (:1584-1587) stay on the Julia side.
    TX, TY, _ = topology(grid)
    if TX === Bounded || TY === Bounded
        bu, bv = model.momentum.ρu.boundary_conditions, model.momentum.ρv.boundary_conditions
        active(bc) = (bc isa BoundaryCondition{<:NormalFlow}) && !(bc.condition isa Nothing)
        rc = ccall((:bz_set_acoustic_lateral_boundaries, libbreeze_hip), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Cdouble), ctx[],
                   TX === Bounded && active(bu.west), TX === Bounded && active(bu.east), TY === Bounded && active(bv.south),
                   TY === Bounded && active(bv.north), a.open_boundary_relaxation)
        check(rc, "bz_set_acoustic_lateral_boundaries", ctx[])
    end
    return ctx[]
end
ccontext(model) = get!(() -> create_compressible_context(model), CONTEXTS, model)

# Whole-step seam (src/TimeSteppers/acoustic_runge_kutta_3.jl:264-319)
function OceananigansTimeSteppers.time_step!(model::HIPAcousticModel, Δt; callbacks=[])
    ctx = ccontext(model)
    OceananigansTimeSteppers.maybe_prepare_first_time_step!(model, Δt, callbacks)
    ts = model.timestepper
    rc = ccall((:bz_time_step_compressible, libbreeze_hip), Cint,
               (Ptr{Cvoid}, Ref{BzCompressibleState}, Ref{BzCompressiblePrognostic}, Ref{BzCompressiblePrognostic}, Ref{BzAcousticSubstepper}, Cdouble),
               ctx, cstate(model), cprognostic(ts.U⁰), cprognostic(ts.Gⁿ), substepper(ts.substepper), Δt)
    check(rc, "bz_time_step_compressible", ctx)
    tick!(model.clock, Δt)
    return nothing
end

# Loop seam (src/CompressibleEquations/acoustic_substepping.jl:1404)
function acoustic_rk3_substep_loop!(model::HIPAcousticModel, a, Δt, β_stage, U⁰)
    ctx = ccontext(model)
    rc = ccall((:bz_acoustic_substep_loop, libbreeze_hip), Cint,
               (Ptr{Cvoid}, Ref{BzCompressibleState}, Ref{BzCompressiblePrognostic}, Ref{BzCompressiblePrognostic}, Ref{BzAcousticSubstepper}, Cdouble, Cdouble),
               ctx, cstate(model), cprognostic(U⁰), cprognostic(model.timestepper.Gⁿ), substepper(a), Δt, β_stage)
    check(rc, "bz_acoustic_substep_loop", ctx)
    return nothing
end

function OceananigansTimeSteppers.update_state!(model::HIPAcousticModel, callbacks=[]; compute_tendencies=true)
    ctx = ccontext(model)
    ts = model.timestepper
    rc = ccall((:bz_compressible_update_state, libbreeze_hip), Cint,
               (Ptr{Cvoid}, Ref{BzCompressibleState}, Ref{BzCompressiblePrognostic}, Ref{BzAcousticSubstepper}, Cint),
               ctx, cstate(model), cprognostic(ts.Gⁿ), substepper(ts.substepper), compute_tendencies ? 1 : 0)
    check(rc, "bz_compressible_update_state", ctx)
    return nothing
end


#####
##### Multi-GPU: one process per GPU, y-slabs, the communicator lives inside libbreeze_hip (bz_comm.hip)
#####
##### The rank-local RectilinearGrid is this rank's y-slab of the global (Periodic, Periodic, Bounded) domain.  MPI.jl (or any
##### launcher) is used for exactly two things: starting the ranks and broadcasting RCCL's 128-byte unique id.
#####

const BZ_UNIQUE_ID_BYTES = 128

"WENO(order = N) of the model's momentum scheme -> the weno_order argument of bz_create (5: tuned kernels; 7, 9: generic kernels)."
weno_order(advection) = 2 * Oceananigans.Advection.required_halo_size_x(advection isa NamedTuple ? advection.momentum : advection) - 1

"The scalars' order of a merged advection NamedTuple (atmosphere_model.jl:279-284); every scalar must share it (bz_set_scalar_advection_order)."
function scalar_weno_order(advection)
    advection isa NamedTuple || return weno_order(advection)
    orders = unique(2 * Oceananigans.Advection.required_halo_size_x(s) - 1 for (k, s) in pairs(advection) if k != :momentum)
    length(orders) <= 1 || error("BreezeHIP: one advection order for all scalars (got $orders)")
    return isempty(orders) ? weno_order(advection) : first(orders)
end

"Create the slab context of this rank and attach the RCCL communicator to it."
function slab_context!(model, y_nranks::Integer, y_rank::Integer, bcast_bytes!::Function)
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    grid, constants, reference = bz_grid(model.grid), bz_constants(model), bz_reference_state(model)
    rc = ccall((:bz_create_slab, libbreeze_hip), Cint,
               (Ref{Ptr{Cvoid}}, Ref{BzGrid}, Ref{BzConstants}, Ref{BzReferenceState}, Cint, Cint, Cint),
               ctx, grid, constants, reference, weno_order(model.advection), y_nranks, y_rank)
    rc == 0 || error("bz_create_slab failed with code $rc")
    id = zeros(UInt8, BZ_UNIQUE_ID_BYTES)
    if y_rank == 0
        rc = ccall((:bz_comm_unique_id, libbreeze_hip), Cint, (Ptr{UInt8},), id)
        rc == 0 || error("bz_comm_unique_id failed with code $rc")
    end
    bcast_bytes!(id)                     # e.g. id -> MPI.Bcast!(id, 0, MPI.COMM_WORLD)
    rc = ccall((:bz_comm_init_rccl, libbreeze_hip), Cint, (Ptr{Cvoid}, Ptr{UInt8}), ctx[], id)
    rc == 0 || error(unsafe_string(ccall((:bz_last_error, libbreeze_hip), Cstring, (Ptr{Cvoid},), ctx[])))
    return ctx[]
end

"set!(model; ...) on slabs ends with update_state! + halo exchange + the initial projection, all inside the library."
function finish_set!(model, ctx; enforce_mass_conservation = true)
    rc = ccall((:bz_comm_update_state_and_project, libbreeze_hip), Cint,
               (Ptr{Cvoid}, Ref{BzState}, Ref{BzPrognostic}, Cdouble, Cint),
               ctx, state(model), prognostic(model.timestepper.Gⁿ), 1.0, enforce_mass_conservation ? 1 : 0)
    rc == 0 || error(unsafe_string(ccall((:bz_last_error, libbreeze_hip), Cstring, (Ptr{Cvoid},), ctx)))
    return nothing
end

# time_step! needs no distributed method: bz_time_step_anelastic on a slab context with a communicator is the distributed step.

"Compressible model on slabs: the same communicator calls on a context made by bz_create_compressible_slab; time_step! is then
bz_time_step_compressible (per-substep halo exchange of the acoustic loop inside the library), update_state! the call below."
function compressible_slab_context!(model, y_nranks::Integer, y_rank::Integer, bcast_bytes!::Function, g, k, r, td)
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:bz_create_compressible_slab, libbreeze_hip), Cint,
               (Ref{Ptr{Cvoid}}, Ref{BzGrid}, Ref{BzConstants}, Ref{BzExnerReference}, Ref{BzSplitExplicit}, Cint, Cint, Cint),
               ctx, g, k, r, td, weno_order(model.advection), y_nranks, y_rank)
    rc == 0 || error("bz_create_compressible_slab failed with code $rc")
    id = zeros(UInt8, BZ_UNIQUE_ID_BYTES)
    if y_rank == 0
        rc = ccall((:bz_comm_unique_id, libbreeze_hip), Cint, (Ptr{UInt8},), id)
        rc == 0 || error("bz_comm_unique_id failed with code $rc")
    end
    bcast_bytes!(id)
    rc = ccall((:bz_comm_init_rccl, libbreeze_hip), Cint, (Ptr{Cvoid}, Ptr{UInt8}), ctx[], id)
    rc == 0 || error(unsafe_string(ccall((:bz_last_error, libbreeze_hip), Cstring, (Ptr{Cvoid},), ctx[])))
    return ctx[]
end

function compressible_update_state!(model, ctx; compute_tendencies = true)
    rc = ccall((:bz_comm_compressible_update_state, libbreeze_hip), Cint,
               (Ptr{Cvoid}, Ref{BzCompressibleState}, Ref{BzCompressiblePrognostic}, Ref{BzAcousticSubstepper}, Cint),
               ctx, cstate(model), cprognostic(model.timestepper.Gⁿ), substepper(model.timestepper.substepper), compute_tendencies ? 1 : 0)
    rc == 0 || error(unsafe_string(ccall((:bz_last_error, libbreeze_hip), Cstring, (Ptr{Cvoid},), ctx)))
    return nothing
end

# hipGraph replay of whole steps (csrc/bz_graph.hip): opt-in, pays with a fixed Δt on launch-bound grids; Oceananigans fields keep
# their device arrays for the life of the model, which is what a recorded step relies on
graph_replay!(ctx, on::Bool = true) = check(ccall((:bz_graph_enable, libbreeze_hip), Cint, (Ptr{Cvoid}, Cint), ctx, on), "bz_graph_enable", ctx)

end # module
