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
                   (Ref{Ptr{Cvoid}}, Ref{BzGrid}, Ref{BzConstants}, Ref{BzReferenceState}, Cint), ctx, g, k, r, 5)
        rc == 0 || error("bz_create failed with code $rc")
    end
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

end # module
