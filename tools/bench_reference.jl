# bench_reference.jl — the true reference CPU number for BASELINE.md §3 (needs Julia + Breeze; cannot run in the build image).
#   julia -t <cores> tools/bench_reference.jl [N]
# Mirrors benchmarking/src/utils.jl:40-172 (warm-up, timed steps, grid_points_per_second).
using Breeze, Oceananigans
N = length(ARGS) > 0 ? parse(Int, ARGS[1]) : 128
grid = RectilinearGrid(CPU(); size=(N, N, N), halo=(3, 3, 3), x=(-10e3, 10e3), y=(-10e3, 10e3), z=(0, 10e3),
                       topology=(Periodic, Periodic, Bounded))
reference_state = ReferenceState(grid, ThermodynamicConstants(); surface_pressure=101325, potential_temperature=300)
model = AtmosphereModel(grid; dynamics=AnelasticDynamics(reference_state), advection=WENO(order=5))
θᵢ(x, y, z) = 300 * exp(1e-6 * z / 9.81) + 10 * max(0, 1 - sqrt(x^2 + y^2 + (z - 3000)^2) / 2000)
set!(model, θ=θᵢ)
for _ in 1:3; time_step!(model, 1.0); end
steps = 10
t = @elapsed for _ in 1:steps; time_step!(model, 1.0); end
println("reference CPU(): ", N^3 * steps / t, " cells/s on ", Threads.nthreads(), " threads")
