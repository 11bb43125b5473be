# dump_goldens.jl — run on a box with Julia >= 1.11, Breeze 0.9 and Oceananigans 0.110.x (NOT available in the
# build image).  Builds the parity-test states with the public Breeze API on CPU(), advances them and writes
# parent(field) arrays as raw Float64 .bin files + a JSON manifest under tests/golden/reference/.  tests/ can then
# compare the oracle and the HIP path against true reference output (closing SURVEY.md Appendix D).
using Breeze, Oceananigans, JSON

function dump(name, size, halo; Δt, steps)
    grid = RectilinearGrid(CPU(); size, halo, x=(-10e3, 10e3), y=(-10e3, 10e3), z=(0, 10e3),
                           topology=(Periodic, Periodic, Bounded))
    constants = ThermodynamicConstants()
    reference_state = ReferenceState(grid, constants; surface_pressure=101325, potential_temperature=300)
    model = AtmosphereModel(grid; dynamics=AnelasticDynamics(reference_state), advection=WENO(order=5))
    g = constants.gravitational_acceleration
    θᵢ(x, y, z) = 300 * exp(1e-6 * z / g) + 10 * max(0, 1 - sqrt(x^2 + y^2 + (z - 3000)^2) / 2000)
    set!(model, θ=θᵢ, u=3, v=-2)
    outdir = joinpath(@__DIR__, "..", "tests", "golden", "reference", name); mkpath(outdir)
    manifest = Dict("size" => size, "halo" => halo, "dt" => Δt, "steps" => steps, "fields" => Dict())
    function save(tag)
        fields = merge(Oceananigans.prognostic_fields(model), model.velocities,
                       (; T=model.temperature, ϕ=model.dynamics.pressure_anomaly), model.timestepper.Gⁿ |> nt -> NamedTuple{Tuple(Symbol("G", k) for k in keys(nt))}(values(nt)))
        for (k, f) in pairs(fields)
            file = "$(tag)_$(k).bin"
            write(joinpath(outdir, file), Array{Float64}(parent(f)))
            manifest["fields"]["$(tag)_$(k)"] = Dict("file" => file, "shape" => collect(Base.size(parent(f))))
        end
    end
    save("step0")
    for n in 1:steps
        time_step!(model, Δt)
        save("step$n")
    end
    open(joinpath(outdir, "manifest.json"), "w") do io; JSON.print(io, manifest, 2); end
end

dump("bubble_32x20x16", (32, 20, 16), (3, 3, 3); Δt=2.0, steps=3)
dump("bubble_64x8x32", (64, 8, 32), (3, 3, 3); Δt=1.0, steps=3)
