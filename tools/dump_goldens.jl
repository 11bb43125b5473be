# dump_goldens.jl — run on a box with Julia >= 1.11, Breeze 0.9 and Oceananigans 0.110.x (NOT available in the
# build image; this script has never been executed).  Builds the parity-test states with the public Breeze API on CPU(),
# advances them and writes parent(field) arrays as raw Float64 .bin files (column-major, i fastest) + a JSON manifest under
# tests/golden/reference/<case>/.  tests/test_reference_goldens.py picks every manifest up automatically and compares
# the oracle (CPU) and the HIP path (GPU) against this true reference output, closing SURVEY.md Appendix D.
#     julia --project tools/dump_goldens.jl
using Breeze, Oceananigans, JSON

const EXTENT = (x=(-10e3, 10e3), y=(-10e3, 10e3), z=(0, 10e3))

function bubble_model(kind, size, halo)
    grid = RectilinearGrid(CPU(); size, halo, EXTENT..., topology=(Periodic, Periodic, Bounded))
    constants = ThermodynamicConstants()
    g = constants.gravitational_acceleration
    θᵢ(x, y, z) = 300 * exp(1e-6 * z / g) + 10 * max(0, 1 - sqrt(x^2 + y^2 + (z - 3000)^2) / 2000)
    if kind == "compressible_weno5"
        dynamics = CompressibleDynamics(SplitExplicitTimeDiscretization(substeps=6); reference_potential_temperature=300)
        model = AtmosphereModel(grid; dynamics, advection=WENO(order=5))
        ρᵣ = model.dynamics.reference_state.density
        set!(model, ρ=ρᵣ, θ=θᵢ, u=3, v=-2)
    else
        reference_state = ReferenceState(grid, constants; surface_pressure=101325, potential_temperature=300)
        advection = kind == "anelastic_centered2" ? Centered(order=2) : WENO(order=5)
        model = AtmosphereModel(grid; dynamics=AnelasticDynamics(reference_state), advection)
        set!(model, θ=θᵢ, u=3, v=-2)
    end
    return model
end

function dump(name, kind, size, halo; Δt, steps)
    model = bubble_model(kind, size, halo)
    outdir = joinpath(@__DIR__, "..", "tests", "golden", "reference", name); mkpath(outdir)
    manifest = Dict("kind" => kind, "size" => collect(size), "halo" => collect(halo), "dt" => Δt, "steps" => steps,
                    "breeze" => string(pkgversion(Breeze)), "oceananigans" => string(pkgversion(Oceananigans)), "fields" => Dict())
    function save(tag)
        G = model.timestepper.Gⁿ
        fields = merge(Oceananigans.prognostic_fields(model), model.velocities, (; T=model.temperature),
                       NamedTuple{Tuple(Symbol("G", k) for k in keys(G))}(values(G)))
        kind == "compressible_weno5" || (fields = merge(fields, (; ϕ=model.dynamics.pressure_anomaly)))
        kind == "compressible_weno5" && (fields = merge(fields, (; p=model.dynamics.pressure)))
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

dump("bubble_32x20x16", "anelastic_weno5", (32, 20, 16), (3, 3, 3); Δt=2.0, steps=3)
dump("bubble_64x8x32", "anelastic_weno5", (64, 8, 32), (3, 3, 3); Δt=1.0, steps=3)
dump("bubble_c2_32x20x16", "anelastic_centered2", (32, 20, 16), (3, 3, 3); Δt=2.0, steps=3)
dump("bubble_cmp_24x16x20", "compressible_weno5", (24, 16, 20), (3, 3, 3); Δt=2.0, steps=2)
