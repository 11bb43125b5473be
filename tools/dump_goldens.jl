# dump_goldens.jl — run on a box with Julia >= 1.11, Breeze 0.9 and Oceananigans 0.110.x (NOT available in the
# build image; this script has never been executed).  Builds the parity-test states with the public Breeze API on CPU(),
# advances them and writes parent(field) arrays as raw Float64 .bin files (column-major, i fastest) + a JSON manifest under
# tests/golden/reference/<case>/.  tests/test_reference_goldens.py picks every manifest up automatically and compares
# the oracle (CPU) and the HIP path (GPU) against this true reference output, closing SURVEY.md Appendix D.
#     julia --project tools/dump_goldens.jl
using Breeze, Oceananigans, JSON

const EXTENT = (x=(-10e3, 10e3), y=(-10e3, 10e3), z=(0, 10e3))

# kind: "anelastic_weno5" | "anelastic_centered2" | "compressible_weno5" (round 1), and the options whose Oceananigans semantics the CPU oracle
# restates from memory (DESIGN.md §2 "parity unpinned"): "anelastic_weno9" (buffer cascade 9 -> 1 next to the z walls),
# "anelastic_walls_y" = (Periodic, Bounded, Bounded) and "anelastic_walls_x" = (Bounded, Flat, Bounded) (wall faces, no-flux halos, DCT
# solve), "anelastic_smagorinsky" (SmagorinskyLilly), "anelastic_mixed_orders" (momentum WENO9 + scalars WENO5)
function bubble_model(kind, size, halo)
    topology = kind == "anelastic_walls_y" ? (Periodic, Bounded, Bounded) : kind == "anelastic_walls_x" ? (Bounded, Flat, Bounded) :
               (Periodic, Periodic, Bounded)
    extent = kind == "anelastic_walls_x" ? (x=EXTENT.x, z=EXTENT.z) : EXTENT
    grid = RectilinearGrid(CPU(); size, halo, extent..., topology)
    constants = ThermodynamicConstants()
    g = constants.gravitational_acceleration
    # round-1 kinds: the centred bubble; the newer kinds: off-centre, next to the west / south walls of the walled cases
    x₀, y₀ = kind in ("anelastic_weno5", "anelastic_centered2", "compressible_weno5") ? (0, 0) : (-4000, -5000)
    θ₃(x, y, z) = 300 * exp(1e-6 * z / g) + 10 * max(0, 1 - sqrt((x - x₀)^2 + (y - y₀)^2 + (z - 3000)^2) / 2000)
    θᵢ = kind == "anelastic_walls_x" ? ((x, z) -> θ₃(x, y₀, z)) : θ₃
    if kind == "compressible_weno5"
        dynamics = CompressibleDynamics(SplitExplicitTimeDiscretization(substeps=6); reference_potential_temperature=300)
        model = AtmosphereModel(grid; dynamics, advection=WENO(order=5))
        ρᵣ = model.dynamics.reference_state.density
        set!(model, ρ=ρᵣ, θ=θᵢ, u=3, v=-2)
    else
        reference_state = ReferenceState(grid, constants; surface_pressure=101325, potential_temperature=300)
        dynamics = AnelasticDynamics(reference_state)
        if kind == "anelastic_mixed_orders"
            model = AtmosphereModel(grid; dynamics, momentum_advection=WENO(order=9), scalar_advection=WENO(order=5))
        elseif kind == "anelastic_smagorinsky"
            model = AtmosphereModel(grid; dynamics, advection=WENO(order=5), closure=SmagorinskyLilly())
        else
            advection = kind == "anelastic_centered2" ? Centered(order=2) : kind == "anelastic_weno9" ? WENO(order=9) : WENO(order=5)
            model = AtmosphereModel(grid; dynamics, advection)
        end
        if kind == "anelastic_walls_x"
            set!(model, θ=θᵢ)
        elseif kind == "anelastic_walls_y"
            set!(model, θ=θᵢ, u=3)                      # no flow through the walls
        else
            set!(model, θ=θᵢ, u=3, v=-2)
        end
    end
    return model
end

function dump(name, kind, size, halo; Δt, steps)
    model = bubble_model(kind, size, halo)
    outdir = joinpath(@__DIR__, "..", "tests", "golden", "reference", name); mkpath(outdir)
    manifest = Dict("kind" => kind, "size" => collect(size), "halo" => collect(halo), "dt" => Δt, "steps" => steps,
                    "breeze" => string(pkgversion(Breeze)), "oceananigans" => string(pkgversion(Oceananigans)),
                    # the scheme's type parameters: WENO{N, FT, FT2, ...} — FT2 is the float type of the weight computation (SURVEY
                    # App. D.1); the reader (tests/test_reference_goldens.py) starts with the matching reading of the oracle / kernels
                    "advection_type" => string(typeof(model.advection)), "eltype" => string(eltype(model.grid)),
                    "fields" => Dict())
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

dump("bubble_w9_32x20x16", "anelastic_weno9", (32, 20, 16), (5, 5, 5); Δt=2.0, steps=3)
dump("bubble_mixed_32x20x16", "anelastic_mixed_orders", (32, 20, 16), (5, 5, 5); Δt=2.0, steps=3)
dump("bubble_smag_32x20x16", "anelastic_smagorinsky", (32, 20, 16), (3, 3, 3); Δt=2.0, steps=3)
dump("bubble_walls_y_32x16x16", "anelastic_walls_y", (32, 16, 16), (3, 3, 3); Δt=2.0, steps=3)
dump("bubble_walls_x_64x32", "anelastic_walls_x", (64, 32), (5, 5); Δt=2.0, steps=3)
