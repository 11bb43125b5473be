#!/usr/bin/env python
"""tools/make_golden_doctests.py — extract the reference-generated numbers of Breeze.jl's own jldoctests.

The reference cannot run in the build container (no Julia), but its docstrings carry outputs that its test suite
(test/doctests.jl) verifies against the real code, so they are genuine reference outputs.  This script reads the doctest
blocks (DATA: inputs written in the docstring + the printed result) from /root/reference and writes them as a small JSON
fixture; no reference source text is copied.

    python tools/make_golden_doctests.py [/root/reference] > tests/golden/reference_doctests.json
"""
import json
import re
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"


def grab(path, pattern, cast=float):
    text = open(f"{root}/{path}", encoding="utf-8").read()
    m = re.search(pattern, text, re.S)
    if not m:
        raise SystemExit(f"pattern not found in {path}: {pattern}")
    return cast(m.group(1))


vs = "src/Thermodynamics/vapor_saturation.jl"
out = {
    "_comment": "Outputs of the reference's jldoctests (reference-generated; inputs as written in the docstrings). "
                "Made by tools/make_golden_doctests.py.",
    "saturation_specific_humidity": {
        "source": f"{vs}:57-91",
        "inputs": {"T": 288.0, "p": 101325.0, "moisture_mass_fractions": [0.0, 0.0, 0.0],
                   "rho": "density(T, p, q, constants) = p / (Rd T)", "constants": "ThermodynamicConstants() defaults"},
        "PlanarLiquidSurface": grab(vs, r"PlanarLiquidSurface\(\)\)\n\n# output\n([0-9.eE+-]+)"),
        "PlanarIceSurface": grab(vs, r"PlanarIceSurface\(\)\)\n([0-9.eE+-]+)"),
        "PlanarMixedPhaseSurface(0.4)": grab(vs, r"mixed_surface\)\n\n# output\n([0-9.eE+-]+)"),
    },
    "pressure_balanced_density": {
        "source": "src/Thermodynamics/reference_states.jl:140-151",
        "inputs": {"rho_background": 1.0, "theta_background": 300.0, "theta_initial": 303.0},
        "output": grab("src/Thermodynamics/reference_states.jl",
                       r"pressure_balanced_density\(ρ_background, θ_background, θ_initial\)\n\n# output\n([0-9.eE+-]+)"),
    },
    "newton_solve": {
        "source": "src/Solvers.jl:178-188",
        "inputs": {"residual": "x^2 - 2", "derivative": "2x", "x0": 1.0, "reltol": 1e-12, "abstol": 1e-4, "maxiter": 20,
                   "round_digits": 10},
        "output": grab("src/Solvers.jl", r"newton_solve\(x -> \(x\^2 - 2, 2x\), solver, 1\.0\)\nround\(x, digits=10\)\n\n# output\n([0-9.eE+-]+)"),
    },
    "secant_solve": {
        "source": "src/Solvers.jl:231-241",
        "inputs": {"residual": "x^2 - 2", "x1": 1.0, "x2": 2.0, "scale": 1.0, "abstol": 1e-12, "reltol": 0.0, "maxiter": 20,
                   "round_digits": 10},
        "output": grab("src/Solvers.jl", r"secant_solve\(x -> x\^2 - 2, solver, 1\.0, 2\.0, 1\.0\)\nround\(x, digits=10\)\n\n# output\n([0-9.eE+-]+)"),
    },
}


def stats(path, after):
    """max / min / mean of the Field summary printed after the first occurrence of `after` in the docstrings of `path`."""
    text = open(f"{root}/{path}", encoding="utf-8").read()
    i = text.index(after)
    m = re.search(r"max=([0-9.eE+-]+), min=([0-9.eE+-]+), mean=([0-9.eE+-]+)", text[i:])
    return {"max": float(m.group(1)), "min": float(m.group(2)), "mean": float(m.group(3))}


# model-level doctests: AtmosphereModel(grid) with every default (ReferenceState p0 = 101325 Pa, theta0 = 288 K, p_st = 1e5 Pa) on
# RectilinearGrid(size = (1, 1, N), extent = (1, 1, 1e3)), i.e. z in (-1000, 0)
out["model_diagnostics"] = {
    "static_energy": {
        "source": "src/AtmosphereModels/Diagnostics/static_energy.jl (jldoctest of StaticEnergy)",
        "inputs": {"size": [1, 1, 8], "extent": [1, 1, 1e3], "set": {"theta": 300}},
        **stats("src/AtmosphereModels/Diagnostics/static_energy.jl", "e = StaticEnergy(model)"),
    },
    "virtual_potential_temperature": {
        "source": "src/AtmosphereModels/Diagnostics/potential_temperatures.jl (jldoctest of VirtualPotentialTemperature)",
        "inputs": {"size": [1, 1, 8], "extent": [1, 1, 1e3], "set": {"theta": 300, "qt": 0.01}},
        **stats("src/AtmosphereModels/Diagnostics/potential_temperatures.jl", "θᵛ = VirtualPotentialTemperature(model)\nField(θᵛ)"),
    },
    "potential_temperature": {
        "source": "src/AtmosphereModels/Diagnostics/potential_temperatures.jl (jldoctest of PotentialTemperature)",
        "inputs": {"size": [1, 1, 8], "extent": [1, 1, 1e3], "set": {"theta": 300, "qt": 0.01}},
        **stats("src/AtmosphereModels/Diagnostics/potential_temperatures.jl", "θ = PotentialTemperature(model)\nField(θ)"),
    },
    "liquid_ice_potential_temperature": {
        "source": "src/AtmosphereModels/Diagnostics/potential_temperatures.jl (jldoctest of LiquidIcePotentialTemperature)",
        "inputs": {"size": [1, 1, 8], "extent": [1, 1, 1e3], "set": {"theta": 300, "qt": 0.01}},
        **stats("src/AtmosphereModels/Diagnostics/potential_temperatures.jl", "θˡⁱ = LiquidIcePotentialTemperature(model)\nField(θˡⁱ)"),
    },
    "equivalent_potential_temperature": {
        "source": "src/AtmosphereModels/Diagnostics/potential_temperatures.jl (jldoctest of EquivalentPotentialTemperature)",
        "inputs": {"size": [1, 1, 8], "extent": [1, 1, 1e3], "set": {"theta": 300, "qt": 0.01}},
        **stats("src/AtmosphereModels/Diagnostics/potential_temperatures.jl", "θᵉ = EquivalentPotentialTemperature(model)\nField(θᵉ)"),
    },
    "stability_equivalent_potential_temperature": {
        "source": "src/AtmosphereModels/Diagnostics/potential_temperatures.jl (jldoctest of StabilityEquivalentPotentialTemperature)",
        "inputs": {"size": [1, 1, 8], "extent": [1, 1, 1e3], "set": {"theta": 300, "qt": 0.01}},
        **stats("src/AtmosphereModels/Diagnostics/potential_temperatures.jl", "θᵇ = StabilityEquivalentPotentialTemperature(model)\nField(θᵇ)"),
    },
    "dewpoint_temperature": {
        "source": "src/AtmosphereModels/Diagnostics/dewpoint_temperature.jl (jldoctest dewpoint): SaturationAdjustment() defaults",
        "inputs": {"size": [1, 1, 8], "extent": [1, 1, 1e3], "set": {"theta": 300, "qt": 0.01}},
        **stats("src/AtmosphereModels/Diagnostics/dewpoint_temperature.jl", "T⁺_field = Field(T⁺)"),
    },
    "relative_humidity": {
        "source": "src/Microphysics/microphysics_diagnostics.jl (jldoctest rh): SaturationAdjustment() defaults",
        "inputs": {"size": [1, 1, 128], "extent": [1e3, 1e3, 1e3], "set": {"theta": 300, "qt": 0.005}},
        **stats("src/Microphysics/microphysics_diagnostics.jl", "ℋ_field = RelativeHumidity(model) |> Field"),
    },
}
json.dump(out, sys.stdout, indent=1, ensure_ascii=False)
sys.stdout.write("\n")
