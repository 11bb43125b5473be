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
json.dump(out, sys.stdout, indent=1, ensure_ascii=False)
sys.stdout.write("\n")
