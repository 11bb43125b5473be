"""True reference output, the day it exists: tools/dump_goldens.jl (run where Julia + Breeze 0.9 + Oceananigans 0.110.x are
installed — not in the build image) writes parent(field) arrays of small bubble runs under tests/golden/reference/<case>/ with a
manifest.json.  This module finds every manifest and compares, step by step,
  * the CPU oracle (always) and
  * the HIP path through the C ABI (-m gpu)
with the reference's own arrays: 1e-12 of the field scale at step 0 (set! + initial projection), 1e-9 after time steps
(SURVEY.md Appendix C last row).  Without manifests the comparisons are skipped; the reader itself is always exercised on a
synthetic case written in the manifest format by the oracle, so the day the real files are dropped in there is a test to turn
green — closing the "parity unpinned" items of SURVEY.md Appendix D (WENO weights, buffer schemes, halo rules, solver digits).

Manifest format: {"kind": "anelastic_weno5" | "anelastic_centered2" | "compressible_weno5", "size": [Nx, Ny, Nz],
"halo": [Hx, Hy, Hz], "dt": .., "steps": n, "fields": {"step<k>_<name>": {"file": .., "shape": [Sx, Sy, Sz]}}}; files are raw
Float64, column-major (i fastest) — read here as (Sz, Sy, Sx) C arrays, the layout of this repo's parent arrays."""
import glob
import json
import os

import numpy as np
import pytest

from helpers import bubble_theta

HERE = os.path.dirname(os.path.abspath(__file__))
EXT = ((-10e3, 10e3), (-10e3, 10e3), (0.0, 10e3))
# reference field name -> oracle attribute (anelastic)
ORACLE_NAMES = {"ρu": "ru", "ρv": "rv", "ρw": "rw", "ρθ": "rtheta", "ρqᵛ": "rq", "ρqᵉ": "rq", "u": "u", "v": "v", "w": "w", "T": "T", "ϕ": "phi"}
HIP_FIELDS = {"ρu": lambda m: m.momentum["ρu"], "ρv": lambda m: m.momentum["ρv"], "ρw": lambda m: m.momentum["ρw"],
              "ρθ": lambda m: m.potential_temperature_density, "ρqᵛ": lambda m: m.moisture_density,
              "u": lambda m: m.velocities["u"], "v": lambda m: m.velocities["v"], "w": lambda m: m.velocities["w"],
              "T": lambda m: m.temperature, "ϕ": lambda m: m.dynamics.pressure_anomaly}


def manifests():
    return sorted(glob.glob(os.path.join(HERE, "golden", "reference", "*", "manifest.json")))


def read_field(case_dir, entry):
    sx, sy, sz = entry["shape"]
    a = np.fromfile(os.path.join(case_dir, entry["file"]), dtype=np.float64)
    assert a.size == sx * sy * sz, (entry, a.size)
    return a.reshape(sz, sy, sx)


def interior(parent, size, halo):
    (Nx, Ny, Nz), (Hx, Hy, Hz) = size, halo
    nz = parent.shape[0] - 2 * Hz                       # Nz, or Nz + 1 for z-face fields
    return parent[Hz:Hz + nz, Hy:Hy + Ny, Hx:Hx + Nx]


def compare_run(manifest_path, get_field, step_fn, names, tol_step0=1e-12, tol_steps=1e-9):
    """get_field(name) -> (Sz, Sy, Sx) parent array of the model under test; step_fn() advances it by one manifest step."""
    case_dir = os.path.dirname(manifest_path)
    man = json.load(open(manifest_path, encoding="utf-8"))
    size, halo = tuple(man["size"]), tuple(man["halo"])
    worst = {}
    for step in range(man["steps"] + 1):
        if step:
            step_fn()
        for name in names:
            key = f"step{step}_{name}"
            if key not in man["fields"]:
                continue
            want = interior(read_field(case_dir, man["fields"][key]), size, halo)
            got = interior(np.asarray(get_field(name)), size, halo)
            scale = max(np.max(np.abs(want)), 1e-3)
            err = float(np.max(np.abs(got - want)) / scale)
            worst[key] = err
            assert err < (tol_step0 if step == 0 else tol_steps), (key, err)
    return worst


def oracle_model_for(oracle, man):
    size, halo = tuple(man["size"]), tuple(man["halo"])
    kind = man["kind"]
    if kind == "compressible_weno5":
        from oracle import oracle_compressible as oc
        g = oracle.Grid(size, x=EXT[0], y=EXT[1], z=EXT[2], halo=halo)
        m = oc.CompressibleOracleModel(g, time_discretization=oc.SplitExplicit(substeps=6), reference_potential_temperature=300.0)
        rho = m.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
        m.set(rho=rho, theta=bubble_theta(300.0, m.constants.g), u=3.0, v=-2.0, w=0.0, qv=0.0)
        return m, {"ρ": "rho_d", "ρu": "ru", "ρv": "rv", "ρw": "rw", "ρθ": "rtheta", "ρqᵛ": "rq", "T": "T", "p": "p"}
    g = oracle.Grid(size, x=EXT[0], y=EXT[1], z=EXT[2], halo=halo)
    m = oracle.OracleModel(g, potential_temperature=300.0, advection="Centered2" if kind == "anelastic_centered2" else "WENO5")
    m.set(theta=bubble_theta(300.0, m.constants.g), u=3.0, v=-2.0)
    return m, ORACLE_NAMES


def write_synthetic_case(oracle, out_dir, size=(16, 8, 8), halo=(3, 3, 3), dt=2.0, steps=2):
    """A case in dump_goldens.jl's format, produced by the oracle itself (column-major raw Float64 + manifest)."""
    os.makedirs(out_dir, exist_ok=True)
    man = {"kind": "anelastic_weno5", "size": list(size), "halo": list(halo), "dt": dt, "steps": steps, "fields": {}}
    m, names = oracle_model_for(oracle, man)
    for step in range(steps + 1):
        if step:
            m.time_step(dt)
        for ref_name in ("ρu", "ρw", "ρθ", "T", "ϕ"):
            a = getattr(m, names[ref_name])
            fn = f"step{step}_{ref_name}.bin"
            a.astype(np.float64).tofile(os.path.join(out_dir, fn))       # C (Sz, Sy, Sx) == Fortran (Sx, Sy, Sz)
            man["fields"][f"step{step}_{ref_name}"] = {"file": fn, "shape": [a.shape[2], a.shape[1], a.shape[0]]}
    path = os.path.join(out_dir, "manifest.json")
    json.dump(man, open(path, "w", encoding="utf-8"), ensure_ascii=False)
    return path


def test_manifest_reader_on_a_synthetic_case(oracle, tmp_path):
    """The comparison machinery on a case the oracle wrote in the reference dump's format: a second oracle run must reproduce it
    bit for bit, and a perturbed file must be caught."""
    path = write_synthetic_case(oracle, str(tmp_path / "synthetic"))
    man = json.load(open(path, encoding="utf-8"))
    m, names = oracle_model_for(oracle, man)
    worst = compare_run(path, lambda n: getattr(m, names[n]), lambda: m.time_step(man["dt"]), list(names), 1e-15, 1e-15)
    assert len(worst) == 15 and max(worst.values()) == 0.0
    f = os.path.join(os.path.dirname(path), "step1_ρθ.bin")
    a = np.fromfile(f)
    a *= 1.0 + 1e-6
    a.tofile(f)
    m2, _ = oracle_model_for(oracle, man)
    with pytest.raises(AssertionError):
        compare_run(path, lambda n: getattr(m2, names[n]), lambda: m2.time_step(man["dt"]), list(names))


@pytest.mark.parametrize("path", manifests() or [None])
def test_oracle_matches_reference_output(oracle, path):
    if path is None:
        pytest.skip("no tests/golden/reference/*/manifest.json: run tools/dump_goldens.jl where Julia + Breeze are installed")
    man = json.load(open(path, encoding="utf-8"))
    m, names = oracle_model_for(oracle, man)
    compare_run(path, lambda n: getattr(m, names[n]), lambda: m.time_step(man["dt"]), list(names))


@pytest.mark.gpu
@pytest.mark.parametrize("path", manifests() or [None])
def test_hip_path_matches_reference_output(bz, path):
    if path is None:
        pytest.skip("no tests/golden/reference/*/manifest.json: run tools/dump_goldens.jl where Julia + Breeze are installed")
    man = json.load(open(path, encoding="utf-8"))
    size, halo, kind = tuple(man["size"]), tuple(man["halo"]), man["kind"]
    grid = bz.RectilinearGrid(size, x=EXT[0], y=EXT[1], z=EXT[2], halo=halo)
    th = bubble_theta(300.0, 9.81)
    if kind == "compressible_weno5":
        dyn = bz.CompressibleDynamics(bz.SplitExplicitTimeDiscretization(substeps=6), reference_potential_temperature=300.0)
        m = bz.CompressibleAtmosphereModel(grid, dyn, advection=bz.WENO())
        rho = m.dynamics.reference_state.density[grid.Hz:grid.Hz + grid.Nz][:, None, None]
        m.set(ρ=rho, θ=th, u=3.0, v=-2.0, w=0.0, qᵗ=0.0)
        fields = dict(HIP_FIELDS, **{"ρ": lambda mm: mm.dynamics.dry_density, "p": lambda mm: mm.dynamics.pressure})
        fields.pop("ϕ")
    else:
        adv = bz.Centered(order=2) if kind == "anelastic_centered2" else bz.WENO()
        m = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(bz.ReferenceState(grid, potential_temperature=300)), advection=adv)
        m.set(θ=th, u=3.0, v=-2.0)
        fields = HIP_FIELDS

    def get(name):
        m.synchronize()
        return fields[name](m).cpu()

    compare_run(path, get, lambda: m.time_step(man["dt"]), list(fields))
