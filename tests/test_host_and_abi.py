"""CPU-side checks of the product: host logic and that the C-ABI library loads and exports every
symbol include/breeze_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "breeze_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bz_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(bz):
    bz.build()
    import torch  # noqa: F401  (its bundled HIP runtime must be the one in the process)
    lib = ctypes.CDLL(bz.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/breeze_hip.h but not exported"
    assert sorted(bz.SYMBOLS) == declared, "ctypes binding out of sync with the header"
    bz.load()
    # the Centered(order = 2) build of the same sources exports the same ABI
    from breeze_jl_amd import _lib
    lib2 = ctypes.CDLL(_lib.CENTERED2_LIB_PATH)
    for name in declared:
        assert hasattr(lib2, name), f"{name} missing from libbreeze_hip_centered2.so"


def test_struct_layouts_match_header(bz):
    from breeze_jl_amd import _lib
    assert ctypes.sizeof(_lib.bz_state) == 12 * ctypes.sizeof(ctypes.c_void_p)
    assert ctypes.sizeof(_lib.bz_prognostic) == 5 * ctypes.sizeof(ctypes.c_void_p)
    assert ctypes.sizeof(_lib.bz_grid) == 6 * 4 + 3 * 4 + 4 + 2 * 8 + 8 + 8
    assert ctypes.sizeof(_lib.bz_constants) == 5 * 8
    assert ctypes.sizeof(_lib.bz_reference_state) == 6 * 8


def test_grid_layout(bz):
    g = bz.RectilinearGrid((8, 6, 4), x=(0, 8), y=(0, 12), z=(0, 2), halo=(3, 4, 4))
    assert (g.Sx, g.Sy) == (14, 14)
    assert g.parent_shape() == (12, 14, 14) and g.parent_shape(zface=True) == (13, 14, 14)
    assert g.Δx == 1.0 and g.Δy == 2.0 and g.Δz == 0.5
    assert np.allclose(g.zᶜ, [0.25, 0.75, 1.25, 1.75])
    with pytest.raises(ValueError):          # Oceananigans: a halo cannot be wider than the domain (Ny < Hy read unfilled rows)
        bz.RectilinearGrid((8, 2, 4), x=(0, 8), y=(0, 12), z=(0, 2))
    with pytest.raises(ValueError):
        bz.RectilinearGrid((8, 6), x=(0, 1), y=(0, 1), z=(0, 1))
    with pytest.raises(ValueError):
        bz.RectilinearGrid((8, 6, 4), x=(0, 1), y=(0, 1), z=(0, 1), topology=("Periodic", "Periodic", "Open"))


def test_reference_state_matches_oracle(bz, oracle):
    g = bz.RectilinearGrid((8, 8, 32), x=(0, 1), y=(0, 1), z=(0, 12e3))
    og = oracle.Grid((8, 8, 32), x=(0, 1), y=(0, 1), z=(0, 12e3))
    r = bz.ReferenceState(g, surface_pressure=101325, potential_temperature=300)
    o = oracle.ReferenceState(og, oracle.Constants(), 101325, 300)
    assert np.array_equal(r.density, o.density)
    assert np.array_equal(r.pressure, o.pressure)
    assert np.array_equal(r.temperature, o.temperature)


def test_model_requires_gpu_and_an_implemented_scheme(bz):
    import torch
    g = bz.RectilinearGrid((16, 16, 16), x=(0, 1), y=(0, 1), z=(0, 1))
    assert bz.WENO(order=9).order == 9 and bz.WENO(order=7).order == 7      # generic kernels (tests/test_weno_orders.py)
    with pytest.raises(NotImplementedError):
        bz.WENO(order=11)
    with pytest.raises(NotImplementedError):
        bz.WENO(order=9, bounds=(0, 1))
    with pytest.raises(NotImplementedError):
        bz.Centered(order=4)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            bz.AtmosphereModel(g, advection=bz.WENO())   # no CPU fallback
        with pytest.raises(RuntimeError):
            bz.AtmosphereModel(g)                        # reference default Centered(order = 2): same rule


def test_product_never_imports_oracle():
    """The product path must not import, link, load or shell out to anything under oracle/ (docstrings may mention
    the oracle in prose)."""
    pkg = os.path.join(ROOT, "breeze.jl_amd")
    bad = re.compile(r"^\s*(import|from)\s+oracle\b|libbreeze_oracle|breeze_oracle\.|oracle/|[\"']oracle[\"']", re.M)
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                m = bad.search(text)
                assert m is None, (dirpath, f, m.group(0))


def test_float32_abi_header_is_committed_and_current():
    """include/breeze_hip_f32.h (the Float32 boundary) is the generator's output for the current Float64 header, declares
    every entry point, and carries no double"""
    import re
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_f32_sources as gen
    with open(os.path.join(ROOT, "include", "breeze_hip_f32.h"), encoding="utf-8") as f:
        text = f.read()
    assert text == gen.f32_header_text()
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    assert not re.search(r"\bdouble\b", code)
    from breeze_jl_amd import _lib
    for name in _lib.SYMBOLS:
        assert re.search(r"\b%s\s*\(" % name, code), name
