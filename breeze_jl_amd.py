"""Import shim: the product package lives in the directory `breeze.jl_amd/` (not a valid dotted
module name), so this module loads it by path and re-exports it as `breeze_jl_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "breeze.jl_amd")
_spec = importlib.util.spec_from_file_location("breeze_jl_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["breeze_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
