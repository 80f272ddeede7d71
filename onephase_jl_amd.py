"""Import shim: loads the package directory `onephase.jl_amd/` (its name has a dot, so a plain
import statement cannot reach it) under the module name `onephase_jl_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "onephase.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "onephase_jl_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["onephase_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
