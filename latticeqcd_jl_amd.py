"""Import alias: `import latticeqcd_jl_amd` loads the package directory `latticeqcd.jl_amd/`
(the dotted directory name cannot be imported directly)."""
import importlib.util
import os
import sys

_d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "latticeqcd.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "latticeqcd_jl_amd", os.path.join(_d, "__init__.py"), submodule_search_locations=[_d])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["latticeqcd_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
