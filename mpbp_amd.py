"""Import shim: the product package lives in the directory ``matrixproductbp.jl_amd/`` whose name
is not a valid Python identifier; ``import mpbp_amd`` loads it under this alias."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "matrixproductbp.jl_amd")
_spec = importlib.util.spec_from_file_location("mpbp_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mpbp_amd"] = _mod
_spec.loader.exec_module(_mod)
