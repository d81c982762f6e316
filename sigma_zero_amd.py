"""Import shim: the product package lives in the directory `sigma-zero_amd/` (a name Python cannot
import directly); `import sigma_zero_amd` loads that directory as the package `sigma_zero_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sigma-zero_amd")
_spec = importlib.util.spec_from_file_location("sigma_zero_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sigma_zero_amd"] = _mod
_spec.loader.exec_module(_mod)
