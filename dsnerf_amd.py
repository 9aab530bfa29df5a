"""Import shim: `import dsnerf_amd` loads the package that lives in ./dual-space-nerf_amd/
(a hyphenated directory name cannot be imported directly)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dual-space-nerf_amd")
_spec = importlib.util.spec_from_file_location("dsnerf_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dsnerf_amd"] = _mod
_spec.loader.exec_module(_mod)
