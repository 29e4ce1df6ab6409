"""Import shim: the package directory is `matrix-eyes_amd/` (not a valid Python identifier), so
`import matrix_eyes_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "matrix-eyes_amd")
_spec = importlib.util.spec_from_file_location(
    "matrix_eyes_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["matrix_eyes_amd"] = _mod
_spec.loader.exec_module(_mod)
