"""Import shim: `import rsp_chains_amd` loads the package that lives in the
directory `rsp-chains_amd/` (the hyphen keeps the reference's repo name but is
not a valid Python identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rsp-chains_amd")
_spec = importlib.util.spec_from_file_location(
    "rsp_chains_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rsp_chains_amd"] = _mod
_spec.loader.exec_module(_mod)
