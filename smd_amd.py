"""Import shim: the package directory is ``symbolic-music-diffusion_amd/`` (not a valid Python
identifier), so ``import smd_amd`` loads it from there under this name."""
import importlib.util as _u
import os as _os
import sys as _sys

_d = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "symbolic-music-diffusion_amd")
_spec = _u.spec_from_file_location("smd_amd", _os.path.join(_d, "__init__.py"), submodule_search_locations=[_d])
_m = _u.module_from_spec(_spec)
_sys.modules["smd_amd"] = _m
_spec.loader.exec_module(_m)
