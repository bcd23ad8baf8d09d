"""Import alias: the package directory is named ``cr-nerf-pytorch_amd`` (not a valid Python
identifier), so ``import crnerf_amd`` loads it from there and replaces this stub in sys.modules."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "cr-nerf-pytorch_amd")
_spec = _ilu.spec_from_file_location(__name__, _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
