"""Import shim: exposes the product package directory ``ace-step-1.5-for-windows_amd/``
(not a valid Python identifier) under the importable name ``ace355``."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "ace-step-1.5-for-windows_amd")
_spec = _ilu.spec_from_file_location("ace355", _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["ace355"] = _mod
_spec.loader.exec_module(_mod)
