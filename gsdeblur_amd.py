"""Import shim: ``import gsdeblur_amd`` loads the package directory ``3dgs-deblur_amd/``
(whose name is not a valid Python identifier) and registers it under this module name."""
import importlib.util
import sys
from pathlib import Path

_pkg_dir = Path(__file__).resolve().parent / "3dgs-deblur_amd"
_spec = importlib.util.spec_from_file_location(
    "gsdeblur_amd", _pkg_dir / "__init__.py", submodule_search_locations=[str(_pkg_dir)]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gsdeblur_amd"] = _mod
_spec.loader.exec_module(_mod)
