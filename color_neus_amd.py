"""Import shim: the package directory is named ``color-neus_amd`` (not a valid Python identifier),
so ``import color_neus_amd`` resolves to this file, which loads that directory as the package."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "color-neus_amd")
_spec = importlib.util.spec_from_file_location(
    "color_neus_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["color_neus_amd"] = _mod
_spec.loader.exec_module(_mod)
