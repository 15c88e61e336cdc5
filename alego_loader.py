"""Imports the hyphen-named package directory `a-lego-loam_amd/` under the module name `alego_amd`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))


def load_package():
    if "alego_amd" in sys.modules:
        return sys.modules["alego_amd"]
    pkg_dir = os.path.join(_ROOT, "a-lego-loam_amd")
    spec = importlib.util.spec_from_file_location(
        "alego_amd", os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["alego_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
