"""Imports the package directory `cuda-efficient-features_amd/` (its name contains '-') as module `cef_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "cuda-efficient-features_amd")


def load():
    if "cef_amd" in sys.modules:
        return sys.modules["cef_amd"]
    spec = importlib.util.spec_from_file_location("cef_amd", os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["cef_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def load_submodule(name):
    """Imports cuda-efficient-features_amd/<name>.py as cef_amd.<name>."""
    load()
    import importlib
    return importlib.import_module("cef_amd." + name)
