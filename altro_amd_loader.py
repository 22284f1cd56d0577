"""Import helper: the product package lives in the directory `altro-mpc-icra2021_amd/`
(the name the project brief fixes), which is not a valid Python identifier.  This module
registers it under the importable name `altro_mpc_icra2021_amd`.

    import altro_amd_loader            # noqa: F401
    import altro_mpc_icra2021_amd as altro
"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "altro-mpc-icra2021_amd")
PKG_NAME = "altro_mpc_icra2021_amd"


def load():
    if PKG_NAME in sys.modules:
        return sys.modules[PKG_NAME]
    spec = importlib.util.spec_from_file_location(
        PKG_NAME, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[PKG_NAME] = mod
    spec.loader.exec_module(mod)
    return mod


load()
