"""Import helper: exposes the package in `gr-fosphor_amd/` (not a valid Python identifier)
as the module `gr_fosphor_amd`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_ROOT, "gr-fosphor_amd")


def _load():
    name = "gr_fosphor_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(
        name, os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


gr_fosphor_amd = _load()
