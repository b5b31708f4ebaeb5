"""Import alias for the package directory `codesign-kernels_amd/`.

The project layout names the package directory with a hyphen, which Python
cannot import by name; this module loads it under the importable name
`codesign_kernels_amd` (submodules resolve inside that directory).
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "codesign-kernels_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
