"""Import shim: registers the package directory ``generalised-gaussian-processes_amd/`` (hyphens are
not importable) under the module name ``ggp_amd``.  ``import ggp_amd`` then behaves like a normal
package import, sub-modules included (``import ggp_amd.core``)."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "generalised-gaussian-processes_amd")
_spec = importlib.util.spec_from_file_location(
    "ggp_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ggp_amd"] = _mod
_spec.loader.exec_module(_mod)
