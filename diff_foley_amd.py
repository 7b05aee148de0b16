"""Import shim: the package directory is ``diff-foley_amd/`` (not a valid Python identifier),
so ``import diff_foley_amd`` loads that directory as the package ``diff_foley_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "diff-foley_amd")
_spec = importlib.util.spec_from_file_location(
    "diff_foley_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["diff_foley_amd"] = _mod
_spec.loader.exec_module(_mod)
