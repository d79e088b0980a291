"""Import shim: exposes the package in `domain-transfer-gan_amd/` (hyphenated, hence not importable
by name) as `dtgan_amd`."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "domain-transfer-gan_amd")
_spec = importlib.util.spec_from_file_location("dtgan_amd", os.path.join(_DIR, "__init__.py"),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dtgan_amd"] = _mod
_spec.loader.exec_module(_mod)
