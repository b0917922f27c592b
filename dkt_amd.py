"""Import shim: the product package lives in `deep-kernel-transfer_amd/` (a directory name Python
cannot import directly); `import dkt_amd` loads it under the module name `dkt_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "deep-kernel-transfer_amd")
_spec = importlib.util.spec_from_file_location(
    "dkt_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dkt_amd"] = _mod
_spec.loader.exec_module(_mod)
