"""Import shim: `import kdcc_amd` loads the package that lives in the directory
`knowledge-distillation-by-replacing-cheap-conv_amd/` (a name Python cannot import directly)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "knowledge-distillation-by-replacing-cheap-conv_amd")
_spec = importlib.util.spec_from_file_location("kdcc_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["kdcc_amd"] = _mod
_spec.loader.exec_module(_mod)
