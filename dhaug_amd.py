"""Import shim: the product package lives in the directory
`dh-aug-dh-forward-kinematics-model-driven-augmentation-for-3d-human-pose-estimation_amd/` (not a valid Python
identifier), so `import dhaug_amd` loads that directory as the package `dhaug_amd`."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "dh-aug-dh-forward-kinematics-model-driven-augmentation-for-3d-human-pose-estimation_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_DIR, "__init__.py"),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
