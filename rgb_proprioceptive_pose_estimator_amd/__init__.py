"""Import alias: ``rgb-proprioceptive-pose-estimator_amd/`` (the package directory the
build contract names) is not a valid Python identifier, so this module loads that
directory under the importable name ``rgb_proprioceptive_pose_estimator_amd``."""
import importlib.util
import os
import sys

_real = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rgb-proprioceptive-pose-estimator_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_real, "__init__.py"), submodule_search_locations=[_real])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
