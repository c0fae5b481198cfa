"""MI355X-native (gfx950) implementation of the RGB + proprioception pose-regression
train step of cremebrule/rgb-proprioceptive-pose-estimator.

Layout
  csrc/        hand-written HIP kernels + the C ABI (include/rpe_hip.h) -> librpe_hip.so
  _lib.py      ctypes binding (fails loudly if the library is missing)
  ops.py       tensor-level wrappers over the C ABI
  engine.py    ResNet-50 trunk engine binding (one native launch plan per batch shape)
  models/      drop-in model classes + PoseDistanceLoss (reference: models/*.py)
  util/        import_resnet, train(), synthetic dataset (reference: util/*.py)
  scripts/     train_model.py / rollout.py entry points (reference: scripts/*.py)
"""
from . import _lib  # noqa: F401  (raises ImportError when librpe_hip.so is absent)

__all__ = ["_lib"]
