"""Diagnostic: train-step time of every model family at 256 images per step (BASELINE.json configs C2..C5: S=4, N=64 for the
sequence models), bf16 trunk.  usage: python tools/time_models.py [steps]"""
import contextlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rgb_proprioceptive_pose_estimator_amd import models as M
from rgb_proprioceptive_pose_estimator_amd.optim import FusedAdam
from rgb_proprioceptive_pose_estimator_amd.util.data_utils import synthetic_batch
from rgb_proprioceptive_pose_estimator_amd.util.learn_utils import train_step

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 15
dt = torch.bfloat16
MODELS = {
    "no   (C2)": (lambda: M.NaiveObjectStateEstimator("cube", [1024, 256, 64], 50, 512, False, (9,), False, False, False, compute_dtype=dt), (256,), False),
    "n": (lambda: M.NaiveEndEffectorStateEstimator([1024, 256, 64], [1024, 256, 64], 50, 512, False, compute_dtype=dt), (256,), False),
    "td   (C3)": (lambda: M.TemporallyDependentStateEstimator(512, 512, 50, 512, 4, 0.1, False, (9,), False, False, compute_dtype=dt), (4, 64), False),
    "tdo  (C4, depth)": (lambda: M.TemporallyDependentObjectStateEstimator("hammer", 512, 50, 512, 4, 0.1, False, (9,), True, False, False, compute_dtype=dt), (4, 64), True),
    "tdo_v2 (C5)": (lambda: M.TemporallyDependentObjectStateEstimatorV2("robot1_eef", 512, 64, 50, 512, 4, 0.1, False, (9,), False, False, compute_dtype=dt), (4, 64), False),
}
crit = {"x0_loss": M.PoseDistanceLoss("combined", 1.0, 0.5, 1e-4, "pose"), "x1_loss": M.PoseDistanceLoss("combined", 1.0, 0.5, 1e-4, "pose"),
        "obj_loss": M.PoseDistanceLoss("combined", 1.0, 0.5, 1e-4, "pose"), "val_loss": M.PoseDistanceLoss(mode="val")}
for name, (make, lead, depth) in MODELS.items():
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):
        model = make()
    model.cuda().train()
    opt = FusedAdam(model.parameters(), lr=1e-3)
    b = synthetic_batch(lead, 1234, with_depth=depth)
    batch = (b["img"], b["depth"], b["x0bar"], b["x0"], b["x1"], b["obj"])
    for _ in range(10):
        train_step(model, batch, crit, opt, hasattr(model, "object_name"), "train", None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        train_step(model, batch, crit, opt, hasattr(model, "object_name"), "train", None)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / STEPS
    print("%-18s %.2f ms/step  %.0f img/s" % (name, t * 1e3, 256 / t))
    del model, opt
    torch.cuda.empty_cache()
