"""Diagnostic: a few train steps of one model family at 256 images (for rocprofv3 --kernel-trace).  usage: python tools/one_model_steps.py td|tdo|tdo_v2|no [steps]"""
import contextlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rgb_proprioceptive_pose_estimator_amd import models as M
from rgb_proprioceptive_pose_estimator_amd.optim import FusedAdam
from rgb_proprioceptive_pose_estimator_amd.util.data_utils import synthetic_batch
from rgb_proprioceptive_pose_estimator_amd.util.learn_utils import train_step

kind = sys.argv[1] if len(sys.argv) > 1 else "td"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dt = torch.bfloat16
make = {
    "no": (lambda: M.NaiveObjectStateEstimator("cube", [1024, 256, 64], 50, 512, False, (9,), False, False, False, compute_dtype=dt), (256,), False),
    "td": (lambda: M.TemporallyDependentStateEstimator(512, 512, 50, 512, 4, 0.1, False, (9,), False, False, compute_dtype=dt), (4, 64), False),
    "tdo": (lambda: M.TemporallyDependentObjectStateEstimator("hammer", 512, 50, 512, 4, 0.1, False, (9,), True, False, False, compute_dtype=dt), (4, 64), True),
    "tdo_v2": (lambda: M.TemporallyDependentObjectStateEstimatorV2("robot1_eef", 512, 64, 50, 512, 4, 0.1, False, (9,), False, False, compute_dtype=dt), (4, 64), False),
}[kind]
crit = {k: M.PoseDistanceLoss("combined", 1.0, 0.5, 1e-4, "pose") for k in ("x0_loss", "x1_loss", "obj_loss")}
crit["val_loss"] = M.PoseDistanceLoss(mode="val")
torch.manual_seed(0)
with contextlib.redirect_stdout(sys.stderr):
    model = make[0]()
model.cuda().train()
opt = FusedAdam(model.parameters(), lr=1e-3)
b = synthetic_batch(make[1], 1234, with_depth=make[2])
batch = (b["img"], b["depth"], b["x0bar"], b["x0"], b["x1"], b["obj"])
model.reset_initial_state(make[1][-1])
for _ in range(steps):
    train_step(model, batch, crit, opt, hasattr(model, "object_name"), "train", None)
torch.cuda.synchronize()
print("done")
