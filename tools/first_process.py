"""Diagnostic: how long does the first process on a freshly leased box run slow?  Windows of 10 train steps (256 images, bf16) for
~SECONDS seconds: per window the device time per step and the host's issue time per step, plus the GPU clock rocm-smi reports.

    python tools/first_process.py [seconds=45]
"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rgb_proprioceptive_pose_estimator_amd import models as M
from rgb_proprioceptive_pose_estimator_amd.optim import FusedAdam
from rgb_proprioceptive_pose_estimator_amd.util.data_utils import synthetic_batch
from rgb_proprioceptive_pose_estimator_amd.util.learn_utils import train_step

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 45.0
t_start = time.perf_counter()
torch.manual_seed(0)
model = M.NaiveObjectStateEstimator("cube", [1024, 256, 64], 50, 512, False, (9,), False, False, False, compute_dtype=torch.bfloat16)
model.cuda().train()
crit = {"obj_loss": M.PoseDistanceLoss("combined", 1.0, 0.5, 1e-4, "pose"), "val_loss": M.PoseDistanceLoss(mode="val")}
opt = FusedAdam(model.parameters(), lr=1e-3)
b = synthetic_batch((256,), 1234)
batch = (b["img"], None, b["x0bar"], b["x0"], None, b["obj"])


def sclk():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
        return " ".join(l.split(":")[-1].strip() for l in out.splitlines() if "sclk" in l or "mclk" in l)[:80]
    except Exception as e:   # noqa: BLE001
        return "n/a (%s)" % type(e).__name__


print("setup %.1f s" % (time.perf_counter() - t_start), flush=True)
w = 0
while time.perf_counter() - t_start < seconds:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        train_step(model, batch, crit, opt, True, "train", None)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    td = time.perf_counter() - t0
    w += 1
    extra = ("  clocks: " + sclk()) if w % 10 == 1 else ""
    print("t=%5.1f s  window %3d: %.2f ms/step (host issue %.2f ms/step)%s" % (time.perf_counter() - t_start, w, td * 100, th * 100, extra), flush=True)
