"""Diagnostic: train step rate with the batch crossing PCIe every step -- 256 raw uint8 frames of 256x256x3 from
pinned host memory through util.data_utils.FramePrefetcher (double buffered, side stream) -- against the resident-data rate.
usage: python tools/h2d_overlap.py [steps]"""
import contextlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rgb_proprioceptive_pose_estimator_amd import models as M
from rgb_proprioceptive_pose_estimator_amd.optim import FusedAdam
from rgb_proprioceptive_pose_estimator_amd.util.data_utils import FramePrefetcher, random_poses
from rgb_proprioceptive_pose_estimator_amd.util.learn_utils import train_step

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = 256
torch.manual_seed(0)
with contextlib.redirect_stdout(sys.stderr):
    model = M.NaiveObjectStateEstimator("cube", [1024, 256, 64], 50, 512, False, (9,), False, False, False, compute_dtype=torch.bfloat16)
model.cuda().train()
crit = {"obj_loss": M.PoseDistanceLoss(distance_metric="combined", scale_factor=1.0, alpha=0.5, mode="pose"), "val_loss": M.PoseDistanceLoss(mode="val")}
opt = FusedAdam(model.parameters(), lr=1e-3)
g = torch.Generator().manual_seed(1)
frames = torch.randint(0, 256, (B, 256, 256, 3), generator=g, dtype=torch.uint8)
frames_pinned = frames.pin_memory()   # what DataLoader(pin_memory=True) hands over
gd = torch.Generator(device="cpu").manual_seed(2)
x0 = random_poses((B,), gd, "cpu")
obj = random_poses((B,), gd, "cpu")


def host_batches(n):
    for _ in range(n):
        yield frames, x0, obj


def run(source, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 0
    for f, xb, tgt in source:
        train_step(model, (f, None, xb, xb, None, tgt), crit, opt, True, "train", None)
        k += 1
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / max(k, 1)


dev = (frames.cuda(), x0.cuda(), obj.cuda())
run((dev for _ in range(5)), 5)
t_res = run((dev for _ in range(STEPS)), STEPS)
t_pre = run(FramePrefetcher(((frames_pinned, x0.pin_memory(), obj.pin_memory()) for _ in range(STEPS)), "cuda"), STEPS)
t_pre_pageable = run(FramePrefetcher(host_batches(STEPS), "cuda"), STEPS)


def sync_copy(n):
    for f, a, b in host_batches(n):
        yield f.cuda(), a.cuda(), b.cuda()


t_sync = run(sync_copy(STEPS), STEPS)
print("resident uint8 frames      : %.2f ms/step  %.0f img/s" % (t_res * 1e3, B / t_res))
print("FramePrefetcher, pinned src : %.2f ms/step  %.0f img/s" % (t_pre * 1e3, B / t_pre))
print("FramePrefetcher, pageable   : %.2f ms/step  %.0f img/s (host memcpy into the staging buffer on the training thread)" % (t_pre_pageable * 1e3, B / t_pre_pageable))
print("synchronous pageable .cuda(): %.2f ms/step  %.0f img/s" % (t_sync * 1e3, B / t_sync))
