"""Diagnostic: is the first process on a freshly leased box slow?  (No: profiles/r03_bench_first_process.txt.)  Windows of 10 train steps (256 images, bf16) for
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


def hbm_probe():
    """TB/s of a 411 MB device-to-device copy (read + write), best of 3"""
    x = torch.empty(411 * 1000 * 1000 // 2, dtype=torch.bfloat16, device="cuda")
    y = torch.empty_like(x)
    best = 0.0
    for _ in range(3):
        a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); y.copy_(x); c.record(); torch.cuda.synchronize()
        best = max(best, 2 * x.numel() * 2 / (a.elapsed_time(c) * 1e-3) / 1e12)
    return best


def mfma_probe():
    """TF/s of a bf16 8192^3 matmul through torch (hipBLASLt), best of 3"""
    a = torch.randn(8192, 8192, device="cuda").bfloat16()
    b = torch.randn(8192, 8192, device="cuda").bfloat16()
    best = 0.0
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); (a @ b); e1.record(); torch.cuda.synchronize()
        best = max(best, 2 * 8192 ** 3 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    return best


print("probes at start: copy %.2f TB/s, matmul %.0f TF/s, clocks %s" % (hbm_probe(), mfma_probe(), sclk()), flush=True)
w, acc, t_rep = 0, [], time.perf_counter()
while time.perf_counter() - t_start < seconds:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        train_step(model, batch, crit, opt, True, "train", None)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    td = time.perf_counter() - t0
    w += 1
    acc.append((td * 100, th * 100))
    if time.perf_counter() - t_rep >= 5.0:   # one line per ~5 s
        ds = sorted(a for a, _ in acc)
        print("t=%5.1f s  %3d windows: device ms/step min %.2f median %.2f max %.2f, host issue median %.2f" %
              (time.perf_counter() - t_start, len(acc), ds[0], ds[len(ds) // 2], ds[-1], sorted(b for _, b in acc)[len(acc) // 2]), flush=True)
        acc, t_rep = [], time.perf_counter()
print("probes at end: copy %.2f TB/s, matmul %.0f TF/s, clocks %s" % (hbm_probe(), mfma_probe(), sclk()), flush=True)
