"""Diagnostic: run N eval-mode rollout frames (batch 1) of one model, to be traced with `rocprofv3 --kernel-trace`.
usage: python tools/rollout_frames.py [kind] [dtype] [n_frames]"""
import contextlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rgb_proprioceptive_pose_estimator_amd import models as M

kind = sys.argv[1] if len(sys.argv) > 1 else "no"
dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
make = {
    "no": lambda: M.NaiveObjectStateEstimator("cube", [1024, 256, 64], 50, 512, False, (9,), False, False, False, compute_dtype=dt),
    "tdo": lambda: M.TemporallyDependentObjectStateEstimator("hammer", 512, 50, 512, 10, 0.1, False, (9,), False, False, False, compute_dtype=dt),
}[kind]
torch.manual_seed(0)
with contextlib.redirect_stdout(sys.stderr):
    model = make()
model.cuda().eval()
model.rollout = True
model.reset_initial_state(1)
seq = getattr(model, "requires_sequence", False)
img = torch.randn((1, 1, 3, 224, 224) if seq else (1, 3, 224, 224), device="cuda")
x0 = torch.randn((1, 1, 7) if seq else (1, 7), device="cuda")
with torch.no_grad():
    for _ in range(n):
        model(img, None, x0)
torch.cuda.synchronize()
print("done", n)
