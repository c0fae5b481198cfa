"""Diagnostic: run ONE conv shape (fwd, dgrad, wgrad) a few times -- target for rocprofv3 --pmc."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rgb_proprioceptive_pose_estimator_amd import ops
ci, co, k, s, h = (int(x) for x in sys.argv[1:6])
B = int(sys.argv[6]) if len(sys.argv) > 6 else 256
which = sys.argv[7] if len(sys.argv) > 7 else "fwd,dgrad,wgrad"
p = k // 2
ho = (h + 2 * p - k) // s + 1
dt = torch.bfloat16
x = torch.randn(B, h, h, ci, device="cuda").to(dt)
w = (torch.randn(co, k, k, ci, device="cuda") / (ci * k * k) ** 0.5).to(dt)
wd = w.permute(3, 1, 2, 0).contiguous()
dy = torch.randn(B, ho, ho, co, device="cuda").to(dt)
for _ in range(3):
    if "fwd" in which: ops.conv2d_fwd(x, w, s, p, want_stats=True)
    if "dgrad" in which: ops.conv2d_dgrad(dy, wd, (B, h, h, ci), s, p)
    if "wgrad" in which: ops.conv2d_wgrad(x, dy, k, s, p)
torch.cuda.synchronize()
