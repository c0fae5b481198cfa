"""Probe: does a conv whose N tile covers only part of an output row write slower than one whose tile covers whole rows?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rgb_proprioceptive_pose_estimator_amd import ops

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

B, h = 256, 56
for ci, co in ((64, 64), (64, 128), (64, 256), (64, 512), (128, 128), (128, 256), (128, 512), (256, 128), (256, 256)):
    x = torch.randn(B, h, h, ci, device="cuda").bfloat16()
    w = (torch.randn(co, 1, 1, ci, device="cuda") / ci ** 0.5).bfloat16()
    t = timeit(lambda: ops.conv2d_fwd(x, w, 1, 0, want_stats=True))
    wd = w.permute(3, 1, 2, 0).contiguous()
    dy = torch.randn(B, h, h, co, device="cuda").bfloat16()
    yprev = torch.randn(B, h, h, ci, device="cuda").bfloat16()
    z = torch.zeros(ci, device="cuda"); o = torch.ones(ci, device="cuda")
    td = timeit(lambda: ops.conv2d_dgrad_bn(dy, wd, (B, h, h, ci), 1, 0, yprev, z, o, scale=o, shift=z))
    mb = (x.numel() + B * h * h * co) * 2 / 1e6
    print("%4d -> %4d  H%d: fwd %.3f ms  %.2f TB/s  (out row %d B) | its fused data gradient (writes %d-B rows) %.3f ms" % (ci, co, h, t, mb / t / 1e3, co * 2, ci * 2, td))
