"""Diagnostic: the 3x3 / stride-1 convs of ResNet-50 (forward, data gradient with the fused BN-backward epilogue) at B images, one line per
shape -- run once as is and once with RPE_NO_HALO=1 to compare the halo form with the per-tap gathered form.

    python tools/bench_conv3.py [B] [bf16|f16]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rgb_proprioceptive_pose_estimator_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dtype = {"bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


print("env:", {k: v for k, v in os.environ.items() if k.startswith("RPE_")})
tot = {"fwd": 0.0, "dgrad_bn": 0.0}
for c, h, cnt in [(64, 56, 3), (128, 28, 3), (256, 14, 5), (512, 7, 2)]:
    x = torch.randn(B, h, h, c, device="cuda").to(dtype)
    w = (torch.randn(c, 3, 3, c, device="cuda") / (c * 9) ** 0.5).to(dtype)
    wd = w.permute(3, 1, 2, 0).contiguous()
    dy = torch.randn(B, h, h, c, device="cuda").to(dtype)
    yprev = torch.randn(B, h, h, c, device="cuda").to(dtype)
    mean = torch.zeros(c, device="cuda"); invstd = torch.ones(c, device="cuda"); sc = torch.ones(c, device="cuda"); sh = torch.zeros(c, device="cuda")
    fl = 2.0 * B * h * h * c * c * 9
    tf = timeit(lambda: ops.conv2d_fwd(x, w, 1, 1, want_stats=True))
    tb = timeit(lambda: ops.conv2d_dgrad_bn(dy, wd, (B, h, h, c), 1, 1, yprev, mean, invstd, scale=sc, shift=sh))
    tot["fwd"] += tf * cnt; tot["dgrad_bn"] += tb * cnt
    print("C%4d H%3d x%d  fwd %.3f ms %5.0f TF/s | dgrad+bn %.3f ms %5.0f TF/s" % (c, h, cnt, tf, fl / tf / 1e9, tb, fl / tb / 1e9), flush=True)
print("weighted per step (ms):", {k: round(v, 3) for k, v in tot.items()})
