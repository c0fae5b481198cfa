"""Diagnostic: the deterministic 3x3 / stride-1 weight gradients of the ResNet trunks at B images, halo form (csrc/wgrad_halo.hip) against the
gathered form (tn_kernel, conv mode), alone on the device -- one line per shape.

    python tools/bench_wgrad3.py [B] [bf16|f16]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rgb_proprioceptive_pose_estimator_amd import ops
from rgb_proprioceptive_pose_estimator_amd._lib import lib

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
tot = [0.0, 0.0]
for ci, co, h, cnt in [(64, 64, 56, 3), (128, 128, 28, 3), (256, 256, 14, 5), (512, 512, 7, 2)]:
    x = torch.randn(B, h, h, ci, device="cuda").to(dtype)
    dy = torch.randn(B, h, h, co, device="cuda").to(dtype)
    fl = 2.0 * B * h * h * ci * co * 9
    res = []
    for minw in (2, 1000):
        prev = lib.rpe_conv2d_wgrad_halo_min_width(minw)
        t = timeit(lambda: ops.conv2d_wgrad(x, dy, 3, 1, 1))
        res.append((t, ops.last_kernel_name()))
        lib.rpe_conv2d_wgrad_halo_min_width(prev)
    tot[0] += res[0][0] * cnt; tot[1] += res[1][0] * cnt
    print("Ci%4d Co%4d H%3d x%d  halo %.3f ms %5.0f TF/s | gathered %.3f ms %5.0f TF/s   (%s | %s)" %
          (ci, co, h, cnt, res[0][0], fl / res[0][0] / 1e9, res[1][0], fl / res[1][0] / 1e9, res[0][1], res[1][1]), flush=True)
print("weighted per step (ms): halo %.3f gathered %.3f" % tuple(tot))
