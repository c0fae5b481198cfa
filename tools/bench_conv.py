"""Diagnostic: per-shape time / TF/s / algorithmic GB/s of the implicit-GEMM kernels on the conv shapes of ResNet-50 (B images).

    python tools/bench_conv.py [B] [bf16|f16|f32] [fwd,dgrad,wgrad,bn]      (default: 256 bf16, everything)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rgb_proprioceptive_pose_estimator_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
which = set((sys.argv[3] if len(sys.argv) > 3 else "fwd,dgrad,wgrad,bn").split(","))
ES = 4 if dtype == torch.float32 else 2
SHAPES = [  # Cin, Cout, k, s, Hin, count
    (64, 64, 1, 1, 56, 1), (64, 64, 3, 1, 56, 3), (64, 256, 1, 1, 56, 4), (256, 64, 1, 1, 56, 2), (256, 128, 1, 1, 56, 1),
    (128, 128, 3, 2, 56, 1), (128, 512, 1, 1, 28, 4), (256, 512, 1, 2, 56, 1), (512, 128, 1, 1, 28, 3), (128, 128, 3, 1, 28, 3),
    (512, 256, 1, 1, 28, 1), (256, 256, 3, 2, 28, 1), (256, 1024, 1, 1, 14, 6), (512, 1024, 1, 2, 28, 1), (1024, 256, 1, 1, 14, 5),
    (256, 256, 3, 1, 14, 5), (1024, 512, 1, 1, 14, 1), (512, 512, 3, 2, 14, 1), (512, 2048, 1, 1, 7, 3), (1024, 2048, 1, 2, 14, 1),
    (2048, 512, 1, 1, 7, 2), (512, 512, 3, 1, 7, 2),
]


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


tot = {}
print("env:", {k: v for k, v in os.environ.items() if k.startswith("RPE_")})
print("%-26s %7s %7s |" % ("shape (Cin->Cout k s H)", "GFLOP", "MB"), " | ".join("%-22s" % w for w in sorted(which)))
for ci, co, k, s, h, cnt in SHAPES:
    p = k // 2
    ho = (h + 2 * p - k) // s + 1
    x = torch.randn(B, h, h, ci, device="cuda").to(dtype)
    w = (torch.randn(co, k, k, ci, device="cuda") / (ci * k * k) ** 0.5).to(dtype)
    wd = w.permute(3, 1, 2, 0).contiguous()
    dy = torch.randn(B, ho, ho, co, device="cuda").to(dtype)
    fl = 2.0 * B * ho * ho * co * ci * k * k
    mb = (x.numel() + dy.numel()) * ES / 1e6   # algorithmic bytes: input once + output once
    cols = []
    res = {}
    if "fwd" in which: res["fwd"] = timeit(lambda: ops.conv2d_fwd(x, w, s, p, want_stats=True))
    if "dgrad" in which: res["dgrad"] = timeit(lambda: ops.conv2d_dgrad(dy, wd, (B, h, h, ci), s, p))
    if "wgrad" in which: res["wgrad"] = timeit(lambda: ops.conv2d_wgrad(x, dy, k, s, p))
    if "wgrad_atomic" in which: res["wgrad_atomic"] = timeit(lambda: ops.conv2d_wgrad(x, dy, k, s, p, deterministic=False))
    if "bn" in which:
        # data gradient with the producing layer's BN-backward reduction fused (mask from y / from a_out + addend)
        yprev = torch.randn(B, h, h, ci, device="cuda").to(dtype)
        aprev = torch.relu(yprev)
        mean = torch.zeros(ci, device="cuda"); invstd = torch.ones(ci, device="cuda"); sc = torch.ones(ci, device="cuda"); sh = torch.zeros(ci, device="cuda")
        res["bn(y)"] = timeit(lambda: ops.conv2d_dgrad_bn(dy, wd, (B, h, h, ci), s, p, yprev, mean, invstd, scale=sc, shift=sh))
        res["bn(a)"] = timeit(lambda: ops.conv2d_dgrad_bn(dy, wd, (B, h, h, ci), s, p, yprev, mean, invstd, a_out=aprev, addend=aprev))
    for key, t in res.items():
        tot[key] = tot.get(key, 0.0) + t * cnt
    print("%4d->%4d k%d s%d H%-3d x%d %7.1f %7.1f |" % (ci, co, k, s, h, cnt, fl / 1e9, mb),
          " | ".join("%-5s %6.3f ms %5.0f TF %4.2f TB/s" % (key, t, fl / t / 1e9, mb / t / 1e3) for key, t in res.items()))
print("weighted totals per step (ms):", {k: round(v, 2) for k, v in tot.items()})
