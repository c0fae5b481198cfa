"""Diagnostic: GB/s of the streaming BatchNorm kernels on the ResNet-50 bs256 activation shapes, beside a torch copy.
env RPE_EW_UNR / RPE_EW_GRID / RPE_EW_NT select kernel variants (read once per process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rgb_proprioceptive_pose_estimator_amd import ops
from rgb_proprioceptive_pose_estimator_amd._lib import lib

dt = torch.bfloat16
SHAPES = [(802816, 64), (802816, 256), (200704, 128), (200704, 512), (50176, 256), (50176, 1024), (12544, 512), (12544, 2048)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


tag = " ".join("%s=%s" % (k, os.environ[k]) for k in ("RPE_EW_UNR", "RPE_EW_GRID", "RPE_EW_NT") if k in os.environ) or "default"
print("variant:", tag)
tot = {"copy": 0.0, "apply": 0.0, "apply_res": 0.0, "dz": 0.0}
for m, c in SHAPES:
    y = torch.randn(m, c, device="cuda").to(dt)
    r = torch.randn(m, c, device="cuda").to(dt)
    o = torch.empty_like(y)
    scale = torch.rand(c, device="cuda") + 0.5
    shift = torch.randn(c, device="cuda")
    mean, invstd, gamma = torch.randn(c, device="cuda"), torch.rand(c, device="cuda") + 0.5, torch.rand(c, device="cuda") + 0.5
    nb = m * c * 2
    t_copy = timeit(lambda: o.copy_(y))
    st = ops._stream()
    code = ops.dtype_code(y)
    t_a = timeit(lambda: lib.rpe_bn_apply(code, ops._p(y), None, ops._p(o), ops._p(scale), ops._p(shift), m, c, 1, st))
    t_ar = timeit(lambda: lib.rpe_bn_apply(code, ops._p(y), ops._p(r), ops._p(o), ops._p(scale), ops._p(shift), m, c, 1, st))
    res = {"copy": (t_copy, 2), "apply": (t_a, 2), "apply_res": (t_ar, 3)}
    if hasattr(ops, "bn_backward_from_dz_raw"):
        pass
    line = "  M=%7d C=%4d" % (m, c)
    for k, (t, mult) in res.items():
        tot[k] += t
        line += "  %s %.3f ms %.2f TB/s" % (k, t, mult * nb / t / 1e9)
    print(line)
print("  totals ms:", {k: round(v, 3) for k, v in tot.items()})
