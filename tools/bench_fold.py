"""Diagnostic: the main-stream cost of a conv3 backward, old form (finalize + streaming dz->dy + data gradient of dy) against the
folded form (coefficients + fold + K-concatenated data gradient of [dz | a_in]), isolated, per ResNet-50 stage at B images."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rgb_proprioceptive_pose_estimator_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dtype = torch.bfloat16


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for p, h, cnt in ((64, 56, 3), (128, 28, 4), (256, 14, 6), (512, 7, 2)):
    co, ci = 4 * p, p
    dz = torch.randn(B, h, h, co, device="cuda").to(dtype)
    y = torch.randn(B, h, h, co, device="cuda").to(dtype)
    a_in = torch.relu(torch.randn(B, h, h, ci, device="cuda")).to(dtype)
    y2 = torch.randn(B, h, h, ci, device="cuda").to(dtype)
    wf = (torch.randn(co, ci, device="cuda") / ci ** 0.5).to(dtype)
    wd = wf.t().contiguous()
    wd4 = wd.reshape(ci, 1, 1, co)
    mean = torch.zeros(co, device="cuda"); invstd = torch.ones(co, device="cuda"); gamma = torch.ones(co, device="cuda")
    m2 = torch.zeros(ci, device="cuda"); r2 = torch.ones(ci, device="cuda"); sc2 = torch.ones(ci, device="cuda"); sh2 = torch.zeros(ci, device="cuda")
    rows = B * h * h
    st = torch.randn((rows + 127) // 128, 2, co, device="cuda")
    dg, db, c1c2 = ops.bn_backward_coeffs(st, rows)
    wk, bias = ops.bn_bwd_fold_conv1x1(wf, wd, gamma, invstd, mean, c1c2)
    t_from = timeit(lambda: ops.bn_backward_from_dz(dz, y, mean, invstd, gamma, st))
    t_apply = timeit(lambda: ops.bn_backward_apply_dz(dz, y, mean, invstd, gamma, c1c2))
    t_coef = timeit(lambda: ops.bn_backward_coeffs(st, rows))
    t_fold = timeit(lambda: ops.bn_bwd_fold_conv1x1(wf, wd, gamma, invstd, mean, c1c2))
    t_dg = timeit(lambda: ops.conv2d_dgrad_bn(y, wd4, (B, h, h, ci), 1, 0, y2, m2, r2, scale=sc2, shift=sh2))
    t_k = timeit(lambda: ops.conv1x1_dgrad_kcat(dz, a_in, wk, bias, bn=dict(y=y2, mean=m2, invstd=r2, scale=sc2, shift=sh2)))
    print("planes %3d H%-2d x%d: old main = from_dz %.3f + dgrad %.3f = %.3f ms | new main = coeffs %.3f + fold %.3f + kcat dgrad %.3f = %.3f ms | side apply %.3f ms" % (
        p, h, cnt, t_from, t_dg, t_from + t_dg, t_coef, t_fold, t_k, t_coef + t_fold + t_k, t_apply))
    wm = wf.float().contiguous()
    t_w_old = timeit(lambda: ops.conv2d_wgrad(a_in, y, 1, 1, 0))
    t_w_new = timeit(lambda: ops.conv1x1_wgrad_folded(dz, a_in, wm, gamma, invstd, mean, c1c2))
    print("          side: old = apply %.3f + wgrad %.3f = %.3f ms | folded wgrad %.3f ms" % (t_apply, t_w_old, t_apply + t_w_old, t_w_new))
