"""Diagnostic (GPU box): BatchNorm statistics of conv3 from the Gram matrix of its input (rpe_gram + rpe_bn_stats_from_gram) against the
direct ones (conv epilogue partial sums + rpe_bn_finalize) and against fp64, on the REAL activations of the configs[0] network
(32 images, bf16): per y3-free block the worst |mean error| / std and the worst relative invstd error of both forms."""
import contextlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pose_oracle as po  # noqa: E402
from rgb_proprioceptive_pose_estimator_amd import ops  # noqa: E402
from _helpers import build, load_values  # noqa: E402
from _helpers_cases import C1  # noqa: E402

dtype = torch.bfloat16
cfg, lead, wseed, dseed = C1
sd = po.make_state("no", cfg, wseed)
with contextlib.redirect_stdout(sys.stderr):
    model = build("no", cfg, dtype)
load_values(model, "no", sd)
model.cuda().train()
b = po.synth_batch(lead, dseed + 1)
with torch.no_grad():
    model.train()
    out = model(b["img"].cuda(), None, b["x0bar"].cuda())
plan = model.trunk._active
for li, nb in ((1, 3), (2, 4), (3, 6)):
    for bi in range(nb):
        name = "layer%d.%d" % (li, bi)
        a2 = plan.tensor(name + ".conv2.a")                       # [rows, p] compute dtype
        blk = getattr(model.trunk, "layer%d" % li)[bi]
        w = blk.conv3.weight.detach().reshape(blk.conv3.weight.shape[0], -1).to(dtype).contiguous()   # [4p, p] as the forward multiplies
        rows, p = a2.shape
        side = int(round((rows // lead[0]) ** 0.5))
        x = a2.reshape(lead[0], side, side, p)
        y64 = a2.double() @ w.double().t()
        m64, v64 = y64.mean(0), y64.var(0, unbiased=False)
        r64 = 1.0 / torch.sqrt(v64 + 1e-5)
        gamma, beta = torch.ones(4 * p, device="cuda"), torch.zeros(4 * p, device="cuda")
        S, s1, buf = ops.gram(x)
        _, _, mg, rg = ops.bn_stats_from_gram(w, buf, rows, gamma, beta)
        y, st = ops.conv2d_fwd(x, w.reshape(4 * p, 1, 1, p), 1, 0, want_stats=True)
        _, _, md, rd = ops.bn_finalize(st, rows, gamma, beta)
        S64 = a2.double().t() @ a2.double()
        e = lambda m, r: (((m.double() - m64).abs() * r64).max().item(), ((r.double() / r64 - 1).abs()).max().item())
        eg, ed = e(mg, rg), e(md, rd)
        sb = ((S.double() - S64) / S64.abs().clamp_min(1e-30))
        print("%s rows %6d p %3d: gram |dmean|/std %.2e invstd rel %.2e | direct %.2e %.2e | S rel err mean %+.2e max %.2e; mean/std of y: max %.1f" % (
            name, rows, p, eg[0], eg[1], ed[0], ed[1], sb.mean().item(), sb.abs().max().item(), (m64.abs() * r64).max().item()))
