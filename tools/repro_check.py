"""Diagnostic: two identical 30-step training runs (same seed, same batch) must end on (nearly) the same parameters -- the only
run-to-run noise allowed is the fp32 atomic order of the weight-gradient accumulation.  A race in a kernel or in the two-stream
schedule shows up here as a large or unstable difference.  usage: python tools/repro_check.py [steps]"""
import contextlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rgb_proprioceptive_pose_estimator_amd import models as M
from rgb_proprioceptive_pose_estimator_amd.optim import FusedAdam
from rgb_proprioceptive_pose_estimator_amd.util.data_utils import synthetic_batch
from rgb_proprioceptive_pose_estimator_amd.util.learn_utils import train_step

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def run():
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):
        model = M.NaiveObjectStateEstimator("cube", [1024, 256, 64], 50, 512, False, (9,), False, False, False, compute_dtype=torch.bfloat16)
    model.cuda().train()
    crit = {"obj_loss": M.PoseDistanceLoss("combined", 1.0, 0.5, 1e-4, "pose"), "val_loss": M.PoseDistanceLoss(mode="val")}
    opt = FusedAdam(model.parameters(), lr=1e-4)
    b = synthetic_batch((64,), 1234)
    batch = (b["img"], None, b["x0bar"], b["x0"], None, b["obj"])
    losses = []
    for _ in range(STEPS):
        loss, _, _ = train_step(model, batch, crit, opt, True, "train", None)
        losses.append(loss)
    torch.cuda.synchronize()
    flat = model._arena.flat.detach().clone()
    return flat, torch.stack(losses).cpu()


def grad_twice():
    """same weights, same batch, two forward+backward passes: gradients may differ by fp32 atomic order only"""
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):
        model = M.NaiveObjectStateEstimator("cube", [1024, 256, 64], 50, 512, False, (9,), False, False, False, compute_dtype=torch.bfloat16)
    model.cuda().train()
    crit = M.PoseDistanceLoss("combined", 1.0, 0.5, 1e-4, "pose")
    b = synthetic_batch((64,), 1234)
    gs = []
    for _ in range(3):
        if model._arena is not None:
            model._arena.zero_grad()
        crit(model(b["img"], None, b["x0bar"]), b["obj"]).backward()
        torch.cuda.synchronize()
        gs.append(model._arena.grad.detach().clone())
    for i in (1, 2):
        d = (gs[i] - gs[0]).abs()
        print("pass %d vs 0: max |dgrad| %.3e (max |grad| %.3e), |d|/|g| (L2) %.3e" % (i, d.max().item(), gs[0].abs().max().item(), (d.norm() / gs[0].norm()).item()))
    ref_path = os.environ.get("GRAD_REF")
    if ref_path:
        if os.environ.get("GRAD_REF_SAVE"):
            torch.save(gs[0].cpu(), ref_path)
        else:
            ref = torch.load(ref_path).cuda()
            off = 0
            for name, prm in model.named_parameters():
                n = prm.numel()
                pad = (n + 3) // 4 * 4
                if "layer1.0.conv2.weight" in name or "layer1.1.conv2.weight" in name or "layer1.2.conv2.weight" in name:
                    r = ref[off:off + n]
                    for i in range(3):
                        g = gs[i][off:off + n]
                        d = (g - r)
                        bad = (d.abs() > 1e-3 * r.abs().max()).nonzero().flatten()
                        print("   %s pass %d vs single-stream reference: rel L2 %.3e, elements off by > 1e-3 of max: %d, first idx %s" % (
                            name.split("module.")[-1], i, (d.norm() / r.norm()).item(), bad.numel(), bad[:8].tolist()))
                        if bad.numel():
                            co = (bad // (9 * 64)).unique()[:8].tolist(); tap = ((bad // 64) % 9).unique().tolist(); ci = (bad % 64).unique()[:12].tolist()
                            print("      out channels %s taps %s in channels %s ; ratio g/r of first: %s" % (co, tap, ci, (g[bad[:4]] / r[bad[:4]]).tolist()))
                off += pad
    if os.environ.get("VERBOSE"):
        off = 0
        rows = []
        for name, prm in model.named_parameters():
            n = prm.numel()
            pad = (n + 3) // 4 * 4
            g0, g1 = gs[0][off:off + n], gs[1][off:off + n]
            rows.append(((g1 - g0).norm().item() / (g0.norm().item() + 1e-30), name, g0.norm().item()))
            off += pad
        for r, name, gn in sorted(rows, reverse=True)[:14]:
            print("   %-60s rel diff %.3e  |g| %.3e" % (name, r, gn))
        print("   params with rel diff > 1e-5: %d of %d" % (sum(1 for r, _, _ in rows if r > 1e-5), len(rows)))


grad_twice()
a, la = run()
b, lb = run()
d = (a - b).abs()
print("params finite: %s / %s" % (bool(torch.isfinite(a).all()), bool(torch.isfinite(b).all())))
print("max |param diff| %.3e, mean %.3e, relative to |param| mean %.3e" % (d.max().item(), d.mean().item(), (d.mean() / a.abs().mean()).item()))
print("loss first/last run A: %.4f -> %.4f ; run B: %.4f -> %.4f" % (la[0], la[-1], lb[0], lb[-1]))
print("per-step |loss diff| max: %.3e" % (la - lb).abs().max().item())
