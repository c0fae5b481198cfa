"""Diagnostic (GPU box): relative l2 error of every block output (conv3.a) of the bf16 trunk against the fp32 trunk on the same weights and
images, y3-free dataflow vs the round-3 dataflow (RPE_NO_Y3FREE=1), configs[0] network in training mode.  usage: python tools/y3_layer_probe.py [seed]"""
import contextlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pose_oracle as po  # noqa: E402
from _helpers import build, load_values  # noqa: E402
from _helpers_cases import C1  # noqa: E402

cfg, lead, wseed, dseed = C1
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sd = po.make_state("no", cfg, wseed + 100 * seed)
b = po.synth_batch(lead, dseed + 1 + 10 * seed)
names = ["layer%d.%d.conv3.a" % (li, bi) for li, nb in ((1, 3), (2, 4), (3, 6), (4, 3)) for bi in range(nb)]


def run(dtype, env):
    for k in ("RPE_NO_Y3FREE", "RPE_Y3_KEEP"):
        os.environ.pop(k, None)
    if env:
        os.environ[env] = "1"
    with contextlib.redirect_stdout(sys.stderr):
        model = build("no", cfg, dtype)
    load_values(model, "no", sd)
    model.cuda().train()
    with torch.no_grad():
        out = model(b["img"].cuda(), None, b["x0bar"].cuda())
    plan = model.trunk._active
    return {n: plan.tensor(n).float().clone() for n in names}, out.float().cpu()


ref, oref = run(torch.float32, None)
new, onew = run(torch.bfloat16, None)
old, oold = run(torch.bfloat16, "RPE_NO_Y3FREE")
for n in names:
    en = ((new[n] - ref[n]).norm() / ref[n].norm()).item()
    eo = ((old[n] - ref[n]).norm() / ref[n].norm()).item()
    # the per-CHANNEL mean of the error (what survives the global average pool and every later BatchNorm's batch statistics), relative to
    # the per-channel mean of the fp32 activations
    cn = ((new[n] - ref[n]).mean(0).norm() / ref[n].mean(0).norm()).item()
    co = ((old[n] - ref[n]).mean(0).norm() / ref[n].mean(0).norm()).item()
    print("%-18s rel l2 error vs fp32: y3-free %.5f  round-3 %.5f  ratio %.3f | channel-mean error: %.6f  %.6f  ratio %.3f" % (n, en, eo, en / eo, cn, co, cn / co))
print("outputs: y3-free %.5f round-3 %.5f" % (((onew - oref).abs().max() / oref.abs().max()).item(), ((oold - oref).abs().max() / oref.abs().max()).item()))
