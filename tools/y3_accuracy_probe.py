"""Diagnostic (GPU box): is the y3-free bottleneck dataflow (BN3 statistics from the Gram matrix, conv3 applying BN in its epilogue,
sum dz*xhat from dz^T a2) as close to the fp32 oracle as the round-3 dataflow?  Train-mode forward + backward of the toy golden cases
and the configs[0] shape over several data seeds, both dataflows, 16-bit types: pose-output error and median gradient error vs the oracle.
usage: python tools/y3_accuracy_probe.py [bf16|f16] [seeds]"""
import contextlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pose_oracle as po  # noqa: E402
from rgb_proprioceptive_pose_estimator_amd import models as M  # noqa: E402
from _helpers import CASES, LOSS_CFG, build, load_values  # noqa: E402
from _helpers_cases import C1  # noqa: E402

dtype = {"bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cases = dict(CASES)
cases["c1"] = C1
cases["bs64"] = (C1[0], (64,), 31, 300)    # tests/test_gpu_parity_sizes.py::test_bs64_flat_gradient_matches_cpu_oracle at seed index 0
if len(sys.argv) > 3:
    cases = {k: v for k, v in cases.items() if k in sys.argv[3].split(",")}


def run(kind, cfg, lead, sd, batch, env):
    for k in ("RPE_NO_Y3FREE", "RPE_Y3_KEEP"):
        os.environ.pop(k, None)
    if env:
        os.environ[env] = "1"
    k2 = "no" if kind in ("c1", "bs64") else kind
    with contextlib.redirect_stdout(sys.stderr):
        model = build(k2, cfg, dtype)
    load_values(model, k2, sd)
    model.cuda().train()
    model.reset_initial_state(lead[-1])
    crit = M.PoseDistanceLoss(distance_metric="combined", scale_factor=1.0, alpha=0.5, mode="pose")
    b = {k: (None if v is None else v.cuda()) for k, v in batch.items()}
    out = model(b["img"], b["depth"], b["x0bar"])
    loss = crit(out[0], b["x0"]) + crit(out[1], b["x1"]) if k2 in ("n", "td") else crit(out, b["obj"])
    loss.backward()
    outs = out if isinstance(out, tuple) else (out,)
    scale = 1.0 if model.loss_scaler is None else model.loss_scaler.get_scale()
    grads = {n: p.grad.detach().float().cpu() / scale for n, p in model.named_parameters() if p.grad is not None}
    return [o.detach().float().cpu() for o in outs], grads


for kind, (cfg, lead, wseed, dseed) in cases.items():
    k2 = "no" if kind in ("c1", "bs64") else kind
    res = {"y3free": [], "round3": [], "emulation": []}
    gres = {"y3free": [], "round3": [], "emulation": []}
    for s in range(nseeds):
        sd = po.make_state(k2, cfg, wseed + 100 * s)
        batch = po.synth_batch(lead, dseed + 1 + 10 * s, with_depth=cfg.get("use_depth", False))
        ref = po.train_step(k2, cfg, {k: v.clone() for k, v in sd.items()}, batch, LOSS_CFG, {}, lr=1e-3, val_metrics=False)
        ro = ref["outputs"] if isinstance(ref["outputs"], tuple) else (ref["outputs"],)
        po.EMULATE = dtype     # the CPU oracle with every stored activation rounded to the compute dtype: the independent yardstick
        try:
            emu = po.train_step(k2, cfg, {k: v.clone() for k, v in sd.items()}, batch, LOSS_CFG, {}, lr=1e-3, val_metrics=False)
        finally:
            po.EMULATE = None
        eo = emu["outputs"] if isinstance(emu["outputs"], tuple) else (emu["outputs"],)
        res["emulation"].append(max(((o - r).abs().max() / r.abs().max().clamp_min(1e-12)).item() for o, r in zip(eo, ro)))
        gres["emulation"].append(float(np.median([((emu["grads"][n].double() - g.double()).norm() / g.double().norm()).item()
                                                  for n, g in ref["grads"].items() if float(g.abs().max()) > 0])))
        for tag, env in (("y3free", None), ("round3", "RPE_NO_Y3FREE")):
            outs, grads = run(kind, cfg, lead, sd, batch, env)
            e = max(((o - r).abs().max() / r.abs().max().clamp_min(1e-12)).item() for o, r in zip(outs, ro))
            res[tag].append(e)
            ge = []
            for n, g in ref["grads"].items():
                if n in grads and float(g.abs().max()) > 0:
                    ge.append(((grads[n].double() - g.double()).norm() / g.double().norm()).item())
            gres[tag].append(float(np.median(ge)))
    for tag in ("y3free", "round3", "emulation"):
        print("%-6s %-7s output rel err: mean %.4f max %.4f  [%s]   median gradient rel err: mean %.3f" % (
            kind, tag, np.mean(res[tag]), np.max(res[tag]), " ".join("%.4f" % e for e in res[tag]), np.mean(gres[tag])))
