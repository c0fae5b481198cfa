"""Diagnostic (GPU box): how closely the HIP path follows the reference through TWO Adam steps, per parameter group -- the numbers the
bars of tests/test_gpu_models.py::test_two_adam_steps_against_reference_final_vectors were set from.  usage: python tools/adam_parity_probe.py [f32|bf16|f16]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pose_oracle as po  # noqa: E402
from rgb_proprioceptive_pose_estimator_amd import models as M  # noqa: E402
from rgb_proprioceptive_pose_estimator_amd.optim import FusedAdam  # noqa: E402
from _helpers import CASES, LOSS_CFG, build, load_values  # noqa: E402

dtype = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "f32"]
LR = 1e-3


def to_dev(b):
    return {k: (None if v is None else v.cuda()) for k, v in b.items()}


for kind in CASES:
    gold = np.load(os.path.join(ROOT, "tests", "golden", "model_%s.npz" % kind))
    cfg, lead, wseed, dseed = CASES[kind]
    sd0 = po.make_state(kind, cfg, wseed)
    model = build(kind, cfg, dtype)
    load_values(model, kind, sd0)
    model.cuda().train()
    crit = M.PoseDistanceLoss(distance_metric="combined", scale_factor=1.0, alpha=0.5, mode="pose")
    opt = FusedAdam(model.parameters(), lr=LR)
    sd_ref = {k: v.clone() for k, v in sd0.items()}
    ost = {}
    for step in (1, 2):
        b = po.synth_batch(lead, dseed + step, with_depth=cfg.get("use_depth", False))
        r = po.train_step(kind, cfg, sd_ref, b, LOSS_CFG, ost, lr=LR)
        model.reset_initial_state(lead[-1])
        opt.zero_grad()
        bd = to_dev(b)
        out = model(bd["img"], bd["depth"], bd["x0bar"])
        loss = crit(out[0], bd["x0"]) + crit(out[1], bd["x1"]) if kind in ("n", "td") else crit(out, bd["obj"])
        loss.backward()
        opt.step()
        msd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
        frac = {"trunk": [], "head": []}
        strong_bad = 0
        for k, g in r["grads"].items():
            u_ref = sd_ref[k].float() - sd0[k].float() if step == 1 else None
            if step != 1:
                continue
            u_gpu = msd[k] - sd0[k].float()
            live = g.abs() > 1e-7 * g.abs().max().clamp_min(1e-30)
            ok = (u_gpu - u_ref).abs() <= 1e-2 * LR
            grp = "trunk" if "feature_net" in k and ".fc." not in k else "head"
            if live.any():
                frac[grp].append((ok & live).sum().item() / live.sum().item())
            strong = g.abs() > 0.05 * g.abs().max()
            strong_bad += int((~ok & strong).sum())
        if step == 1:
            print("%s step1: update agreement (|du| <= 1e-2 lr) trunk min %.4f median %.4f | head min %.4f median %.4f | strong-gradient misses %d" % (
                kind, min(frac["trunk"]), float(np.median(frac["trunk"])), min(frac["head"]), float(np.median(frac["head"])), strong_bad))
    worst = {"trunk": 0.0, "head": 0.0}
    relw = {"trunk": 0.0, "head": 0.0}
    for k in gold.files:
        if not k.startswith("final::"):
            continue
        name = k[7:]
        if name.endswith("num_batches_tracked"):
            continue
        ref = torch.from_numpy(gold[k]).float()
        grp = "trunk" if "feature_net" in name and ".fc." not in name else "head"
        d = (msd[name] - ref).abs()
        worst[grp] = max(worst[grp], d.max().item())
        relw[grp] = max(relw[grp], (d / ref.abs().clamp_min(1e-3)).max().item())
    dg = gold["final_digest"]
    dn = 0.0
    for name, ref in zip(gold["keys"], dg):
        t = msd[str(name)].double()
        if ref[1] > 0:
            dn = max(dn, abs(t.norm().item() - ref[1]) / ref[1])
    print("%s 2 steps vs reference final:: vectors: max abs diff trunk %.3e head %.3e (rel-to-max(|ref|,1e-3): %.3e / %.3e); l2-norm digest rel diff max %.3e" % (
        kind, worst["trunk"], worst["head"], relw["trunk"], relw["head"], dn))
