"""Convergence record: loss per train step of the bench workload, HIP path in bf16 / fp16 / fp32 from identical initial weights, and the
CPU oracle for the first steps from the same state.  Two data modes: ONE seeded batch repeated (the model must overfit it: round 3) or,
with `fresh`, a NEW seeded batch every step drawn from the synthetic distribution (what training on a stream of simulator episodes is:
util/learn_utils.py:152-184 of the reference; the loss then converges to the distribution's irreducible level and the three compute
types can be compared on the same sample path).  NaN loss VALUES (an all-zero post-ReLU quaternion: models/losses.py:68-69 of the
reference, pinned by tests/golden/model_no_nanloss.npz) are recorded as null.

    python tools/loss_record.py [images 32] [steps 200] [oracle_steps 20] [fresh] > profiles/r04_loss_trace_bs256.json
"""
import contextlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from oracle import pose_oracle as po
from rgb_proprioceptive_pose_estimator_amd import models as M
from rgb_proprioceptive_pose_estimator_amd.optim import FusedAdam
from rgb_proprioceptive_pose_estimator_amd.util.learn_utils import train_step

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ORACLE = int(sys.argv[3]) if len(sys.argv) > 3 else 20
FRESH = len(sys.argv) > 4 and sys.argv[4] == "fresh"
CFG = dict(latent_dim=512, hidden=[1024, 256, 64], use_depth=False, no_proprioception=False)
LOSS = dict(metric="combined", scale=1.0, alpha=0.5, mode="pose")
sd0 = po.make_state("no", CFG, 0)
batch = po.synth_batch((B,), 1234)


def clean(x):
    return None if x != x else round(x, 5)


out = {"workload": "NaiveObjectStateEstimator latent 512 hidden [1024,256,64], %d images of 224x224, %s, Adam lr 1e-3, "
                   "PoseDistanceLoss(combined, alpha 0.5)" % (B, "a FRESH seeded batch every step (seed 1234 + step; identical sample path for every compute type)"
                                                                 if FRESH else "the SAME seeded batch every step"), "steps": STEPS}


def dev_batch(i):
    bb = po.synth_batch((B,), 1234 + i) if FRESH else batch
    return tuple(None if t is None else t.cuda() for t in (bb["img"], None, bb["x0bar"], bb["x0"], None, bb["obj"]))

DTYPES = (("f32", torch.float32), ("bf16", torch.bfloat16), ("f16", torch.float16))
models, opts, crits, hist = {}, {}, {}, {}
for name, dtype in DTYPES:
    with contextlib.redirect_stdout(sys.stderr):
        m = M.NaiveObjectStateEstimator("cube", [1024, 256, 64], 50, 512, False, (9,), False, False, False, compute_dtype=dtype)
    m.load_state_dict({k: v.clone() for k, v in sd0.items()})
    m.cuda().train()
    models[name], opts[name] = m, FusedAdam(m.parameters(), lr=1e-3)
    crits[name] = {"obj_loss": M.PoseDistanceLoss(distance_metric="combined", scale_factor=1.0, alpha=0.5, mode="pose"), "val_loss": M.PoseDistanceLoss(mode="val")}
    hist[name] = ([], [])
b = dev_batch(0)
for i in range(STEPS):          # steps outside, compute types inside: every type sees the same batch, generated once
    if FRESH and i:
        b = dev_batch(i)
    for name, _ in DTYPES:
        loss, pe, oe = train_step(models[name], b, crits[name], opts[name], True, "train", None)
        hist[name][0].append(clean(float(loss.item())))
        hist[name][1].append(round(float(pe.item()) / B, 5))
    if i % 25 == 0:
        print("[loss_record] step %d: %s" % (i, {n: hist[n][0][-1] for n, _ in DTYPES}), file=sys.stderr, flush=True)
for name, _ in DTYPES:
    ls, pos = hist[name]
    tail = [x for x in ls[-50:] if x is not None]
    out[name] = {"loss": ls, "mean_pos_err_m": pos, "nan_loss_steps": sum(1 for x in ls if x is None),
                 "mean_loss_last_50": round(sum(tail) / max(1, len(tail)), 4), "mean_pos_err_m_last_50": round(sum(pos[-50:]) / len(pos[-50:]), 5),
                 "params_finite": bool(torch.isfinite(models[name]._arena.flat).all().item())}
    print("[loss_record] %s: first %.4f last %s min %s" % (name, ls[0], ls[-1], min(x for x in ls if x is not None)), file=sys.stderr)
del models, opts
torch.cuda.empty_cache()
if ORACLE:
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    sd, opt, ls = {k: v.clone() for k, v in sd0.items()}, {}, []
    for i in range(ORACLE):
        r = po.train_step("no", CFG, sd, po.synth_batch((B,), 1234 + i) if FRESH else batch, LOSS, opt, val_metrics=False)
        ls.append(clean(float(r["loss"])))
    out["oracle_f32_cpu"] = {"loss": ls}
print(json.dumps(out))
