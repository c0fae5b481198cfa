"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz by running the REFERENCE's
own classes (imported from /root/reference through oracle/ref_stubs.py) on seeded
synthetic inputs.  Run in the build container only:

    python -B oracle/gen_golden.py

The reference tree never travels; only the resulting input/output vectors do.
Weights are not stored: both sides regenerate them with
oracle.pose_oracle.make_state(kind, cfg, seed) (a per-key seeded fill, independent
of module construction order).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import ref_stubs  # noqa: E402
from oracle import pose_oracle as po  # noqa: E402

ref_stubs.install()
sys.path.insert(0, "/root/reference")
from models.losses import PoseDistanceLoss  # noqa: E402
from models.naive import NaiveEndEffectorStateEstimator, NaiveObjectStateEstimator  # noqa: E402
from models.time_sensitive import (  # noqa: E402
    TemporallyDependentObjectStateEstimator,
    TemporallyDependentObjectStateEstimatorV2,
    TemporallyDependentStateEstimator,
)

OUT = os.path.join(ROOT, "tests", "golden")

CASES = {
    # kind: (cfg, lead, weight seed, data seed)
    "no": (dict(latent_dim=64, hidden=[32, 16], use_depth=False, no_proprioception=False), (2,), 11, 101),
    "n": (dict(latent_dim=64, hidden=[32]), (2,), 12, 102),
    "td": (dict(latent_dim=64, hidden=32, use_depth=False), (2, 2), 13, 103),
    "tdo": (dict(latent_dim=64, hidden=32, use_depth=True, no_proprioception=False), (2, 2), 14, 104),
    "tdo_v2": (dict(latent_dim=64, hidden=32, proprio_hidden=8, use_depth=False), (2, 2), 15, 105),
}
LOSS_CFG = dict(distance_metric="combined", scale_factor=1.0, alpha=0.5, mode="pose")


def build(kind, cfg):
    L = cfg["latent_dim"]
    if kind == "n":
        return NaiveEndEffectorStateEstimator(list(cfg["hidden"]), list(cfg["hidden"]), 50, L, False)
    if kind == "no":
        return NaiveObjectStateEstimator("cube", list(cfg["hidden"]), 50, L, False, (9,), cfg["use_depth"], False,
                                         cfg["no_proprioception"])
    if kind == "td":
        return TemporallyDependentStateEstimator(cfg["hidden"], cfg["hidden"], 50, L, 2, 0.1, False, (9,),
                                                 cfg["use_depth"], False)
    if kind == "tdo":
        return TemporallyDependentObjectStateEstimator("hammer", cfg["hidden"], 50, L, 2, 0.1, False, (9,),
                                                       cfg["use_depth"], False, cfg["no_proprioception"])
    return TemporallyDependentObjectStateEstimatorV2("robot1_eef", cfg["hidden"], cfg["proprio_hidden"], 50, L, 2, 0.1,
                                                     False, (9,), cfg["use_depth"], False)


def load_values(model, kind, sd):
    real = {k: v for k, v in sd.items() if not k.startswith("~")}
    missing = model.load_state_dict(real, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    if kind == "td":  # unregistered heads (plain lists): assign directly
        with torch.no_grad():
            model.aux_nets[0][0].weight.copy_(sd["~aux_nets.0.0.weight"])
            model.aux_nets[0][0].bias.copy_(sd["~aux_nets.0.0.bias"])
            model.depth_nets[0][2].weight.copy_(sd["~depth_nets.0.2.weight"])
            model.depth_nets[0][2].bias.copy_(sd["~depth_nets.0.2.bias"])


def digest(t):
    t = t.detach().double().flatten()
    return np.array([t.sum().item(), t.norm().item(), t.abs().max().item() if t.numel() else 0.0])


def run_case(kind):
    cfg, lead, wseed, dseed = CASES[kind]
    torch.manual_seed(0)
    model = build(kind, cfg)
    sd = po.make_state(kind, cfg, wseed)
    # the key table itself is part of what is pinned
    ref_keys = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    mine = [(k, tuple(s)) for k, s in po.model_keys(kind, cfg) if not k.startswith("~")]
    assert ref_keys == mine, "state_dict key table mismatch for %s" % kind
    load_values(model, kind, sd)
    model.train()
    model.reset_initial_state(lead[-1])
    crit = PoseDistanceLoss(**LOSS_CFG)
    val = PoseDistanceLoss(mode="val")
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    rec = {"keys": np.array([k for k, _ in ref_keys])}

    def eval_pass(tag):
        # eval-mode forward (BN running-stats path) and the rollout contract (carried (h, c))
        model.eval()
        model.rollout = False
        model.reset_initial_state(lead[-1])
        with torch.no_grad():
            b = po.synth_batch(lead, dseed + 9, with_depth=cfg.get("use_depth", False))
            depth = b["depth"] if b["depth"] is not None else torch.empty(*b["img"].shape)
            out = model(b["img"], depth, b["x0bar"])
            if isinstance(out, tuple):
                rec[tag + "eval_out0"], rec[tag + "eval_out1"] = out[0].numpy(), out[1].numpy()
            else:
                rec[tag + "eval_out0"] = out.numpy()
            if kind in ("td", "tdo", "tdo_v2"):
                model.rollout = True
                model.reset_initial_state(lead[-1])
                o1 = model(b["img"][:1], depth[:1], b["x0bar"][:1])
                o2 = model(b["img"][1:], depth[1:], b["x0bar"][1:])
                o1 = o1[-1] if isinstance(o1, tuple) else o1
                o2 = o2[-1] if isinstance(o2, tuple) else o2
                rec[tag + "rollout_out"] = torch.cat([o1, o2], 0).numpy()
                model.rollout = False
        model.train()
        model.reset_initial_state(lead[-1])

    eval_pass("pre_")
    for step in (1, 2):
        b = po.synth_batch(lead, dseed + step, with_depth=cfg.get("use_depth", False))
        depth = b["depth"] if b["depth"] is not None else torch.empty(*b["img"].shape)
        opt.zero_grad()
        out = model(b["img"], depth, b["x0bar"])
        if kind in ("n", "td"):
            loss = crit(out[0], b["x0"]) + crit(out[1], b["x1"])
            pos_err, ori_err = val(out[1], b["x1"])
            rec["out0_s%d" % step] = out[0].detach().numpy()
            rec["out1_s%d" % step] = out[1].detach().numpy()
        else:
            loss = crit(out, b["obj"])
            pos_err, ori_err = val(out, b["obj"])
            rec["out0_s%d" % step] = out.detach().numpy()
        loss.backward()
        rec["loss_s%d" % step] = np.array(loss.item())
        rec["pos_err_s%d" % step] = np.array(float(pos_err))
        rec["ori_err_s%d" % step] = np.array(float(ori_err))
        gnames, gdig = [], []
        for name, p in model.named_parameters():
            if p.grad is not None:
                gnames.append(name)
                gdig.append(digest(p.grad))
        rec["grad_keys_s%d" % step] = np.array(gnames)
        rec["grad_digest_s%d" % step] = np.stack(gdig)
        if step == 1:  # a few raw gradient tensors (small ones) for element-wise checks
            named = dict(model.named_parameters())
            for name in gnames:
                if named[name].numel() <= 4096:
                    rec["grad::" + name] = named[name].grad.detach().numpy().copy()
        opt.step()
    fin = model.state_dict()
    rec["final_digest"] = np.stack([digest(fin[k]) for k, _ in ref_keys])
    for k, _ in ref_keys:  # small tensors kept element-wise (biases, BN stats of the stem, heads)
        if fin[k].numel() <= 4096 and fin[k].dtype.is_floating_point:
            rec["final::" + k] = fin[k].detach().numpy().copy()
    eval_pass("post_")
    np.savez_compressed(os.path.join(OUT, "model_%s.npz" % kind), **rec)
    print(kind, "loss", rec["loss_s1"], rec["loss_s2"], "pos/ori err", rec["pos_err_s1"], rec["ori_err_s1"])


# BASELINE.json configs[0] at its own size: NaiveObjectStateEstimator('cube', [1024, 256, 64], 50, 512, ...), 32 images of 224x224
# (scripts/train_no.sbatch:68-82 hyper-parameters; models/naive.py:298-352 forward).  One pristine eval forward and ONE train
# step: outputs, loss, val metrics, a digest of every gradient, small gradients whole and a strided sample of the large ones.
C1 = (dict(latent_dim=512, hidden=[1024, 256, 64], use_depth=False, no_proprioception=False), (32,), 21, 201)
SAMPLE_STRIDE, SAMPLE_MAX = 997, 4096


def run_c1():
    cfg, lead, wseed, dseed = C1
    torch.manual_seed(0)
    model = build("no", cfg)
    sd = po.make_state("no", cfg, wseed)
    ref_keys = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    assert ref_keys == [(k, tuple(s)) for k, s in po.model_keys("no", cfg)], "state_dict key table mismatch (C1)"
    load_values(model, "no", sd)
    crit = PoseDistanceLoss(**LOSS_CFG)
    val = PoseDistanceLoss(mode="val")
    rec = {"keys": np.array([k for k, _ in ref_keys])}
    model.eval()
    with torch.no_grad():
        b = po.synth_batch(lead, dseed + 9)
        rec["pre_eval_out0"] = model(b["img"], torch.empty(*b["img"].shape), b["x0bar"]).numpy()
    model.train()
    b = po.synth_batch(lead, dseed + 1)
    out = model(b["img"], torch.empty(*b["img"].shape), b["x0bar"])
    loss = crit(out, b["obj"])
    pos_err, ori_err = val(out, b["obj"])
    loss.backward()
    rec["out0_s1"] = out.detach().numpy()
    rec["loss_s1"], rec["pos_err_s1"], rec["ori_err_s1"] = np.array(loss.item()), np.array(float(pos_err)), np.array(float(ori_err))
    gnames, gdig = [], []
    for name, p in model.named_parameters():
        if p.grad is None:
            continue
        gnames.append(name)
        gdig.append(digest(p.grad))
        g = p.grad.detach().flatten()
        if g.numel() <= 4096:
            rec["grad::" + name] = g.numpy().copy()
        else:
            rec["gsample::" + name] = g[::SAMPLE_STRIDE][:SAMPLE_MAX].numpy().copy()
    rec["grad_keys_s1"], rec["grad_digest_s1"] = np.array(gnames), np.stack(gdig)
    fin = model.state_dict()   # BN running statistics after the one training forward
    for k, _ in ref_keys:
        if k.endswith("running_mean") or k.endswith("running_var"):
            rec["final::" + k] = fin[k].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "model_no_c1.npz"), **rec)
    print("no_c1 loss", rec["loss_s1"], "pos/ori err", rec["pos_err_s1"], rec["ori_err_s1"])


# The deeper bottleneck trunk import_resnet also offers (util/model_utils.py:130-136): NaiveObjectStateEstimator on ResNet-101,
# two images.  Key table, pristine eval output, step-1 outputs / loss / gradient digests.
R101 = (dict(latent_dim=64, hidden=[32], use_depth=False, no_proprioception=False, depth=101), (2,), 41, 401)


# ... and the BasicBlock member it reaches: ResNet-18 (its option set {18, 32, 50, 101, 152} spells 34 as 32, which torchvision does not have)
R18 = (dict(latent_dim=64, hidden=[32], use_depth=False, no_proprioception=False, depth=18), (2,), 45, 451)


def run_r101(case=None, tag="r101"):
    cfg, lead, wseed, dseed = case or R101
    torch.manual_seed(0)
    model = NaiveObjectStateEstimator("cube", list(cfg["hidden"]), cfg["depth"], cfg["latent_dim"], False, (9,), False, False, False)
    sd = po.make_state("no", cfg, wseed)
    ref_keys = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    assert ref_keys == [(k, tuple(s)) for k, s in po.model_keys("no", cfg)], "state_dict key table mismatch (ResNet-%d)" % cfg["depth"]
    load_values(model, "no", sd)
    rec = {"keys": np.array([k for k, _ in ref_keys])}
    model.eval()
    with torch.no_grad():
        b = po.synth_batch(lead, dseed + 9)
        rec["pre_eval_out0"] = model(b["img"], torch.empty(*b["img"].shape), b["x0bar"]).numpy()
    model.train()
    b = po.synth_batch(lead, dseed + 1)
    out = model(b["img"], torch.empty(*b["img"].shape), b["x0bar"])
    loss = PoseDistanceLoss(**LOSS_CFG)(out, b["obj"])
    loss.backward()
    rec["out0_s1"], rec["loss_s1"] = out.detach().numpy(), np.array(loss.item())
    gn, gd = [], []
    for name, p in model.named_parameters():
        if p.grad is not None:
            gn.append(name)
            gd.append(digest(p.grad))
    rec["grad_keys_s1"], rec["grad_digest_s1"] = np.array(gn), np.stack(gd)
    np.savez_compressed(os.path.join(OUT, "model_no_%s.npz" % tag), **rec)
    print("no_%s loss" % tag, rec["loss_s1"])


# BASELINE.json configs[2]: the two-arm TD model on sequences of FOUR frames (lead dims (S, N) = (4, 2); the toy cases above run
# S = 2).  Pristine eval output, a 4-frame rollout with the LSTM state carried between single-frame calls, step-1 outputs / loss /
# val metrics / gradient digests.
TD_S4 = (dict(latent_dim=64, hidden=32, use_depth=False), (4, 2), 51, 501)


def run_td_s4():
    cfg, lead, wseed, dseed = TD_S4
    torch.manual_seed(0)
    model = TemporallyDependentStateEstimator(cfg["hidden"], cfg["hidden"], 50, cfg["latent_dim"], 4, 0.1, False, (9,), False, False)
    sd = po.make_state("td", cfg, wseed)
    ref_keys = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    assert ref_keys == [(k, tuple(s)) for k, s in po.model_keys("td", cfg) if not k.startswith("~")], "state_dict key table mismatch (TD, S=4)"
    load_values(model, "td", sd)
    rec = {"keys": np.array([k for k, _ in ref_keys])}
    model.eval()
    model.reset_initial_state(lead[-1])
    with torch.no_grad():
        b = po.synth_batch(lead, dseed + 9)
        blank = torch.empty(*b["img"].shape)
        out = model(b["img"], blank, b["x0bar"])
        rec["pre_eval_out0"], rec["pre_eval_out1"] = out[0].numpy(), out[1].numpy()
        model.rollout = True
        model.reset_initial_state(lead[-1])
        frames = [model(b["img"][t:t + 1], blank[t:t + 1], b["x0bar"][t:t + 1])[-1] for t in range(lead[0])]
        rec["pre_rollout_out"] = torch.cat(frames, 0).numpy()
        model.rollout = False
    model.train()
    model.reset_initial_state(lead[-1])
    b = po.synth_batch(lead, dseed + 1)
    out = model(b["img"], torch.empty(*b["img"].shape), b["x0bar"])
    crit = PoseDistanceLoss(**LOSS_CFG)
    loss = crit(out[0], b["x0"]) + crit(out[1], b["x1"])
    pe, oe = PoseDistanceLoss(mode="val")(out[1], b["x1"])
    loss.backward()
    rec["out0_s1"], rec["out1_s1"], rec["loss_s1"] = out[0].detach().numpy(), out[1].detach().numpy(), np.array(loss.item())
    rec["pos_err_s1"], rec["ori_err_s1"] = np.array(float(pe)), np.array(float(oe))
    gn, gd = [], []
    for name, p in model.named_parameters():
        if p.grad is not None:
            gn.append(name)
            gd.append(digest(p.grad))
    rec["grad_keys_s1"], rec["grad_digest_s1"] = np.array(gn), np.stack(gd)
    np.savez_compressed(os.path.join(OUT, "model_td_s4.npz"), **rec)
    print("td_s4 loss", rec["loss_s1"], "pos/ori err", rec["pos_err_s1"], rec["ori_err_s1"])


# feature_layer_nums other than the scripts' (9,) (models/naive.py:196-240): every hook the reference can run at 224x224 -- conv1's raw
# output, bn1, layer1..3 (given out of order: aux_nets follow the order the hooks FIRE in) -- with the depth heads on; and None (no
# early features).  Key table, pristine eval output, step-1 outputs / loss / gradient digests + the head gradients whole.
HOOKS = (dict(latent_dim=64, hidden=[32], use_depth=True, no_proprioception=False, hooks=(3, 0, 9, 2, 1)), (2,), 61, 601)
HOOKS18 = (dict(latent_dim=64, hidden=[32], use_depth=True, no_proprioception=False, hooks=(3, 0, 9, 2, 1), depth=18), (2,), 63, 603)   # the same on resnet18
NOHOOK = (dict(latent_dim=64, hidden=[32], use_depth=False, no_proprioception=False, hooks=None), (2,), 62, 602)


def run_hooks(tag, case):
    cfg, lead, wseed, dseed = case
    torch.manual_seed(0)
    model = NaiveObjectStateEstimator("cube", list(cfg["hidden"]), cfg.get("depth", 50), cfg["latent_dim"], False, cfg["hooks"], cfg["use_depth"], False, False)
    sd = po.make_state("no", cfg, wseed)
    ref_keys = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    assert ref_keys == [(k, tuple(s)) for k, s in po.model_keys("no", cfg)], "state_dict key table mismatch (%s)" % tag
    load_values(model, "no", sd)
    rec = {"keys": np.array([k for k, _ in ref_keys])}
    model.eval()
    with torch.no_grad():
        b = po.synth_batch(lead, dseed + 9, with_depth=cfg["use_depth"])
        depth = b["depth"] if b["depth"] is not None else torch.empty(*b["img"].shape)
        rec["pre_eval_out0"] = model(b["img"], depth, b["x0bar"]).numpy()
    model.train()
    b = po.synth_batch(lead, dseed + 1, with_depth=cfg["use_depth"])
    depth = b["depth"] if b["depth"] is not None else torch.empty(*b["img"].shape)
    out = model(b["img"], depth, b["x0bar"])
    loss = PoseDistanceLoss(**LOSS_CFG)(out, b["obj"])
    loss.backward()
    rec["out0_s1"], rec["loss_s1"] = out.detach().numpy(), np.array(loss.item())
    gn, gd = [], []
    for name, p in model.named_parameters():
        if p.grad is not None:
            gn.append(name)
            gd.append(digest(p.grad))
            if name.startswith("aux_nets") or name.startswith("depth_nets") or name.endswith("conv1.weight") and "layer" not in name:
                rec["grad::" + name] = p.grad.detach().numpy().copy()
    rec["grad_keys_s1"], rec["grad_digest_s1"] = np.array(gn), np.stack(gd)
    np.savez_compressed(os.path.join(OUT, "model_no_%s.npz" % tag), **rec)
    print("no_%s loss" % tag, rec["loss_s1"])


# BASELINE.json configs[2..4] at the head sizes the reference's scripts train (scripts/train_model.py:25,180-181 latent 512 / hidden 512;
# scripts/train_tdo_v2.sbatch:68-70 proprio hidden 64): TD, TDO (+ depth head) and TDO-V2 on sequences of four frames, eight episodes
# (lead dims (S, N) = (4, 8): 32 images -- what the CPU reference steps through in seconds).  Pristine eval outputs, step-1 outputs /
# loss / val metrics, a digest of every gradient, the small gradients whole and a strided sample of the large ones (the LSTM input
# projections are 2048 x 3648..3655: 30 MB each).
SEQ_CFG = {
    "td": (dict(latent_dim=512, hidden=512, use_depth=False), (4, 8), 71, 701),
    "tdo": (dict(latent_dim=512, hidden=512, use_depth=True, no_proprioception=False), (4, 8), 72, 702),
    "tdo_v2": (dict(latent_dim=512, hidden=512, proprio_hidden=64, use_depth=False), (4, 8), 73, 703),
}
SEQ_SAMPLE_STRIDE, SEQ_SAMPLE_MAX = 499, 16384


def run_seq_cfg(kind):
    cfg, lead, wseed, dseed = SEQ_CFG[kind]
    torch.manual_seed(0)
    L, S = cfg["latent_dim"], lead[0]
    if kind == "td":
        model = TemporallyDependentStateEstimator(cfg["hidden"], cfg["hidden"], 50, L, S, 0.1, False, (9,), cfg["use_depth"], False)
    elif kind == "tdo":
        model = TemporallyDependentObjectStateEstimator("hammer", cfg["hidden"], 50, L, S, 0.1, False, (9,), cfg["use_depth"], False, cfg["no_proprioception"])
    else:
        model = TemporallyDependentObjectStateEstimatorV2("robot1_eef", cfg["hidden"], cfg["proprio_hidden"], 50, L, S, 0.1, False, (9,), cfg["use_depth"], False)
    sd = po.make_state(kind, cfg, wseed)
    ref_keys = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    assert ref_keys == [(k, tuple(s)) for k, s in po.model_keys(kind, cfg) if not k.startswith("~")], "state_dict key table mismatch (%s, config size)" % kind
    load_values(model, kind, sd)
    rec = {"keys": np.array([k for k, _ in ref_keys])}
    use_depth = cfg.get("use_depth", False)
    model.eval()
    model.reset_initial_state(lead[-1])
    with torch.no_grad():
        b = po.synth_batch(lead, dseed + 9, with_depth=use_depth)
        depth = b["depth"] if b["depth"] is not None else torch.empty(*b["img"].shape)
        out = model(b["img"], depth, b["x0bar"])
        if isinstance(out, tuple):
            rec["pre_eval_out0"], rec["pre_eval_out1"] = out[0].numpy(), out[1].numpy()
        else:
            rec["pre_eval_out0"] = out.numpy()
    model.train()
    model.reset_initial_state(lead[-1])
    b = po.synth_batch(lead, dseed + 1, with_depth=use_depth)
    depth = b["depth"] if b["depth"] is not None else torch.empty(*b["img"].shape)
    out = model(b["img"], depth, b["x0bar"])
    crit = PoseDistanceLoss(**LOSS_CFG)
    val = PoseDistanceLoss(mode="val")
    if kind == "td":
        loss = crit(out[0], b["x0"]) + crit(out[1], b["x1"])
        pe, oe = val(out[1], b["x1"])
        rec["out0_s1"], rec["out1_s1"] = out[0].detach().numpy(), out[1].detach().numpy()
    else:
        loss = crit(out, b["obj"])
        pe, oe = val(out, b["obj"])
        rec["out0_s1"] = out.detach().numpy()
    loss.backward()
    rec["loss_s1"], rec["pos_err_s1"], rec["ori_err_s1"] = np.array(loss.item()), np.array(float(pe)), np.array(float(oe))
    gn, gd = [], []
    for name, p in model.named_parameters():
        if p.grad is None:
            continue
        gn.append(name)
        gd.append(digest(p.grad))
        g = p.grad.detach().flatten()
        head = not name.startswith("feature_net") or ".fc." in name
        if g.numel() <= 4096:
            rec["grad::" + name] = g.numpy().copy()
        elif head:
            rec["gsample::" + name] = g[::SEQ_SAMPLE_STRIDE][:SEQ_SAMPLE_MAX].numpy().copy()
    rec["grad_keys_s1"], rec["grad_digest_s1"] = np.array(gn), np.stack(gd)
    np.savez_compressed(os.path.join(OUT, "model_%s_cfg.npz" % kind), **rec)
    print("%s_cfg loss" % kind, rec["loss_s1"], "pos/ori err", rec["pos_err_s1"], rec["ori_err_s1"])


# models/losses.py:68-69 normalises the predicted quaternion without an epsilon, and NaiveObjectStateEstimator ends in a ReLU
# (models/naive.py:343-345): a sample whose four quaternion outputs are all clipped to zero makes the LOSS VALUE NaN, while the
# ReLU's backward (a select on the output sign) keeps every gradient finite and the step is still applied.  Forced here for every
# sample through the last layer's bias / weights (quaternion rows: zero weights, bias -1); recorded: outputs, the NaN loss, every gradient, the parameters after one Adam step.
def run_nan_loss():
    cfg, lead, wseed, dseed = CASES["no"]
    torch.manual_seed(0)
    model = build("no", cfg)
    sd = po.make_state("no", cfg, wseed)
    last = "fc%d.module" % len(cfg["hidden"])
    sd[last + ".weight"][3:7] = 0.0
    sd[last + ".bias"][3:7] = -1.0
    load_values(model, "no", sd)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    b = po.synth_batch(lead, dseed + 1)
    out = model(b["img"], torch.empty(*b["img"].shape), b["x0bar"])
    loss = PoseDistanceLoss(**LOSS_CFG)(out, b["obj"])
    opt.zero_grad()
    loss.backward()
    rec = {"out0_s1": out.detach().numpy(), "loss_s1": np.array(loss.item())}
    assert np.isnan(rec["loss_s1"]) and float(out[:, 3:].abs().max()) == 0.0
    gn, gd = [], []
    for name, p in model.named_parameters():
        if p.grad is None:
            continue
        assert torch.isfinite(p.grad).all(), name
        gn.append(name)
        gd.append(digest(p.grad))
        if p.numel() <= 4096:
            rec["grad::" + name] = p.grad.detach().numpy().copy()
    rec["grad_keys_s1"], rec["grad_digest_s1"] = np.array(gn), np.stack(gd)
    opt.step()
    fin = model.state_dict()
    for k, v in fin.items():
        if v.dtype.is_floating_point:
            assert torch.isfinite(v).all(), k
            if v.numel() <= 4096:
                rec["final::" + k] = v.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "model_no_nanloss.npz"), **rec)
    print("no_nanloss loss", rec["loss_s1"], "max |grad|", max(d[2] for d in gd))


# The configuration every published job of the reference runs (scripts/train_no.sbatch:83, train_tdo.sbatch:83, train_tdo_v2.sbatch:84
# pass --use_pretrained, and feature_extract=True is the constructors' default): util/model_utils.py:110-113,136-137 then FREEZES the
# whole ResNet body (requires_grad False) and only the replaced fc, the aux head and the fusion layers train -- with BatchNorm still in
# train mode (batch statistics, running statistics updated).  The stub's resnet50(pretrained=True) returns its seeded init instead of
# fetching.  Recorded: which parameters have a gradient, outputs / loss, those gradients, BN running statistics after the step.
def run_frozen():
    cfg, lead, wseed, dseed = CASES["no"]
    torch.manual_seed(0)
    model = NaiveObjectStateEstimator("cube", list(cfg["hidden"]), 50, cfg["latent_dim"], True, (9,), False, True, False)
    sd = po.make_state("no", cfg, wseed)
    load_values(model, "no", sd)
    model.train()
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=1e-3)
    b = po.synth_batch(lead, dseed + 1)
    out = model(b["img"], torch.empty(*b["img"].shape), b["x0bar"])
    loss = PoseDistanceLoss(**LOSS_CFG)(out, b["obj"])
    opt.zero_grad()
    loss.backward()
    rec = {"out0_s1": out.detach().numpy(), "loss_s1": np.array(loss.item())}
    rec["trainable"] = np.array([n for n, p in model.named_parameters() if p.requires_grad])
    rec["frozen"] = np.array([n for n, p in model.named_parameters() if not p.requires_grad])
    gn, gd = [], []
    for name, p in model.named_parameters():
        if p.grad is None:   # frozen, or the depth head (trainable but unused without use_depth)
            continue
        assert p.requires_grad, name
        gn.append(name)
        gd.append(digest(p.grad))
        g = p.grad.detach().flatten()
        rec[("grad::" if g.numel() <= 4096 else "gsample::") + name] = (g if g.numel() <= 4096 else g[::SAMPLE_STRIDE][:SAMPLE_MAX]).numpy().copy()
    rec["grad_keys_s1"], rec["grad_digest_s1"] = np.array(gn), np.stack(gd)
    opt.step()
    fin = model.state_dict()
    for k, v in fin.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            if v.numel() <= 256:
                rec["final::" + k] = v.numpy().copy()
    rec["final_frozen_digest"] = np.stack([digest(fin[k]) for k in [str(n) for n in rec["frozen"]]])
    np.savez_compressed(os.path.join(OUT, "model_no_frozen.npz"), **rec)
    print("no_frozen loss", rec["loss_s1"], "trainable", len(rec["trainable"]), "frozen", len(rec["frozen"]))


def run_loss():
    g = torch.Generator().manual_seed(7)
    rec = {}
    pred = torch.randn(3, 5, 7, generator=g)
    pred[0, 0, 3:] *= -1  # make sure some predicted w are negative (penalty branch)
    truth = po._rand_pose((3, 5), g)
    rec["pred"], rec["truth"] = pred.numpy(), truth.numpy()
    for metric in ("l1", "l2", "linf", "combined"):
        for mode in ("position", "pose"):
            for scale, alpha in ((1.0, 1.0), (2.5, 0.5)):
                p = pred.clone().requires_grad_(True)
                l = PoseDistanceLoss(metric, scale, alpha, 1e-4, mode)(p, truth)
                l.backward()
                tag = "%s_%s_%g_%g" % (metric, mode, scale, alpha)
                rec["loss::" + tag] = np.array(l.item())
                rec["grad::" + tag] = p.grad.numpy().copy()
    pe, oe = PoseDistanceLoss(mode="val")(pred, truth)
    rec["val_pos"], rec["val_ori"] = np.array(float(pe)), np.array(float(oe))
    for bad in (dict(distance_metric="l3"), dict(mode="train")):
        try:
            PoseDistanceLoss(**bad)
            raise AssertionError("expected ValueError")
        except ValueError:
            pass
    np.savez_compressed(os.path.join(OUT, "pose_loss.npz"), **rec)
    print("loss fixtures written")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["loss"] + list(CASES) + ["c1", "r101", "r18", "td_s4", "hooks", "hooks18", "nohook", "td_cfg", "tdo_cfg", "tdo_v2_cfg", "nanloss", "frozen"]
    for w in which:
        if w == "loss":
            run_loss()
        elif w == "c1":
            run_c1()
        elif w == "r101":
            run_r101()
        elif w == "r18":
            run_r101(R18, "r18")
        elif w == "td_s4":
            run_td_s4()
        elif w == "hooks":
            run_hooks("hooks", HOOKS)
        elif w == "hooks18":
            run_hooks("hooks18", HOOKS18)
        elif w == "nohook":
            run_hooks("nohook", NOHOOK)
        elif w == "frozen":
            run_frozen()
        elif w == "nanloss":
            run_nan_loss()
        elif w.endswith("_cfg"):
            run_seq_cfg(w[:-4])
        else:
            run_case(w)
