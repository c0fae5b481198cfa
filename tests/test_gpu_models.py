"""GPU parity of the five drop-in model classes (forward, loss, gradients, Adam step, eval / rollout) against
(a) the golden vectors produced by the reference's own classes and (b) the oracle on the same seeded inputs.

fp32 compute path: step-1 outputs within 1e-4 relative (the bar BASELINE.json states), gradients within 2e-3 of
each tensor's max (a 50-layer train-mode-BN network at batch 2-4 amplifies fp32 summation-order noise: the
oracle itself differs from an fp64 run by ~2e-2 on those tensors, see tests/test_oracle_golden.py).
bf16 compute path: outputs within 5e-2 relative, loss within 5e-2.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pose_oracle as po
from rgb_proprioceptive_pose_estimator_amd import models as M
from rgb_proprioceptive_pose_estimator_amd.optim import FusedAdam

from _helpers import CASES, LOSS_CFG, build, load_values

DEV = "cuda"


def rel(a, b):
    a, b = a.detach().float().cpu(), torch.as_tensor(b).float()
    assert torch.isfinite(a).all()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def to_dev(b):
    return {k: (None if v is None else v.to(DEV)) for k, v in b.items()}


def run_step(model, kind, b, crit, val):
    depth = b["depth"] if b["depth"] is not None else None
    out = model(b["img"], depth, b["x0bar"])
    if kind in ("n", "td"):
        loss = crit(out[0], b["x0"]) + crit(out[1], b["x1"])
        pe, oe = val(out[1], b["x1"])
        outs = out
    else:
        loss = crit(out, b["obj"])
        pe, oe = val(out, b["obj"])
        outs = (out,)
    loss.backward()
    return outs, loss, pe, oe


@pytest.mark.parametrize("kind", list(CASES))
def test_model_fp32_matches_reference_and_oracle(kind, golden_dir):
    gold = np.load(os.path.join(golden_dir, "model_%s.npz" % kind))
    cfg, lead, wseed, dseed = CASES[kind]
    sd = po.make_state(kind, cfg, wseed)
    model = build(kind, cfg, torch.float32)
    assert [k for k in model.state_dict().keys()] == list(gold["keys"])
    load_values(model, kind, sd)
    model.cuda()
    crit = M.PoseDistanceLoss(distance_metric="combined", scale_factor=1.0, alpha=0.5, mode="pose")
    val = M.PoseDistanceLoss(mode="val")

    # pristine eval-mode forward + rollout contract
    model.eval()
    model.reset_initial_state(lead[-1])
    b9 = to_dev(po.synth_batch(lead, dseed + 9, with_depth=cfg.get("use_depth", False)))
    with torch.no_grad():
        out = model(b9["img"], b9["depth"], b9["x0bar"])
        outs = out if isinstance(out, tuple) else (out,)
        for i, o in enumerate(outs):
            assert rel(o, gold["pre_eval_out%d" % i]) < 2e-4, "eval out%d" % i
        if "pre_rollout_out" in gold.files:
            model.rollout = True
            model.reset_initial_state(lead[-1])
            d = b9["depth"]
            o1 = model(b9["img"][:1], None if d is None else d[:1], b9["x0bar"][:1])
            o2 = model(b9["img"][1:], None if d is None else d[1:], b9["x0bar"][1:])
            o1 = o1[-1] if isinstance(o1, tuple) else o1
            o2 = o2[-1] if isinstance(o2, tuple) else o2
            assert rel(torch.cat([o1, o2], 0), gold["pre_rollout_out"]) < 2e-4
            model.rollout = False

    # training step 1: against the reference vectors and, tensor by tensor, against the oracle
    model.train()
    model.reset_initial_state(lead[-1])
    opt = FusedAdam(model.parameters(), lr=1e-3)
    b1c = po.synth_batch(lead, dseed + 1, with_depth=cfg.get("use_depth", False))
    ref = po.train_step(kind, cfg, {k: v.clone() for k, v in sd.items()}, b1c, LOSS_CFG, {}, lr=1e-3)
    # fp64 run of the oracle: the yardstick for how much an fp32 implementation may deviate on this network
    sd64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    b64 = {k: (None if v is None else v.double()) for k, v in b1c.items()}
    ref64 = po.train_step(kind, cfg, sd64, b64, LOSS_CFG, {}, lr=1e-3, val_metrics=False)
    opt.zero_grad()
    outs, loss, pe, oe = run_step(model, kind, to_dev(b1c), crit, val)
    for i, o in enumerate(outs):
        assert rel(o, gold["out%d_s1" % i]) < 1e-4, "out%d" % i
    np.testing.assert_allclose(loss.item(), gold["loss_s1"], rtol=1e-4)
    np.testing.assert_allclose(float(pe), gold["pos_err_s1"], rtol=1e-4)
    np.testing.assert_allclose(oe, gold["ori_err_s1"], rtol=1e-4, atol=1e-4)
    # Gradients.  With random weights this 50-layer train-mode-BN network amplifies fp32 rounding ~1e5x: torch's own
    # CPU fp32 gradients differ from an fp64 run by ~2e-2 (median over tensors) and up to ~0.2 (tools/debug_grads.py).
    # So the bar is set by that yardstick: the HIP fp32 path must be as close to fp64 as the CPU fp32 reference is.
    named = dict(model.named_parameters())
    e_gpu, e_cpu, cos = {}, {}, {}
    for name, g64 in ref64["grads"].items():
        g = named[name].grad
        assert g is not None, name
        gc = g.detach().double().cpu()
        e_gpu[name] = ((gc - g64).abs().max() / g64.abs().max().clamp_min(1e-30)).item()
        e_cpu[name] = ((ref["grads"][name].double() - g64).abs().max() / g64.abs().max().clamp_min(1e-30)).item()
        cos[name] = (torch.dot(gc.flatten(), g64.flatten()) / (gc.norm() * g64.norm()).clamp_min(1e-300)).item()
    mg, mc = np.median(list(e_gpu.values())), np.median(list(e_cpu.values()))
    xg, xc = max(e_gpu.values()), max(e_cpu.values())
    assert mg <= 2.0 * mc + 1e-4, "median gradient error %.3g vs CPU-fp32 yardstick %.3g" % (mg, mc)
    assert xg <= 3.0 * xc + 1e-3, "max gradient error %.3g vs CPU-fp32 yardstick %.3g (%s)" % (xg, xc, max(e_gpu, key=e_gpu.get))
    assert min(cos.values()) > 0.97, "gradient direction off: %s" % (min(cos, key=cos.get),)
    for name in e_gpu:  # the layers after the trunk are well conditioned: tight
        if "feature_net" not in name and "aux_nets" not in name and "depth_nets" not in name:
            assert e_gpu[name] < 1e-4 + 3 * e_cpu[name], (name, e_gpu[name], e_cpu[name])
    for name, p in named.items():  # parameters the reference leaves without a gradient stay untouched by Adam
        if name not in ref["grads"]:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
    # Adam step 1: the UPDATE of every parameter against the oracle's.  The first Adam step is -lr g / (|g| + eps) = -lr sign(g) wherever
    # |g| >> eps, so two implementations agree on an element to ~1e-6 or differ by 2 lr (a gradient element at the fp32 noise floor of
    # this net -- see the gradient bars above -- with the other sign); an optimizer that did nothing scores 0 on every tensor here
    # (round 3's check, allclose with atol 2.1e-3 > lr, could not tell).  Head / fc / aux tensors are well conditioned: every live
    # element must agree.  Trunk tensors: measured per-tensor agreement 0.961-0.977 at worst, 0.990-0.995 median over the five families
    # (tools/adam_parity_probe.py, profiles/r04_adam_parity.txt); elements whose reference gradient is above 5 % of the tensor's
    # largest may disagree in fewer than 1e-4 of all cases (measured 31-169 elements of 23.5 M).
    lr = 1e-3
    sd_after = {k: v.clone() for k, v in sd.items()}
    ost = {}
    r1 = po.train_step(kind, cfg, sd_after, b1c, LOSS_CFG, ost, lr=lr)
    opt.step()
    msd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    trunk_frac, strong_bad, strong_n = [], 0, 0
    for k, g in r1["grads"].items():
        if k not in msd:   # (td's unregistered aux / depth heads: no optimizer sees them, models/time_sensitive.py:102-115)
            continue
        u_ref = sd_after[k].float() - sd[k].float()
        u_gpu = msd[k] - sd[k].float()
        live = g.abs() > 1e-7 * g.abs().max().clamp_min(1e-30)
        if not live.any():
            continue
        ok = (u_gpu - u_ref).abs() <= 1e-2 * lr
        frac = (ok & live).sum().item() / live.sum().item()
        strong = g.abs() > 0.05 * g.abs().max()
        strong_bad += int((~ok & strong).sum())
        strong_n += int(strong.sum())
        if "feature_net" in k and ".fc." not in k:
            trunk_frac.append(frac)
        else:
            assert frac >= 0.9999, "Adam step 1, %s: %.5f of the live elements moved as the reference's" % (k, frac)
        assert float(u_gpu.abs().max()) <= lr * 1.001, k
    assert min(trunk_frac) >= 0.95 and np.median(trunk_frac) >= 0.985, "Adam step 1, trunk tensors: agreement min %.4f median %.4f" % (min(trunk_frac), np.median(trunk_frac))
    assert strong_bad <= 1e-4 * strong_n, "Adam step 1: %d of %d strong-gradient elements moved against the reference" % (strong_bad, strong_n)
    for k in msd:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert rel(msd[k], sd_after[k]) < 1e-4, k
        if k.endswith("num_batches_tracked"):
            assert int(msd[k]) == 1
    # Step 2 on the reference's second batch, then the reference's own post-2-step vectors (`final::` / `final_digest` of the golden
    # file).  From step 2 on the two runs see slightly different trunks (the sign flips above), so head gradients differ at the 1e-2
    # level and an element may be one full step apart: every head / fc / aux element within 2.1 lr, half of them within 5e-5 (measured:
    # max 2.4e-4 .. 1.6e-3); every tensor's l2 norm within 1 % of the reference's digest (measured <= 5.1e-3, the deep running_var's).
    b2c = po.synth_batch(lead, dseed + 2, with_depth=cfg.get("use_depth", False))
    model.reset_initial_state(lead[-1])
    opt.zero_grad()
    run_step(model, kind, to_dev(b2c), crit, val)
    opt.step()
    msd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    for k in gold.files:
        if not k.startswith("final::") or k.endswith("num_batches_tracked"):
            continue
        name = k[7:]
        if "feature_net" in name and ".fc." not in name:
            continue
        d = (msd[name] - torch.from_numpy(gold[k]).float()).abs()
        assert float(d.max()) <= 2.1 * lr and float(d.median()) <= 5e-5, "after 2 Adam steps, %s: max %.3e median %.3e off the reference" % (name, d.max(), d.median())
    for name, ref_d in zip(gold["keys"], gold["final_digest"]):
        t = msd[str(name)].double()
        if ref_d[1] > 0:
            assert abs(t.norm().item() - ref_d[1]) <= 1e-2 * ref_d[1], "after 2 Adam steps, l2 norm of %s: %.6g vs %.6g" % (name, t.norm().item(), ref_d[1])


@pytest.mark.parametrize("kind", ["no", "tdo"])
def test_model_bf16_tracks_fp32_reference(kind, golden_dir):
    gold = np.load(os.path.join(golden_dir, "model_%s.npz" % kind))
    cfg, lead, wseed, dseed = CASES[kind]
    sd = po.make_state(kind, cfg, wseed)
    model = build(kind, cfg, torch.bfloat16)
    load_values(model, kind, sd)
    model.cuda().train()
    model.reset_initial_state(lead[-1])
    crit = M.PoseDistanceLoss(distance_metric="combined", scale_factor=1.0, alpha=0.5, mode="pose")
    val = M.PoseDistanceLoss(mode="val")
    b1 = to_dev(po.synth_batch(lead, dseed + 1, with_depth=cfg.get("use_depth", False)))
    outs, loss, pe, oe = run_step(model, kind, b1, crit, val)
    assert rel(outs[0], gold["out0_s1"]) < 5e-2
    np.testing.assert_allclose(loss.item(), gold["loss_s1"], rtol=5e-2)
    for name, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), name


@pytest.mark.parametrize("tag,dtype", [("r101", torch.float32), ("r18", torch.float32), ("r18", torch.bfloat16), ("r18", torch.float16)])
def test_resnet101_trunk_matches_reference(golden_dir, tag, dtype):
    """import_resnet's other members on the same native engine -- the deeper bottleneck option ([3,4,23,3] blocks) and the BasicBlock
    network resnet18 ([2,2,2,2] blocks of two 3x3 convs, fc input 512; util/model_utils.py:130-136): state_dict keys, pristine eval
    output and train step 1 against the reference's vectors (fp32 path, 1e-4 bar; 16-bit: the output bars of the other model tests),
    gradients against the oracle."""
    import _helpers_cases as hc
    gold = np.load(os.path.join(golden_dir, "model_no_%s.npz" % tag))
    cfg, lead, wseed, dseed = getattr(hc, tag.upper())
    f32 = dtype == torch.float32
    bar = 1e-4 if f32 else (5e-2 if dtype == torch.bfloat16 else 1.5e-2)
    sd = po.make_state("no", cfg, wseed)
    model = build("no", cfg, dtype)
    assert list(model.state_dict().keys()) == list(gold["keys"])
    assert model.trunk.fc.in_features == (512 if tag == "r18" else 2048)
    load_values(model, "no", sd)
    model.cuda().eval()
    b9 = to_dev(po.synth_batch(lead, dseed + 9))
    with torch.no_grad():
        assert rel(model(b9["img"], None, b9["x0bar"]), gold["pre_eval_out0"]) < (2e-4 if f32 else bar)
    model.train()
    crit = M.PoseDistanceLoss(distance_metric="combined", scale_factor=1.0, alpha=0.5, mode="pose")
    b1c = po.synth_batch(lead, dseed + 1)
    b1 = to_dev(b1c)
    out = model(b1["img"], None, b1["x0bar"])
    loss = crit(out, b1["obj"])
    loss.backward()
    assert rel(out, gold["out0_s1"]) < bar
    np.testing.assert_allclose(loss.item(), gold["loss_s1"], rtol=bar)
    named = dict(model.named_parameters())
    for name, p in named.items():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
    if not f32:
        return
    ref = po.train_step("no", cfg, {k: v.clone() for k, v in sd.items()}, b1c, LOSS_CFG, {}, lr=1e-3, val_metrics=False)
    cos = []
    for name, g in ref["grads"].items():
        if float(g.abs().max()) == 0.0:
            continue
        a, b = named[name].grad.detach().cpu().double().flatten(), g.double().flatten()
        cos.append((torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-300)).item())
    # 101 layers of train-mode BN at batch 2 amplify fp32 rounding further than ResNet-50 does (see the module docstring); the 18-layer
    # network is held to a tighter bar
    if tag == "r18":
        assert np.median(cos) > 0.999 and min(cos) > 0.95, (np.median(cos), min(cos))
        for name, refd in zip(gold["grad_keys_s1"], gold["grad_digest_s1"]):   # every gradient's norm as in the reference
            np.testing.assert_allclose(named[str(name)].grad.double().norm().item(), refd[1], rtol=5e-2, atol=1e-7, err_msg=str(name))
    else:
        assert np.median(cos) > 0.95 and min(cos) > 0.5, (np.median(cos), min(cos))


def test_resnet152_trunk_runs_and_matches_the_oracle():
    """The deepest bottleneck option of import_resnet (util/model_utils.py:130-136; [3,8,36,3] blocks = 154 packed convs -- more than
    the 127-entry weight-packing table took in round 2, so neither a train nor an eval forward could run): one eval forward and one
    train step on two images, against the oracle (itself pinned on the 50- and 101-layer trunks)."""
    cfg = dict(latent_dim=64, hidden=[32], use_depth=False, no_proprioception=False, depth=152)
    sd = po.make_state("no", cfg, 43)
    model = build("no", cfg, torch.float32)
    assert list(model.state_dict().keys()) == [k for k, _ in po.model_keys("no", cfg)]
    load_values(model, "no", sd)
    model.cuda().eval()
    b9c = po.synth_batch((2,), 409)
    b9 = to_dev(b9c)
    with torch.no_grad():
        want = po.model_forward("no", cfg, sd, b9c["img"], None, b9c["x0bar"], train=False)
        assert rel(model(b9["img"], None, b9["x0bar"]), want) < 2e-4
    model.train()
    crit = M.PoseDistanceLoss(distance_metric="combined", scale_factor=1.0, alpha=0.5, mode="pose")
    b1c = po.synth_batch((2,), 401)
    b1 = to_dev(b1c)
    out = model(b1["img"], None, b1["x0bar"])
    loss = crit(out, b1["obj"])
    loss.backward()
    ref = po.train_step("no", cfg, {k: v.clone() for k, v in sd.items()}, b1c, LOSS_CFG, {}, lr=1e-3, val_metrics=False)
    assert rel(out, ref["outputs"]) < 1e-4
    np.testing.assert_allclose(loss.item(), ref["loss"].item(), rtol=1e-4)
    named = dict(model.named_parameters())
    cos = []
    for name, g in ref["grads"].items():
        assert torch.isfinite(named[name].grad).all(), name
        if float(g.abs().max()) == 0.0:
            continue
        a, b = named[name].grad.detach().cpu().double().flatten(), g.double().flatten()
        cos.append((torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-300)).item())
    # 152 layers of train-mode BN at batch 2: rounding noise dominates the early layers' gradients in ANY fp32 implementation
    assert np.median(cos) > 0.9, (np.median(cos), min(cos))


@pytest.mark.parametrize("dtype,bar", [(torch.float32, 1e-4), (torch.bfloat16, 5e-2), (torch.float16, 1.5e-2)], ids=["f32", "bf16", "f16"])
def test_td_four_frame_sequences_match_reference(dtype, bar, golden_dir):
    """BASELINE configs[2]: TemporallyDependentStateEstimator on sequences of FOUR frames (lead dims (4, 2)) against the reference's
    vectors: pristine eval outputs, a 4-frame rollout with the LSTM state carried on the device, step-1 outputs / loss / val metrics."""
    from _helpers_cases import TD_S4
    gold = np.load(os.path.join(golden_dir, "model_td_s4.npz"))
    cfg, lead, wseed, dseed = TD_S4
    sd = po.make_state("td", cfg, wseed)
    model = M.TemporallyDependentStateEstimator(cfg["hidden"], cfg["hidden"], 50, cfg["latent_dim"], 4, 0.1, False, (9,), False, False,
                                                compute_dtype=dtype)
    assert model.sequence_length == 4 and list(model.state_dict().keys()) == list(gold["keys"])
    load_values(model, "td", sd)
    model.cuda().eval()
    model.reset_initial_state(lead[-1])
    ebar = 2e-4 if dtype == torch.float32 else bar
    b9 = to_dev(po.synth_batch(lead, dseed + 9))
    with torch.no_grad():
        pre, post = model(b9["img"], None, b9["x0bar"])
        assert rel(pre, gold["pre_eval_out0"]) < ebar and rel(post, gold["pre_eval_out1"]) < ebar
        model.rollout = True
        model.reset_initial_state(lead[-1])
        frames = [model(b9["img"][t:t + 1], None, b9["x0bar"][t:t + 1])[-1].clone() for t in range(lead[0])]
        assert rel(torch.cat(frames, 0), gold["pre_rollout_out"]) < ebar
        model.rollout = False
    model.train()
    model.reset_initial_state(lead[-1])
    crit = M.PoseDistanceLoss(distance_metric="combined", scale_factor=1.0, alpha=0.5, mode="pose")
    val = M.PoseDistanceLoss(mode="val")
    outs, loss, pe, oe = run_step(model, "td", to_dev(po.synth_batch(lead, dseed + 1)), crit, val)
    assert rel(outs[0], gold["out0_s1"]) < bar and rel(outs[1], gold["out1_s1"]) < bar
    np.testing.assert_allclose(loss.item(), gold["loss_s1"], rtol=bar)
    np.testing.assert_allclose(float(pe), gold["pos_err_s1"], rtol=max(bar, 1e-4))
    if dtype != torch.float16:      # (fp16 gradients carry the loss scale until the optimizer unscales and tests them)
        for name, p in model.named_parameters():
            if p.grad is not None:
                assert torch.isfinite(p.grad).all(), name


@pytest.mark.parametrize("tag,dtype,bar", [("hooks", torch.float32, 1e-4), ("hooks", torch.bfloat16, 5e-2), ("hooks", torch.float16, 1.5e-2),
                                           ("nohook", torch.float32, 1e-4), ("nohook", torch.bfloat16, 5e-2),
                                           ("hooks18", torch.float32, 1e-4), ("hooks18", torch.bfloat16, 5e-2), ("hooks18", torch.float16, 1.5e-2)])
def test_other_hook_layers_match_reference(tag, dtype, bar, golden_dir):
    """feature_layer_nums other than the scripts' (9,) (models/naive.py:196-240): hooks on conv1, bn1 and layer1..3 -- given out of
    order, depth heads on -- and None, against the reference's vectors: state_dict keys (aux_nets in hook FIRING order), pristine
    eval output, step-1 output / loss; fp32: every gradient's norm and the head gradients element-wise."""
    import _helpers_cases as hc
    cfg, lead, wseed, dseed = {"hooks": hc.HOOKS, "hooks18": hc.HOOKS18, "nohook": hc.NOHOOK}[tag]   # (hooks18: the same hooks on the BasicBlock trunk)
    gold = np.load(os.path.join(golden_dir, "model_no_%s.npz" % tag))
    sd = po.make_state("no", cfg, wseed)
    model = M.NaiveObjectStateEstimator("cube", list(cfg["hidden"]), cfg.get("depth", 50), cfg["latent_dim"], False, cfg["hooks"], cfg["use_depth"], False, False,
                                        compute_dtype=dtype)
    assert list(model.state_dict().keys()) == list(gold["keys"])
    load_values(model, "no", sd)
    model.cuda().eval()
    b9 = to_dev(po.synth_batch(lead, dseed + 9, with_depth=cfg["use_depth"]))
    with torch.no_grad():
        assert rel(model(b9["img"], b9["depth"], b9["x0bar"]), gold["pre_eval_out0"]) < (2e-4 if dtype == torch.float32 else bar)
    model.train()
    crit = M.PoseDistanceLoss(distance_metric="combined", scale_factor=1.0, alpha=0.5, mode="pose")
    b1 = to_dev(po.synth_batch(lead, dseed + 1, with_depth=cfg["use_depth"]))
    out = model(b1["img"], b1["depth"], b1["x0bar"])
    loss = crit(out, b1["obj"])
    loss.backward()
    assert rel(out, gold["out0_s1"]) < bar
    np.testing.assert_allclose(loss.item(), gold["loss_s1"], rtol=bar)
    if dtype != torch.float32:
        return
    named = dict(model.named_parameters())
    # every hooked feature's gradient reached the trunk: all gradient norms as in the reference (noise-amplifying early layers: 5 %)
    for name, ref in zip(gold["grad_keys_s1"], gold["grad_digest_s1"]):
        g = named[str(name)].grad
        assert g is not None, name
        np.testing.assert_allclose(g.double().norm().item(), ref[1], rtol=5e-2, atol=1e-7, err_msg=str(name))
    for name in gold.files:
        if name.startswith("grad::"):
            g = named[name[6:]].grad.detach().cpu().numpy()
            scale = max(1e-6, float(np.abs(gold[name]).max()))
            assert np.abs(g - gold[name]).max() / scale < (3e-2 if "conv1" in name else 2e-3), name


def test_hook_on_layer4_is_refused_like_the_reference():
    """the reference sizes fc0 with 7*7//4 = 12 aux columns for layer4 while its aux head yields 9: its forward raises; so does the constructor here"""
    with pytest.raises(ValueError):
        M.NaiveObjectStateEstimator("cube", [32], 50, 64, False, (4,), False, False, False)
    with pytest.raises(ValueError):
        M.NaiveObjectStateEstimator("cube", [32], 50, 64, False, (5,), False, False, False)


def test_sequence_model_with_extra_hooks_matches_oracle():
    """The hook wiring is shared by all model classes: TDO (LSTM over 2x2 frames) with hooks on layer1 and bn1 against the oracle
    (pinned to the reference for these hooks by tests/test_oracle_golden.py), fp32 path."""
    cfg = dict(latent_dim=64, hidden=32, use_depth=True, no_proprioception=False, hooks=(1, 9))
    lead, wseed, dseed = (2, 2), 71, 701
    sd = po.make_state("tdo", cfg, wseed)
    model = M.TemporallyDependentObjectStateEstimator("hammer", cfg["hidden"], 50, cfg["latent_dim"], 2, 0.1, False, cfg["hooks"], True, False, False,
                                                      compute_dtype=torch.float32)
    assert list(model.state_dict().keys()) == [k for k, _ in po.model_keys("tdo", cfg)]
    load_values(model, "tdo", sd)
    model.cuda().train()
    model.reset_initial_state(lead[-1])
    b1c = po.synth_batch(lead, dseed + 1, with_depth=True)
    ref = po.train_step("tdo", cfg, {k: v.clone() for k, v in sd.items()}, b1c, LOSS_CFG, {}, lr=1e-3, val_metrics=False)
    crit = M.PoseDistanceLoss(distance_metric="combined", scale_factor=1.0, alpha=0.5, mode="pose")
    b1 = to_dev(b1c)
    out = model(b1["img"], b1["depth"], b1["x0bar"])
    loss = crit(out, b1["obj"])
    loss.backward()
    assert rel(out, ref["outputs"]) < 1e-4
    np.testing.assert_allclose(loss.item(), ref["loss"].item(), rtol=1e-4)
    named = dict(model.named_parameters())
    for name in ("aux_nets.0.module.0.weight", "aux_nets.1.module.0.weight", "depth_nets.1.module.3.weight", "rnn.module.weight_ih_l0"):
        g, r = named[name].grad.detach().cpu(), ref["grads"][name]
        assert ((g - r).abs().max() / r.abs().max().clamp_min(1e-12)).item() < 2e-3, name


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_nan_loss_value_keeps_gradients_finite_like_the_reference(dtype, golden_dir):
    """The reference normalises the predicted quaternion without an epsilon (models/losses.py:68-69) behind the final ReLU of
    NaiveObjectStateEstimator (models/naive.py:343-345): all-zero quaternion outputs give a NaN loss VALUE, finite gradients (the
    ReLU backward is a select) and an applied optimizer step.  Pinned by the reference's own vectors (model_no_nanloss.npz): the HIP
    loss kernel and heads must do the same -- no epsilon, no skipped step, no NaN leaking into a gradient."""
    gold = np.load(os.path.join(golden_dir, "model_no_nanloss.npz"))
    cfg, lead, wseed, dseed = CASES["no"]
    sd = po.make_state("no", cfg, wseed)
    last = "fc%d.module" % len(cfg["hidden"])
    sd[last + ".weight"][3:7] = 0.0
    sd[last + ".bias"][3:7] = -1.0
    model = build("no", cfg, dtype)
    load_values(model, "no", sd)
    model.cuda().train()
    opt = FusedAdam(model.parameters(), lr=1e-3)
    crit = M.PoseDistanceLoss(distance_metric="combined", scale_factor=1.0, alpha=0.5, mode="pose")
    b = to_dev(po.synth_batch(lead, dseed + 1))
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    out = model(b["img"], None, b["x0bar"])
    loss = crit(out, b["obj"])
    opt.zero_grad()
    loss.backward()
    assert np.isnan(gold["loss_s1"]) and torch.isnan(loss).item()
    assert float(out[:, 3:].abs().max()) == 0.0
    assert rel(out, gold["out0_s1"]) < (1e-4 if dtype == torch.float32 else 5e-2)
    named = dict(model.named_parameters())
    for name, ref in zip(gold["grad_keys_s1"], gold["grad_digest_s1"]):
        g = named[str(name)].grad
        assert g is not None and torch.isfinite(g).all(), name
        if dtype == torch.float32 and not str(name).startswith("feature_net") and "grad::" + str(name) in gold.files:
            want = torch.from_numpy(gold["grad::" + str(name)])
            assert ((g.cpu() - want).abs().max() <= 2e-3 * want.abs().max().clamp_min(1e-6)).item(), name
    opt.step()
    torch.cuda.synchronize()
    moved = 0
    for k, p in model.named_parameters():
        assert torch.isfinite(p).all(), k
        if p.grad is not None and float(p.grad.abs().max()) > 0.0:
            moved += int((p.detach() != before[k]).any().item())
    assert moved > 100   # the step was applied (Adam moves every element with a non-zero gradient by ~lr)
    if dtype == torch.float32:
        msd = model.state_dict()
        for k in gold.files:   # head parameters one Adam step later (trunk tensors: gradient signs near zero are rounding noise)
            if k.startswith("final::fc") or k.startswith("final::aux_nets"):
                np.testing.assert_allclose(msd[k[7:]].cpu().numpy(), gold[k], rtol=0, atol=2.5e-4, err_msg=k)


@pytest.mark.parametrize("dtype,bar", [(torch.float32, 1e-4), (torch.bfloat16, 5e-2)], ids=["f32", "bf16"])
def test_frozen_trunk_matches_reference_and_skips_the_body_backward(dtype, bar, golden_dir):
    """feature_extract=True, use_pretrained=True -- the constructors' defaults and the setting of every published job
    (scripts/train_no.sbatch:83, train_tdo.sbatch:83, train_tdo_v2.sbatch:84): util/model_utils.py:110-113,137 freezes the ResNet
    body, BatchNorm stays in train mode.  Against the reference's own run (model_no_frozen.npz): the same parameters are frozen,
    outputs / loss agree, the frozen ones have grad None, the trainable ones the reference's gradients, BN running statistics
    move; and the engine runs the fc-only backward (no data-gradient / weight-gradient launch for the body)."""
    import warnings
    from rgb_proprioceptive_pose_estimator_amd import ops
    gold = np.load(os.path.join(golden_dir, "model_no_frozen.npz"))
    cfg, lead, wseed, dseed = CASES["no"]
    sd = po.make_state("no", cfg, wseed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")            # (no local ImageNet checkpoint: the seeded values are loaded below)
        model = M.NaiveObjectStateEstimator("cube", list(cfg["hidden"]), 50, cfg["latent_dim"], True, (9,), False, True, False, compute_dtype=dtype)
    load_values(model, "no", sd)
    assert [n for n, p in model.named_parameters() if not p.requires_grad] == [str(k) for k in gold["frozen"]]
    assert [n for n, p in model.named_parameters() if p.requires_grad] == [str(k) for k in gold["trainable"]]
    model.cuda().train()
    opt = FusedAdam(model.parameters(), lr=1e-3)
    crit = M.PoseDistanceLoss(distance_metric="combined", scale_factor=1.0, alpha=0.5, mode="pose")
    b = to_dev(po.synth_batch(lead, dseed + 1))
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    out = model(b["img"], None, b["x0bar"])
    loss = crit(out, b["obj"])
    opt.zero_grad()
    loss.backward()
    assert model.trunk.body_frozen()
    assert rel(out, gold["out0_s1"]) < bar
    np.testing.assert_allclose(loss.item(), gold["loss_s1"], rtol=bar)
    named = dict(model.named_parameters())
    for k in gold["frozen"]:
        assert named[str(k)].grad is None, k
    for name, dig in zip(gold["grad_keys_s1"], gold["grad_digest_s1"]):
        name = str(name)
        g = named[name].grad
        assert g is not None and torch.isfinite(g).all(), name
        want = torch.from_numpy(gold["grad::" + name] if "grad::" + name in gold.files else gold["gsample::" + name])
        got = g.detach().float().cpu().flatten()
        got = got if "grad::" + name in gold.files else got[::997][:4096]
        tol = 2e-3 if dtype == torch.float32 else 0.25
        assert ((got - want).norm() / want.norm().clamp_min(1e-12)).item() < tol, name
    opt.step()
    torch.cuda.synchronize()
    after = model.state_dict()
    for k in gold["frozen"]:                         # frozen parameters do not move
        assert torch.equal(after[str(k)], before[str(k)]), k
    assert not torch.equal(after["feature_net.module.fc.weight"], before["feature_net.module.fc.weight"])
    if dtype == torch.float32:
        for k in gold.files:                         # BatchNorm ran in train mode: running statistics as the reference's
            if k.startswith("final::"):
                assert rel(after[k[7:]], gold[k]) < 1e-4, k
