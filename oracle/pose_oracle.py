"""TEST INFRASTRUCTURE ONLY -- CPU restatement (fp32 torch-CPU ops + numpy) of the
reference's pose-regression train step.  Nothing in the product imports this file;
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may.

What is restated (citations are into /root/reference):

* ResNet-50 v1.5 body           -- third-party torchvision, constructed at
                                   util/model_utils.py:136-141 (architecture restated
                                   from the public paper/torchvision layout; see
                                   oracle/ref_stubs.py)
* bn1 forward hook quirk        -- models/naive.py:211,282-283 (hook output is mutated
                                   by the following in-place ReLU, so the aux head sees
                                   relu(bn1(conv1(x))))
* aux / depth heads             -- models/naive.py:223-240,318-333
* NaiveEndEffectorStateEstimator.forward  -- models/naive.py:68-112
* NaiveObjectStateEstimator.forward       -- models/naive.py:298-352
* TemporallyDependentStateEstimator.forward         -- models/time_sensitive.py:165-254
* TemporallyDependentObjectStateEstimator.forward   -- models/time_sensitive.py:453-517
* TemporallyDependentObjectStateEstimatorV2.forward -- models/time_sensitive.py:714-786
* PoseDistanceLoss.forward      -- models/losses.py:47-128
* Adam                          -- scripts/train_model.py:228 (torch.optim.Adam defaults)
* train() step body             -- util/learn_utils.py:152-179

Pinning: tests/test_oracle_golden.py checks every function here against
tests/golden/*.npz, which oracle/gen_golden.py produced in the build container by
running the reference's OWN classes (imported from /root/reference through the
stand-ins in oracle/ref_stubs.py).  The reference has no tests or golden vectors
of its own, and the val-mode angle (third-party robosuite)
rests on a restatement: for that one quantity parity is unpinned.
"""
import math
import zlib

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
STAGES = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))  # planes, blocks, stride  (ResNet-50)
BLOCKS = {18: (2, 2, 2, 2), 34: (3, 4, 6, 3), 50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}
BASIC = (18, 34)   # torchvision BasicBlock members (two 3x3 convs, expansion 1); util/model_utils.py:130-136 reaches 18 only ("32" is no torchvision model)


def stages(depth=50):
    return tuple((pl, n, st) for (pl, _, st), n in zip(STAGES, BLOCKS[depth]))

MODEL_KINDS = ("n", "no", "td", "tdo", "tdo_v2")


# ----------------------------------------------------------------------------
# key tables (state_dict layout of the reference classes)
# ----------------------------------------------------------------------------
def resnet_keys(latent_dim, depth=50):
    """Ordered (key, shape) list of a torchvision ResNet (BasicBlock 18 / 34, bottleneck 50 / 101 / 152) with fc -> latent_dim."""
    out = []

    def conv(name, cout, cin, k):
        out.append((name + ".weight", (cout, cin, k, k)))

    def bn(name, c):
        out.append((name + ".weight", (c,)))
        out.append((name + ".bias", (c,)))
        out.append((name + ".running_mean", (c,)))
        out.append((name + ".running_var", (c,)))
        out.append((name + ".num_batches_tracked", ()))

    conv("conv1", 64, 3, 7)
    bn("bn1", 64)
    inpl = 64
    if depth in BASIC:
        for li, (planes, nblk, stride) in enumerate(stages(depth), start=1):
            for b in range(nblk):
                p = "layer%d.%d" % (li, b)
                conv(p + ".conv1", planes, inpl, 3)
                bn(p + ".bn1", planes)
                conv(p + ".conv2", planes, planes, 3)
                bn(p + ".bn2", planes)
                if b == 0 and (stride != 1 or inpl != planes):   # (layer1.0 keeps the identity shortcut)
                    conv(p + ".downsample.0", planes, inpl, 1)
                    bn(p + ".downsample.1", planes)
                inpl = planes
        out.append(("fc.weight", (latent_dim, 512)))
        out.append(("fc.bias", (latent_dim,)))
        return out
    for li, (planes, nblk, stride) in enumerate(stages(depth), start=1):
        for b in range(nblk):
            p = "layer%d.%d" % (li, b)
            conv(p + ".conv1", planes, inpl, 1)
            bn(p + ".bn1", planes)
            conv(p + ".conv2", planes, planes, 3)
            bn(p + ".bn2", planes)
            conv(p + ".conv3", planes * 4, planes, 1)
            bn(p + ".bn3", planes * 4)
            if b == 0:
                conv(p + ".downsample.0", planes * 4, inpl, 1)
                bn(p + ".downsample.1", planes * 4)
            inpl = planes * 4
    out.append(("fc.weight", (latent_dim, 2048)))
    out.append(("fc.bias", (latent_dim,)))
    return out


def _lstm_keys(prefix, inp, hid):
    return [
        (prefix + "weight_ih_l0", (4 * hid, inp)),
        (prefix + "weight_hh_l0", (4 * hid, hid)),
        (prefix + "bias_ih_l0", (4 * hid,)),
        (prefix + "bias_hh_l0", (4 * hid,)),
    ]


# Hooked feature maps (models/naive.py:196-240): layer number -> (channels, spatial size at 224x224), in the order the forward
# hooks fire (= the order of aux_nets / depth_nets, whatever the order of the feature_layer_nums tuple).  Layer 4 is left out: the
# reference sizes its fc input with H*W//4 = 12 columns for the 7x7 map while MaxPool2d(2) yields 3x3 = 9 (its forward raises).
HOOK_ORDER = (0, 9, 1, 2, 3)
HOOK_SHAPE = {0: (64, 112), 9: (64, 112), 1: (256, 56), 2: (512, 28), 3: (1024, 14)}


def hooks_of(cfg):
    """cfg['hooks']: the feature_layer_nums tuple (default (9,), None = no early features) -> layers in firing order"""
    h = cfg.get("hooks", (9,))
    if h is None:
        return []
    h = list(h)
    assert len(set(h)) == len(h) and all(x in HOOK_SHAPE for x in h), h
    return [x for x in HOOK_ORDER if x in h]


def hook_channels(layer, cfg):
    """channels of a hooked map: the layer outputs of a BasicBlock trunk have planes, not 4 x planes, channels"""
    c = HOOK_SHAPE[layer][0]
    return c // 4 if (layer in (1, 2, 3) and cfg.get("depth", 50) in BASIC) else c


def hook_pools(layer):
    """number of AvgPool2d(2) steps of the depth head (models/naive.py:235-236)"""
    c, hw = HOOK_SHAPE[layer]
    return int(math.log(224 ** 2 / (hw * hw // 4), 4))


def aux_dim(cfg):
    return sum(HOOK_SHAPE[h][1] ** 2 // 4 for h in hooks_of(cfg))


def model_keys(kind, cfg):
    """Ordered (key, shape) list of the reference model's state_dict.

    cfg: dict(latent_dim, hidden (list for n/no, int otherwise), proprio_hidden,
              no_proprioception).  hooks: the feature_layer_nums tuple (default (9,): aux dim 3136).
    For kind 'td' the aux/depth heads are NOT in the state_dict (plain Python
    lists, models/time_sensitive.py:102-115); they are listed under the
    pseudo-prefix '~' so callers can still carry their values.
    """
    L = cfg["latent_dim"]
    aux = aux_dim(cfg)
    hooks = hooks_of(cfg)
    keys = []
    if kind in ("n", "td"):
        fpre = "feature_net."
    else:
        fpre = "feature_net.module."
    keys += [(fpre + k, s) for k, s in resnet_keys(L, cfg.get("depth", 50))]
    if kind in ("no", "tdo", "tdo_v2"):   # nn.ModuleList(aux_nets) is registered before nn.ModuleList(depth_nets)
        for i, h in enumerate(hooks):
            keys += [("aux_nets.%d.module.0.weight" % i, (1, hook_channels(h, cfg), 1, 1)), ("aux_nets.%d.module.0.bias" % i, (1,))]
        for i, h in enumerate(hooks):
            keys += [("depth_nets.%d.module.%d.weight" % (i, hook_pools(h)), (1,)), ("depth_nets.%d.module.%d.bias" % (i, hook_pools(h)), (1,))]
    if kind == "td":
        for i, h in enumerate(hooks):
            keys += [("~aux_nets.%d.0.weight" % i, (1, hook_channels(h, cfg), 1, 1)), ("~aux_nets.%d.0.bias" % i, (1,))]
        for i, h in enumerate(hooks):
            keys += [("~depth_nets.%d.%d.weight" % (i, hook_pools(h)), (1,)), ("~depth_nets.%d.%d.bias" % (i, hook_pools(h)), (1,))]
    if kind == "n":
        pre = [L] + list(cfg["hidden"]) + [7]
        for i in range(len(pre) - 1):
            keys += [("pre_fc%d.weight" % i, (pre[i + 1], pre[i])), ("pre_fc%d.bias" % i, (pre[i + 1],))]
        post = [L + 7] + list(cfg["hidden"]) + [7]
        for i in range(len(post) - 1):
            keys += [("post_fc%d.weight" % i, (post[i + 1], post[i])), ("post_fc%d.bias" % i, (post[i + 1],))]
    elif kind == "no":
        inp = L + aux + (0 if cfg.get("no_proprioception") else 7)
        dims = [inp] + list(cfg["hidden"]) + [7]
        for i in range(len(dims) - 1):
            keys += [("fc%d.module.weight" % i, (dims[i + 1], dims[i])), ("fc%d.module.bias" % i, (dims[i + 1],))]
    elif kind == "td":
        H = cfg["hidden"]
        keys += _lstm_keys("pre_measurement_rnn.", L + aux, H)
        keys += [("pre_measurement_fc.weight", (7, H)), ("pre_measurement_fc.bias", (7,))]
        keys += _lstm_keys("post_measurement_rnn.", L + aux + 7, H)
        keys += [("post_measurement_fc.weight", (7, H)), ("post_measurement_fc.bias", (7,))]
    elif kind == "tdo":
        H = cfg["hidden"]
        inp = L + aux + (0 if cfg.get("no_proprioception") else 7)
        keys += _lstm_keys("rnn.module.", inp, H)
        keys += [
            ("fc.module.0.weight", (H // 4, H)),
            ("fc.module.0.bias", (H // 4,)),
            ("fc.module.1.weight", (7, H // 4)),
            ("fc.module.1.bias", (7,)),
        ]
    elif kind == "tdo_v2":
        H, P = cfg["hidden"], cfg["proprio_hidden"]
        keys += _lstm_keys("img_rnn.module.", L + aux, H)
        keys += _lstm_keys("proprio_rnn.module.", 7, P)
        keys += [
            ("fc.module.0.weight", ((H + P) // 4, H + P)),
            ("fc.module.0.bias", ((H + P) // 4,)),
            ("fc.module.1.weight", (7, (H + P) // 4)),
            ("fc.module.1.bias", (7,)),
        ]
    else:
        raise ValueError(kind)
    return keys


# ----------------------------------------------------------------------------
# deterministic, construction-order-independent parameter values
# ----------------------------------------------------------------------------
def value_for_key(key, shape, seed):
    """Deterministic tensor for a state_dict entry, seeded by (seed, crc32(key)).

    Scales follow the reference's initialisers (kaiming fan_out conv, 1/sqrt(fan_in)
    Linear/LSTM) but BN affine/running stats are made non-trivial so a parity
    check exercises them.
    """
    g = torch.Generator().manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) % (2**31 - 1))
    leaf = key.split(".")[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros((), dtype=torch.long)
    if leaf == "running_mean":
        return (torch.rand(shape, generator=g) - 0.5) * 0.2
    if leaf == "running_var":
        return 0.8 + 0.4 * torch.rand(shape, generator=g)
    if len(shape) == 4:  # conv
        fan_out = shape[0] * shape[2] * shape[3]
        if "aux_nets" in key:
            fan_out = shape[1]  # keep the 64->1 projection O(1)
        return torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_out)
    if "depth_nets" in key:  # InstanceNorm2d(1, affine)
        return (0.9 + 0.2 * torch.rand(shape, generator=g)) if leaf == "weight" else (torch.rand(shape, generator=g) - 0.5) * 0.2
    if len(shape) == 1 and leaf in ("weight", "bias") and (".bn" in key or "downsample.1" in key or key.endswith("bn1.weight") or key.endswith("bn1.bias")):
        if leaf == "weight":
            return 0.8 + 0.4 * torch.rand(shape, generator=g)
        return (torch.rand(shape, generator=g) - 0.5) * 0.2
    if len(shape) == 2:  # Linear / LSTM matrices
        bound = 1.0 / math.sqrt(shape[1])
        if "weight_ih" in key or "weight_hh" in key:
            bound = 1.0 / math.sqrt(shape[0] // 4)
        return (torch.rand(shape, generator=g) * 2 - 1) * bound
    # biases of Linear / LSTM / aux conv: small positive-leaning so final ReLU outputs are not all dead
    return torch.rand(shape, generator=g) * 0.2 + 0.05


def make_state(kind, cfg, seed):
    """state_dict-like dict (reference key names) with deterministic values."""
    return {k: value_for_key(k, s, seed) for k, s in model_keys(kind, cfg)}


# ----------------------------------------------------------------------------
# synthetic Robosuite-shaped inputs (util/data_utils.py:48-54,162-176,207-211)
# ----------------------------------------------------------------------------
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def _rand_pose(lead, g):
    pos = torch.rand(*lead, 3, generator=g)
    pos = pos * torch.tensor([0.7, 0.7, 0.5]) + torch.tensor([-0.35, -0.35, 0.8])
    q = torch.randn(*lead, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True)
    q = torch.where(q[..., 3:4] < 0, -q, q)  # standardize_quat: w >= 0
    return torch.cat([pos, q], dim=-1)


def synth_batch(lead, seed, hw=224, with_depth=False):
    """lead = (N,) for one-shot models or (S, N) for sequence models."""
    lead = tuple(lead)
    g = torch.Generator().manual_seed(int(seed))
    u8 = torch.randint(0, 256, (*lead, hw, hw, 3), generator=g, dtype=torch.uint8)
    img = u8.float() / 255.0
    img = (img - torch.tensor(IMAGENET_MEAN)) / torch.tensor(IMAGENET_STD)
    img = img.movedim(-1, -3).contiguous()  # (..., 3, H, W)
    depth = torch.rand(*lead, 1, hw, hw, generator=g) if with_depth else None
    x0 = _rand_pose(lead, g)
    x1 = _rand_pose(lead, g)
    obj = _rand_pose(lead, g)
    x0bar = x0 + math.sqrt(0.001) * torch.randn(*lead, 7, generator=g)
    qb = x0bar[..., 3:]
    x0bar = torch.cat([x0bar[..., :3], qb / qb.norm(dim=-1, keepdim=True)], dim=-1)
    return {"img": img, "depth": depth, "x0bar": x0bar, "x0": x0, "x1": x1, "obj": obj}


# ----------------------------------------------------------------------------
# ResNet-50 (functional).  `sd` maps key -> tensor; BN running stats are updated
# in place in train mode, exactly as nn.BatchNorm2d does.
# ----------------------------------------------------------------------------
# Reduced-precision emulation (test yardstick only).  With EMULATE set to torch.bfloat16 / torch.float16, resnet50_forward
# rounds what the HIP trunk keeps in that type -- the staged image, the conv weight copies, every raw conv output y and every
# BN(+residual)(+ReLU) output a -- and, in the backward, the gradients flowing through the same points (straight-through
# rounding).  16-bit gradients of this 50-layer train-mode-BN network at random initialisation are dominated by amplified
# rounding noise (the BN backward subtracts the common-mode part of the gradient at every layer, the rounding noise stays);
# the emulation tells a parity test how far ANY implementation with these storage types lands from the fp64 gradient.
EMULATE = None


class _RoundSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return x.to(dt).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt).to(g.dtype), None


def _q(x):
    return x if EMULATE is None else _RoundSTE.apply(x, EMULATE)


def _qw(w):
    """compute-dtype copy of an fp32 master weight: rounded in the forward, the gradient reaches the master unrounded"""
    return w if EMULATE is None else w + (w.to(EMULATE).to(w.dtype) - w).detach()


def _bn(sd, name, x, train):
    rm, rv = sd[name + ".running_mean"], sd[name + ".running_var"]
    if train:
        sd[name + ".num_batches_tracked"] = sd[name + ".num_batches_tracked"] + 1
    return F.batch_norm(x, rm, rv, sd[name + ".weight"], sd[name + ".bias"], train, BN_MOMENTUM, BN_EPS)


def resnet50_forward(sd, pre, x, train, depth=50):
    """Returns (latent features (B, L), hooked maps {0: conv1 x, 9: relu(bn1(conv1 x)) (the in-place ReLU reaches the hooked
    tensor), 1..4: layer outputs})."""
    x = x if EMULATE is None else x.to(EMULATE).to(x.dtype)
    y = _q(F.conv2d(x, _qw(sd[pre + "conv1.weight"]), None, 2, 3))
    early = _q(F.relu(_bn(sd, pre + "bn1", y, train)))
    maps = {0: y, 9: early}
    y = F.max_pool2d(early, 3, 2, 1)
    for li, (planes, nblk, stride) in enumerate(stages(depth), start=1):
        for b in range(nblk):
            p = "%slayer%d.%d" % (pre, li, b)
            s = stride if b == 0 else 1
            if depth in BASIC:   # torchvision BasicBlock: conv3x3(s) - bn - relu - conv3x3 - bn, + identity / projection, relu
                o = _q(F.relu(_bn(sd, p + ".bn1", _q(F.conv2d(y, _qw(sd[p + ".conv1.weight"]), None, s, 1)), train)))
                o = _bn(sd, p + ".bn2", _q(F.conv2d(o, _qw(sd[p + ".conv2.weight"]), None, 1, 1)), train)
                if (p + ".downsample.0.weight") in sd:
                    idn = _q(_bn(sd, p + ".downsample.1", _q(F.conv2d(y, _qw(sd[p + ".downsample.0.weight"]), None, s)), train))
                else:
                    idn = y
                y = _q(F.relu(o + idn))
                continue
            o = _q(F.relu(_bn(sd, p + ".bn1", _q(F.conv2d(y, _qw(sd[p + ".conv1.weight"]))), train)))
            o = _q(F.relu(_bn(sd, p + ".bn2", _q(F.conv2d(o, _qw(sd[p + ".conv2.weight"]), None, s, 1)), train)))
            o = _bn(sd, p + ".bn3", _q(F.conv2d(o, _qw(sd[p + ".conv3.weight"]))), train)
            if b == 0:
                idn = _q(_bn(sd, p + ".downsample.1", _q(F.conv2d(y, _qw(sd[p + ".downsample.0.weight"]), None, s)), train))
            else:
                idn = y
            y = _q(F.relu(o + idn))
        maps[li] = y
    y = F.adaptive_avg_pool2d(y, 1).flatten(1)
    return F.linear(y, sd[pre + "fc.weight"], sd[pre + "fc.bias"]), maps


def aux_head(early, w, b):
    """Conv2d(64->1, 1x1, bias) -> MaxPool2d(2) -> Flatten (models/naive.py:223-231)."""
    return F.max_pool2d(F.conv2d(early, w, b), 2).flatten(1)


def depth_head(depth, w, b, pools=2):
    """AvgPool2d(2) x pools -> InstanceNorm2d(1, affine) -> Flatten (models/naive.py:233-240)."""
    d = depth
    for _ in range(pools):
        d = F.avg_pool2d(d, 2)
    return F.instance_norm(d, None, None, w, b, True, 0.1, 1e-5).flatten(1)


def lstm_forward(x, w_ih, w_hh, b_ih, b_hh, h0=None, c0=None):
    """Single-layer LSTM over x (S, N, I), torch gate order i,f,g,o.  Returns (out, h, c)."""
    S, N, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(N, H) if h0 is None else h0
    c = x.new_zeros(N, H) if c0 is None else c0
    outs = []
    xg = F.linear(x, w_ih, b_ih)
    for t in range(S):
        g = xg[t] + F.linear(h, w_hh, b_hh)
        i, f, gg, o = g.chunk(4, dim=-1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs.append(h)
    return torch.stack(outs, 0), h, c


def _features(kind, cfg, sd, img, depth, train):
    """ResNet + aux(+depth) concat shared by no/td/tdo/tdo_v2. img (B,3,H,W)."""
    fpre = "feature_net." if kind in ("n", "td") else "feature_net.module."
    feat, maps = resnet50_forward(sd, fpre, img, train, cfg.get("depth", 50))
    if kind == "n":
        return feat
    parts = [feat]
    for i, h in enumerate(hooks_of(cfg)):
        k = hook_pools(h)
        if kind == "td":
            aw, ab = sd["~aux_nets.%d.0.weight" % i], sd["~aux_nets.%d.0.bias" % i]
            dw, db = sd["~depth_nets.%d.%d.weight" % (i, k)], sd["~depth_nets.%d.%d.bias" % (i, k)]
        else:
            aw, ab = sd["aux_nets.%d.module.0.weight" % i], sd["aux_nets.%d.module.0.bias" % i]
            dw, db = sd["depth_nets.%d.module.%d.weight" % (i, k)], sd["depth_nets.%d.module.%d.bias" % (i, k)]
        a = aux_head(maps[h], aw, ab)
        if cfg.get("use_depth"):
            a = a * depth_head(depth, dw, db, k)
        parts.append(a)
    return torch.cat(parts, dim=-1)


def model_forward(kind, cfg, sd, img, depth, x0bar, train=True, state=None):
    """Forward of the five reference models.  `state` (dict) carries (h, c) in rollout
    mode (models/time_sensitive.py:503-507); None = zero initial state (training)."""
    if kind in ("n", "no"):
        f = _features(kind, cfg, sd, img, depth, train)
        if kind == "n":
            pre = f
            n_pre = len(cfg["hidden"]) + 1
            for i in range(n_pre):
                pre = F.relu(F.linear(pre, sd["pre_fc%d.weight" % i], sd["pre_fc%d.bias" % i]))
            post = torch.cat([f, pre - x0bar], dim=1)
            for i in range(n_pre):
                post = F.relu(F.linear(post, sd["post_fc%d.weight" % i], sd["post_fc%d.bias" % i]))
            return pre, post
        out = f if cfg.get("no_proprioception") else torch.cat((f, x0bar), dim=-1)
        for i in range(len(cfg["hidden"]) + 1):
            out = F.relu(F.linear(out, sd["fc%d.module.weight" % i], sd["fc%d.module.bias" % i]))
        return out
    S, N = img.shape[:2]
    d = None if depth is None else depth.reshape(S * N, *depth.shape[2:])
    f = _features(kind, cfg, sd, img.reshape(S * N, *img.shape[2:]), d, train).view(S, N, -1)

    def run(prefix, x, tag):
        h0 = c0 = None
        if state is not None and tag in state:
            h0, c0 = state[tag]
        o, h, c = lstm_forward(x, sd[prefix + "weight_ih_l0"], sd[prefix + "weight_hh_l0"],
                               sd[prefix + "bias_ih_l0"], sd[prefix + "bias_hh_l0"], h0, c0)
        if state is not None:
            state[tag] = (h, c)
        return o

    if kind == "td":
        h1 = run("pre_measurement_rnn.", f, "pre")
        pre = F.linear(h1, sd["pre_measurement_fc.weight"], sd["pre_measurement_fc.bias"])
        h2 = run("post_measurement_rnn.", torch.cat([f, pre - x0bar], dim=-1), "post")
        post = F.linear(h2, sd["post_measurement_fc.weight"], sd["post_measurement_fc.bias"])
        return pre, post
    if kind == "tdo":
        x = f if cfg.get("no_proprioception") else torch.cat((f, x0bar), dim=-1)
        h = run("rnn.module.", x, "rnn")
    else:
        h = torch.cat((run("img_rnn.module.", f, "img"), run("proprio_rnn.module.", x0bar, "proprio")), dim=-1)
    h = F.linear(h, sd["fc.module.0.weight"], sd["fc.module.0.bias"])
    return F.linear(h, sd["fc.module.1.weight"], sd["fc.module.1.bias"])


# ----------------------------------------------------------------------------
# PoseDistanceLoss (models/losses.py:47-128)
# ----------------------------------------------------------------------------
def _pos_dist(pp, tp, metric, eps):
    d = pp - tp
    l2 = torch.sqrt((d * d).sum(-1) + eps).sum()
    l1 = d.abs().sum()
    linf = d.abs().max(dim=-1)[0].sum()
    return {"l2": l2, "l1": l1, "linf": linf, "combined": l2 + l1 + linf}[metric]


def quat_angle_np(qp, qt):
    """|angle| of quat_distance(qp, qt) wrapped to [-pi, pi], per row (xyzw).
    Follows models/losses.py:104-111 + robosuite ~v1.0 quat_distance/quat2axisangle."""
    qp = np.asarray(qp, dtype=np.float32)
    qt = np.asarray(qt, dtype=np.float32)
    inv = qt * np.array([-1, -1, -1, 1], dtype=np.float32) / (qt * qt).sum(-1, keepdims=True)
    w = (-qp[:, 0] * inv[:, 0] - qp[:, 1] * inv[:, 1] - qp[:, 2] * inv[:, 2] + qp[:, 3] * inv[:, 3]).astype(np.float32)
    w = np.clip(w, -1.0, 1.0).astype(np.float64)
    den = np.sqrt(1.0 - w * w)
    ang = np.where(np.isclose(den, 0.0, rtol=1e-9, atol=0.0), 0.0, 2.0 * np.arccos(w))
    ang = np.where(ang > np.pi, ang - 2 * np.pi, ang)
    return np.abs(ang)


def pose_loss(pred, truth, metric="l2", scale=1.0, alpha=1.0, eps=1e-4, mode="pose"):
    if metric not in ("l1", "l2", "linf", "combined"):
        raise ValueError("Invalid distance metric specified: %r" % (metric,))
    if mode not in ("position", "pose", "val"):
        raise ValueError("Invalid loss mode specified: %r" % (mode,))
    pp, pq = pred[..., :3], pred[..., 3:]
    tp, tq = truth[..., :3], truth[..., 3:]
    pq = pq / torch.sqrt((pq * pq).sum(-1, keepdim=True))  # no eps: NaN for an all-zero quat
    pos = _pos_dist(pp, tp, metric, eps)
    if mode == "val":
        ang = quat_angle_np(pq.reshape(-1, 4).detach().numpy(), tq.reshape(-1, 4).detach().numpy())
        return pos.detach().numpy(), float(ang.sum())
    if mode == "pose":
        ip = (pq * tq).sum(-1)
        ori = (1 - ip * ip).sum() + torch.clamp(-pq[..., -1], min=0).sum()
    else:
        ori = 0
    return scale * (pos + alpha * ori)


# ----------------------------------------------------------------------------
# Adam (torch.optim.Adam defaults) and the train step
# ----------------------------------------------------------------------------
def adam_update(p, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """In-place single-tensor Adam as torch.optim.Adam computes it (no amsgrad/decay)."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


NON_PARAM_LEAVES = ("running_mean", "running_var", "num_batches_tracked")


def trainable_keys(kind, cfg):
    ks = []
    for k, _ in model_keys(kind, cfg):
        if k.split(".")[-1] in NON_PARAM_LEAVES or k.startswith("~"):
            continue
        ks.append(k)
    return ks


def train_step(kind, cfg, sd, batch, loss_cfg, opt, lr=1e-3, val_metrics=True):
    """One optimisation step (util/learn_utils.py:152-179).  `sd` and `opt` (dict with
    'step', 'm', 'v') are updated in place.  Returns dict(outputs, loss, grads, pos_err, ori_err)."""
    keys = trainable_keys(kind, cfg)
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    work = dict(sd)
    work.update(leaves)
    out = model_forward(kind, cfg, work, batch["img"], batch.get("depth"), batch["x0bar"], train=True)
    if kind in ("n", "td"):
        loss = pose_loss(out[0], batch["x0"], **loss_cfg) + pose_loss(out[1], batch["x1"], **loss_cfg)
        vo, vt = out[1], batch["x1"]
    else:
        loss = pose_loss(out, batch["obj"], **loss_cfg)
        vo, vt = out, batch["obj"]
    pos_err = ori_err = None
    if val_metrics:
        pos_err, ori_err = pose_loss(vo.detach(), vt, mode="val")
    grads = torch.autograd.grad(loss, [leaves[k] for k in keys], allow_unused=True)
    # BN running statistics were updated inside `work`
    for k in work:
        if k.split(".")[-1] in NON_PARAM_LEAVES:
            sd[k] = work[k].detach()
    opt["step"] = opt.get("step", 0) + 1
    gd = {}
    with torch.no_grad():
        for k, g in zip(keys, grads):
            if g is None:  # e.g. depth head when use_depth=False: Adam skips it
                continue
            gd[k] = g
            m = opt.setdefault("m", {}).setdefault(k, torch.zeros_like(sd[k]))
            v = opt.setdefault("v", {}).setdefault(k, torch.zeros_like(sd[k]))
            adam_update(sd[k], g, m, v, opt["step"], lr)
    outs = tuple(o.detach() for o in out) if isinstance(out, tuple) else out.detach()
    return {"outputs": outs, "loss": loss.detach(), "grads": gd, "pos_err": pos_err, "ori_err": ori_err}
