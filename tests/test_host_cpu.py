"""CPU-only checks of the host side: the C-ABI library loads and exports every declared symbol, the drop-in
classes keep the reference's state_dict keys / shapes / constructor validation, and the product refuses to
run without the GPU path (no silent CPU fallback)."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import pose_oracle as po
from _helpers import CASES, build, load_values

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from rgb_proprioceptive_pose_estimator_amd import _lib
    header = open(os.path.join(ROOT, "include", "rpe_hip.h")).read()
    declared = set(re.findall(r"\b(rpe_[a-z0-9_]+)\s*\(", header))
    declared -= {"rpe_resnet50_t"}
    assert declared, "no declarations parsed"
    missing = [n for n in sorted(declared) if not hasattr(_lib.raw, n)]
    assert not missing, missing
    assert set(_lib.EXPORTS) == declared, sorted(set(_lib.EXPORTS) ^ declared)
    assert _lib.lib.rpe_abi_version() == _lib.ABI_VERSION


@pytest.mark.parametrize("kind", list(CASES))
def test_state_dict_keys_and_shapes_match_reference(kind, golden_dir):
    gold = np.load(os.path.join(golden_dir, "model_%s.npz" % kind))
    cfg, lead, wseed, _ = CASES[kind]
    model = build(kind, cfg, torch.float32)
    sd = model.state_dict()
    assert list(sd.keys()) == [str(k) for k in gold["keys"]]  # keys as the reference's own classes produce them
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == [(k, tuple(s)) for k, s in po.model_keys(kind, cfg) if not k.startswith("~")]
    load_values(model, kind, po.make_state(kind, cfg, wseed))
    # reference-visible attributes (util/learn_utils.py:58,75,118,296,323)
    assert model.requires_sequence == (kind in ("td", "tdo", "tdo_v2"))
    assert hasattr(model, "object_name") == (kind in ("no", "tdo", "tdo_v2"))
    assert model.rollout is False
    model.reset_initial_state(3)
    if kind != "n":
        assert model.aux_latent_dim == 3136
        aux, dep = model.aux_nets[0], model.depth_nets[0]
        if kind != "td":
            assert aux.module[0].weight.shape == (1, 64, 1, 1) and dep.module[2].weight.shape == (1,)
    if kind == "td":  # quirk: heads are not registered (models/time_sensitive.py:102-115)
        assert not any("aux_nets" in k or "depth_nets" in k for k in sd)


def test_no_cpu_fallback():
    cfg, lead, wseed, dseed = CASES["no"]
    model = build("no", cfg, torch.float32)
    b = po.synth_batch(lead, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback|HIP path"):
        model(b["img"], None, b["x0bar"])
    from rgb_proprioceptive_pose_estimator_amd.models import PoseDistanceLoss
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        PoseDistanceLoss()(b["obj"], b["obj"])
    with pytest.raises(ValueError):
        PoseDistanceLoss(distance_metric="l3")
    with pytest.raises(ValueError):
        PoseDistanceLoss(mode="train")
    out = PoseDistanceLoss(mode="val")
    assert out.distance_metric == "l2" and out.epsilon == 1e-4


def test_import_resnet_contract():
    from rgb_proprioceptive_pose_estimator_amd.util.model_utils import import_resnet
    with pytest.raises(AssertionError):
        import_resnet(34, 10)  # the reference's option set spells 34 as 32 (util/model_utils.py:130)
    m, size = import_resnet(50, 12, feature_extract=True, use_pretrained=False)
    assert size == 224 and m.fc.out_features == 12 and m.fc.in_features == 2048
    assert all(p.requires_grad for p in m.parameters())  # frozen only iff feature_extract AND use_pretrained
    with pytest.warns(UserWarning):
        m, _ = import_resnet(50, 12, feature_extract=True, use_pretrained=True)
    assert [n for n, p in m.named_parameters() if p.requires_grad] == ["fc.weight", "fc.bias"]
    # "32" passes the reference's assert and then fails in getattr(models, "resnet32") (util/model_utils.py:136); 18 is the BasicBlock
    # network: fc -> Linear(512, output_dim), identity shortcut in layer1.0, projection shortcuts in layer2..4.0, torchvision's key table
    with pytest.raises(AttributeError):
        import_resnet(32, 10, False, False)
    net, size = import_resnet(18, 8, True, False)
    assert size == 224 and net.fc.in_features == 512 and net.fc.out_features == 8
    assert not hasattr(net.layer1[0], "downsample") and hasattr(net.layer2[0], "downsample") and not hasattr(net.layer1[0], "conv3")
    from oracle import pose_oracle as po
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == [(k, tuple(s)) for k, s in po.resnet_keys(8, 18)]


def test_pretrained_checkpoint_ingestion_and_freezing(tmp_path, monkeypatch):
    """import_resnet with use_pretrained=True (util/model_utils.py:136 of the reference fetches torchvision's ImageNet weights): here a
    LOCAL checkpoint in torchvision's resnet50 state_dict format, named by RPE_RESNET50_WEIGHTS, is loaded into the trunk; with
    feature_extract as well every body parameter is frozen (util/model_utils.py:110-113,137) and the replaced fc stays trainable
    (:140-141)."""
    import torch

    from oracle import pose_oracle as po
    from rgb_proprioceptive_pose_estimator_amd.util.model_utils import PRETRAINED_ENV, import_resnet

    g = torch.Generator().manual_seed(3)
    ckpt = {}
    for k, shape in po.resnet_keys(1000):          # torchvision's key table: conv1.weight, bn1.*, layer*.*, fc.weight [1000, 2048], fc.bias
        if k.endswith("num_batches_tracked"):
            ckpt[k] = torch.tensor(7)
        elif k.endswith("running_var"):
            ckpt[k] = torch.rand(shape, generator=g) + 0.5
        else:
            ckpt[k] = torch.randn(shape, generator=g) * 0.05
    path = tmp_path / "resnet50_imagenet_format.pth"
    torch.save(ckpt, path)
    monkeypatch.setenv(PRETRAINED_ENV, str(path))
    trunk, size = import_resnet(50, 64, feature_extract=True, use_pretrained=True, compute_dtype=torch.float32)
    assert size == 224
    sd = trunk.state_dict()
    for k, v in ckpt.items():
        if k.startswith("fc."):
            continue                                # replaced by Linear(2048, 64)
        assert torch.equal(sd[k], v), k
    assert tuple(sd["fc.weight"].shape) == (64, 2048)
    frozen = [n for n, p in trunk.named_parameters() if not p.requires_grad]
    trainable = [n for n, p in trunk.named_parameters() if p.requires_grad]
    assert trainable == ["fc.weight", "fc.bias"] and len(frozen) == 159
    assert trunk.body_frozen()
    trunk2, _ = import_resnet(50, 64, feature_extract=False, use_pretrained=True, compute_dtype=torch.float32)   # fine-tuning: loaded, not frozen
    assert torch.equal(trunk2.state_dict()["layer3.2.conv2.weight"], ckpt["layer3.2.conv2.weight"]) and not trunk2.body_frozen()


def test_torch_ops_are_registered_for_the_device_only():
    """torch_ops.py: every `torch.ops.rpe.*` entry exists in the dispatcher with a schema and a fake (shape) implementation, and has a
    kernel for the CUDA (= HIP) device type ONLY -- a CPU call fails in the dispatcher instead of reaching a fallback."""
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode

    import rgb_proprioceptive_pose_estimator_amd.torch_ops as T
    for n in T.NAMES:
        assert getattr(torch.ops.rpe, n).default._schema.name == "rpe::" + n
    x, w = torch.randn(1, 8, 8, 8), torch.randn(8, 3, 3, 8)
    for call in (lambda: torch.ops.rpe.conv2d_fwd(x, w, 1, 1), lambda: torch.ops.rpe.conv2d(x, w, 1, 1),
                 lambda: torch.ops.rpe.pose_loss(torch.randn(2, 7), torch.randn(2, 7), 3, 1, 1.0, 0.5, 1e-4),
                 lambda: torch.ops.rpe.adam_step(torch.zeros(8), torch.zeros(8), torch.zeros(8), torch.zeros(8), 1e-3, 0.9, 0.999, 1e-8, 1)):
        with pytest.raises(NotImplementedError):
            call()
    with FakeTensorMode():
        xf, wf = torch.empty(2, 16, 16, 8, device="cuda", dtype=torch.bfloat16), torch.empty(16, 3, 3, 8, device="cuda", dtype=torch.bfloat16)
        y = torch.ops.rpe.conv2d(xf, wf, 2, 1)
        assert y.shape == (2, 8, 8, 16) and y.dtype == torch.bfloat16
        assert torch.ops.rpe.conv2d_wgrad(xf, y, 3, 2, 1).shape == (16, 3, 3, 8)
        assert torch.ops.rpe.conv2d_dgrad(y, wf.permute(3, 1, 2, 0), [2, 16, 16, 8], 2, 1).shape == (2, 16, 16, 8)


@pytest.mark.parametrize("cfg", [(2, 5, 7, 3, 4), (1, 8, 8, 2, 2), (3, 4, 60, 2, 3)])
def test_wgrad_halo_padded_grid_algebra(cfg):
    """The algorithm of csrc/wgrad_halo.hip restated on the CPU (no HIP call): the 3x3 / stride-1 / pad-1 weight gradient as a reduction over
    the positions g of the zero-padded pixel grid [B][H+2][W+2] with ONE uniform shift per tap, x read from a ring of 64-position blocks that
    starts W + 3 positions before the split -- against torch's weight gradient; and the ring conditions the launcher relies on (a step reads
    blocks t .. t + E only; PF + E + 1 blocks fit the ring of every configuration it launches)."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    b, h, w, ci, co = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = torch.randn(b, ci, h, w, generator=g, dtype=torch.float64)
    wt = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, wt, None, 1, 1)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    (ref,) = torch.autograd.grad(y, wt, dy)
    pw, ph = w + 2, h + 2
    G, hl = b * ph * pw, w + 3
    E = (63 + 2 * hl) // 64
    xp = np.zeros((b, ph, pw, ci)); xp[:, 1:-1, 1:-1] = x.permute(0, 2, 3, 1).numpy()
    dyp = np.zeros((b, ph, pw, co)); dyp[:, 1:-1, 1:-1] = dy.permute(0, 2, 3, 1).numpy()
    xp, dyp = xp.reshape(G, ci), dyp.reshape(G, co)

    def x_at(p):   # what the buffer descriptor's range check returns outside the tensor: zeros
        return xp[p] if 0 <= p < G else np.zeros(ci)

    dw = np.zeros((co, 3, 3, ci))
    rps = 128   # two steps per split
    for g0 in range(0, G, rps):
        g1 = min(G, g0 + rps)
        for step in range((g1 - g0 + 63) // 64):
            for row in range(64):
                gg = g0 + 64 * step + row
                if gg >= g1:
                    continue
                for t in range(9):
                    pr = 64 * step + row + (t // 3) * pw + (t % 3)          # stream position: the stream starts at grid position g0 - (W + 3)
                    assert step <= pr // 64 <= step + E                      # the blocks a step may touch
                    p = g0 - hl + pr
                    assert p == gg + (t // 3 - 1) * pw + (t % 3 - 1)         # = the tap's uniform shift on the padded grid
                    dw[:, t // 3, t % 3, :] += np.outer(dyp[gg], x_at(p))
    np.testing.assert_allclose(dw, ref.permute(0, 2, 3, 1).numpy(), rtol=1e-10, atol=1e-10)
    # ring capacity of the four launch configurations (wgrad_halo.hip, conv_wgrad_halo): (WI, PF, RINGP) by E
    for wi in (2, 1):
        for e in (1, 2):
            pf, ring = (2, 512 if e == 2 else 256) if wi == 2 else ((1, 256) if e == 2 else (2, 256))
            assert (pf + e + 1) * 64 <= ring
    assert (63 + 2 * (60 + 3)) // 64 == 2   # the widest map the form takes (60 pixels) still reads two blocks ahead at most
