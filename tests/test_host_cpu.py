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
