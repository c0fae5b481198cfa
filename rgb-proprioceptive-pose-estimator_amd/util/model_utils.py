"""Model construction helpers.  Mirrors util/model_utils.py:110-147 of the reference
(`set_parameter_requires_grad`, `import_resnet`); the visualisation helpers of that file are out of
scope."""
import os
import warnings

import torch
import torch.nn as nn

from ..engine import ResNet50Trunk

# An ImageNet checkpoint in torchvision's resnet50 state_dict format, if one is available locally.
# The reference fetches it over the network (torchvision pretrained=True); there is no egress here.
PRETRAINED_ENV = "RPE_RESNET50_WEIGHTS"


def set_parameter_requires_grad(model, feature_extracting):
    """Freeze every parameter when feature extracting (util/model_utils.py:110-113)."""
    if feature_extracting:
        for param in model.parameters():
            param.requires_grad = False


def import_resnet(num_layers, output_dim, feature_extract=True, use_pretrained=True, compute_dtype=torch.bfloat16):
    """ResNet feature extractor with its fc replaced by Linear(fc.in_features, output_dim) (2048; 512 for resnet18).

    Same contract as util/model_utils.py:116-147: validates `num_layers` against the reference's
    option set (which spells 34 as 32), freezes the body iff `feature_extract and use_pretrained`,
    the new fc is always trainable, returns (model, 224).  Every member the reference can build has a native launch
    plan: the bottleneck networks 50 / 101 / 152 and the BasicBlock network 18 (fc input 512); "32" passes the reference's
    assert and then fails in `getattr(models, "resnet32")` -- the same AttributeError is raised here.  Every caller of the
    reference passes 50 (scripts/train_model.py:63).
    """
    options = {18, 32, 50, 101, 152}
    assert num_layers in options, "Invalid layer size specified. Options are: {}".format(options)
    if num_layers == 32:   # util/model_utils.py:136: getattr(models, "resnet32") -- torchvision has no such model
        raise AttributeError("module 'torchvision.models' has no attribute 'resnet32'")
    model = ResNet50Trunk(1000, compute_dtype=compute_dtype, depth=num_layers)
    if use_pretrained:
        path = os.environ.get(PRETRAINED_ENV)
        if path and os.path.exists(path):
            model.load_state_dict(torch.load(path, map_location="cpu"))
        else:
            warnings.warn("use_pretrained=True but no local ImageNet checkpoint (set %s): the network fetch the reference "
                          "performs is unavailable offline; continuing from the random initialisation" % PRETRAINED_ENV)
    set_parameter_requires_grad(model, (feature_extract and use_pretrained))
    model.fc = nn.Linear(model.fc.in_features, output_dim)
    return model, 224
