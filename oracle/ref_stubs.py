"""TEST INFRASTRUCTURE ONLY -- build-owned stand-ins for two third-party packages.

The reference (``/root/reference``) imports ``torchvision`` (util/model_utils.py:6,
util/data_utils.py:2) and ``robosuite.utils.transform_utils`` (models/losses.py:4).
Neither is installed in this image and neither is part of the reference tree
(requirements.txt:1-6, unpinned).  To run the reference's *own* wiring code
(hooks, concatenations, LSTM/FC topology, loss formula, DataParallel key names)
in this container we register the two modules below in ``sys.modules`` before
importing it.  They restate the PUBLISHED algorithms:

* ``torchvision.models.resnet50`` (and 101 / 152; 18 / 34 with BasicBlock):
  ResNet v1.5 (He et al. 2015, stride on the 3x3 conv), attribute names conv1/bn1/relu/maxpool/layer1-4/avgpool/fc,
  kaiming_normal_(fan_out, relu) conv init, BN weight 1 / bias 0.
* ``robosuite.utils.transform_utils`` (~v1.0, mid 2020): xyzw quaternions,
  ``quat_distance(q1, q0) = q1 * q0^-1`` and ``quat2axisangle`` returning the
  2-tuple ``(axis, angle)`` that models/losses.py:105 unpacks.  Restated from
  the public release; *val-mode orientation parity therefore rests on this
  restatement* (SURVEY.md section 8c, item 2).

Only ``oracle/gen_golden.py`` installs these; nothing in the product imports
this file.
"""
import math
import sys
import types

import numpy as np
import torch
import torch.nn as nn


# ----------------------------------------------------------------------------
# torchvision-shaped ResNet-50 v1.5
# ----------------------------------------------------------------------------
class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out += identity
        return self.relu(out)


class _BasicBlock(nn.Module):
    """torchvision BasicBlock (ResNet-18 / 34): two 3x3 convs, stride on the first, expansion 1."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out += identity
        return self.relu(out)


class _ResNet(nn.Module):
    def __init__(self, blocks, num_classes=1000, block=_Bottleneck):
        super().__init__()
        self.block = block
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._stage(64, blocks[0], 1)
        self.layer2 = self._stage(128, blocks[1], 2)
        self.layer3 = self._stage(256, blocks[2], 2)
        self.layer4 = self._stage(512, blocks[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _stage(self, planes, n, stride):
        down, block = None, self.block
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * block.expansion
        for _ in range(1, n):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def _resnet50(pretrained=False, **kw):
    # pretrained=True would be a network fetch: never attempted (no egress);
    # the caller gets the seeded random init instead.
    return _ResNet([3, 4, 6, 3])


def _resnet18(pretrained=False, **kw):
    return _ResNet([2, 2, 2, 2], block=_BasicBlock)


def _resnet34(pretrained=False, **kw):
    return _ResNet([3, 4, 6, 3], block=_BasicBlock)


def _resnet101(pretrained=False, **kw):
    return _ResNet([3, 4, 23, 3])


def _resnet152(pretrained=False, **kw):
    return _ResNet([3, 8, 36, 3])


class _Inert:
    def __init__(self, *a, **k):
        pass

    def __call__(self, x):
        return x


# ----------------------------------------------------------------------------
# robosuite.utils.transform_utils (xyzw)
# ----------------------------------------------------------------------------
def quat_multiply(q1, q0):
    x0, y0, z0, w0 = q0
    x1, y1, z1, w1 = q1
    return np.array(
        (
            x1 * w0 + y1 * z0 - z1 * y0 + w1 * x0,
            -x1 * z0 + y1 * w0 + z1 * x0 + w1 * y0,
            x1 * y0 - y1 * x0 + z1 * w0 + w1 * z0,
            -x1 * x0 - y1 * y0 - z1 * z0 + w1 * w0,
        ),
        dtype=np.float32,
    )


def quat_conjugate(q):
    return np.array((-q[0], -q[1], -q[2], q[3]), dtype=np.float32)


def quat_inverse(q):
    return quat_conjugate(q) / np.dot(q, q)


def quat_distance(q1, q0):
    return quat_multiply(q1, quat_inverse(q0))


def quat2axisangle(q):
    w = q[3]
    if w > 1.0:
        w = 1.0
    elif w < -1.0:
        w = -1.0
    den = np.sqrt(1.0 - w * w)
    if math.isclose(den, 0.0):
        return np.zeros(3), 0.0
    return q[:3] / den, 2.0 * math.acos(w)


def install():
    """Register the stand-ins in sys.modules (idempotent)."""
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvm = types.ModuleType("torchvision.models")
        tvt = types.ModuleType("torchvision.transforms")
        tvm.resnet50, tvm.resnet101, tvm.resnet152 = _resnet50, _resnet101, _resnet152
        tvm.resnet18, tvm.resnet34 = _resnet18, _resnet34   # (no resnet32: util/model_utils.py:130's "32" raises AttributeError there too)
        for name in ("Compose", "ToPILImage", "Resize", "CenterCrop", "ToTensor", "Normalize"):
            setattr(tvt, name, _Inert)
        tv.models, tv.transforms = tvm, tvt
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.models"] = tvm
        sys.modules["torchvision.transforms"] = tvt
    if "robosuite" not in sys.modules:
        rs = types.ModuleType("robosuite")
        rsu = types.ModuleType("robosuite.utils")
        rst = types.ModuleType("robosuite.utils.transform_utils")
        for f in (quat_multiply, quat_conjugate, quat_inverse, quat_distance, quat2axisangle):
            setattr(rst, f.__name__, f)
        rs.utils, rsu.transform_utils = rsu, rst
        sys.modules["robosuite"] = rs
        sys.modules["robosuite.utils"] = rsu
        sys.modules["robosuite.utils.transform_utils"] = rst
