"""Oracle: the ResNet-FPN model as a pure function of a ``state_dict`` (plain torch CPU ops).  TEST INFRASTRUCTURE ONLY.

Restates ``src/models/semanticFCN.py:8-40,266-354`` (resnet18/34/50 branch :145-153,:305-314) together with the public
torchvision 0.19 ``BasicBlock`` / ``Bottleneck`` / ``resnet18`` / ``resnet34`` / ``resnet50`` architecture (BasicBlock:
conv3x3(stride)-BN-ReLU-conv3x3-BN; Bottleneck (v1.5): conv1x1-BN-ReLU-conv3x3(stride)-BN-ReLU-conv1x1(x4)-BN;
1x1-stride conv + BN downsample, add, ReLU; layers [2,2,2,2] / [3,4,6,3]; widths 64-128-256-512 (x4 for resnet50)), which is a third-party
dependency absent from the reference tree (pinned torchvision 0.19.1, docker/Dockerfile:195).

Pinning: torchvision cannot be imported here, so the backbone half is restated from its published definition (parity of
the backbone UNPINNED by reference tests); the FPN / attention / decoder wiring IS pinned -- ``tools/gen_golden.py`` imports
the reference's own ``models.semanticFCN`` with a stub ``torchvision.models`` that serves ``ResNetRef`` below, loads the
same state_dict and compares (see tests/golden/fpn_*.npz).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

LAYERS = {"resnet18": [2, 2, 2, 2], "resnet34": [3, 4, 6, 3], "resnet50": [3, 4, 6, 3]}
EXPANSION = {"resnet18": 1, "resnet34": 1, "resnet50": 4}


# ---- a CPU-runnable torchvision-shaped ResNet (served to the reference through a stub `torchvision.models`) ----
class BasicBlockRef(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample, self.stride = downsample, stride

    def forward(self, x):
        idn = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idn)


class BottleneckRef(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)      # v1.5: the stride sits on the 3x3
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample, self.stride = downsample, stride

    def forward(self, x):
        idn = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + idn)


class ResNetRef(nn.Module):
    def __init__(self, layers, block=None, **_ignored):
        super().__init__()
        self.block = block or BasicBlockRef
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make(64, layers[0], 1)
        self.layer2 = self._make(128, layers[1], 2)
        self.layer3 = self._make(256, layers[2], 2)
        self.layer4 = self._make(512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * self.block.expansion, 1000)

    def _make(self, planes, blocks, stride):
        down, e = None, self.block.expansion
        if stride != 1 or self.inplanes != planes * e:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * e, 1, stride, bias=False), nn.BatchNorm2d(planes * e))
        seq = [self.block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * e
        seq += [self.block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*seq)


def torchvision_models_stub():
    """A module object that can stand in for `torchvision.models` (resnet18 / resnet34 / resnet50 only; `pretrained` ignored)."""
    import types
    m = types.ModuleType("torchvision.models")
    m.resnet18 = lambda *a, **k: ResNetRef(LAYERS["resnet18"])
    m.resnet34 = lambda *a, **k: ResNetRef(LAYERS["resnet34"])
    m.resnet50 = lambda *a, **k: ResNetRef(LAYERS["resnet50"], BottleneckRef)
    from oracle import effnet as _effnet
    _effnet.add_to_stub(m)          # efficientnet_v2_{s,m,l}
    return m


# ---- functional restatement ----
BN_TRAIN = False      # True: batch statistics (a training step's forward; the running statistics are not updated by this functional form)


def _bn(x, sd, p):
    if BN_TRAIN:
        return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.0, 1e-5)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def _block(x, sd, p, stride):
    idn = x
    if (p + ".downsample.0.weight") in sd:
        idn = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride=stride), sd, p + ".downsample.1")
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], None, stride=stride, padding=1), sd, p + ".bn1"))
    out = _bn(F.conv2d(out, sd[p + ".conv2.weight"], None, padding=1), sd, p + ".bn2")
    return F.relu(out + idn)


def _bottleneck(x, sd, p, stride):
    idn = x
    if (p + ".downsample.0.weight") in sd:
        idn = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride=stride), sd, p + ".downsample.1")
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], None), sd, p + ".bn1"))
    out = F.relu(_bn(F.conv2d(out, sd[p + ".conv2.weight"], None, stride=stride, padding=1), sd, p + ".bn2"))
    out = _bn(F.conv2d(out, sd[p + ".conv3.weight"], None), sd, p + ".bn3")
    return F.relu(out + idn)


def _stage(x, sd, name, nblocks, stride):
    for b in range(nblocks):
        p = f"backbone.{name}.{b}"
        x = (_bottleneck if (p + ".conv3.weight") in sd else _block)(x, sd, p, stride if b == 0 else 1)
    return x


def _cbr(x, sd, p):            # Sequential(conv3x3, BN, ReLU)
    return F.relu(_bn(F.conv2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], padding=1), sd, p + ".1"))


def _attention(x, sd, p):      # semanticFCN.py:25-40
    q = F.conv2d(x, sd[p + ".query_conv.weight"], sd[p + ".query_conv.bias"])
    k = F.conv2d(x, sd[p + ".key_conv.weight"], sd[p + ".key_conv.bias"])
    v = F.conv2d(x, sd[p + ".value_conv.weight"], sd[p + ".value_conv.bias"])
    s = F.conv2d(torch.tanh(q + k), sd[p + ".attention_conv.weight"], sd[p + ".attention_conv.bias"])
    return v * torch.softmax(s, dim=-1)


def fpn_forward(sd, x, meta, backbone="resnet18", attention=True, multi_scale_meta=True):
    """[B,num_classes,H,W] = SemanticNetworkWithFPN(x, meta)  -- semanticFCN.py:266-354, resnet branch."""
    layers = LAYERS[backbone]
    m = meta.shape[1]
    h = torch.cat([x, meta], 1)
    xs = F.max_pool2d(F.relu(F.conv2d(h, sd["backbone.conv1.weight"], None, padding=1)), 3, 2, 1)     # bn1 skipped (:149)
    x1 = _stage(xs, sd, "layer1", layers[0], 1)
    if multi_scale_meta:
        m1, m2, m3 = (F.interpolate(meta, scale_factor=s, mode="nearest") for s in (1 / 2, 1 / 4, 1 / 8))
        x2 = _stage(torch.cat([x1[:, :-m], m1], 1), sd, "layer2", layers[1], 2)
        x3 = _stage(torch.cat([x2[:, :-m], m2], 1), sd, "layer3", layers[2], 2)
        x4 = _stage(torch.cat([x3[:, :-m], m3], 1), sd, "layer4", layers[3], 2)
    else:
        x2 = _stage(x1, sd, "layer2", layers[1], 2)
        x3 = _stage(x2, sd, "layer3", layers[2], 2)
        x4 = _stage(x3, sd, "layer4", layers[3], 2)
    f4, f3, f2, f1 = _cbr(x4, sd, "fpn_block4"), _cbr(x3, sd, "fpn_block3"), _cbr(x2, sd, "fpn_block2"), _cbr(x1, sd, "fpn_block1")
    if attention:
        f4, f3 = _attention(f4, sd, "attention4"), _attention(f3, sd, "attention3")
        f2, f1 = _attention(f2, sd, "attention2"), _attention(f1, sd, "attention1")
    u4 = F.conv_transpose2d(f4, sd["upsample_layer_x4.weight"], sd["upsample_layer_x4.bias"], stride=8)
    u3 = F.conv_transpose2d(f3, sd["upsample_layer_x3.weight"], sd["upsample_layer_x3.bias"], stride=4)
    u2 = F.conv_transpose2d(f2, sd["upsample_layer_x2.weight"], sd["upsample_layer_x2.bias"], stride=2)
    y = torch.cat([f1, u2, u3, u4], 1)
    y = F.relu(_bn(F.conv2d(y, sd["decoder_semantic.0.weight"], sd["decoder_semantic.0.bias"], padding=1), sd, "decoder_semantic.1"))
    y = F.relu(_bn(F.conv2d(y, sd["decoder_semantic.3.weight"], sd["decoder_semantic.3.bias"], padding=1), sd, "decoder_semantic.4"))
    y = F.conv_transpose2d(y, sd["decoder_semantic.6.weight"], sd["decoder_semantic.6.bias"], stride=2, padding=1)
    return F.elu(y) + 1.0
