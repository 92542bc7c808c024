"""EfficientNetV2 backbone container for the `semanticFCN_opt` segmenter (reference baselines/Reichert/semanticFCN_opt.py:170-180 builds
torchvision's `efficientnet_v2_{s,m,l}`, :238-247 replaces `features[0][0]` by a 3x3 / stride-1 conv over input + meta channels and wires
`stem = features[0]`, `layer1..3 = features[2..4]`, `layer4 = features[6:]`; its forward (:396-404) never calls layer4 nor features[1] / [5]).

torchvision is not needed: the classes below re-create the public architecture of torchvision 0.19's `efficientnet.py` -- module names,
parameter shapes and therefore `state_dict` keys -- without a forward (the arithmetic lives in fpn_opt.py on the HIP kernels); `pretrained`
weights are not downloaded, load a checkpoint.  **Parity of the block internals is unpinned** (no reference-held fixture): the
oracle (oracle/effnet.py) restates the same public definition, and the reference's OWN head / meta-injection wiring is pinned through it."""
from __future__ import annotations

import math
from typing import List, Tuple

import torch.nn as nn

# (block kind, expand ratio, kernel, stride, input channels, output channels, layers) -- torchvision's _efficientnet_conf for the V2 family
_CONFIGS = {
    "efficientnet_v2_s": ([("fused", 1, 3, 1, 24, 24, 2), ("fused", 4, 3, 2, 24, 48, 4), ("fused", 4, 3, 2, 48, 64, 4), ("mb", 4, 3, 2, 64, 128, 6),
                           ("mb", 6, 3, 1, 128, 160, 9), ("mb", 6, 3, 2, 160, 256, 15)], 1280, 0.2),
    "efficientnet_v2_m": ([("fused", 1, 3, 1, 24, 24, 3), ("fused", 4, 3, 2, 24, 48, 5), ("fused", 4, 3, 2, 48, 80, 5), ("mb", 4, 3, 2, 80, 160, 7),
                           ("mb", 6, 3, 1, 160, 176, 14), ("mb", 6, 3, 2, 176, 304, 18), ("mb", 6, 3, 1, 304, 512, 5)], 1280, 0.3),
    "efficientnet_v2_l": ([("fused", 1, 3, 1, 32, 32, 4), ("fused", 4, 3, 2, 32, 64, 7), ("fused", 4, 3, 2, 64, 96, 7), ("mb", 4, 3, 2, 96, 192, 10),
                           ("mb", 6, 3, 1, 192, 224, 19), ("mb", 6, 3, 2, 224, 384, 25), ("mb", 6, 3, 1, 384, 640, 7)], 1280, 0.4),
}
# channel ladder the reference hard-codes per backbone (semanticFCN_opt.py:170-180): [x4, fpn4 out, fpn3 out, fpn2 out, fpn1 out]
BASE_CHANNELS = {"efficientnet_v2_s": [128, 128, 64, 48, 168], "efficientnet_v2_m": [160, 160, 80, 48, 168], "efficientnet_v2_l": [192, 192, 96, 64, 168]}
BN_EPS = 1e-3      # torchvision builds the V2 family with norm_layer = partial(nn.BatchNorm2d, eps=1e-03)


def _make_divisible(v: float, divisor: int = 8) -> int:
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


class Conv2dNormActivation(nn.Sequential):
    """conv (bias-free) -> BatchNorm2d(eps 1e-3) -> SiLU (children '0', '1', '2' as in torchvision.ops.misc)."""

    def __init__(self, cin: int, cout: int, kernel_size: int = 3, stride: int = 1, groups: int = 1, act: bool = True):
        layers: List[nn.Module] = [nn.Conv2d(cin, cout, kernel_size, stride, (kernel_size - 1) // 2, groups=groups, bias=False),
                                   nn.BatchNorm2d(cout, eps=BN_EPS)]
        if act:
            layers.append(nn.SiLU(inplace=True))
        super().__init__(*layers)
        self.out_channels = cout


class SqueezeExcitation(nn.Module):
    def __init__(self, channels: int, squeeze: int):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = nn.Conv2d(channels, squeeze, 1)
        self.fc2 = nn.Conv2d(squeeze, channels, 1)
        self.activation = nn.SiLU(inplace=True)
        self.scale_activation = nn.Sigmoid()


class StochasticDepth(nn.Module):
    def __init__(self, p: float, mode: str = "row"):
        super().__init__()
        self.p, self.mode = p, mode


class FusedMBConv(nn.Module):
    def __init__(self, expand, kernel, stride, cin, cout, sd_prob):
        super().__init__()
        self.use_res_connect = stride == 1 and cin == cout
        expanded = _make_divisible(cin * expand)
        if expanded != cin:
            layers = [Conv2dNormActivation(cin, expanded, kernel, stride), Conv2dNormActivation(expanded, cout, 1, act=False)]
        else:
            layers = [Conv2dNormActivation(cin, cout, kernel, stride)]
        self.block = nn.Sequential(*layers)
        self.stochastic_depth = StochasticDepth(sd_prob)
        self.out_channels, self.stride = cout, stride


class MBConv(nn.Module):
    def __init__(self, expand, kernel, stride, cin, cout, sd_prob):
        super().__init__()
        self.use_res_connect = stride == 1 and cin == cout
        expanded = _make_divisible(cin * expand)
        layers: List[nn.Module] = []
        if expanded != cin:
            layers.append(Conv2dNormActivation(cin, expanded, 1))
        layers.append(Conv2dNormActivation(expanded, expanded, kernel, stride, groups=expanded))
        layers.append(SqueezeExcitation(expanded, max(1, cin // 4)))
        layers.append(Conv2dNormActivation(expanded, cout, 1, act=False))
        self.block = nn.Sequential(*layers)
        self.stochastic_depth = StochasticDepth(sd_prob)
        self.out_channels, self.stride = cout, stride


class EfficientNetContainer(nn.Module):
    """`features` / `avgpool` / `classifier` of torchvision's EfficientNet for the V2 configurations (names and shapes only)."""

    def __init__(self, name: str, num_classes: int = 1000, stochastic_depth_prob: float = 0.2):
        super().__init__()
        conf, last_channel, dropout = _CONFIGS[name]
        layers: List[nn.Module] = [Conv2dNormActivation(3, conf[0][4], 3, 2)]
        total = float(sum(c[6] for c in conf))
        bid = 0
        for kind, expand, kernel, stride, cin, cout, n in conf:
            stage = []
            for k in range(n):
                blk = FusedMBConv if kind == "fused" else MBConv
                stage.append(blk(expand, kernel, stride if k == 0 else 1, cin if k == 0 else cout, cout, stochastic_depth_prob * bid / total))
                bid += 1
            layers.append(nn.Sequential(*stage))
        layers.append(Conv2dNormActivation(conf[-1][5], last_channel, 1))
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.classifier = nn.Sequential(nn.Dropout(p=dropout, inplace=True), nn.Linear(last_channel, num_classes))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                r = 1.0 / math.sqrt(m.out_features)
                nn.init.uniform_(m.weight, -r, r)
                nn.init.zeros_(m.bias)


def stage_channels(name: str) -> Tuple[int, int, int, int]:
    """(stem, features[2], features[3], features[4]) output channels."""
    conf = _CONFIGS[name][0]
    return conf[0][4], conf[1][5], conf[2][5], conf[3][5]
