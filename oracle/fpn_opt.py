"""Oracle: the `semanticFCN_opt` model as a pure function of a ``state_dict`` (plain torch CPU ops).  TEST INFRASTRUCTURE ONLY.

Restates ``src/baselines/Reichert/semanticFCN_opt.py`` -- ``UpsampleBlock`` :10-28, ``SpatialAttention`` :73-85, ``GN`` :66-70, the
head of ``__init__`` :256-296 and ``forward`` :366-455 (resnet branch) -- on top of ``oracle.fpn``'s restated torchvision BasicBlock
encoder (backbone parity unpinned, see there).  The head wiring IS pinned: ``tools/gen_golden_r02.py`` imports the reference's own
class through the stub ``torchvision.models``, loads the same state_dict and compares (tests/golden/fpn_opt_*.npz).  The
``dropout_pyramid`` (nn.Dropout2d(0.1), :266,450) is an explicit per-(sample, channel) multiplier."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from oracle.fpn import LAYERS, _bn, _cbr, _stage


def _gn(x, sd, p, groups):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _groups(sd, p, cap):
    import math
    c = sd[p + ".weight"].numel()
    return math.gcd(min(cap, c), c) or 1


def _spatial_attention(x, sd, p):                    # :80-85
    s = F.conv2d(F.relu(F.conv2d(x, sd[p + ".proj.weight"])), sd[p + ".score.weight"])
    b, _, h, w = s.shape
    wgt = torch.softmax(s.view(b, 1, h * w), dim=-1).view(b, 1, h, w)
    return x * wgt + x


def _upsample_block(x, sd, p, scale):                 # :24-28, gn_groups = gcd(8, out_ch)
    x = F.interpolate(x, scale_factor=scale, mode="bilinear", align_corners=False)
    y = F.conv2d(x, sd[p + ".block.0.weight"], None, padding=1)
    return F.relu(_gn(y, sd, p + ".block.1", _groups(sd, p + ".block.1", 8)))


def fpn_opt_forward(sd, x, meta, backbone="resnet18", attention=True, multi_scale_meta=True, dropout_scale=None):
    """logits [B,num_classes,H,W] = SemanticNetworkWithFPN(x, meta) of semanticFCN_opt.py:366-455; dropout_scale [B,C,1,1] or None."""
    if backbone.startswith("efficientnet_v2"):
        from oracle import effnet
        x1, x2, x3, x4 = effnet.encode(sd, x, meta, backbone, multi_scale_meta)
        return _head(sd, x1, x2, x3, x4, attention, dropout_scale, (4, 4, 2))
    layers = LAYERS[backbone]
    m = meta.shape[1]
    h = torch.cat([x, meta], 1)
    xs = F.max_pool2d(F.relu(F.conv2d(h, sd["backbone.conv1.weight"], None, padding=1)), 3, 2, 1)
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
    return _head(sd, x1, x2, x3, x4, attention, dropout_scale, (8, 4, 2))


def _head(sd, x1, x2, x3, x4, attention, dropout_scale, scales):
    """FPN blocks, SpatialAttention, UpsampleBlocks (scale factors per backbone family, semanticFCN_opt.py:270-285), pyramid dropout, decoder."""
    f4, f3, f2, f1 = _cbr(x4, sd, "fpn_block4"), _cbr(x3, sd, "fpn_block3"), _cbr(x2, sd, "fpn_block2"), _cbr(x1, sd, "fpn_block1")
    if attention:
        f4, f3 = _spatial_attention(f4, sd, "attention4"), _spatial_attention(f3, sd, "attention3")
        f2, f1 = _spatial_attention(f2, sd, "attention2"), _spatial_attention(f1, sd, "attention1")
    u4, u3, u2 = (_upsample_block(f4, sd, "upsample_layer_x4", scales[0]), _upsample_block(f3, sd, "upsample_layer_x3", scales[1]),
                  _upsample_block(f2, sd, "upsample_layer_x2", scales[2]))
    y = torch.cat([f1, u2, u3, u4], 1)
    if dropout_scale is not None:
        y = y * dropout_scale.reshape(y.shape[0], y.shape[1], 1, 1)
    d = "decoder_semantic"
    y = F.relu(_gn(F.conv2d(y, sd[d + ".0.weight"], None, padding=1), sd, d + ".1", _groups(sd, d + ".1", 32)))
    y = F.relu(_gn(F.conv2d(y, sd[d + ".3.weight"], None, padding=1), sd, d + ".4", _groups(sd, d + ".4", 32)))
    y = _upsample_block(y, sd, d + ".6", 2)
    return F.conv2d(y, sd[d + ".7.weight"], sd[d + ".7.bias"])
