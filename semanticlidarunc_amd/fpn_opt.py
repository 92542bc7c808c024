"""The `semanticFCN_opt` variant of the ResNet-FPN segmenter on MI355X -- what the reference's `train_semantics.py:134` builds for
`baseline: Reichert` (src/baselines/Reichert/semanticFCN_opt.py:109-455) -- for the resnet18 / resnet34 / resnet50 backbones.

Same encoder as `fpn.SemanticNetworkWithFPN` (shared code); the head differs:
  SpatialAttention (:73-85)   1x1 (C -> C/8, no bias) + ReLU -> 1x1 (-> 1) -> softmax over H*W -> x * w + x
                              = two fused conv launches + `slu_spatial_softmax_gate`
  UpsampleBlock (:10-28)      bilinear x8 / x4 / x2 (align_corners=False) -> conv3x3 (no bias) -> GroupNorm -> ReLU
                              = `slu_bilinear_upsample` + one conv launch + `slu_groupnorm_fwd` (ReLU fused)
  dropout_pyramid (:266)      nn.Dropout2d(0.1) on cat([x1, x2, x3, x4]): drawn by the real child on a [B,C,1,1] tensor of ones and folded
                              into the first decoder conv as per-(sample, channel) input multipliers -- so MC-dropout
                              (`utils.mc_dropout`) works on this model, unlike on `models/semanticFCN.py` which has no dropout
  decoder (:286-296)          conv3x3 -> GN -> ReLU -> conv3x3 -> GN -> ReLU -> UpsampleBlock(x2) -> conv1x1: RAW LOGITS (no ELU)
Contract kept: constructor keywords, `forward(x, meta) -> logits [B,num_classes,H,W]`, `state_dict` keys / shapes of the reference
class (checked against it in tools/gen_golden_r02.py), genuine `nn.Dropout2d` / `nn.GroupNorm` children.  Inference: folded launches
(`_forward`); training / any gradient: one autograd node per layer (`_forward_train_opt`), ResNet and EfficientNetV2 encoders alike.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from . import effnet as _eff
from . import ops
from .fpn import SemanticNetworkWithFPN as _FPNBase
from .fpn import _RESNETS
from .ops import ConvSource


class UpsampleBlock(nn.Module):
    """Interpolate -> 3x3 conv -> GroupNorm -> ReLU (module layout of the reference, :10-28)."""

    def __init__(self, in_ch: int, out_ch: int, scale: int, mode: str = "bilinear", groups: int = 8):
        super().__init__()
        self.scale, self.mode = scale, mode
        self.block = nn.Sequential(nn.Conv2d(in_ch, out_ch, 3, padding=1, bias=False), nn.GroupNorm(math.gcd(groups, out_ch) or 1, out_ch),
                                   nn.ReLU(inplace=True))


def GN(channels, groups=32):
    g = min(groups, channels)
    return nn.GroupNorm(math.gcd(g, channels) or 1, channels)


class SpatialAttention(nn.Module):
    def __init__(self, in_ch, reduction=8):
        super().__init__()
        hid = max(1, in_ch // reduction)
        self.proj = nn.Conv2d(in_ch, hid, kernel_size=1, bias=False)
        self.score = nn.Conv2d(hid, 1, kernel_size=1, bias=False)


class SemanticNetworkWithFPN(_FPNBase):
    def __init__(self, backbone="resnet18", input_channels=2, meta_channel_dim=3, interpolation_mode="nearest", num_classes=3,
                 attention=True, multi_scale_meta=True):
        nn.Module.__init__(self)
        self.is_effnet = backbone in _eff.BASE_CHANNELS
        if backbone not in _RESNETS and not self.is_effnet:
            known = ("regnet_y_400mf", "regnet_y_800mf", "regnet_y_1_6gf", "regnet_y_3_2gf", "shufflenet_v2_x0_5", "shufflenet_v2_x1_0",
                     "shufflenet_v2_x1_5", "shufflenet_v2_x2_0", "squeezenet1_0")
            if backbone in known:
                raise NotImplementedError(f"backbone '{backbone}' is not implemented on the HIP path yet (resnet18 / 34 / 50 and efficientnet_v2_s / m / l are)")
            raise ValueError("Invalid ResNet type. Supported types: 'resnet18', 'resnet34', 'resnet50', 'regnet_y_400mf','regnet_y_800mf', "
                             "'regnet_y_1_6gf', 'regnet_y_3_2gf', 'shufflenet_v2_x0_5', 'shufflenet_v2_x1_0', 'shufflenet_v2_x1_5', "
                             "'shufflenet_v2_x2_0.")
        if interpolation_mode != "nearest":
            raise NotImplementedError("only interpolation_mode='nearest' (the reference default) runs on the HIP path")
        self.backbone_name, self.interpolation_mode = backbone, interpolation_mode
        self.num_classes, self.attention, self.multi_scale_meta = num_classes, attention, multi_scale_meta
        bc = self._build_effnet(backbone, input_channels, meta_channel_dim) if self.is_effnet else self._build_encoder(backbone, input_channels, meta_channel_dim)
        self.attention4, self.attention3 = SpatialAttention(bc[1]), SpatialAttention(bc[2])
        self.attention2, self.attention1 = SpatialAttention(bc[3]), SpatialAttention(bc[4])
        self.fpn_block4, self.fpn_block3 = self._fpn(bc[0], bc[1]), self._fpn(bc[1], bc[2])
        self.fpn_block2, self.fpn_block1 = self._fpn(bc[2], bc[3]), self._fpn(bc[3], bc[4])
        self.dropout_pyramid = nn.Dropout2d(p=0.1)
        # semanticFCN_opt.py:270-285: the efficientnet branch sets is_shuffle (x4 and x3 both sit at 1/8 resolution there)
        scales, out_chs = ([4, 4, 2], [bc[1] // 4, bc[2] // 4, bc[3] // 2]) if self.is_effnet else ([8, 4, 2], [bc[1] // 8, bc[2] // 4, bc[3] // 2])
        self.upsample_layer_x4 = UpsampleBlock(bc[1], out_chs[0], scale=scales[0], mode="bilinear")
        self.upsample_layer_x3 = UpsampleBlock(bc[2], out_chs[1], scale=scales[1], mode="bilinear")
        self.upsample_layer_x2 = UpsampleBlock(bc[3], out_chs[2], scale=scales[2], mode="bilinear")
        self.decoder_semantic = nn.Sequential(
            nn.Conv2d(sum(out_chs) + bc[4], bc[4], 3, padding=1, bias=False), GN(bc[4]), nn.ReLU(inplace=True),
            nn.Conv2d(bc[4], bc[4], 3, padding=1, bias=False), GN(bc[4]), nn.ReLU(inplace=True),
            UpsampleBlock(bc[4], bc[4] // 2, scale=2),
            nn.Conv2d(bc[4] // 2, num_classes, kernel_size=1))

    # ---------------- EfficientNetV2 encoder (semanticFCN_opt.py:170-180,238-247,396-404) ----------------
    def _build_effnet(self, backbone, input_channels, meta_channel_dim):
        """torchvision-shaped efficientnet_v2_{s,m,l} with the reference's surgery: features[0][0] replaced by a 3x3 / stride-1 conv over
        input + meta channels; stem = features[0], layer1..3 = features[2..4], layer4 = features[6:] (built, never called by forward)."""
        self.meta_channel_dim = meta_channel_dim
        self.backbone = _eff.EfficientNetContainer(backbone)
        f = self.backbone.features
        f[0][0] = nn.Conv2d(input_channels + meta_channel_dim, f[0][0].out_channels, kernel_size=3, stride=1, padding=1, bias=False)
        self.stem, self.layer1, self.layer2, self.layer3, self.layer4 = f[0], f[2], f[3], f[4], f[6:]
        return list(_eff.BASE_CHANNELS[backbone])

    def _cna(self, name, cna, srcs, resid=None, tail_first=0):
        """Conv2dNormActivation with a dense stride-1 conv: conv -> folded BatchNorm -> SiLU (or none) [+ resid] as one launch."""
        return self._conv(name, cna[0], cna[1], srcs, act="silu" if len(cna) > 2 else "none", resid=resid, tail_first=tail_first)

    def _cna_s2(self, name, cna, s2d, cin):
        """... with the 3x3 / stride-2 conv of a stage's first FusedMBConv: 2x2 stride-1 conv on the space-to-depth image."""
        return self._conv_s2(name, cna[0], cna[1], s2d, cin, act="silu" if len(cna) > 2 else "none")

    def _dw(self, name, cna):
        """(w [C, 9], bias [C]) of a depthwise Conv2dNormActivation with the eval BatchNorm folded in (cached per parameter version)."""
        conv, bn = cna[0], cna[1]

        def make():
            w, b = self._fold(conv.weight, None, bn)
            return w, b, 3, 1, 1
        cache = self.__dict__.setdefault("_dw_cache", {})
        key = tuple((t.data_ptr(), t._version) for t in (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var))
        hit = cache.get(name)
        if hit is None or hit[0] != key:
            w, b, *_ = make()
            hit = cache[name] = (key, w.detach().float().reshape(w.shape[0], 9).contiguous(), b.detach().float().contiguous())
        return hit[1], hit[2]

    def _eff_block(self, name, blk, x, meta_k=None):
        """One FusedMBConv / MBConv block, eval mode (StochasticDepth = identity).  meta_k: the meta channels that replace the last m input
        channels (first block of a stage, semanticFCN_opt.py:399-403)."""
        seq = blk.block
        cx = x.shape[1]
        m = 0 if meta_k is None else meta_k.shape[1]
        resid = x if blk.use_res_connect else None
        if isinstance(blk, _eff.FusedMBConv):
            first = seq[0]
            if first[0].stride[0] == 2:
                s2d = ops.space_to_depth2(x) if meta_k is None else ops.space_to_depth2_cat(x, cx - m, meta_k)
                h = self._cna_s2(name + ".0", first, s2d, cx)
            else:
                src = [ConvSource(x)] if meta_k is None else [ConvSource(meta_k), ConvSource(x, None, False, 0, cx - m)]
                h = self._cna(name + ".0", first, src, resid=resid if len(seq) == 1 else None, tail_first=m)
            return h if len(seq) == 1 else self._cna(name + ".1", seq[1], [ConvSource(h)], resid=resid)
        i = 0
        h = x
        src = [ConvSource(x)] if meta_k is None else [ConvSource(meta_k), ConvSource(x, None, False, 0, cx - m)]
        if len(seq) == 4:                                               # 1x1 expansion
            h, i = self._cna(name + ".0", seq[0], src, tail_first=m), 1
        elif meta_k is not None:
            raise NotImplementedError("meta injection into an MBConv without expansion does not occur in the V2 configurations")
        dw, se, proj = seq[i], seq[i + 1], seq[i + 2]
        w9, b9 = self._dw(f"{name}.{i}", dw)
        h = ops.dwconv3x3(h, w9, b9, dw[0].stride[0], "silu")
        c, s = se.fc1.in_channels, se.fc1.out_channels
        scale = ops.se_scale(h, se.fc1.weight.detach().reshape(s, c).contiguous(), se.fc1.bias.detach(), se.fc2.weight.detach().reshape(c, s).contiguous(),
                             se.fc2.bias.detach())
        return self._cna(f"{name}.{i + 2}", proj, [ConvSource(h, scale)], resid=resid)      # SE multiplies the projection's input per (sample, channel)

    def _eff_stage(self, lname, stage, x, meta_k):
        for bi, blk in enumerate(stage):
            x = self._eff_block(f"{lname}.{bi}", blk, x, meta_k if bi == 0 else None)
        return x

    def _encode_effnet(self, x, meta):
        if not self.multi_scale_meta:
            raise NotImplementedError("efficientnet backbones run with multi_scale_meta=True only (the reference's other branch feeds layer4 = "
                                      "features[6:] a tensor of the wrong channel count)")
        m1, m2, m3 = ops.nearest_down(meta, 2), ops.nearest_down(meta, 4), ops.nearest_down(meta, 8)
        xs = self._cna("stem", self.stem, [ConvSource(x), ConvSource(meta)])
        x1 = self._eff_stage("layer1", self.layer1, xs, None)
        x2 = self._eff_stage("layer2", self.layer2, x1, m1)
        x3 = self._eff_stage("layer3", self.layer3, x2, m2)
        x4 = ops.replace_tail(x3, m3)                                     # :404: x4 = cat(x3[:, :-m], meta3) -- layer4 is never applied
        return x1, x2, x3, x4

    # ---------------- head pieces ----------------
    def _spatial_attention(self, name, att: SpatialAttention, x):
        hid = self._conv(name + ".proj", att.proj, None, [ConvSource(x)], act="relu")
        score = self._conv(name + ".score", att.score, None, [ConvSource(hid)], act="none")
        return ops.spatial_softmax_gate(x, score)

    def _upsample_block(self, name, up: UpsampleBlock, srcs_before_interp):
        """UpsampleBlock.forward: the (single) input is interpolated, then conv -> GroupNorm -> ReLU."""
        if up.mode != "bilinear":
            raise NotImplementedError("UpsampleBlock: only mode='bilinear' (what the reference constructs) runs on the HIP path")
        xi = ops.bilinear_upsample(srcs_before_interp, up.scale)
        conv, gn = up.block[0], up.block[1]
        y = self._conv(name, conv, None, [ConvSource(xi)], act="none")
        return ops.groupnorm(y, gn.num_groups, gn.weight.detach(), gn.bias.detach(), gn.eps, relu=True, inplace=True)

    def _pyramid_dropout(self, n, c, device, scale_override: Optional[torch.Tensor]):
        """[n, c] multipliers of dropout_pyramid (None = identity), drawn by the real nn.Dropout2d child (its .training flag is what
        utils.mc_dropout.set_dropout_mode toggles)."""
        if scale_override is not None:
            return scale_override.reshape(n, c).to(device=device, dtype=torch.float32).contiguous()
        d = self.dropout_pyramid
        if not d.training or d.p == 0.0:
            return None
        return d(torch.ones((n, c, 1, 1), dtype=torch.float32, device=device)).reshape(n, c)

    def forward(self, x, meta_channel):
        return self._forward(x, meta_channel, None)

    def forward_with_dropout_scale(self, x, meta_channel, scale: torch.Tensor):
        """forward() with the dropout_pyramid multipliers given explicitly ([B, C_pyramid] or [B, C_pyramid, 1, 1]); parity tests."""
        return self._forward(x, meta_channel, scale)

    # ---------------- training path (fpn_autograd.py nodes; trainer.py:783-786 calls loss.backward() on this model) ----------------
    def _t_spatial_attention(self, name, att: SpatialAttention, x):
        from . import fpn_autograd as fa
        hid = fa.conv2d([x], att.proj.weight, None, 1, 0, 1, 0.0, None, None, self._dg(name + ".proj"))            # 1x1 -> ReLU in one node
        score = fa.conv2d([hid], att.score.weight, None, 1, 0, 1, None, None, None, self._dg(name + ".score"))
        return fa.SpatialGateFn.apply(x, score)

    def _t_upsample_block(self, name, up: UpsampleBlock, x):
        from . import fpn_autograd as fa
        if up.mode != "bilinear":
            raise NotImplementedError("UpsampleBlock: only mode='bilinear' (what the reference constructs) runs on the HIP path")
        conv, gn = up.block[0], up.block[1]
        y = fa.conv2d([fa.BilinearUpFn.apply(x, up.scale)], conv.weight, None, 3, 1, 1, None, None, None, self._dg(name))
        return fa.group_norm(gn, y, relu_after=True)

    # EfficientNetV2 encoder, training: the same blocks as _eff_block, one autograd node per layer, batch-statistics BatchNorm (eps 1e-3),
    # StochasticDepth('row') live on the residual blocks (torchvision: `result = stochastic_depth(result); result += input`)
    def _t_cna(self, name, cna, srcs, resid=None):
        """Conv2dNormActivation (dense, stride 1): conv -> BatchNorm [+ resid] -> SiLU.  The residual form is only legal without activation (the
        projection of a block): the one configuration that adds it AFTER the SiLU (a FusedMBConv with expansion 1) sits in features[1], which the
        reference never calls."""
        from . import fpn_autograd as fa
        conv, bn = cna[0], cna[1]
        if resid is not None and len(cna) > 2:
            raise NotImplementedError("residual after the activation (FusedMBConv with expansion ratio 1) is not on the reference's path")
        y = fa.conv2d(srcs, conv.weight, None, conv.kernel_size[0], conv.padding[0], 1, None, bn, resid, self._dg(name))
        return fa.silu(y) if len(cna) > 2 else y

    def _sd_noise(self, name, blk, n, c, device):
        """[n, c] multipliers of torchvision's stochastic_depth(mode='row') for this block in train mode, None when it is the identity."""
        sd = blk.stochastic_depth
        if not sd.training or sd.p == 0.0:
            return None
        override = self.__dict__.get("_sd_noise_override")
        if override is not None:
            noise = override[name].to(device=device, dtype=torch.float32).reshape(n, 1)
        else:
            survival = 1.0 - sd.p
            noise = torch.empty((n, 1), dtype=torch.float32, device=device).bernoulli_(survival)
            if survival > 0.0:
                noise.div_(survival)
        return noise.expand(n, c).contiguous()

    def _t_project(self, name, blk, proj, h, x):
        """The block's last Conv2dNormActivation (no activation) + StochasticDepth + residual."""
        from . import fpn_autograd as fa
        if not blk.use_res_connect:
            return self._t_cna(name, proj, [h])
        noise = self._sd_noise(name.rsplit(".", 1)[0], blk, x.shape[0], proj[0].out_channels, x.device)
        if noise is None:
            return self._t_cna(name, proj, [h], resid=x)
        return fa.ScaleAddFn.apply(self._t_cna(name, proj, [h]), noise, x)

    def _t_eff_block(self, name, blk, x):
        import torch.nn.functional as F
        from . import fpn_autograd as fa
        seq = blk.block
        if isinstance(blk, _eff.FusedMBConv):
            first = seq[0]
            if len(seq) == 1:
                raise NotImplementedError("FusedMBConv with expansion ratio 1 (features[1]) is not on the reference's path")
            h = fa.silu(self._t_conv_s2(name + ".0", first[0], first[1], x)) if first[0].stride[0] == 2 else self._t_cna(name + ".0", first, [x])
            return self._t_project(name + ".1", blk, seq[1], h, x)
        i, h = 0, x
        if len(seq) == 4:                                               # 1x1 expansion
            h, i = self._t_cna(name + ".0", seq[0], [x]), 1
        dw, se, proj = seq[i], seq[i + 1], seq[i + 2]
        d = fa.DepthwiseConv3x3Fn.apply(h, dw[0].weight)
        if dw[0].stride[0] == 2:
            d = fa.NearestDownFn.apply(d, 2)                            # the stride-2 depthwise conv = the stride-1 one sampled at even pixels
        d = fa.silu(fa.batch_norm(dw[1], d))
        c, sq = se.fc1.in_channels, se.fc1.out_channels
        pooled = fa.GlobalAvgPoolFn.apply(d)                            # [N, C]; the two tiny fully-connected layers run as library GEMMs
        gate = torch.sigmoid(F.linear(F.silu(F.linear(pooled, se.fc1.weight.view(sq, c), se.fc1.bias)), se.fc2.weight.view(c, sq), se.fc2.bias))
        return self._t_project(f"{name}.{i + 2}", blk, proj, fa.ChannelGateFn.apply(d, gate), x)

    def _t_encode_effnet(self, x, meta):
        from . import fpn_autograd as fa
        if not self.multi_scale_meta:
            raise NotImplementedError("efficientnet backbones run with multi_scale_meta=True only (the reference's other branch feeds layer4 = "
                                      "features[6:] a tensor of the wrong channel count)")
        m1, m2, m3 = (fa.NearestDownFn.apply(meta, f) for f in (2, 4, 8))
        h = self._t_cna("stem", self.stem, [x, meta])
        outs = []
        for lname, stage, mk in (("layer1", self.layer1, None), ("layer2", self.layer2, m1), ("layer3", self.layer3, m2)):
            if mk is not None:
                h = fa.ReplaceTailFn.apply(h, mk)                       # :399-403: the stage sees cat(x[:, :-m], meta_k)
            for bi, blk in enumerate(stage):
                h = self._t_eff_block(f"{lname}.{bi}", blk, h)
            outs.append(h)
        return outs[0], outs[1], outs[2], fa.ReplaceTailFn.apply(outs[2], m3)      # :404: x4 = cat(x3[:, :-m], meta3) -- layer4 is never applied

    def _forward_train_opt(self, x, meta, drop_scale):
        from . import fpn_autograd as fa
        from . import autograd as _ag
        _ag.nbt_scope_enter()
        try:
            x1, x2, x3, x4 = self._t_encode_effnet(x, meta) if self.is_effnet else self._t_encode(x, meta)
            f4 = self._t_cbr("fpn4", self.fpn_block4[0], self.fpn_block4[1], [x4])
            f3 = self._t_cbr("fpn3", self.fpn_block3[0], self.fpn_block3[1], [x3])
            f2 = self._t_cbr("fpn2", self.fpn_block2[0], self.fpn_block2[1], [x2])
            f1 = self._t_cbr("fpn1", self.fpn_block1[0], self.fpn_block1[1], [x1])
            if self.attention:
                f4, f3 = self._t_spatial_attention("att4", self.attention4, f4), self._t_spatial_attention("att3", self.attention3, f3)
                f2, f1 = self._t_spatial_attention("att2", self.attention2, f2), self._t_spatial_attention("att1", self.attention1, f1)
            u4 = self._t_upsample_block("up4", self.upsample_layer_x4, f4)
            u3 = self._t_upsample_block("up3", self.upsample_layer_x3, f3)
            u2 = self._t_upsample_block("up2", self.upsample_layer_x2, f2)
            n = f1.shape[0]
            c1, c2, c3, c4 = f1.shape[1], u2.shape[1], u3.shape[1], u4.shape[1]
            s = self._pyramid_dropout(n, c1 + c2 + c3 + c4, f1.device, drop_scale)
            u34 = fa.DepthToSpaceCatFn.apply((1, 1), u3, u4)              # channel concatenation (the fused conv takes up to three sources)
            sc = None if s is None else (s[:, :c1].contiguous(), s[:, c1:c1 + c2].contiguous(), s[:, c1 + c2:].contiguous())
            d = self.decoder_semantic
            y = fa.conv2d([f1, u2, u34], d[0].weight, None, 3, 1, 1, None, None, None, self._dg("dec0"), scales=sc)
            y = fa.group_norm(d[1], y, relu_after=True)
            y = fa.conv2d([y], d[3].weight, None, 3, 1, 1, None, None, None, self._dg("dec1"))
            y = fa.group_norm(d[4], y, relu_after=True)
            y = self._t_upsample_block("dec_up", d[6], y)
            return fa.conv2d([y], d[7].weight, d[7].bias, 1, 0, 1, None, None, None, self._dg("dec_out"))
        finally:
            _ag.nbt_scope_exit()

    def _forward(self, x, meta_channel, drop_scale):
        x, meta = self._check_inputs(x, meta_channel)
        if self._wants_autograd(x, meta):
            return self._forward_train_opt(x, meta, drop_scale)
        if self.is_effnet:
            x1, x2, x3, x4 = self._encode_effnet(x, meta)
        else:
            x1, x2, x3, x4 = self._encode(x, meta)
        f4 = self._conv("fpn4", self.fpn_block4[0], self.fpn_block4[1], [ConvSource(x4)])
        f3 = self._conv("fpn3", self.fpn_block3[0], self.fpn_block3[1], [ConvSource(x3)])
        f2 = self._conv("fpn2", self.fpn_block2[0], self.fpn_block2[1], [ConvSource(x2)])
        f1 = self._conv("fpn1", self.fpn_block1[0], self.fpn_block1[1], [ConvSource(x1)])
        if self.attention:
            f4, f3 = self._spatial_attention("att4", self.attention4, f4), self._spatial_attention("att3", self.attention3, f3)
            f2, f1 = self._spatial_attention("att2", self.attention2, f2), self._spatial_attention("att1", self.attention1, f1)
        u4 = self._upsample_block("up4", self.upsample_layer_x4, f4)
        u3 = self._upsample_block("up3", self.upsample_layer_x3, f3)
        u2 = self._upsample_block("up2", self.upsample_layer_x2, f2)
        # cat([x1, x2, x3, x4]) -> dropout_pyramid -> decoder conv: three sources (x1 | x2 | x3 + x4 share a buffer), multipliers per source
        n = f1.shape[0]
        c1, c2, c3, c4 = f1.shape[1], u2.shape[1], u3.shape[1], u4.shape[1]
        s = self._pyramid_dropout(n, c1 + c2 + c3 + c4, f1.device, drop_scale)
        u34 = torch.cat([u3, u4], dim=1)              # the fused conv takes up to three sources
        sc = (None, None, None) if s is None else (s[:, :c1].contiguous(), s[:, c1:c1 + c2].contiguous(), s[:, c1 + c2:].contiguous())
        d = self.decoder_semantic
        y = self._conv("dec0", d[0], None, [ConvSource(f1, sc[0]), ConvSource(u2, sc[1]), ConvSource(u34, sc[2])], act="none")
        y = ops.groupnorm(y, d[1].num_groups, d[1].weight.detach(), d[1].bias.detach(), d[1].eps, relu=True, inplace=True)
        y = self._conv("dec1", d[3], None, [ConvSource(y)], act="none")
        y = ops.groupnorm(y, d[4].num_groups, d[4].weight.detach(), d[4].bias.detach(), d[4].eps, relu=True, inplace=True)
        y = self._upsample_block("dec_up", d[6], y)
        return self._conv("dec_out", d[7], None, [ConvSource(y)], act="none")
