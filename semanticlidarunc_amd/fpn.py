"""ResNet-FPN range-image segmenter on MI355X: the reference's `SemanticNetworkWithFPN` contract
(src/models/semanticFCN.py:76-354) over the HIP conv kernel plus a few data-movement kernels.

Contract kept: constructor keywords (plus the stale `resnet_type=` alias that src/inference_ouster.py:35 still passes),
`forward(x[B,cin,H,W], meta[B,m,H,W]) -> [B,num_classes,H,W]` (ELU + 1 > 0), and the `state_dict` layout including the
aliased `stem.0.* / layerN.*` duplicates of `backbone.*` and the never-used `backbone.bn1.* / backbone.fc.*`
(semanticFCN.py:146-153).  torchvision is not needed: the backbone container below re-creates the public
resnet18/34 BasicBlock architecture (names, shapes); `pretrained` weights are not downloaded -- load a checkpoint.

How the layers map onto kernels (inference; BatchNorm folded into the preceding conv's weights and bias):
  conv3x3/s1 + BN + ReLU ........ one fused conv launch (ReLU = leaky slope 0)
  BasicBlock tail ............... conv + folded BN + identity, ReLU applied after the residual add (late activation)
  conv3x3/s2 (stage entry) ...... space-to-depth of cat(x[:, :-m], meta_k) (one kernel) + a 2x2 stride-1 conv with
                                  re-indexed weights; the 1x1/s2 downsample conv reads phase (0,0) of the same tensor
  MaxPool(3,2,1), nearest 1/2,1/4,1/8 of meta ... one kernel each
  attention ..................... q + k = ONE 1x1 conv with summed weights, tanh in its epilogue; 1x1 -> score;
                                  softmax over azimuth fused with the multiply into the value map
  ConvTranspose2d k = s ......... 1x1 conv to Cout*s*s channels + depth-to-space
  ConvTranspose2d k4 s2 p1 ...... 3x3 conv to 4*classes sub-pixel channels + depth-to-space fused with ELU + 1
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from . import salsanext as _sn
from .ops import ConvSource


# ----------------------------------------------------------------------------------------------------------------
# torchvision-compatible ResNet container (names and shapes only; forward lives in the FPN class)
# ----------------------------------------------------------------------------------------------------------------
class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)      # torchvision's v1.5: the stride sits on the 3x3
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


class ResNetContainer(nn.Module):
    def __init__(self, layers: List[int], block=None):
        super().__init__()
        self.block = block or BasicBlock
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * self.block.expansion, 1000)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, planes, blocks, stride):
        down, e = None, self.block.expansion
        if stride != 1 or self.inplanes != planes * e:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * e, 1, stride, bias=False), nn.BatchNorm2d(planes * e))
        seq = [self.block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * e
        seq += [self.block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*seq)


_RESNETS = {"resnet18": ([2, 2, 2, 2], BasicBlock), "resnet34": ([3, 4, 6, 3], BasicBlock), "resnet50": ([3, 4, 6, 3], Bottleneck)}


class AttentionModule(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.query_conv = nn.Conv2d(in_channels, out_channels, 1)
        self.key_conv = nn.Conv2d(in_channels, out_channels, 1)
        self.value_conv = nn.Conv2d(in_channels, out_channels, 1)
        self.attention_conv = nn.Conv2d(out_channels, 1, 1)
        self.softmax = nn.Softmax(dim=-1)


# ----------------------------------------------------------------------------------------------------------------
class _Packed:
    __slots__ = ("key", "wpack", "bias", "cout", "k", "dil", "pad", "precision")


def _key(*ts):
    return tuple((t.data_ptr(), t._version) for t in ts if t is not None) + (_sn.get_conv_precision(),)


class SemanticNetworkWithFPN(nn.Module):
    def __init__(self, backbone="resnet18", input_channels=2, meta_channel_dim=3, interpolation_mode="nearest", num_classes=3,
                 attention=True, multi_scale_meta=True, resnet_type: Optional[str] = None):
        super().__init__()
        if resnet_type is not None:          # stale keyword of src/inference_ouster.py:35
            backbone = resnet_type
        if backbone not in _RESNETS:
            if backbone in ("regnet_y_400mf", "regnet_y_800mf", "regnet_y_1_6gf", "regnet_y_3_2gf", "shufflenet_v2_x0_5",
                            "shufflenet_v2_x1_0", "shufflenet_v2_x1_5", "shufflenet_v2_x2_0", "squeezenet1_0", "efficientnet_v2_s",
                            "efficientnet_v2_m", "efficientnet_v2_l"):
                raise NotImplementedError(f"backbone '{backbone}' is not implemented on the HIP path yet (resnet18 / resnet34 / resnet50 are)")
            raise ValueError("Invalid ResNet type. Supported types: 'resnet18', 'resnet34', 'resnet50', 'regnet_y_400mf','regnet_y_800mf', "
                             "'regnet_y_1_6gf', 'regnet_y_3_2gf', 'shufflenet_v2_x0_5', 'shufflenet_v2_x1_0', 'shufflenet_v2_x1_5', "
                             "'shufflenet_v2_x2_0.")
        if interpolation_mode != "nearest":
            raise NotImplementedError("only interpolation_mode='nearest' (the reference default) runs on the HIP path")
        self.backbone_name, self.interpolation_mode = backbone, interpolation_mode
        self.num_classes, self.attention, self.multi_scale_meta = num_classes, attention, multi_scale_meta
        bc = self._build_encoder(backbone, input_channels, meta_channel_dim)
        self.attention4, self.attention3 = AttentionModule(bc[1], bc[1]), AttentionModule(bc[2], bc[2])
        self.attention2, self.attention1 = AttentionModule(bc[3], bc[3]), AttentionModule(bc[4], bc[4])
        self.fpn_block4, self.fpn_block3 = self._fpn(bc[0], bc[1]), self._fpn(bc[1], bc[2])
        self.fpn_block2, self.fpn_block1 = self._fpn(bc[2], bc[3]), self._fpn(bc[3], bc[4])
        self.upsample_layer_x4 = nn.ConvTranspose2d(bc[1], bc[1] // 8, 8, 8, 0)
        self.upsample_layer_x3 = nn.ConvTranspose2d(bc[2], bc[2] // 4, 4, 4, 0)
        self.upsample_layer_x2 = nn.ConvTranspose2d(bc[3], bc[3] // 2, 2, 2, 0)
        up = bc[1] // 8 + bc[2] // 4 + bc[3] // 2
        self.decoder_semantic = nn.Sequential(
            nn.Conv2d(up + bc[4], bc[4], 3, 1, 1), nn.BatchNorm2d(bc[4]), nn.ReLU(inplace=True),
            nn.Conv2d(bc[4], bc[4], 3, 1, 1), nn.BatchNorm2d(bc[4]), nn.ReLU(inplace=True),
            nn.ConvTranspose2d(bc[4], num_classes, 4, 2, 1), nn.ELU(inplace=True))

    def _build_encoder(self, backbone, input_channels, meta_channel_dim):
        """The torchvision-shaped ResNet with the reference's surgery (semanticFCN.py:146-153): conv1 replaced by a 3x3 / stride-1 conv
        over input + meta channels, `stem` = conv1 -> relu -> maxpool (bn1 skipped), layerN aliases.  Returns the channel ladder."""
        self.meta_channel_dim = meta_channel_dim
        layers, block = _RESNETS[backbone]
        self.backbone = ResNetContainer(layers, block)
        self.backbone.conv1 = nn.Conv2d(input_channels + meta_channel_dim, 64, 3, 1, 1, bias=False)
        self.stem = nn.Sequential(self.backbone.conv1, self.backbone.relu, self.backbone.maxpool)
        self.layer1, self.layer2 = self.backbone.layer1, self.backbone.layer2
        self.layer3, self.layer4 = self.backbone.layer3, self.backbone.layer4
        top = 512 * block.expansion                      # semanticFCN.py:85-96: 512 (resnet18 / 34), 2048 (resnet50)
        return [top, top // 2, top // 4, top // 8, top // 16]

    def _check_inputs(self, x, meta_channel):
        if not (isinstance(x, torch.Tensor) and isinstance(meta_channel, torch.Tensor)) or x.dim() != 4 or meta_channel.dim() != 4:
            raise RuntimeError("SemanticNetworkWithFPN expects x[B,c,H,W] and meta_channel[B,m,H,W]")
        if not x.is_cuda:
            raise RuntimeError(f"semanticlidarunc_amd FPN runs on MI355X only: input is on '{x.device}' and there is no CPU fallback")
        if x.shape[2] % 16 or x.shape[3] % 16:
            raise RuntimeError("SemanticNetworkWithFPN needs H and W divisible by 16")
        return x.contiguous().float(), meta_channel.contiguous().float()

    def _wants_autograd(self, x, meta) -> bool:
        """The layer-by-layer autograd path (fpn_autograd.py) instead of the folded inference launches: whenever a gradient may be asked for, or
        a BatchNorm is in train mode (batch statistics cannot be folded into the conv weights)."""
        if any(isinstance(m, nn.modules.batchnorm._BatchNorm) and m.training for m in self.modules()):
            return True
        return torch.is_grad_enabled() and (x.requires_grad or meta.requires_grad or any(p.requires_grad for p in self.parameters()))

    def _encode(self, x, meta):
        """(x1, x2, x3, x4): stem + the four ResNet stages with the multi-scale meta injection (semanticFCN.py:279-314)."""
        m1 = m2 = m3 = None
        if self.multi_scale_meta:
            m1, m2, m3 = ops.nearest_down(meta, 2), ops.nearest_down(meta, 4), ops.nearest_down(meta, 8)
        xs = self._conv("stem", self.backbone.conv1, None, [ConvSource(x), ConvSource(meta)])       # bn1 is skipped by the reference stem
        xs = ops.maxpool3s2(xs)
        x1 = self._stage("layer1", self.layer1, xs, None)
        x2 = self._stage("layer2", self.layer2, x1, m1)
        x3 = self._stage("layer3", self.layer3, x2, m2)
        x4 = self._stage("layer4", self.layer4, x3, m3)
        return x1, x2, x3, x4

    @staticmethod
    def _fpn(cin, cout):
        return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))

    # ---------------- weight preparation (cached per parameter version and conv precision) ----------------
    def _prep(self, name: str, maker, *deps) -> _Packed:
        cache: Dict[str, _Packed] = self.__dict__.setdefault("_packed", {})
        p = cache.get(name)
        k = _key(*deps)
        if p is None or p.key != k:
            w, b, ksz, dil, pad = maker()
            p = _Packed()
            p.key, p.cout, p.k, p.dil, p.pad = k, w.shape[0], ksz, dil, pad
            p.precision = _sn.get_conv_precision()
            w = w.detach().float().contiguous()
            p.wpack = ops.pack_conv_weight_f16x3(w) if p.precision == "f16x3" else ops.pack_conv_weight(w)
            p.bias = None if b is None else b.detach().float().contiguous()
            cache[name] = p
        return p

    @staticmethod
    def _fold(conv_w, conv_b, bn: Optional[nn.BatchNorm2d]):
        """(w', b') with w'*x + b' == bn(conv(x))  (eval-mode BatchNorm after the conv)."""
        if bn is None:
            return conv_w, conv_b
        if bn.training:
            raise NotImplementedError("the FPN model runs in eval mode only on the HIP path (BatchNorm is folded into the convs)")
        a = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
        b = bn.bias.detach() - bn.running_mean * a
        if conv_b is not None:
            b = b + conv_b.detach() * a
        return conv_w.detach() * a.view(-1, 1, 1, 1), b

    def _conv(self, name, conv: nn.Conv2d, bn, srcs, act="relu", resid=None, late=False, tail_first: int = 0):
        """tail_first = m > 0: the sources are given as (last m input channels, the rest) -- the packed weight's input channels are rotated
        to match (a channel prefix `cuse` is only honoured on the LAST source of a fused conv)."""
        def make():
            w, b = self._fold(conv.weight, conv.bias, bn)
            if tail_first:
                w = torch.cat([w[:, -tail_first:], w[:, :-tail_first]], 1)
            return w, b, conv.kernel_size[0], conv.dilation[0], conv.padding[0]
        p = self._prep(name, make,
                       conv.weight, conv.bias, *(() if bn is None else (bn.weight, bn.bias, bn.running_mean, bn.running_var)))
        return ops.conv2d_fused(srcs, p.wpack, p.cout, p.k, p.dil, p.pad, bias=p.bias, resid=resid, precision=p.precision,
                                act=act, act_after_resid=late)

    def _conv_s2(self, name, conv: nn.Conv2d, bn, s2d, cin, act="relu"):
        """3x3 / stride 2 / pad 1 conv of the tensor whose space-to-depth image is `s2d` ([N, 4*cin, H/2, W/2])."""
        def make():
            w, b = self._fold(conv.weight, conv.bias, bn)
            cout = w.shape[0]
            w2 = torch.zeros((cout, 4, cin, 2, 2), dtype=w.dtype, device=w.device)
            tap = {0: (1, 0), 1: (0, 1), 2: (1, 1)}          # kernel index -> (phase, 2x2 tap): rows 2y-1, 2y, 2y+1
            for i, (p, a) in tap.items():
                for j, (q, bb) in tap.items():
                    w2[:, 2 * p + q, :, a, bb] = w[:, :, i, j]
            return w2.reshape(cout, 4 * cin, 2, 2), b, 2, 1, 1
        p = self._prep(name, make, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)
        return ops.conv2d_fused([ConvSource(s2d)], p.wpack, p.cout, 2, 1, 1, bias=p.bias, precision=p.precision, act=act)

    def _convT_eq_stride(self, name, ct: nn.ConvTranspose2d, x, out=None, c_off=0):
        s = ct.stride[0]

        def make():
            w = ct.weight.detach()                               # [Cin, Cout, s, s]
            cin, cout = w.shape[0], w.shape[1]
            wc = w.permute(1, 2, 3, 0).reshape(cout * s * s, cin, 1, 1)
            b = None if ct.bias is None else ct.bias.detach().repeat_interleave(s * s)
            return wc, b, 1, 1, 0
        p = self._prep(name, make, ct.weight, ct.bias)
        y = ops.conv2d_fused([ConvSource(x)], p.wpack, p.cout, 1, 1, 0, bias=p.bias, precision=p.precision, act="none")
        return ops.depth_to_space(y, s, False, out, c_off)

    def _convT_k4s2p1(self, name, ct: nn.ConvTranspose2d, x, elu_plus_one):
        def make():
            w = ct.weight.detach()                               # [Cin, Cout, 4, 4]
            cin, cout = w.shape[0], w.shape[1]
            wf = torch.zeros((cout, 2, 2, cin, 3, 3), dtype=w.dtype, device=w.device)
            pairs = {0: ((0, 1), (-1, 3)), 1: ((1, 0), (0, 2))}   # output parity -> ((input offset, kernel index), ...)
            for py, ys in pairs.items():
                for px, xs in pairs.items():
                    for dy, i in ys:
                        for dx, j in xs:
                            wf[:, py, px, :, dy + 1, dx + 1] = w[:, :, i, j].t()
            b = None if ct.bias is None else ct.bias.detach().repeat_interleave(4)
            return wf.reshape(cout * 4, cin, 3, 3), b, 3, 1, 1
        p = self._prep(name, make, ct.weight, ct.bias)
        y = ops.conv2d_fused([ConvSource(x)], p.wpack, p.cout, 3, 1, 1, bias=p.bias, precision=p.precision, act="none")
        return ops.depth_to_space(y, 2, elu_plus_one)

    # ---------------- forward ----------------
    def _stage(self, lname: str, layer: nn.Sequential, x, meta_k):
        """One ResNet stage.  meta_k: the down-sampled meta channels that overwrite the last m channels of x (or None)."""
        for bi, blk in enumerate(layer):
            n = f"{lname}.{bi}"
            if isinstance(blk, Bottleneck):
                x = self._bottleneck(n, blk, x, meta_k if bi == 0 else None)
                continue
            if bi == 0 and blk.stride == 2:
                cx = x.shape[1]
                if meta_k is not None:
                    s2d, cin = ops.space_to_depth2_cat(x, cx - meta_k.shape[1], meta_k), cx
                else:
                    s2d, cin = ops.space_to_depth2(x), cx
                o1 = self._conv_s2(n + ".conv1", blk.conv1, blk.bn1, s2d, cin)
                dconv, dbn = blk.downsample[0], blk.downsample[1]
                idn = self._conv(n + ".down", dconv, dbn, [ConvSource(s2d, None, False, 0, cin)], act="none")   # phase (0,0) == stride-2 sampling
            else:
                if bi == 0 and meta_k is not None:
                    srcs = [ConvSource(x, None, False, 0, x.shape[1] - meta_k.shape[1]), ConvSource(meta_k)]
                    raise NotImplementedError("meta injection into a stride-1 stage does not occur for resnet18/34")
                o1 = self._conv(n + ".conv1", blk.conv1, blk.bn1, [ConvSource(x)])
                idn = x
            x = self._conv(n + ".conv2", blk.conv2, blk.bn2, [ConvSource(o1)], act="relu", resid=idn, late=True)
        return x

    def _bottleneck(self, n: str, blk: Bottleneck, x, meta_k):
        """conv1x1-BN-ReLU, conv3x3(stride)-BN-ReLU, conv1x1-BN, + identity (1x1-stride conv + BN where the shape changes), ReLU.  meta_k: the
        meta channels that overwrite the last m channels of x on the way in (first block of a stride-2 stage)."""
        cx = x.shape[1]
        m = 0 if meta_k is None else meta_k.shape[1]
        src = [ConvSource(x)] if meta_k is None else [ConvSource(meta_k), ConvSource(x, None, False, 0, cx - m)]      # (meta, x[:, :-m])
        o1 = self._conv(n + ".conv1", blk.conv1, blk.bn1, src, tail_first=m)
        if blk.stride == 2:
            o2 = self._conv_s2(n + ".conv2", blk.conv2, blk.bn2, ops.space_to_depth2(o1), o1.shape[1])
            s2d = ops.space_to_depth2(x) if meta_k is None else ops.space_to_depth2_cat(x, cx - meta_k.shape[1], meta_k)
            idn = self._conv(n + ".down", blk.downsample[0], blk.downsample[1], [ConvSource(s2d, None, False, 0, cx)], act="none")   # phase (0,0)
        else:
            o2 = self._conv(n + ".conv2", blk.conv2, blk.bn2, [ConvSource(o1)])
            idn = x if blk.downsample is None else self._conv(n + ".down", blk.downsample[0], blk.downsample[1], src, act="none", tail_first=m)
        return self._conv(n + ".conv3", blk.conv3, blk.bn3, [ConvSource(o2)], act="relu", resid=idn, late=True)

    def _attend(self, name, att: AttentionModule, x):
        def make_qk():
            return att.query_conv.weight.detach() + att.key_conv.weight.detach(), att.query_conv.bias.detach() + att.key_conv.bias.detach(), 1, 1, 0
        pqk = self._prep(name + ".qk", make_qk, att.query_conv.weight, att.key_conv.weight, att.query_conv.bias, att.key_conv.bias)
        t = ops.conv2d_fused([ConvSource(x)], pqk.wpack, pqk.cout, 1, 1, 0, bias=pqk.bias, precision=pqk.precision, act="tanh")
        score = self._conv(name + ".score", att.attention_conv, None, [ConvSource(t)], act="none")
        value = self._conv(name + ".value", att.value_conv, None, [ConvSource(x)], act="none")
        return ops.row_softmax_mul(score, value)

    # ---------------- training path: one autograd node per layer (fpn_autograd.py) ----------------
    def _dg(self, name: str) -> dict:
        """Per-layer cache of the packed data-gradient weights (keyed by the weight's version inside ConvLayerFn)."""
        return self.__dict__.setdefault("_dgrad_cache", {}).setdefault(name, {})

    def _t_cbr(self, name, conv: nn.Conv2d, bn, srcs, resid=None, act=True):
        """conv (stride 1) -> BatchNorm -> [+ resid] -> ReLU, the ResNet / FPN order (activation AFTER the normalisation)."""
        from . import fpn_autograd as fa
        y = fa.conv2d(srcs, conv.weight, conv.bias, conv.kernel_size[0], conv.padding[0], conv.dilation[0], None, bn, resid, self._dg(name))
        return fa.relu(y) if act else y

    def _t_conv_s2(self, name, conv: nn.Conv2d, bn, x):
        """A stride-2 conv as the stride-1 conv sub-sampled (y[::2, ::2]), then BatchNorm on the sub-sampled map: 4x the conv arithmetic of
        the strided form on these layers, but every piece (forward, data gradient, weight gradient) is the stock stride-1 kernel family."""
        from . import fpn_autograd as fa
        if conv.kernel_size[0] == 1:
            y = fa.conv2d([fa.NearestDownFn.apply(x, 2)], conv.weight, conv.bias, 1, 0, 1, None, None, None, self._dg(name))
        else:
            full = fa.conv2d([x], conv.weight, conv.bias, conv.kernel_size[0], conv.padding[0], conv.dilation[0], None, None, None, self._dg(name))
            y = fa.NearestDownFn.apply(full, 2)
        return fa.batch_norm(bn, y)

    def _t_stage(self, lname, layer, x, meta_k):
        from . import fpn_autograd as fa
        if meta_k is not None:
            x = fa.ReplaceTailFn.apply(x, meta_k)                      # torch.cat([x[:, :-m], meta_k], 1)  (semanticFCN.py:309-313)
        for bi, blk in enumerate(layer):
            n = f"{lname}.{bi}"
            if isinstance(blk, Bottleneck):
                o = self._t_cbr(n + ".conv1", blk.conv1, blk.bn1, [x])
                o = fa.relu(self._t_conv_s2(n + ".conv2", blk.conv2, blk.bn2, o)) if blk.stride == 2 else self._t_cbr(n + ".conv2", blk.conv2, blk.bn2, [o])
                last, last_bn = blk.conv3, blk.bn3
            else:
                o = fa.relu(self._t_conv_s2(n + ".conv1", blk.conv1, blk.bn1, x)) if blk.stride == 2 else self._t_cbr(n + ".conv1", blk.conv1, blk.bn1, [x])
                last, last_bn = blk.conv2, blk.bn2
            idn = x
            if blk.downsample is not None:
                dconv, dbn = blk.downsample[0], blk.downsample[1]
                idn = self._t_conv_s2(n + ".down", dconv, dbn, x) if dconv.stride[0] == 2 else self._t_cbr(n + ".down", dconv, dbn, [x], act=False)
            x = self._t_cbr(n + ".tail", last, last_bn, [o], resid=idn)
        return x

    def _t_attend(self, name, att: AttentionModule, x):
        from . import fpn_autograd as fa
        wqk, bqk = att.query_conv.weight + att.key_conv.weight, att.query_conv.bias + att.key_conv.bias      # tanh(q + k): one conv
        t = fa.tanh(fa.conv2d([x], wqk, bqk, 1, 0, 1, None, None, None, self._dg(name + ".qk")))
        score = fa.conv2d([t], att.attention_conv.weight, att.attention_conv.bias, 1, 0, 1, None, None, None, self._dg(name + ".score"))
        value = fa.conv2d([x], att.value_conv.weight, att.value_conv.bias, 1, 0, 1, None, None, None, self._dg(name + ".value"))
        return fa.RowSoftmaxMulFn.apply(score, value)

    def _t_convT_eq_stride(self, name, ct: nn.ConvTranspose2d, x):
        """ConvTranspose2d(k = s) = 1x1 conv to Cout s s channels (+ depth-to-space by the caller); the weight re-arrangement is a
        differentiable view of the parameter."""
        from . import fpn_autograd as fa
        s = ct.stride[0]
        cin, cout = ct.weight.shape[0], ct.weight.shape[1]
        wc = ct.weight.permute(1, 2, 3, 0).reshape(cout * s * s, cin, 1, 1)
        b = None if ct.bias is None else ct.bias.repeat_interleave(s * s)
        return fa.conv2d([x], wc, b, 1, 0, 1, None, None, None, self._dg(name))

    def _t_convT_k4s2p1(self, name, ct: nn.ConvTranspose2d, x):
        from . import fpn_autograd as fa
        w = ct.weight                                            # [Cin, Cout, 4, 4]
        cin, cout = w.shape[0], w.shape[1]
        wf = torch.zeros((cout, 2, 2, cin, 3, 3), dtype=w.dtype, device=w.device)
        pairs = {0: ((0, 1), (-1, 3)), 1: ((1, 0), (0, 2))}       # output parity -> ((input offset, kernel index), ...)
        for py, ys in pairs.items():
            for px, xs in pairs.items():
                for dy, i in ys:
                    for dx, j in xs:
                        wf[:, py, px, :, dy + 1, dx + 1] = w[:, :, i, j].t()
        b = None if ct.bias is None else ct.bias.repeat_interleave(4)
        y = fa.conv2d([x], wf.reshape(cout * 4, cin, 3, 3), b, 3, 1, 1, None, None, None, self._dg(name))
        return fa.depth_to_space(y, 2)

    def _t_encode(self, x, meta):
        """(x1, x2, x3, x4) of the training path: stem + the four stages with the multi-scale meta injection (semanticFCN.py:279-314)."""
        from . import fpn_autograd as fa
        m1 = m2 = m3 = None
        if self.multi_scale_meta:
            m1, m2, m3 = (fa.NearestDownFn.apply(meta, f) for f in (2, 4, 8))
        conv1 = self.backbone.conv1                                                        # bn1 is skipped by the reference stem
        xs = fa.conv2d([x, meta], conv1.weight, conv1.bias, 3, 1, 1, 0.0, None, None, self._dg("stem"))       # conv -> ReLU in one node
        xs = fa.MaxPoolFn.apply(xs)
        x1 = self._t_stage("layer1", self.layer1, xs, None)
        x2 = self._t_stage("layer2", self.layer2, x1, m1)
        x3 = self._t_stage("layer3", self.layer3, x2, m2)
        x4 = self._t_stage("layer4", self.layer4, x3, m3)
        return x1, x2, x3, x4

    def _forward_train(self, x, meta):
        from . import fpn_autograd as fa
        from . import autograd as _ag
        _ag.nbt_scope_enter()
        try:
            x1, x2, x3, x4 = self._t_encode(x, meta)
            f4 = self._t_cbr("fpn4", self.fpn_block4[0], self.fpn_block4[1], [x4])
            f3 = self._t_cbr("fpn3", self.fpn_block3[0], self.fpn_block3[1], [x3])
            f2 = self._t_cbr("fpn2", self.fpn_block2[0], self.fpn_block2[1], [x2])
            f1 = self._t_cbr("fpn1", self.fpn_block1[0], self.fpn_block1[1], [x1])
            if self.attention:
                f4, f3 = self._t_attend("att4", self.attention4, f4), self._t_attend("att3", self.attention3, f3)
                f2, f1 = self._t_attend("att2", self.attention2, f2), self._t_attend("att1", self.attention1, f1)
            u2 = self._t_convT_eq_stride("up2", self.upsample_layer_x2, f2)
            u3 = self._t_convT_eq_stride("up3", self.upsample_layer_x3, f3)
            u4 = self._t_convT_eq_stride("up4", self.upsample_layer_x4, f4)
            ups = fa.DepthToSpaceCatFn.apply((self.upsample_layer_x2.stride[0], self.upsample_layer_x3.stride[0], self.upsample_layer_x4.stride[0]),
                                             u2, u3, u4)
            d = self.decoder_semantic
            y = self._t_cbr("dec0", d[0], d[1], [f1, ups])
            y = self._t_cbr("dec1", d[3], d[4], [y])
            return fa.elu_plus_one(self._t_convT_k4s2p1("dec_out", d[6], y))
        finally:
            _ag.nbt_scope_exit()

    def forward(self, x, meta_channel):
        x, meta = self._check_inputs(x, meta_channel)
        if self._wants_autograd(x, meta):
            return self._forward_train(x, meta)
        if self.backbone_name == "resnet50" and _sn.get_conv_precision() == "f16x3":
            raise RuntimeError("models/semanticFCN with the resnet50 backbone does not run with conv precision 'f16x3': split-fp16 products miss the "
                               "1e-3 parity bar on the 50-layer stack without GroupNorm (4e-3 of the output scale); use set_conv_precision('fp32')")
        x1, x2, x3, x4 = self._encode(x, meta)
        f4 = self._conv("fpn4", self.fpn_block4[0], self.fpn_block4[1], [ConvSource(x4)])
        f3 = self._conv("fpn3", self.fpn_block3[0], self.fpn_block3[1], [ConvSource(x3)])
        f2 = self._conv("fpn2", self.fpn_block2[0], self.fpn_block2[1], [ConvSource(x2)])
        f1 = self._conv("fpn1", self.fpn_block1[0], self.fpn_block1[1], [ConvSource(x1)])
        if self.attention:
            f4, f3 = self._attend("att4", self.attention4, f4), self._attend("att3", self.attention3, f3)
            f2, f1 = self._attend("att2", self.attention2, f2), self._attend("att1", self.attention1, f1)
        # the three up-sampled maps land in channel slices of one buffer: cat([x1, x2, x3, x4]) is read as 2 sources
        c2, c3, c4 = (m.out_channels for m in (self.upsample_layer_x2, self.upsample_layer_x3, self.upsample_layer_x4))
        ups = torch.empty((f1.shape[0], c2 + c3 + c4, f1.shape[2], f1.shape[3]), dtype=torch.float32, device=f1.device)
        self._convT_eq_stride("up2", self.upsample_layer_x2, f2, ups, 0)
        self._convT_eq_stride("up3", self.upsample_layer_x3, f3, ups, c2)
        self._convT_eq_stride("up4", self.upsample_layer_x4, f4, ups, c2 + c3)
        d = self.decoder_semantic
        y = self._conv("dec0", d[0], d[1], [ConvSource(f1), ConvSource(ups)])
        y = self._conv("dec1", d[3], d[4], [ConvSource(y)])
        return self._convT_k4s2p1("dec_out", d[6], y, elu_plus_one=True)
