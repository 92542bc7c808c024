"""SalsaNext on MI355X: the reference's module contract over hand-written HIP kernels.

Contract mirrored from ``src/baselines/SalsaNext/SalsaNext.py:10-215`` of the reference
(constructor ``SalsaNext(nclasses, nchannels=5)``, ``forward(x[B,nch,H,W]) -> logits[B,ncls,H,W]``,
identical ``state_dict`` keys / OIHW fp32 shapes, genuine ``nn.Dropout2d`` children whose
``.training`` flag is honoured per call so that ``utils.mc_dropout.set_dropout_mode`` works).

What is different underneath: a block is not a chain of ATen ops but a handful of fused launches
(``ops.conv2d_fused``): conv + bias + LeakyReLU + eval-BatchNorm + residual in one kernel,
``torch.cat`` / ``PixelShuffle`` never materialised (the consumer conv reads its 2-3 sources
directly) and every ``Dropout2d`` folded into a per-(sample, channel) multiplier that the consumer
applies while staging its input tile.  The multipliers are drawn by the real ``nn.Dropout2d``
children (on a ``[B,C,1,1]`` tensor of ones, i.e. the same Bernoulli noise shape ATen's feature
dropout draws), so torch's RNG stream and the ``.training`` flags behave as in the reference.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import autograd as _autograd
from . import h8, ops
from .autograd import AvgPoolFn, ConvLayerFn, LayerCfg
from .ops import ConvSource

_SLOPE = 0.01  # nn.LeakyReLU() default used throughout the reference model

# Multiply precision of the fused inference convs: "fp32" = exact fp32 MFMA; "f16x3" = split-fp16 (three f16
# MFMAs per K-step, ~2^-22 relative product error, 5x fewer matrix-core cycles); "f16" = fp16 storage in channel
# blocks of 8 + one f16 MFMA per K-step with fp32 accumulation (BASELINE.json configs[2],[4]: half-precision
# inference, logits within 1e-3 of the fp32 reference).  Training / autograd always runs the exact kernels.
# Override per process with SLU_CONV_PRECISION or set_conv_precision().
CONV_PRECISIONS = ("fp32", "f16x3", "f16")
_CONV_PRECISION = os.environ.get("SLU_CONV_PRECISION", "fp32")
# Products of the TRAINING forward convs: "fp32" = exact MFMA, "f16x3" = split-fp16 (fp32 storage and accumulation, ~2^-22
# relative product error; activations are O(1)).  Data- and weight-gradient kernels are always exact fp32: gradients fall
# below fp16's range.
_TRAIN_CONV_PRECISION = os.environ.get("SLU_TRAIN_CONV_PRECISION", "fp32")
# half-precision inference: run a block's 2x2-dilated conv and its concat 1x1 conv as one launch (0: two launches; A/B switch)
_FUSE_TAIL = os.environ.get("SLU_FUSE_TAIL", "1") != "0"
# the fused kernel also covers 128 channels, but there (MFMA-bound layers, 4-row tiles) it measured slower than the two launches
_FUSE_TAIL_MAX_C = int(os.environ.get("SLU_FUSE_TAIL_MAX_C", "64"))
# half-precision inference: a ResContextBlock (1x1 -> 3x3 -> 3x3 dilated + shortcut) as one launch (0: three launches; A/B switch)
_FUSE_SHORTCUT = os.environ.get("SLU_FUSE_SHORTCUT", "1") != "0"      # A/B: ResBlock shortcut conv inside the fused tail
_FUSE_CTX = os.environ.get("SLU_FUSE_CTX", "1") != "0"
_PACK_MULTI = os.environ.get("SLU_PACK_MULTI", "1") != "0"          # A/B: training step packs all conv weights in one launch
# half-precision MC inference: head conv + softmax / entropy / MI reduction over the T passes as one launch (0: logits + slu_mc_reduce)
_FUSE_HEAD_MC = os.environ.get("SLU_FUSE_HEAD_MC", "1") != "0"


def set_train_conv_precision(precision: str) -> None:
    global _TRAIN_CONV_PRECISION
    if precision not in ("fp32", "f16x3"):
        raise ValueError(f"unknown training conv precision {precision!r}; choose 'fp32' or 'f16x3'")
    _TRAIN_CONV_PRECISION = precision



def set_conv_precision(precision: str) -> None:
    global _CONV_PRECISION
    if precision not in CONV_PRECISIONS:
        raise ValueError(f"unknown conv precision {precision!r}; choose from {CONV_PRECISIONS}")
    _CONV_PRECISION = precision


def get_conv_precision() -> str:
    return _CONV_PRECISION


class _Prepared:
    """Device-side derived constants of one conv (+ its BatchNorm): the MFMA-ordered weight image and
    the folded BN affine.  Rebuilt lazily whenever the owning parameters/buffers change."""

    __slots__ = ("key", "wpack", "bn_key", "bn_a", "bn_b", "dgrad", "key16", "wpack16", "key8", "wpack8")

    def __init__(self):
        self.key = self.bn_key = None
        self.wpack = self.bn_a = self.bn_b = None
        self.key16 = self.wpack16 = None
        self.key8 = self.wpack8 = None
        self.dgrad = {}          # packed data-gradient weights, keyed by the weight version


def _tkey(*ts):
    return tuple((t.data_ptr(), t._version, t.device) for t in ts)


class _FusedBlock(nn.Module):
    """Shared machinery: run ``conv -> LeakyReLU -> BatchNorm(eval) [-> + resid]`` as one launch."""

    def __call__(self, *args, **kwargs):
        # train-mode BatchNorm bookkeeping of all layers inside the outermost call is flushed as one launch when it returns (autograd.py)
        _autograd.nbt_scope_enter()
        try:
            return super().__call__(*args, **kwargs)
        finally:
            _autograd.nbt_scope_exit()

    def _folded_bn(self, p: _Prepared, bn: Optional[nn.BatchNorm2d]):
        if bn is None:
            return None, None
        bkey = _tkey(bn.weight, bn.bias, bn.running_mean, bn.running_var)
        if p.bn_key != bkey:
            p.bn_a, p.bn_b = ops.bn_fold(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
            p.bn_key = bkey
        return p.bn_a, p.bn_b

    def _run_h8(self, p: _Prepared, conv: nn.Conv2d, bn, srcs, resid, act, out_f32):
        """Half-precision inference form: sources / residual / result are h8 tensors (see h8.py)."""
        wkey = _tkey(conv.weight)
        if p.key8 != wkey:
            p.wpack8 = h8.pack_conv_weight_h8(conv.weight.detach().contiguous())
            p.key8 = wkey
        bn_a, bn_b = self._folded_bn(p, bn)
        return h8.conv2d_h8([h8.H8Source(s.tensor, s.scale, s.nbatch) for s in srcs], p.wpack8, conv.in_channels,
                            conv.out_channels, conv.kernel_size[0], conv.dilation[0], conv.padding[0],
                            bias=None if conv.bias is None else conv.bias.detach(), slope=_SLOPE if act else None,
                            bn_a=bn_a, bn_b=bn_b, resid=resid, out_f32_nchw=out_f32)

    def _prepared(self, conv: nn.Conv2d) -> _Prepared:
        cache: Dict[str, _Prepared] = self.__dict__.setdefault("_prep", {})
        p = cache.get(str(id(conv)))
        if p is None:
            p = cache[str(id(conv))] = _Prepared()
        return p

    def _tail_is_fused(self, conv_a: nn.Conv2d, conv_b: nn.Conv2d, a1) -> bool:
        c = conv_a.out_channels
        return bool(_FUSE_TAIL and a1.dtype == torch.float16 and a1.dim() == 5 and conv_a.kernel_size == (2, 2) and conv_a.dilation == (2, 2)
                    and conv_b.kernel_size == (1, 1) and conv_b.in_channels == 3 * c and conv_b.out_channels == c
                    and c <= _FUSE_TAIL_MAX_C and h8.conv_tail_supported(c, a1.shape[2], a1.shape[3]))

    def _run_tail(self, conv_a: nn.Conv2d, bn_a, conv_b: nn.Conv2d, bn_b, a1, a2, resid=None, shortcut=None):
        """The last two layers of a block, `conv_b(cat(a1, a2, conv_a(a2)))`, each followed by LeakyReLU and eval BatchNorm.
        Half-precision inference with 32 / 64 channels: one fused launch that keeps conv_a's output on chip (csrc/conv_tail_h8.hip);
        otherwise the two layers one after the other."""
        if self._tail_is_fused(conv_a, conv_b, a1):
            packs, folded = [], []
            for conv, bn in ((conv_a, bn_a), (conv_b, bn_b)):
                p = self._prepared(conv)
                wkey = _tkey(conv.weight)
                if p.key8 != wkey:
                    p.wpack8 = h8.pack_conv_weight_h8(conv.weight.detach().contiguous())
                    p.key8 = wkey
                packs.append(p.wpack8)
                fa, fb = self._folded_bn(p, bn)
                folded.append(None if fa is None else (fa, fb))
            sc = None
            if shortcut is not None:      # (x, conv): the block's shortcut branch leaky(conv(x)) computed inside the tail (h8.conv_tail_h8)
                sx, sconv = shortcut
                p = self._prepared(sconv)
                wkey = _tkey(sconv.weight)
                if p.key8 != wkey:
                    p.wpack8 = h8.pack_conv_weight_h8(sconv.weight.detach().contiguous())
                    p.key8 = wkey
                sc = (sx, p.wpack8, None if sconv.bias is None else sconv.bias.detach(), _SLOPE, sconv.in_channels)
            return h8.conv_tail_h8(a1, a2, packs[0], packs[1], None if conv_a.bias is None else conv_a.bias.detach(), _SLOPE, folded[0],
                                   None if conv_b.bias is None else conv_b.bias.detach(), _SLOPE, folded[1], resid=resid, shortcut=sc)
        if shortcut is not None:
            resid = self._run(shortcut[1], None, [ConvSource(shortcut[0])])
        a3 = self._run(conv_a, bn_a, [ConvSource(a2)])
        return self._run(conv_b, bn_b, [ConvSource(a1), ConvSource(a2), ConvSource(a3)], resid=resid)

    def _run(self, conv: nn.Conv2d, bn: Optional[nn.BatchNorm2d], srcs, resid=None, act=True, out_f32=False):
        cache: Dict[str, _Prepared] = self.__dict__.setdefault("_prep", {})
        name = str(id(conv))
        p = cache.get(name)
        if p is None:
            p = cache[name] = _Prepared()
        if srcs[0].tensor.dtype == torch.float16:
            return self._run_h8(p, conv, bn, srcs, resid, act, out_f32)
        wkey = _tkey(conv.weight)
        if p.key != wkey:
            p.wpack = ops.pack_conv_weight(conv.weight.detach().contiguous())
            p.key = wkey
        k = conv.kernel_size[0]
        slope = _SLOPE if act else None
        wants_grad = torch.is_grad_enabled() and (
            conv.weight.requires_grad or any(s.tensor.requires_grad for s in srcs)
            or (resid is not None and resid.requires_grad) or (bn is not None and bn.weight.requires_grad))
        if wants_grad or (bn is not None and bn.training):
            # autograd node per layer (also the train-mode BatchNorm forward): conv -> stats -> affine
            wp = p.wpack
            if _TRAIN_CONV_PRECISION == "f16x3":
                if p.key16 != wkey:
                    p.wpack16 = ops.pack_conv_weight_f16x3(conv.weight.detach().contiguous())
                    p.key16 = wkey
                wp = p.wpack16
            cfg = LayerCfg(k, conv.dilation[0], conv.padding[0], slope, [s.scale for s in srcs],
                           [s.pixel_shuffle for s in srcs], bn, conv.out_channels, wp, p.dgrad, _TRAIN_CONV_PRECISION)
            return ConvLayerFn.apply(cfg, conv.weight, conv.bias, None if bn is None else bn.weight,
                                     None if bn is None else bn.bias, resid, *[s.tensor for s in srcs])
        bn_a, bn_b = self._folded_bn(p, bn)
        wpack, precision = p.wpack, "fp32"
        if _CONV_PRECISION == "f16x3":
            precision = "f16x3"
            if p.key16 != wkey:
                p.wpack16 = ops.pack_conv_weight_f16x3(conv.weight.detach().contiguous())
                p.key16 = wkey
            wpack = p.wpack16
        return ops.conv2d_fused(srcs, wpack, conv.out_channels, k, conv.dilation[0], conv.padding[0],
                                bias=None if conv.bias is None else conv.bias.detach(),
                                slope=slope, bn_a=bn_a, bn_b=bn_b, resid=resid, precision=precision)


# A/B switch: 0 = draw the MC-dropout multipliers through the nn.Dropout2d children (ATen launches on a side stream) instead of the one-launch kernel
_DROPOUT_KERNEL = os.environ.get("SLU_DROPOUT_KERNEL", "1") != "0"


def _draw(drop: nn.Dropout2d, n: int, c: int, device, override: Optional[Dict[str, torch.Tensor]], name: str):
    """[n,c] multiplier of one Dropout2d site, or None when it is the identity."""
    if override is not None:
        s = override.get(name)
        return None if s is None else s.reshape(n, c).to(device=device, dtype=torch.float32).contiguous()
    if not drop.training or drop.p == 0.0:
        return None
    return drop(torch.ones((n, c, 1, 1), dtype=torch.float32, device=device)).reshape(n, c)


def _mul(a, b):
    if a is None:
        return b
    if b is None:
        return a
    return a * b


class ResContextBlock(_FusedBlock):
    def __init__(self, in_filters, out_filters):
        super().__init__()
        self.conv1 = nn.Conv2d(in_filters, out_filters, 1)
        self.conv2 = nn.Conv2d(out_filters, out_filters, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(out_filters)
        self.conv3 = nn.Conv2d(out_filters, out_filters, 3, dilation=2, padding=2)
        self.bn2 = nn.BatchNorm2d(out_filters)

    def forward(self, x):
        if (_FUSE_CTX and x.dtype == torch.float16 and x.dim() == 5 and self.conv1.out_channels == 32
                and h8.ctx_block_supported(self.conv1.in_channels, 32, x.shape[2], x.shape[3])):
            # half-precision inference: the whole block as one launch, shortcut and a1 never leave the CU (csrc/ctx_block_h8.hip)
            packs, folded = [], []
            for conv, bn in ((self.conv1, None), (self.conv2, self.bn1), (self.conv3, self.bn2)):
                p = self._prepared(conv)
                wkey = _tkey(conv.weight)
                if p.key8 != wkey:
                    p.wpack8 = h8.pack_conv_weight_h8(conv.weight.detach().contiguous())
                    p.key8 = wkey
                packs.append(p.wpack8)
                fa, fb = self._folded_bn(p, bn)
                folded.append(None if fa is None else (fa, fb))
            b = [None if c.bias is None else c.bias.detach() for c in (self.conv1, self.conv2, self.conv3)]
            return h8.ctx_block_h8(x, self.conv1.in_channels, packs[0], packs[1], packs[2], b[0], b[1], folded[1], b[2], folded[2], _SLOPE)
        shortcut = self._run(self.conv1, None, [ConvSource(x)])
        a1 = self._run(self.conv2, self.bn1, [ConvSource(shortcut)])
        return self._run(self.conv3, self.bn2, [ConvSource(a1)], resid=shortcut)


class ResBlock(_FusedBlock):
    def __init__(self, in_filters, out_filters, dropout_rate, kernel_size=(3, 3), stride=1, pooling=True, drop_out=True):
        super().__init__()
        if stride != 1 or tuple(kernel_size) != (3, 3):
            raise ValueError("only the stride-1 / 3x3-pool configuration of the reference is implemented")
        self.pooling, self.drop_out = pooling, drop_out
        self.conv1 = nn.Conv2d(in_filters, out_filters, 1)
        self.conv2 = nn.Conv2d(in_filters, out_filters, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(out_filters)
        self.conv3 = nn.Conv2d(out_filters, out_filters, 3, dilation=2, padding=2)
        self.bn2 = nn.BatchNorm2d(out_filters)
        self.conv4 = nn.Conv2d(out_filters, out_filters, 2, dilation=2, padding=1)
        self.bn3 = nn.BatchNorm2d(out_filters)
        self.conv5 = nn.Conv2d(out_filters * 3, out_filters, 1)
        self.bn4 = nn.BatchNorm2d(out_filters)
        self.dropout = nn.Dropout2d(p=dropout_rate)

    def features(self, x):
        """The block up to (not including) dropout / pooling: deterministic given x."""
        src = [ConvSource(x)]
        # half precision, 32 -> 64 (resBlock1): the shortcut conv runs inside the fused tail, from x, and its tensor never exists
        fuse_sc = (_FUSE_SHORTCUT and x.dtype == torch.float16 and x.dim() == 5 and x.shape[1] * 8 == self.conv1.in_channels
                   and h8.conv_tail_shortcut_supported(self.conv1.out_channels, self.conv1.in_channels))
        shortcut = None if fuse_sc else self._run(self.conv1, None, src)
        a1 = self._run(self.conv2, self.bn1, src)
        a2 = self._run(self.conv3, self.bn2, [ConvSource(a1)])
        if fuse_sc and self._tail_is_fused(self.conv4, self.conv5, a1):
            return self._run_tail(self.conv4, self.bn3, self.conv5, self.bn4, a1, a2, shortcut=(x, self.conv1))
        if shortcut is None:
            shortcut = self._run(self.conv1, None, src)
        return self._run_tail(self.conv4, self.bn3, self.conv5, self.bn4, a1, a2, resid=shortcut)

    def forward(self, x, _scales=None, _name=""):
        """pooling: (pooled, full_res);  else: (full_res, deferred dropout multiplier or None)."""
        full = self.features(x)
        s = None
        if self.drop_out:
            s = _draw(self.dropout, full.shape[0], self.conv5.out_channels, full.device, _scales, _name + ".dropout")
        if self.pooling:
            if full.dtype == torch.float16:
                return h8.avgpool3s2_h8(full, s), full
            pooled = AvgPoolFn.apply(full, s) if (torch.is_grad_enabled() and full.requires_grad) else ops.avgpool3s2(full, s)
            return pooled, full
        return full, s


class UpBlock(_FusedBlock):
    def __init__(self, in_filters, out_filters, dropout_rate, drop_out=True):
        super().__init__()
        self.drop_out, self.in_filters, self.out_filters = drop_out, in_filters, out_filters
        self.dropout1 = nn.Dropout2d(p=dropout_rate)
        self.dropout2 = nn.Dropout2d(p=dropout_rate)
        self.conv1 = nn.Conv2d(in_filters // 4 + 2 * out_filters, out_filters, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(out_filters)
        self.conv2 = nn.Conv2d(out_filters, out_filters, 3, dilation=2, padding=2)
        self.bn2 = nn.BatchNorm2d(out_filters)
        self.conv3 = nn.Conv2d(out_filters, out_filters, 2, dilation=2, padding=1)
        self.bn3 = nn.BatchNorm2d(out_filters)
        self.conv4 = nn.Conv2d(out_filters * 3, out_filters, 1)
        self.bn4 = nn.BatchNorm2d(out_filters)
        self.dropout3 = nn.Dropout2d(p=dropout_rate)

    def _compose_scales(self, x_scale, d1, d2):
        """(multiplier of the stored x [n, in_filters], multiplier of the skip [n, skip channels]) from the producer's deferred multiplier
        and this block's dropout1 / dropout2 draws: dropout1 acts on the shuffled x, dropout2 on cat(shuffled x, skip) (SalsaNext.py:141-149)."""
        cu = self.in_filters // 4
        sx, ss = x_scale, None
        up = d1
        if d2 is not None:
            up = _mul(up, d2[:, :cu])
            ss = d2[:, cu:].contiguous()
        if up is not None:      # shuffled channel c is fed by stored channels 4c..4c+3
            sx = _mul(sx, up.repeat_interleave(4, dim=1))
        return sx, ss

    def forward(self, x, skip, x_scale=None, _scales=None, _name="", skip_nbatch=0):
        """x is read through PixelShuffle(2); x_scale is the producer's deferred dropout multiplier.
        skip_nbatch > 0: `skip` holds that many images shared by the stacked MC passes.
        Returns (out, deferred multiplier of dropout3 or None)."""
        n, cx, dev = x.shape[0], self.in_filters, x.device
        cu, cs = cx // 4, self.conv1.in_channels - cx // 4
        if _scales is not None and _scales.get(_name + "._composed"):      # composed with the draws, on the side stream (SalsaNext._predraw_dropout)
            sx, ss = _scales.get(_name + "._sx"), _scales.get(_name + "._ss")
        elif self.drop_out:
            sx, ss = self._compose_scales(x_scale, _draw(self.dropout1, n, cu, dev, _scales, _name + ".dropout1"),
                                          _draw(self.dropout2, n, cu + cs, dev, _scales, _name + ".dropout2"))
        else:
            sx, ss = x_scale, None
        if x.dtype == torch.float16:
            # h8: PixelShuffle is a (tiny) data-movement launch that also applies the producer's multiplier and dropout1/2
            xs = h8.pixel_shuffle_h8(x, None if sx is None else sx.contiguous())
            e1 = self._run(self.conv1, self.bn1, [ConvSource(xs), ConvSource(skip, ss, False, skip_nbatch)])
        else:
            if sx is not None:
                sx = sx.contiguous()
            e1 = self._run(self.conv1, self.bn1, [ConvSource(x, sx, True), ConvSource(skip, ss, False, skip_nbatch)])
        e2 = self._run(self.conv2, self.bn2, [ConvSource(e1)])
        out = self._run_tail(self.conv3, self.bn3, self.conv4, self.bn4, e1, e2)
        s3 = None
        if self.drop_out:
            s3 = _draw(self.dropout3, n, self.out_filters, dev, _scales, _name + ".dropout3")
        return out, s3


class SalsaNext(_FusedBlock):
    def __init__(self, nclasses, nchannels=5):
        super().__init__()
        self.nclasses = nclasses
        self.downCntx = ResContextBlock(nchannels, 32)
        self.downCntx2 = ResContextBlock(32, 32)
        self.downCntx3 = ResContextBlock(32, 32)
        self.resBlock1 = ResBlock(32, 64, 0.2, pooling=True, drop_out=False)
        self.resBlock2 = ResBlock(64, 128, 0.2, pooling=True)
        self.resBlock3 = ResBlock(128, 256, 0.2, pooling=True)
        self.resBlock4 = ResBlock(256, 256, 0.2, pooling=True)
        self.resBlock5 = ResBlock(256, 256, 0.2, pooling=False)
        self.upBlock1 = UpBlock(256, 128, 0.2)
        self.upBlock2 = UpBlock(128, 128, 0.2)
        self.upBlock3 = UpBlock(128, 64, 0.2)
        self.upBlock4 = UpBlock(64, 32, 0.2, drop_out=False)
        self.logits = nn.Conv2d(32, nclasses, 1)
        self.tail_act = nn.Softmax(dim=1)   # kept for module-tree parity; the reference forward never applies it

    def forward(self, x):
        """Raw logits [B, nclasses, H, W] (reference SalsaNext.py:197-215)."""
        return self._forward(x, None)

    def forward_with_dropout_scales(self, x, scales: Dict[str, torch.Tensor]):
        """Same as forward() but with the Dropout2d multipliers given explicitly
        (site name -> [B,C,1,1] or [B,C]; missing site = identity).  Used by the parity tests."""
        return self._forward(x, scales)

    def _head(self, u1):
        """The 1x1 logits conv; with `_features_only` set (mc_predict_fused) the decoder output is handed back instead."""
        if self.__dict__.get("_features_only", False):
            return u1
        return self._run(self.logits, None, [ConvSource(u1)], act=False, out_f32=True)

    def mc_fused_ok(self, x, T: int) -> bool:
        """The fused head + MC reduction covers half-precision inference on images whose pixel count is a multiple of 32."""
        return (_FUSE_HEAD_MC and _CONV_PRECISION == "f16" and isinstance(x, torch.Tensor) and x.dim() == 4 and x.is_cuda
                and (x.shape[2] * x.shape[3]) % 32 == 0 and self.logits.out_channels <= 32 and self._inference_only())

    @torch.no_grad()
    def mc_predict_fused(self, x, T: int, eps: float = 1e-12, share_prefix: bool = False, scales=None):
        """(p_bar, H_norm, MI_norm, preds) of T stochastic passes with the head conv and the MC reduction in one launch
        (csrc/head_mc_h8.hip): the T*B fp32 logit maps are never written.  Call under utils.mc_dropout.dropout_sampling."""
        b = x.shape[0]
        self.__dict__["_features_only"] = True
        try:
            u1 = self.forward_mc(x, T, scales) if share_prefix else self._forward(x.repeat(int(T), 1, 1, 1), scales)
        finally:
            self.__dict__["_features_only"] = False
        p = self._prepared(self.logits)
        wkey = _tkey(self.logits.weight)
        if p.key8 != wkey:
            p.wpack8 = h8.pack_conv_weight_h8(self.logits.weight.detach().contiguous())
            p.key8 = wkey
        return h8.head_mc_h8(u1, p.wpack8, None if self.logits.bias is None else self.logits.bias.detach(), self.logits.out_channels,
                             int(T), b, eps)

    @torch.no_grad()
    def forward_mc(self, x, T: int, scales: Optional[Dict[str, torch.Tensor]] = None):
        """T stochastic passes of a batch x[B,...] -> logits [T*B, ncls, H, W] (pass-major), computing the part of
        the network that no active Dropout2d can reach ONCE: the three context blocks, resBlock1 and the convs of
        resBlock2 see the same input in every pass (resBlock1 is built with drop_out=False and resBlock2's dropout
        sits after its convs, reference SalsaNext.py:98-101,183-184), so their outputs -- and the two skip tensors
        they feed to upBlock3 / upBlock4 -- are identical in all T passes.  Results equal T full forwards with the
        same multipliers; 48 % of the conv FLOPs are not repeated T times.  Inference only (BatchNorm must be frozen)."""
        if self.training or any(m.training for m in self.modules() if isinstance(m, nn.BatchNorm2d)):
            raise RuntimeError("forward_mc needs eval-mode BatchNorm (use utils.mc_dropout.mc_forward)")
        if not isinstance(x, torch.Tensor) or x.dim() != 4 or not x.is_cuda:
            raise RuntimeError("forward_mc expects a [B, C, H, W] tensor on the GPU")
        if x.shape[2] % 16 or x.shape[3] % 16:
            raise RuntimeError("SalsaNext needs H and W divisible by 16")
        b = x.shape[0]
        n = int(T) * b
        x = x.contiguous().float()
        half = _CONV_PRECISION == "f16"
        if scales is None and not torch.cuda.is_current_stream_capturing():
            scales = self._predraw_dropout(n, x.device)
        if half:
            x = h8.to_h8(x)
        d = self.downCntx3(self.downCntx2(self.downCntx(x)))
        d0c, d0b = self.resBlock1(d, scales, "resBlock1")               # no dropout in this block
        full2 = self.resBlock2.features(d0c)                             # deterministic; also the skip of upBlock3
        self._join_dropout()
        s2 = _draw(self.resBlock2.dropout, n, self.resBlock2.conv5.out_channels, x.device, scales, "resBlock2.dropout")
        # from here on: T*B stacked passes
        d1c = h8.avgpool3s2_h8(full2, s2, n) if half else ops.avgpool3s2_bcast(full2, s2, n)
        d2c, d2b = self.resBlock3(d1c, scales, "resBlock3")
        d3c, d3b = self.resBlock4(d2c, scales, "resBlock4")
        d5c, s5 = self.resBlock5(d3c, scales, "resBlock5")
        u4, s = self.upBlock1(d5c, d3b, s5, scales, "upBlock1")
        u3, s = self.upBlock2(u4, d2b, s, scales, "upBlock2")
        u2, s = self.upBlock3(u3, full2, s, scales, "upBlock3", skip_nbatch=b)
        u1, _ = self.upBlock4(u2, d0b, s, scales, "upBlock4", skip_nbatch=b)
        return self._head(u1)

    def _inference_only(self) -> bool:
        """True when no autograd graph is wanted and every BatchNorm is frozen (the half-precision path has no backward)."""
        if any(m.training for m in self.modules() if isinstance(m, nn.BatchNorm2d)):
            return False
        return not (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()))

    # (site, channels) of every Dropout2d applied in forward, in the reference's call order (SalsaNext.py:98,106,145,149,168)
    _DROPOUT_SITES = (("resBlock2", "dropout", 128), ("resBlock3", "dropout", 256), ("resBlock4", "dropout", 256), ("resBlock5", "dropout", 256),
                      ("upBlock1", "dropout1", 64), ("upBlock1", "dropout2", 320), ("upBlock1", "dropout3", 128),
                      ("upBlock2", "dropout1", 32), ("upBlock2", "dropout2", 288), ("upBlock2", "dropout3", 128),
                      ("upBlock3", "dropout1", 32), ("upBlock3", "dropout2", 160), ("upBlock3", "dropout3", 64))

    def _predraw_dropout(self, n: int, device):
        """Inference with live Dropout2d (MC sampling): draw the multipliers of all 13 sites up front, through the real nn.Dropout2d
        children and in the reference's call order (same RNG consumption as drawing them inside the blocks), on a side stream, so the
        ~60 tiny launches overlap the context blocks instead of sitting between the convs.  None when no site is active."""
        sites = [(f"{blk}.{name}", getattr(getattr(self, blk), name), c) for blk, name, c in self._DROPOUT_SITES]
        if not any(d.training and d.p > 0.0 for _, d, _ in sites):
            return None
        if _DROPOUT_KERNEL:
            return self._predraw_dropout_kernel(n, device, sites)
        main = torch.cuda.current_stream(device)
        side = self.__dict__.get("_drop_stream")
        if side is None or side.device != device:
            side = self.__dict__["_drop_stream"] = torch.cuda.Stream(device)
        out: Dict[str, torch.Tensor] = {}
        with torch.cuda.stream(side):
            for key, drop, c in sites:
                s = _draw(drop, n, c, device, None, key)
                if s is not None:
                    s.record_stream(main)
                    out[key] = s
            # the multipliers the decoder's convs actually consume are products of these masks (UpBlock.forward): compose them here too, so
            # that the main stream carries no tiny elementwise launches between the convs
            prod = out.get("resBlock5.dropout")
            for blk in ("upBlock1", "upBlock2", "upBlock3", "upBlock4"):
                sx, ss = getattr(self, blk)._compose_scales(prod, out.get(f"{blk}.dropout1"), out.get(f"{blk}.dropout2"))
                for key, t in ((f"{blk}._sx", sx), (f"{blk}._ss", ss)):
                    if t is not None:
                        t.record_stream(main)
                        out[key] = t
                out[f"{blk}._composed"] = True
                prod = out.get(f"{blk}.dropout3")
        self.__dict__["_drop_event"] = side.record_event()
        return out

    def _predraw_dropout_kernel(self, n: int, device, sites):
        """The same dictionary from ONE hand-written launch (csrc/dropout_draw.hip): every multiplier a consumer needs -- the plain masks of the
        encoder sites and of the deferred dropout3 sites, and the decoder's composed products -- is a stateless Philox function of torch's CUDA
        generator state, so no intermediate mask is materialised and no ATen launch (bernoulli, mul, fill, copy: ~64 per MC step before) is left
        on the path.  The generator's offset is advanced, so `torch.manual_seed` reproduces the masks and successive calls differ."""
        sig = (n, str(device), tuple((bool(d.training), float(d.p)) for _, d, _ in sites))
        plan = self.__dict__.get("_drop_plan")
        if plan is None or plan[0] != sig:
            index = {key: i for i, (key, _, _) in enumerate(sites)}
            active = {key: bool(d.training and d.p > 0.0) for key, d, _ in sites}
            chans = {key: c for key, _, c in sites}
            outs = []

            def add(key, c, refs, shuffled=False):
                refs = [(index[k], off) for k, off in refs if k is not None and active.get(k, False)]
                if refs:
                    outs.append((key, c, refs, shuffled))

            for blk in ("resBlock2", "resBlock3", "resBlock4", "resBlock5"):
                add(f"{blk}.dropout", chans[f"{blk}.dropout"], [(f"{blk}.dropout", 0)])
            prod = "resBlock5.dropout"
            for blk in ("upBlock1", "upBlock2", "upBlock3", "upBlock4"):
                m = getattr(self, blk)
                cu, cs = m.in_filters // 4, m.conv1.in_channels - m.in_filters // 4
                d1, d2, d3 = (f"{blk}.dropout{k}" if f"{blk}.dropout{k}" in index else None for k in (1, 2, 3))
                add(f"{blk}._sx", m.in_filters, [(prod, 0), (d1, 0), (d2, 0)], shuffled=True)
                # (the shuffled flag applies to the second and third factor: the producer's multiplier is per STORED channel)
                add(f"{blk}._ss", cs, [(d2, cu)])
                if d3 is not None:
                    add(f"{blk}.dropout3", chans[d3], [(d3, 0)])
                prod = d3
            fixed = []
            for key, c, refs, shuffled in outs:      # slu_dropout_out: factor a is read unshuffled, b and c through c / 4
                if shuffled and (len(refs) < 1 or refs[0][0] != index.get(self._sx_producer(key), -2)):
                    refs = [(-1, 0)] + refs          # no (active) producer multiplier: keep the shuffled factors in slots b / c
                fixed.append((key, c, refs[:3], shuffled))
            plan = (sig, ops.DropoutPlan(n, [(c, float(d.p), bool(d.training and d.p > 0.0)) for _, d, c in sites], fixed, device),
                    [f"{blk}._composed" for blk in ("upBlock1", "upBlock2", "upBlock3", "upBlock4")])
            self.__dict__["_drop_plan"] = plan
        out = plan[1].run()
        for k in plan[2]:
            out[k] = True
        return out

    @staticmethod
    def _sx_producer(key: str):
        """The site whose deferred multiplier reaches `<upBlockK>._sx` per stored channel."""
        return {"upBlock1._sx": "resBlock5.dropout", "upBlock2._sx": "upBlock1.dropout3", "upBlock3._sx": "upBlock2.dropout3",
                "upBlock4._sx": "upBlock3.dropout3"}.get(key)

    def _pack_training_weights(self):
        """Training step (exact-fp32 products): the forward and data-gradient MFMA images of ALL conv weights in one launch
        (ops.WeightPackPlan) instead of one pack launch per layer in the forward and two per layer in the backward; the per-layer caches
        (`_Prepared.wpack`, `.dgrad`) are pointed at the plan's buffers, so the layers find them current."""
        if not _PACK_MULTI or _TRAIN_CONV_PRECISION != "fp32":
            return
        owners = [(m, c) for m in self.modules() if isinstance(m, _FusedBlock) for c in m.children() if isinstance(c, nn.Conv2d)]
        weights = [c.weight for _, c in owners]
        if not weights or not all(w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() for w in weights):
            return
        keys = [_tkey(w) for w in weights]
        plan = self.__dict__.get("_pack_plan")
        if plan is None or not plan.matches(weights):
            plan = self.__dict__["_pack_plan"] = ops.WeightPackPlan([w.detach() for w in weights])
            self.__dict__["_pack_keys"] = None
        if self.__dict__.get("_pack_keys") != keys:
            plan.run()
            self.__dict__["_pack_keys"] = keys
        for i, ((m, c), w, k) in enumerate(zip(owners, weights, keys)):
            p = m._prepared(c)
            p.wpack, p.key = plan.fwd[i], k
            p.dgrad["pack"], p.dgrad["key"] = plan.dgrad[i], (w.data_ptr(), w._version)

    def _join_dropout(self):
        ev = self.__dict__.pop("_drop_event", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def _forward(self, x, scales):
        if not isinstance(x, torch.Tensor) or x.dim() != 4:
            raise RuntimeError("SalsaNext expects a [B, C, H, W] tensor")
        if not x.is_cuda:
            raise RuntimeError("semanticlidarunc_amd.SalsaNext runs on MI355X only: input is on "
                               f"'{x.device}' and there is no CPU fallback")
        if x.shape[2] % 16 or x.shape[3] % 16:
            raise RuntimeError("SalsaNext needs H and W divisible by 16")
        x = x.contiguous().float()
        if not self._inference_only():
            self._pack_training_weights()
        # all 13 Dropout2d sites drawn (and the decoder's products composed) up front: in inference always; in training through the one-launch
        # kernel (the per-site nn.Dropout2d path costs ~60 tiny launches per step there as well).  Not under HIP-graph capture: the Philox
        # offset is a launch argument and would be frozen into the graph.
        if scales is None and (self._inference_only() or _DROPOUT_KERNEL) and not torch.cuda.is_current_stream_capturing():
            scales = self._predraw_dropout(x.shape[0], x.device)
        if _CONV_PRECISION == "f16" and self._inference_only():
            x = h8.to_h8(x)             # everything downstream stays in the fp16 channel-blocked layout
        d = self.downCntx(x)
        d = self.downCntx2(d)
        d = self.downCntx3(d)
        d0c, d0b = self.resBlock1(d, scales, "resBlock1")
        self._join_dropout()                      # the first consumer of a multiplier is resBlock2's pooling
        d1c, d1b = self.resBlock2(d0c, scales, "resBlock2")
        d2c, d2b = self.resBlock3(d1c, scales, "resBlock3")
        d3c, d3b = self.resBlock4(d2c, scales, "resBlock4")
        d5c, s5 = self.resBlock5(d3c, scales, "resBlock5")
        u4, s = self.upBlock1(d5c, d3b, s5, scales, "upBlock1")
        u3, s = self.upBlock2(u4, d2b, s, scales, "upBlock2")
        u2, s = self.upBlock3(u3, d1b, s, scales, "upBlock3")
        u1, _ = self.upBlock4(u2, d0b, s, scales, "upBlock4")
        return self._head(u1)
