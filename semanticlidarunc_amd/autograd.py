"""`torch.autograd.Function`s for the conv stack: one node per fused layer
(cat / PixelShuffle / Dropout2d multipliers -> Conv2d -> LeakyReLU -> BatchNorm -> + residual) and one for the pool.

Used whenever gradients are required or BatchNorm is in train mode; pure inference takes the single fused
kernel instead (`salsanext._FusedBlock._run`).  Backward reads only saved tensors, so it is re-entrant
(`torch.autograd.grad(..., retain_graph=True)` as in the reference's utils/grad_norm.py:52).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

import os

from . import ops
from .ops import ConvSource

# training-path fusions (A/B switch): batch statistics accumulated by the conv kernel itself, da written in both layouts by one pass
# train-mode BatchNorm statistics accumulated by the conv kernel itself instead of a second pass over its output (A/B switch; -1.2 ms of a
# 37.6 ms step.  A fused "da in both layouts" gradient pass was also built and measured: slower than the two launches, not kept)
_FUSE_STATS = os.environ.get("SLU_TRAIN_FUSE", "1") != "0"
_WGRAD_NCHW = os.environ.get("SLU_WGRAD_NCHW", "1") != "0"      # A/B: 0 = weight gradients from channel-last copies


class LayerCfg:
    """Non-tensor description of one fused layer call (plain Python, ignored by autograd)."""
    __slots__ = ("ksize", "dil", "pad", "slope", "scales", "shuffles", "bn", "cout", "wpack", "wdpack_cache", "precision")

    def __init__(self, ksize, dil, pad, slope, scales, shuffles, bn, cout, wpack, wdpack_cache, precision="fp32"):
        self.ksize, self.dil, self.pad, self.slope = ksize, dil, pad, slope
        self.scales, self.shuffles, self.bn, self.cout = scales, shuffles, bn, cout
        self.wpack, self.wdpack_cache = wpack, wdpack_cache
        self.precision = precision          # forward conv products: "fp32" (exact MFMA) or "f16x3" (split-fp16, fp32 storage)


# BatchNorm.num_batches_tracked += 1 is one tiny launch per layer (42 per SalsaNext step, 0.2 ms): inside a model call the increments are
# collected and applied as ONE multi-tensor add when the outermost module call returns (salsanext._FusedBlock.__call__), i.e. before anything
# outside the model can look at the buffers; outside any such call the increment is immediate.
_NBT_DEPTH = 0
_NBT_PENDING: list = []


def bump_num_batches_tracked(bn) -> None:
    if bn.num_batches_tracked is None:
        return
    if _NBT_DEPTH > 0:
        _NBT_PENDING.append(bn.num_batches_tracked)
    else:
        with torch.no_grad():
            bn.num_batches_tracked += 1


def nbt_scope_enter() -> None:
    global _NBT_DEPTH
    _NBT_DEPTH += 1


def nbt_scope_exit() -> None:
    global _NBT_DEPTH
    _NBT_DEPTH -= 1
    if _NBT_DEPTH == 0 and _NBT_PENDING:
        pend = list(_NBT_PENDING)
        _NBT_PENDING.clear()
        with torch.no_grad():
            torch._foreach_add_(pend, 1)


# A/B switches: 0 = the BatchNorm coefficient launches + separate apply passes (+ the fp64 -> fp32 bias-gradient conversion) of round 2
_BN_FUSED_FWD = os.environ.get("SLU_BN_FUSED_FWD", "1") != "0"
_BN_FUSED_BWD = os.environ.get("SLU_BN_FUSED_BWD", "1") != "0"


class ConvLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg: LayerCfg, weight, bias, gamma, beta, resid, *tensors):
        srcs = [ConvSource(t.detach().contiguous(), s, ps) for t, s, ps in zip(tensors, cfg.scales, cfg.shuffles)]
        bn: Optional[nn.BatchNorm2d] = cfg.bn
        train_stats = bn is not None and bool(bn.training)
        # train-mode BatchNorm: the exact-fp32 conv kernel adds the batch statistics of what it stores while it stores it
        fused_stats = ops.zeros_f64((2, cfg.cout), weight.device) if (_FUSE_STATS and train_stats and cfg.precision == "fp32") else None
        y = ops.conv2d_fused(srcs, cfg.wpack, cfg.cout, cfg.ksize, cfg.dil, cfg.pad,
                             bias=None if bias is None else bias.detach(), slope=cfg.slope, precision=cfg.precision, stats=fused_stats)
        mean = invstd = None
        if bn is not None:
            n, c, h, w = y.shape
            m = n * h * w
            sums = ((fused_stats[0], fused_stats[1]) if fused_stats is not None else ops.bn_stats(y)) if train_stats else None
            mom = 0.0
            track = train_stats and bn.track_running_stats and bn.running_mean is not None
            if track:
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
            rm = bn.running_mean if (track or not train_stats) else None
            rv = bn.running_var if (track or not train_stats) else None
            rs = None if resid is None else resid.detach().contiguous()
            if _BN_FUSED_FWD:
                z, mean, invstd = ops.bn_apply_fwd(y, sums, m, gamma.detach(), beta.detach(), bn.eps, mom, rm, rv, train_stats, rs)
            else:
                mean, invstd, a, b = ops.bn_coeffs_fwd(sums, m, gamma.detach(), beta.detach(), bn.eps, mom, rm, rv, train_stats)
                z = ops.affine(y, a, b, rs)
            if track:
                bump_num_batches_tracked(bn)
        elif resid is not None:
            z = ops.affine(y, None, None, resid.detach().contiguous())
        else:
            z = y
        ctx.cfg, ctx.train_stats, ctx.has_bn, ctx.has_resid = cfg, train_stats, bn is not None, resid is not None
        ctx.src_shapes = [tuple(t.shape) for t in tensors]
        ctx.has_bias = bias is not None
        saved = [weight, y] + [t.detach() for t in tensors]
        if bn is not None:
            saved += [gamma, mean, invstd]
        ctx.save_for_backward(*saved)
        return z

    @staticmethod
    def backward(ctx, dz):
        cfg: LayerCfg = ctx.cfg
        saved = list(ctx.saved_tensors)
        weight, y = saved[0], saved[1]
        nsrc = len(ctx.src_shapes)
        tensors = saved[2:2 + nsrc]
        dz = dz.contiguous().float()
        n, c, h, w = y.shape
        need = ctx.needs_input_grad                      # (cfg, weight, bias, gamma, beta, resid, *tensors)
        d_resid = dz if (ctx.has_resid and need[5]) else None
        dgamma = dbeta = None
        if not _BN_FUSED_BWD:
            k1 = k2 = k3 = None
            if ctx.has_bn:
                gamma, mean, invstd = saved[2 + nsrc:5 + nsrc]
                s1, s2 = ops.bn_bwd_reduce(dz, y, mean, invstd)
                k1, k2, k3, dgamma, dbeta = ops.bn_coeffs_bwd(s1, s2, float(n * h * w), gamma.detach(), mean, invstd, ctx.train_stats)
                if not ctx.train_stats:
                    k2 = k3 = None
            da, db = ops.act_affine_bwd(dz, y if (cfg.slope is not None or k3 is not None) else None, k1, k2, k3, cfg.slope, ctx.has_bias)
            dbias = db.float() if ctx.has_bias else None
        elif ctx.has_bn:
            gamma, mean, invstd = saved[2 + nsrc:5 + nsrc]
            s1, s2 = ops.bn_bwd_reduce(dz, y, mean, invstd)
            da, dbias, dgamma, dbeta = ops.bn_act_bwd(dz, y, s1, s2, float(n * h * w), gamma.detach(), mean, invstd, ctx.train_stats, cfg.slope,
                                                      ctx.has_bias)
        else:
            da, dbias, _, _ = ops.bn_act_bwd(dz, y if cfg.slope is not None else None, slope=cfg.slope, want_dbias=ctx.has_bias)
        if not need[2]:
            dbias = None
        srcs = [ConvSource(t.contiguous(), s, ps) for t, s, ps in zip(tensors, cfg.scales, cfg.shuffles)]
        cin = weight.shape[1]
        dweight = None
        if need[1]:
            # straight from the NCHW tensors when the sources are plain and 32-aligned (all of SalsaNext's layers but UpBlock.conv1, whose
            # inputs carry PixelShuffle / dropout multipliers): no channel-last copies of da and of the concatenated input
            if _WGRAD_NCHW:
                dweight = ops.conv1x1_wgrad_nchw(da, srcs) if cfg.ksize == 1 else ops.conv2d_wgrad_nchw(da, srcs, cfg.ksize, cfg.dil, cfg.pad)
            if dweight is None:
                dweight = ops.conv2d_wgrad(ops.nchw_to_nhwc(da), ops.gather_nhwc(srcs), n, h, w, cfg.cout, cin, cfg.ksize, cfg.dil, cfg.pad)
        dsrc: List[Optional[torch.Tensor]] = [None] * nsrc
        if any(need[6:6 + nsrc]):
            # always the exact fp32 kernel: gradients reach 1e-8 and below, outside fp16's range (the split-fp16 products of
            # the forward are fine because activations are O(1))
            key = (weight.data_ptr(), weight._version)
            if cfg.wdpack_cache.get("key") != key:
                cfg.wdpack_cache["pack"] = ops.pack_conv_weight(ops.dgrad_weight(weight.detach().contiguous()))
                cfg.wdpack_cache["key"] = key
            dcat = ops.conv2d_fused([ConvSource(da)], cfg.wdpack_cache["pack"], cin, cfg.ksize, cfg.dil, cfg.pad)
            cbeg = 0
            for i, (shape, s, ps) in enumerate(zip(ctx.src_shapes, cfg.scales, cfg.shuffles)):
                contributed = shape[1] // 4 if ps else shape[1]
                if need[6 + i]:
                    if not ps and s is None:
                        # a plain source's gradient is a channel slice of dcat: handed out as a VIEW.  Where the tensor has a second consumer
                        # (a1 / a2 of every block feed the next conv AND the concat) the engine's accumulation reads the strided slice directly
                        # and the copy `split_grad` made is gone; a sole consumer's node makes it contiguous on entry, which costs what the copy did
                        dsrc[i] = dcat if nsrc == 1 else dcat.narrow(1, cbeg, contributed)
                    else:
                        dsrc[i] = ops.split_grad(dcat, cbeg, shape, ps, s)
                cbeg += contributed
        return (None, dweight, dbias, dgamma if need[3] else None, dbeta if need[4] else None, d_resid, *dsrc)


class AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        ctx.shape = tuple(x.shape)
        ctx.save_for_backward(scale if scale is not None else torch.empty(0, device=x.device))
        ctx.has_scale = scale is not None
        return ops.avgpool3s2(x.detach().contiguous(), scale)

    @staticmethod
    def backward(ctx, dy):
        (scale,) = ctx.saved_tensors
        return ops.avgpool3s2_bwd(dy.contiguous().float(), scale if ctx.has_scale else None, ctx.shape), None
