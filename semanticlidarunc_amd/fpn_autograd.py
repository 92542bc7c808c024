"""Autograd nodes of the ResNet-FPN training path (reference models/semanticFCN.py:266-354 and baselines/Reichert/semanticFCN_opt.py:366-455;
`trainer.py:783-786` calls `loss.backward()` on them, `utils/grad_norm.py:52` repeated `autograd.grad(..., retain_graph=True)`).

Every node's arithmetic is a HIP kernel (csrc/fpn_train.hip, backward.hip, wgrad.hip, conv2d.hip); backward reads saved tensors only, so
the nodes are re-entrant.  The conv itself is `autograd.ConvLayerFn` (conv [+ bias] [+ LeakyReLU] [+ BatchNorm] [+ residual]); what the
ResNet / FPN graphs need on top of it:

  conv2d(...)            ConvLayerFn with a weight that may be a (differentiable) re-arrangement of a parameter
  batch_norm(...)        BatchNorm2d (train or eval statistics) [+ residual] on its own -- after a sub-sampled conv output
  relu / tanh / elu1     PointwiseFn
  max_pool / nearest_down / replace_tail / row_softmax_mul / depth_to_space / bilinear_up / group_norm / spatial_gate
  silu / depthwise_conv3x3 / global_avg_pool / channel_gate / scale_add      the EfficientNetV2 blocks (MBConv: depthwise conv, squeeze-excitation;
                                                                            StochasticDepth in train mode)
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import ops
from .autograd import ConvLayerFn, LayerCfg, bump_num_batches_tracked
from .ops import ConvSource


def conv2d(srcs: Sequence[torch.Tensor], weight: torch.Tensor, bias: Optional[torch.Tensor], ksize: int, pad: int, dil: int = 1,
           slope: Optional[float] = None, bn: Optional[nn.BatchNorm2d] = None, resid: Optional[torch.Tensor] = None, cache: Optional[dict] = None,
           scales: Optional[Sequence[Optional[torch.Tensor]]] = None):
    """conv(cat(srcs)) [+ bias] [-> LeakyReLU(slope)] [-> bn] [+ resid], stride 1, as one autograd node.  `weight` [Cout, Cin, k, k] may be any
    tensor (a parameter or a re-arrangement of one: autograd carries its gradient on).  cache: a dict that lives as long as the layer (packed
    data-gradient weights keyed by the weight's version).  scales: per source, an [N, C] multiplier without gradient (a Dropout2d draw) or None."""
    w = weight.contiguous()
    wpack = ops.pack_conv_weight(w.detach().float())
    if cache is None or not weight.is_leaf:
        # the packed data-gradient weights are cached under (data_ptr, version) of the weight: only safe for a long-lived parameter -- a
        # re-arranged copy is a fresh tensor every forward, and the allocator hands the same address out again after an optimizer step
        cache = {}
    cfg = LayerCfg(ksize, dil, pad, slope, list(scales) if scales is not None else [None] * len(srcs), [False] * len(srcs), bn, w.shape[0], wpack, cache, "fp32")
    return ConvLayerFn.apply(cfg, w, bias, None if bn is None else bn.weight, None if bn is None else bn.bias, resid, *srcs)


class BatchNormFn(torch.autograd.Function):
    """z = BatchNorm2d(y) [+ resid], batch statistics in train mode (running statistics updated as nn.BatchNorm2d does), else the running ones."""

    @staticmethod
    def forward(ctx, bn: nn.BatchNorm2d, y, gamma, beta, resid):
        y = y.detach().contiguous()
        n, c, h, w = y.shape
        train = bool(bn.training)
        sums = ops.bn_stats(y) if train else None
        track = train and bn.track_running_stats and bn.running_mean is not None
        mom = 0.0
        if track:
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
        mean, invstd, a, b = ops.bn_coeffs_fwd(sums, n * h * w, gamma.detach(), beta.detach(), bn.eps, mom,
                                               bn.running_mean if (track or not train) else None, bn.running_var if (track or not train) else None, train)
        if track:
            bump_num_batches_tracked(bn)
        z = ops.affine(y, a, b, None if resid is None else resid.detach().contiguous())
        ctx.train, ctx.has_resid = train, resid is not None
        ctx.save_for_backward(y, gamma, mean, invstd)
        return z

    @staticmethod
    def backward(ctx, dz):
        y, gamma, mean, invstd = ctx.saved_tensors
        dz = dz.contiguous().float()
        n, c, h, w = y.shape
        s1, s2 = ops.bn_bwd_reduce(dz, y, mean, invstd)
        k1, k2, k3, dgamma, dbeta = ops.bn_coeffs_bwd(s1, s2, float(n * h * w), gamma.detach(), mean, invstd, ctx.train)
        if not ctx.train:
            k2 = k3 = None
        dy, _ = ops.act_affine_bwd(dz, y if k3 is not None else None, k1, k2, k3, None, False)
        need = ctx.needs_input_grad          # (bn, y, gamma, beta, resid)
        return None, dy if need[1] else None, dgamma if need[2] else None, dbeta if need[3] else None, dz if (ctx.has_resid and need[4]) else None


def batch_norm(bn: nn.BatchNorm2d, y, resid=None):
    return BatchNormFn.apply(bn, y, bn.weight, bn.bias, resid)


class PointwiseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op: str, slope: float):
        y = ops.pointwise_fwd(x.detach().contiguous(), op, slope)
        ctx.op, ctx.slope = op, slope
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.pointwise_bwd(dy.contiguous().float(), y, ctx.op, ctx.slope), None, None


def relu(x):
    return PointwiseFn.apply(x, "leaky", 0.0)


def tanh(x):
    return PointwiseFn.apply(x, "tanh", 0.0)


def elu_plus_one(x):
    return PointwiseFn.apply(x, "elu+1", 0.0)


class MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.detach().contiguous()
        ctx.save_for_backward(x)
        return ops.maxpool3s2(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.maxpool3s2_bwd(x, dy.contiguous().float())


class NearestDownFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, factor: int):
        ctx.factor = int(factor)
        return ops.nearest_down(x.detach().contiguous(), int(factor))

    @staticmethod
    def backward(ctx, dy):
        return ops.nearest_down_bwd(dy.contiguous().float(), ctx.factor), None


class ReplaceTailFn(torch.autograd.Function):
    """cat(x[:, :C-m], meta): the reference's meta-channel injection (semanticFCN.py:309-313)."""

    @staticmethod
    def forward(ctx, x, meta):
        ctx.m = int(meta.shape[1])
        return ops.replace_tail(x.detach().contiguous(), meta.detach().contiguous())

    @staticmethod
    def backward(ctx, dout):
        need = ctx.needs_input_grad
        dx, dmeta = ops.replace_tail_bwd(dout.contiguous().float(), ctx.m, need[0], need[1])
        return dx, dmeta


class RowSoftmaxMulFn(torch.autograd.Function):
    """value * softmax(score, dim=-1)  (AttentionModule, semanticFCN.py:35-38)."""

    @staticmethod
    def forward(ctx, score, value):
        score, value = score.detach().contiguous(), value.detach().contiguous()
        ctx.save_for_backward(score, value)
        return ops.row_softmax_mul(score, value)

    @staticmethod
    def backward(ctx, dout):
        score, value = ctx.saved_tensors
        need = ctx.needs_input_grad
        return ops.row_softmax_mul_bwd(score, value, dout.contiguous().float(), need[0], need[1])


class DepthToSpaceCatFn(torch.autograd.Function):
    """cat([depth_to_space(y_k, r_k) for k], dim=1): the up-sampled maps of the three ConvTranspose2d(k = s) layers in one buffer."""

    @staticmethod
    def forward(ctx, rs, *ys):
        outs = [y.shape[1] // (r * r) for y, r in zip(ys, rs)]
        n, _, h, w = ys[0].shape
        buf = torch.empty((n, sum(outs), h * rs[0], w * rs[0]), dtype=torch.float32, device=ys[0].device)
        off = 0
        for y, r, c in zip(ys, rs, outs):
            ops.depth_to_space(y.detach().contiguous(), r, False, buf, off)
            off += c
        ctx.rs, ctx.outs = tuple(rs), tuple(outs)
        return buf

    @staticmethod
    def backward(ctx, dbuf):
        dbuf = dbuf.contiguous().float()
        grads, off = [], 0
        for k, (r, c) in enumerate(zip(ctx.rs, ctx.outs)):
            grads.append(ops.depth_to_space_bwd(dbuf, c, r, off) if ctx.needs_input_grad[1 + k] else None)
            off += c
        return (None, *grads)


def depth_to_space(y, r: int):
    return DepthToSpaceCatFn.apply((int(r),), y)


class BilinearUpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale: int):
        ctx.scale = int(scale)
        return ops.bilinear_upsample(x.detach().contiguous(), int(scale))

    @staticmethod
    def backward(ctx, dy):
        return ops.bilinear_upsample_bwd(dy.contiguous().float(), ctx.scale), None


class GroupNormFn(torch.autograd.Function):
    """nn.GroupNorm(groups, C, eps) [-> ReLU] (UpsampleBlock / decoder of semanticFCN_opt)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups: int, eps: float, relu_after: bool):
        x = x.detach().contiguous()
        y, stats = ops.groupnorm(x, groups, None if gamma is None else gamma.detach(), None if beta is None else beta.detach(), eps, relu_after,
                                 return_stats=True)
        ctx.groups, ctx.relu = int(groups), bool(relu_after)
        ctx.has_affine = gamma is not None
        ctx.save_for_backward(x, y, stats, gamma if gamma is not None else torch.empty(0, device=x.device))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, stats, gamma = ctx.saved_tensors
        need = ctx.needs_input_grad
        dx, dg, db = ops.groupnorm_bwd(x, y if ctx.relu else None, dy.contiguous().float(), gamma.detach() if ctx.has_affine else None, stats, ctx.groups,
                                       ctx.relu, ctx.has_affine and (need[1] or need[2]))
        return (dx if need[0] else None, dg.float() if (dg is not None and need[1]) else None, db.float() if (db is not None and need[2]) else None,
                None, None, None)


def group_norm(gn: nn.GroupNorm, x, relu_after: bool = False):
    return GroupNormFn.apply(x, gn.weight, gn.bias, gn.num_groups, gn.eps, relu_after)


class SpatialGateFn(torch.autograd.Function):
    """x * softmax(score over H*W) + x  (SpatialAttention, semanticFCN_opt.py:80-85)."""

    @staticmethod
    def forward(ctx, x, score):
        x, score = x.detach().contiguous(), score.detach().contiguous()
        out, stats = ops.spatial_softmax_gate(x, score, return_stats=True)
        ctx.save_for_backward(x, score, stats)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, score, stats = ctx.saved_tensors
        dx, dscore = ops.spatial_softmax_gate_bwd(x, score, stats, dout.contiguous().float())
        return dx, dscore


class ScaleChannelsFn(torch.autograd.Function):
    """x * scale[n, c] with a multiplier that carries no gradient (a Dropout2d draw): `dropout_pyramid` of semanticFCN_opt."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.save_for_backward(scale)
        return _scale_nc(x.detach().contiguous(), scale)

    @staticmethod
    def backward(ctx, dy):
        (scale,) = ctx.saved_tensors
        return _scale_nc(dy.contiguous().float(), scale), None


def _scale_nc(x, scale):
    """x[n, c] * scale[n, c]: the per-(sample, channel) form of slu_affine_fwd (which takes per-channel vectors) -- the batch is folded into the
    channel axis."""
    n, c, h, w = x.shape
    return ops.affine(x.view(1, n * c, h, w), scale.reshape(n * c).contiguous().float(), None, None).view(n, c, h, w)


# ---------------------------------------------------------------------------------------------------------------------------------------
# EfficientNetV2 blocks (torchvision's FusedMBConv / MBConv; semanticFCN_opt.py:170-180,238-247,396-404 uses features[0], [2], [3], [4])
# ---------------------------------------------------------------------------------------------------------------------------------------
class SiluFn(torch.autograd.Function):
    """x * sigmoid(x); the backward needs the INPUT (SiLU is not monotone, its output does not determine it)."""

    @staticmethod
    def forward(ctx, x):
        x = x.detach().contiguous()
        ctx.save_for_backward(x)
        return ops.pointwise_fwd(x, "silu")

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.pointwise_bwd(dy.contiguous().float(), x, "silu")


def silu(x):
    return SiluFn.apply(x)


class DepthwiseConv3x3Fn(torch.autograd.Function):
    """Depthwise 3x3 / stride 1 / pad 1, no bias: weight [C, 1, 3, 3].  Data gradient = the same kernel with the nine taps reversed."""

    @staticmethod
    def forward(ctx, x, weight):
        x = x.detach().contiguous()
        w9 = weight.detach().float().reshape(weight.shape[0], 9).contiguous()
        ctx.save_for_backward(x, w9)
        return ops.dwconv3x3(x, w9, None, 1, "none")

    @staticmethod
    def backward(ctx, dy):
        x, w9 = ctx.saved_tensors
        dy = dy.contiguous().float()
        need = ctx.needs_input_grad
        dx = ops.dwconv3x3(dy, w9.flip(1).contiguous(), None, 1, "none") if need[0] else None
        dw = ops.dwconv3x3_wgrad(x, dy).view(-1, 1, 3, 3) if need[1] else None
        return dx, dw


class GlobalAvgPoolFn(torch.autograd.Function):
    """AdaptiveAvgPool2d(1) as [N, C] (SqueezeExcitation.avgpool)."""

    @staticmethod
    def forward(ctx, x):
        ctx.hw = (int(x.shape[2]), int(x.shape[3]))
        return ops.global_avgpool(x.detach().contiguous())

    @staticmethod
    def backward(ctx, ds):
        h, w = ctx.hw
        n, c = ds.shape
        return (ds.float() / float(h * w)).view(n, c, 1, 1).expand(n, c, h, w)      # a broadcast view: the engine's accumulation materialises it


class ChannelGateFn(torch.autograd.Function):
    """x * g[n, c] with a gradient for BOTH (SqueezeExcitation's `scale * input`)."""

    @staticmethod
    def forward(ctx, x, g):
        x, g = x.detach().contiguous(), g.detach().contiguous().float()
        ctx.save_for_backward(x, g)
        return _scale_nc(x, g)

    @staticmethod
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        dy = dy.contiguous().float()
        need = ctx.needs_input_grad
        n, c, h, w = x.shape
        dx = _scale_nc(dy, g) if need[0] else None
        dg = None
        if need[1]:
            # sum over H W of dy * x per (sample, channel): the BatchNorm-backward reduction with mean 0 / invstd 1 on the [1, N C, H, W] view
            zero, one = torch.zeros(n * c, dtype=torch.float32, device=x.device), torch.ones(n * c, dtype=torch.float32, device=x.device)
            _, s2 = ops.bn_bwd_reduce(dy.view(1, n * c, h, w), x.view(1, n * c, h, w), zero, one)
            dg = s2.float().view(n, c)
        return dx, dg


class ScaleAddFn(torch.autograd.Function):
    """y * noise[n] + x: torchvision's StochasticDepth('row') in train mode followed by the residual add (noise carries no gradient)."""

    @staticmethod
    def forward(ctx, y, noise_nc, x):
        y, x = y.detach().contiguous(), x.detach().contiguous()
        n, c, h, w = y.shape
        ctx.save_for_backward(noise_nc)
        return ops.affine(y.view(1, n * c, h, w), noise_nc.reshape(n * c).contiguous().float(), None, x.view(1, n * c, h, w)).view(n, c, h, w)

    @staticmethod
    def backward(ctx, dout):
        (noise_nc,) = ctx.saved_tensors
        dout = dout.contiguous().float()
        need = ctx.needs_input_grad
        return _scale_nc(dout, noise_nc) if need[0] else None, None, dout if need[2] else None
