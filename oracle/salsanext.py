"""Oracle: SalsaNext forward as a pure function of a ``state_dict``.  TEST INFRASTRUCTURE ONLY.

Restates ``src/baselines/SalsaNext/SalsaNext.py:10-215`` of the reference with
``torch.nn.functional`` CPU ops.  The layer order is conv -> LeakyReLU(0.01) -> BatchNorm
(reference ``SalsaNext.py:30-36``), channel dropout is expressed as an explicit
per-(sample, channel) multiplier (``Dropout2d`` == ``x * bernoulli(1-p)/(1-p)`` broadcast
over H, W) so that a test can hand the very same multipliers to the HIP path.

Pinned by ``tools/gen_golden.py`` against the imported reference class (max abs diff
<= 1e-6 on seeded inputs, eval mode and train-mode BatchNorm).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

SLOPE = 0.01          # nn.LeakyReLU() default, SalsaNext.py:14
BN_EPS = 1e-5         # nn.BatchNorm2d default
BN_MOMENTUM = 0.1

# (module prefix, channel count) of every Dropout2d that is applied in forward, in call order
# (SalsaNext.py:98,106 for ResBlock; :145,149,168 for UpBlock; resBlock1/upBlock4 have drop_out=False)
DROPOUT_SITES = (
    ("resBlock2.dropout", 128), ("resBlock3.dropout", 256), ("resBlock4.dropout", 256),
    ("resBlock5.dropout", 256),
    ("upBlock1.dropout1", 64), ("upBlock1.dropout2", 320), ("upBlock1.dropout3", 128),
    ("upBlock2.dropout1", 32), ("upBlock2.dropout2", 288), ("upBlock2.dropout3", 128),
    ("upBlock3.dropout1", 32), ("upBlock3.dropout2", 160), ("upBlock3.dropout3", 64),
)


def draw_dropout_scales(n: int, p: float = 0.2, generator: torch.Generator | None = None):
    """Per-(sample, channel) multipliers {0, 1/(1-p)} for every active dropout site."""
    out = {}
    for name, c in DROPOUT_SITES:
        keep = torch.bernoulli(torch.full((n, c, 1, 1), 1.0 - p), generator=generator)
        out[name] = keep / (1.0 - p)
    return out


class _Net:
    def __init__(self, sd, bn_train, scales):
        self.sd, self.bn_train, self.scales = sd, bn_train, scales or {}
        self.bn_batch_stats = {}

    def conv(self, x, name, pad=0, dil=1):
        return F.conv2d(x, self.sd[name + ".weight"], self.sd[name + ".bias"], padding=pad, dilation=dil)

    def bn(self, x, name):
        w, b = self.sd[name + ".weight"], self.sd[name + ".bias"]
        rm, rv = self.sd[name + ".running_mean"], self.sd[name + ".running_var"]
        if self.bn_train:
            # batch statistics (biased variance), exactly nn.BatchNorm2d in train mode; the running-stat
            # update (momentum 0.1, unbiased variance) is left to the caller via bn_batch_stats
            self.bn_batch_stats[name] = (x.detach().mean(dim=(0, 2, 3)), x.detach().var(dim=(0, 2, 3), unbiased=False))
            return F.batch_norm(x, None, None, w, b, True, 0.0, BN_EPS)
        return F.batch_norm(x, rm, rv, w, b, False, 0.0, BN_EPS)

    def cab(self, x, conv, bn, pad=0, dil=1):
        """conv -> LeakyReLU -> (BatchNorm)"""
        y = F.leaky_relu(self.conv(x, conv, pad, dil), SLOPE)
        return self.bn(y, bn) if bn else y

    def drop(self, x, name):
        s = self.scales.get(name)
        return x if s is None else x * s

    def context(self, x, p):                       # ResContextBlock, SalsaNext.py:25-39
        sc = self.cab(x, p + ".conv1", None)
        a1 = self.cab(sc, p + ".conv2", p + ".bn1", 1, 1)
        a2 = self.cab(a1, p + ".conv3", p + ".bn2", 2, 2)
        return sc + a2

    def res(self, x, p, pooling, drop):            # ResBlock, SalsaNext.py:73-109
        sc = self.cab(x, p + ".conv1", None)
        a1 = self.cab(x, p + ".conv2", p + ".bn1", 1, 1)
        a2 = self.cab(a1, p + ".conv3", p + ".bn2", 2, 2)
        a3 = self.cab(a2, p + ".conv4", p + ".bn3", 1, 2)
        a = self.cab(torch.cat((a1, a2, a3), 1), p + ".conv5", p + ".bn4")
        a = sc + a
        b = self.drop(a, p + ".dropout") if drop else a
        if pooling:
            return F.avg_pool2d(b, 3, 2, 1), a
        return b

    def up(self, x, skip, p, drop):                # UpBlock, SalsaNext.py:142-170
        u = F.pixel_shuffle(x, 2)
        if drop:
            u = self.drop(u, p + ".dropout1")
        u = torch.cat((u, skip), 1)
        if drop:
            u = self.drop(u, p + ".dropout2")
        e1 = self.cab(u, p + ".conv1", p + ".bn1", 1, 1)
        e2 = self.cab(e1, p + ".conv2", p + ".bn2", 2, 2)
        e3 = self.cab(e2, p + ".conv3", p + ".bn3", 1, 2)
        e = self.cab(torch.cat((e1, e2, e3), 1), p + ".conv4", p + ".bn4")
        if drop:
            e = self.drop(e, p + ".dropout3")
        return e


def salsanext_forward(sd, x, dropout_scales=None, bn_train=False, return_bn_stats=False):
    """logits[B,ncls,H,W] = SalsaNext(x[B,nch,H,W]) -- SalsaNext.py:197-215.

    ``dropout_scales``: dict site-name -> [B,C,1,1] multiplier (see DROPOUT_SITES); missing
    sites are the identity (== dropout in eval mode).
    """
    net = _Net(sd, bn_train, dropout_scales)
    d = net.context(x, "downCntx")
    d = net.context(d, "downCntx2")
    d = net.context(d, "downCntx3")
    d0c, d0b = net.res(d, "resBlock1", True, False)
    d1c, d1b = net.res(d0c, "resBlock2", True, True)
    d2c, d2b = net.res(d1c, "resBlock3", True, True)
    d3c, d3b = net.res(d2c, "resBlock4", True, True)
    d5c = net.res(d3c, "resBlock5", False, True)
    u4 = net.up(d5c, d3b, "upBlock1", True)
    u3 = net.up(u4, d2b, "upBlock2", True)
    u2 = net.up(u3, d1b, "upBlock3", True)
    u1 = net.up(u2, d0b, "upBlock4", False)
    logits = net.conv(u1, "logits")
    if return_bn_stats:
        return logits, net.bn_batch_stats
    return logits


# ----------------------------------------------------------------------------------------------
# single ops (per-kernel oracles)
# ----------------------------------------------------------------------------------------------
def pixel_shuffle2(x):
    """out[n,c,2h+i,2w+j] = in[n,4c+2i+j,h,w]  (nn.PixelShuffle(2), SalsaNext.py:143)"""
    return F.pixel_shuffle(x, 2)


def fused_conv(srcs, weight, bias, pad, dil, slope=None, bn_a=None, bn_b=None, resid=None):
    """The fused operator the HIP conv kernel implements.

    srcs: list of (tensor[N,C,h,w], scale[N,C] or None, pixel_shuffle: bool); the sources are
    concatenated along channels after optional PixelShuffle(2) and per-(n,c) scaling.
    out = resid + bn_a * leaky(conv(cat) + bias) + bn_b
    """
    parts = []
    for t, s, ps in srcs:
        if s is not None:
            t = t * s.reshape(s.shape[0], s.shape[1], 1, 1)
        if ps:
            t = F.pixel_shuffle(t, 2)
        parts.append(t)
    x = torch.cat(parts, 1) if len(parts) > 1 else parts[0]
    y = F.conv2d(x, weight, bias, padding=pad, dilation=dil)
    if slope is not None:
        y = F.leaky_relu(y, slope)
    if bn_a is not None:
        y = y * bn_a[None, :, None, None] + bn_b[None, :, None, None]
    if resid is not None:
        y = y + resid
    return y


def avgpool3s2(x, scale=None):
    """AvgPool2d(3, stride 2, pad 1, count_include_pad=True) of x * scale[n,c] (SalsaNext.py:69,98-101)."""
    if scale is not None:
        x = x * scale.reshape(scale.shape[0], scale.shape[1], 1, 1)
    return F.avg_pool2d(x, 3, 2, 1)
