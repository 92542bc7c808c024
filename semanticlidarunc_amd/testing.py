"""Deterministic synthetic inputs shared by tests, bench.py and tools/gen_golden.py
(definition: SURVEY.md section 8(d) "Synthetic inputs")."""
from __future__ import annotations

import torch
import torch.nn as nn


def synthetic_scan(batch: int, h: int, w: int, seed: int = 1234, n_classes: int = 20):
    """(x[B,5,H,W] = [range, reflectivity, x, y, z], labels[B,H,W] int64) on the CPU.
    xyz ~ N(0,1)*[20,20,2] m, range = |xyz|, reflectivity ~ U(0,1); 10 % of the pixels are empty
    returns (all channels 0, label 0 = unlabeled)."""
    g = torch.Generator().manual_seed(seed)
    xyz = torch.randn(batch, 3, h, w, generator=g) * torch.tensor([20.0, 20.0, 2.0]).view(1, 3, 1, 1)
    rng = xyz.norm(dim=1, keepdim=True)
    refl = torch.rand(batch, 1, h, w, generator=g)
    x = torch.cat([rng, refl, xyz], dim=1)
    empty = torch.rand(batch, 1, h, w, generator=g) < 0.10
    x = x.masked_fill(empty, 0.0)
    labels = torch.randint(1, n_classes, (batch, h, w), generator=g)
    labels = labels.masked_fill(empty[:, 0], 0)
    return x.contiguous(), labels.contiguous()


def randomize_bn_(model: nn.Module, seed: int = 1) -> nn.Module:
    """Make eval-mode BatchNorm non-trivial: gamma~U(.5,1.5), beta~N(0,.1), mean~N(0,.1), var~U(.5,1.5).
    Modules are visited in sorted-name order so any class with the reference's key names gets the
    same values."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, m in sorted(model.named_modules(), key=lambda kv: kv[0]):
            if isinstance(m, nn.BatchNorm2d):
                c = m.num_features
                m.weight.copy_(torch.rand(c, generator=g) + 0.5)
                m.bias.copy_(torch.randn(c, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(c, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(c, generator=g) + 0.5)
    return model


def seeded_model(cls, nclasses: int = 20, nchannels: int = 5, seed: int = 0, bn_seed: int = 1):
    """cls(nclasses, nchannels) with torch's default init under manual_seed(seed) + randomize_bn_."""
    torch.manual_seed(seed)
    return randomize_bn_(cls(nclasses, nchannels), bn_seed).eval()
