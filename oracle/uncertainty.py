"""Oracle: MC-dropout reduction and per-pixel entropy maps.  TEST INFRASTRUCTURE ONLY.

The reference keeps these as closures inside ``Trainer.test_one_epoch`` (not importable:
``models/trainer.py`` needs cv2/tensorboard), so they are restated from
``src/models/trainer.py:1105-1136,1143-1154`` (MC) and ``:1180-1214`` (single pass) and pinned
by hand-derived known answers in ``tests/test_oracle.py`` plus the importable twin
``utils.mc_dropout.predictive_entropy_mc`` (``src/utils/mc_dropout.py:121-133``).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def mc_reduce(mc_logits: torch.Tensor, eps: float = 1e-12):
    """mc_logits [T,B,C,H,W] -> (p_bar[B,C,H,W], H_norm[B,H,W], MI_norm[B,H,W], preds[B,H,W] int64).

    trainer.py:1143-1154: probs = exp(log_softmax(dim=2)); p_bar = mean_T;
    H = -sum_c clamp(p_bar,eps) log clamp(p_bar,eps) / ln C;
    MI = clamp_min((H_bar - mean_T H[p_t]) / ln C, 0); preds = argmax_c p_bar.
    """
    probs = F.log_softmax(mc_logits, dim=2).exp()
    p_bar = probs.mean(dim=0)
    c = p_bar.size(1)
    pb = p_bar.clamp_min(eps)
    h_bar = -(pb * pb.log()).sum(dim=1)
    pt = probs.clamp_min(eps)
    h_t = -(pt * pt.log()).sum(dim=2)
    mi = ((h_bar - h_t.mean(dim=0)) / math.log(c)).clamp_min(0.0)
    return p_bar, h_bar / math.log(c), mi, p_bar.argmax(dim=1)


def single_pass(logits: torch.Tensor, eps: float = 1e-8):
    """logits [B,C,H,W] -> (probs, H_norm, preds).  trainer.py:1180-1184,1211-1214:
    probs = exp(log_softmax); H = -sum p * log(clamp(p, min=eps)) / ln C  (eps = _EPS = 1e-8)."""
    probs = F.log_softmax(logits, dim=1).exp()
    h = -(probs * torch.clamp(probs, min=eps).log()).sum(dim=1)
    return probs, h / math.log(logits.size(1)), probs.argmax(dim=1)
