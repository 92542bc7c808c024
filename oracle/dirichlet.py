"""Oracle: Dirichlet head and its uncertainty measures.  TEST INFRASTRUCTURE ONLY.

Restates ``src/models/probability_helper.py:89-105`` (alpha from shape / scale logits), ``:116-136``
(predictive entropy, aleatoric, epistemic), ``:148-153`` (normalised entropy) and the channel split /
``p_hat`` of ``src/models/trainer.py:533-538`` with torch CPU ops.  Pinned by ``tools/gen_golden.py``
against the imported reference module (0.0 max-abs difference).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch.special import digamma

EPS = 1e-8      # probability_helper.py:14
T = 1.0         # probability_helper.py:15


def alpha_from_shape_and_scale(shape_logits, scale_logits, t: float = T, eps: float = EPS):
    """alpha = 1 + softplus(scale / T) * softmax(shape, dim=1) + eps   (probability_helper.py:89-105)"""
    return 1.0 + F.softplus(scale_logits / t) * F.softmax(shape_logits, dim=1) + eps


def predictive_entropy(alpha, eps: float = EPS):
    a0 = alpha.sum(dim=1, keepdim=True) + eps
    p = alpha / a0
    return -(p * torch.log(p + eps)).sum(dim=1)


def predictive_entropy_norm(alpha, eps: float = EPS):
    return predictive_entropy(alpha, eps) / math.log(alpha.shape[1])


def aleatoric(alpha, eps: float = EPS):
    a0 = alpha.sum(dim=1, keepdim=True) + eps
    return -((alpha / a0) * (digamma(alpha + 1.0) - digamma(a0 + 1.0))).sum(dim=1)


def epistemic(alpha, eps: float = EPS):
    return predictive_entropy(alpha, eps) - aleatoric(alpha, eps)


def head(outputs, num_classes: int, t: float = T, eps: float = EPS):
    """trainer.py:533-538: outputs [B, C+1, H, W] -> (alpha, p_hat, H_norm, preds)."""
    alpha = alpha_from_shape_and_scale(outputs[:, :num_classes], outputs[:, num_classes:num_classes + 1], t, eps)
    p_hat = alpha / (alpha.sum(dim=1, keepdim=True) + eps)
    return alpha, p_hat, predictive_entropy_norm(alpha, eps), alpha.argmax(dim=1)
