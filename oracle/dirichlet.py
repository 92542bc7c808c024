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


# ---- per-pixel Dirichlet losses (src/losses/dirichlet_losses.py:73-221,317-385; src/losses/regularizers.py:291-389) -----------
def _masked_mean(per_pix, target, ignore_index):
    valid = torch.ones_like(target, dtype=torch.bool) if ignore_index is None else target != ignore_index
    w = valid.float()
    return (per_pix * w).sum() / w.sum().clamp_min(1.0)


def loss_nll_dircat(alpha, target, ignore_index=None, eps=1e-12):
    a0, ay = alpha.sum(dim=1), alpha.gather(1, target.unsqueeze(1)).squeeze(1)
    return _masked_mean(-(torch.log(ay + eps) - torch.log(a0 + eps)), target, ignore_index)


def loss_digamma_ce(alpha, target, ignore_index=None):
    a0, ay = alpha.sum(dim=1), alpha.gather(1, target.unsqueeze(1)).squeeze(1)
    return _masked_mean(torch.digamma(a0) - torch.digamma(ay), target, ignore_index)


def loss_brier(alpha, target, ignore_index=None, s_ref=None, eps=1e-12):
    a0 = alpha.sum(dim=1, keepdim=True)
    p = alpha / (a0 + eps)
    sum_p2 = (p * p).sum(dim=1, keepdim=True)
    s = a0 if s_ref is None else torch.as_tensor(float(s_ref), dtype=alpha.dtype)
    per = ((s * sum_p2 + 1.0) / (s + 1.0) - 2.0 * p.gather(1, target.unsqueeze(1)) + 1.0).squeeze(1)
    return _masked_mean(per, target, ignore_index)


def loss_mse(alpha, target, ignore_index=None, eps=1e-8):
    a0 = alpha.sum(dim=1, keepdim=True)
    p = alpha / (a0 + eps)
    y = torch.zeros_like(alpha).scatter_(1, target.unsqueeze(1), 1.0)
    var = alpha * (a0 - alpha) / ((a0 * a0 + eps) * (a0 + 1.0))
    return _masked_mean(((y - p) ** 2 + var).sum(dim=1), target, ignore_index)


def loss_kl_off_uniform(alpha, target, ignore_index=None, eps=1e-8, with_conf_weighting=False, gamma=1.0):
    """losses/regularizers.py:291-389; with_conf_weighting (:375-385): per-pixel weight (1 - p_y)^gamma, detached, mean over max(sum w, 1)."""
    valid = torch.ones_like(target, dtype=torch.bool) if ignore_index is None else target != ignore_index
    y = torch.zeros_like(alpha).scatter_(1, target.unsqueeze(1), 1.0)
    at = (y + (1.0 - y) * alpha).permute(0, 2, 3, 1).reshape(-1, alpha.shape[1])[valid.reshape(-1)]
    a = at.clamp_min(eps)
    s = a.sum(dim=1, keepdim=True)
    kl = torch.lgamma(s) - torch.lgamma(a).sum(dim=1, keepdim=True) + ((a - 1.0) * (torch.digamma(a) - torch.digamma(s))).sum(dim=1, keepdim=True)
    kl = kl.squeeze(1)
    if not with_conf_weighting:
        return kl.mean()
    p_y = (alpha / (alpha.sum(dim=1, keepdim=True) + eps)).gather(1, target.unsqueeze(1)).squeeze(1)
    w = ((1.0 - p_y).clamp(0.0, 1.0) ** gamma).reshape(alpha.shape[0], -1)[valid.reshape(alpha.shape[0], -1)].reshape(-1).detach()
    return (kl * w).sum() / w.sum().clamp_min(1.0)


def loss_complement_kl(alpha, target, ignore_index=0, gamma=2.0, tau=0.55, sigma=0.12, s_target=None, normalize=True, eps=1e-8, detach_uncert=True):
    """losses/dirichlet_losses.py:228-314: gate(p_y) * KL(off-class conditional || uniform over C-1), mean over valid pixels."""
    c = alpha.shape[1]
    valid = torch.ones_like(target, dtype=torch.bool) if ignore_index is None else target != ignore_index
    if int(valid.sum()) == 0 or c <= 2:
        return alpha.sum() * 0.0
    idx = torch.where(valid, target, torch.zeros_like(target)).unsqueeze(1)
    a0 = alpha.sum(dim=1, keepdim=True) + eps
    p = alpha / a0
    py = p.gather(1, idx).clamp_min(eps)
    cond = p.scatter(1, idx, 0.0) / (1.0 - py).clamp_min(eps)              # distribution over the C-1 other classes
    kl = (cond * cond.clamp_min(eps).log()).sum(dim=1) + math.log(c - 1)
    if normalize:
        kl = kl / math.log(c - 1)
    pg = py.detach() if detach_uncert else py
    w = ((1.0 - pg).pow(gamma) * torch.sigmoid((tau - pg) / sigma)).squeeze(1)
    if s_target is not None:
        w = w * (float(s_target) / (a0.detach().squeeze(1) + float(s_target)))
    return _masked_mean(w * kl, target, ignore_index)


def loss_wrong_low_evidence(alpha, target, ignore_index=None, s_low=0.0, margin=0.05, soft_margin_k=0.08, eps=1e-8):
    """losses/regularizers.py:218-289: squared hinge of ln(alpha0) above ln(C + s_low + eps) on (softly) confidently wrong pixels,
    averaged over the sum of the gates; the gates carry no gradient."""
    c = alpha.shape[1]
    valid = torch.ones_like(target, dtype=torch.bool) if ignore_index is None else target != ignore_index
    if int(valid.sum()) == 0:
        return alpha.sum() * 0.0
    a0 = alpha.sum(dim=1, keepdim=True).clamp_min(eps)
    with torch.no_grad():
        p = alpha.detach() / a0.detach()
        top, pred = p.max(dim=1)
        gap = top.clamp_min(eps) - p.gather(1, target.unsqueeze(1)).squeeze(1).clamp_min(eps)
        if margin > 0.0:
            soft = torch.sigmoid((gap - margin) / soft_margin_k) if soft_margin_k > 0.0 else (gap > margin).float()
        else:
            soft = torch.ones_like(gap)
        gate = (pred != target).float() * soft * valid.float()
    hinge = torch.relu(a0.log().squeeze(1) - math.log(c + s_low + eps)).pow(2) * gate
    return hinge.sum() / gate.sum().clamp_min(1.0)
