"""Drop-ins for the per-pixel Dirichlet losses of the reference (``src/losses/dirichlet_losses.py``:
``NLLDirichletCategorical`` :73-119, ``DigammaDirichletCE`` :122-167, ``BrierDirichlet`` :174-221, ``ComplementKLUniform`` :228-314,
``DirichletMSELoss`` :317-385):
same constructors and ``forward(alpha, target)``; value and d/d alpha are one fused HIP pass each (``csrc/dirichlet_loss.hip``)
behind a re-entrant ``torch.autograd.Function``.  ``ignore_index``: None or one int (what the reference's Trainer passes)."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from semanticlidarunc_amd import ops


class _DirichletLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alpha, target, kind, param, eps, ignore_index):
        a = alpha.detach().float().contiguous()
        if target.dim() == 4 and target.size(1) == 1:
            target = target[:, 0]
        lab = target.detach().to(device=a.device, dtype=torch.int64).contiguous()
        s, n = ops.dirichlet_loss_fwd(a, lab, kind, param, eps, ignore_index)
        ctx.save_for_backward(a, lab, n)
        ctx.cfg = (kind, param, eps, ignore_index)
        return (s / n.clamp_min(1).to(torch.float64)).to(torch.float32).reshape(())       # 0 valid pixels -> 0, as the reference

    @staticmethod
    def backward(ctx, g):
        a, lab, n = ctx.saved_tensors
        kind, param, eps, ignore_index = ctx.cfg
        gscale = (g.detach().reshape(1).to(torch.float64) / n.clamp_min(1).to(torch.float64)).to(torch.float32)
        return ops.dirichlet_loss_bwd(a, lab, kind, param, eps, ignore_index, gscale), None, None, None, None, None


class _DirichletLossExFn(torch.autograd.Function):
    """The parameter-vector form (kinds "complement_kl", "wrong_low_evidence"); `gated`: the mean runs over the sum of the gates."""

    @staticmethod
    def forward(ctx, alpha, target, kind, params, eps, ignore_index, gated):
        a = alpha.detach().float().contiguous()
        if target.dim() == 4 and target.size(1) == 1:
            target = target[:, 0]
        lab = target.detach().to(device=a.device, dtype=torch.int64).contiguous()
        s, n = ops.dirichlet_loss_fwd_ex(a, lab, kind, params, eps, ignore_index)
        den = s[1:2].clamp_min(1.0) if gated else n.clamp_min(1).to(torch.float64)
        ctx.save_for_backward(a, lab, den)
        ctx.cfg = (kind, tuple(params), eps, ignore_index)
        return (s[0:1] / den).to(torch.float32).reshape(())

    @staticmethod
    def backward(ctx, g):
        a, lab, den = ctx.saved_tensors
        kind, params, eps, ignore_index = ctx.cfg
        gscale = (g.detach().reshape(1).to(torch.float64) / den).to(torch.float32)
        return ops.dirichlet_loss_bwd_ex(a, lab, kind, params, eps, ignore_index, gscale), None, None, None, None, None, None


def _check_ignore(ignore_index):
    if ignore_index is not None and not isinstance(ignore_index, int):
        raise NotImplementedError("the HIP Dirichlet losses take ignore_index = None or one int")
    return ignore_index


class NLLDirichletCategorical(nn.Module):
    """-log E[p_y] = -(log(alpha_y + eps) - log(alpha0 + eps)), mean over valid pixels."""

    def __init__(self, ignore_index: Optional[int] = None, eps: float = 1e-12):
        super().__init__()
        self.ignore_index, self.eps = _check_ignore(ignore_index), eps

    def forward(self, alpha, target):
        return _DirichletLossFn.apply(alpha, target, "nll_dircat", 0.0, self.eps, self.ignore_index)


class DigammaDirichletCE(nn.Module):
    """E[-log p_y] = psi(alpha0) - psi(alpha_y), mean over valid pixels."""

    def __init__(self, ignore_index: Optional[int] = None, eps: float = 1e-8):
        super().__init__()
        self.ignore_index, self.eps = _check_ignore(ignore_index), eps

    def forward(self, alpha, target):
        return _DirichletLossFn.apply(alpha, target, "digamma_ce", 0.0, self.eps, self.ignore_index)


class BrierDirichlet(nn.Module):
    """Expected Brier score under the Dirichlet; ``s_ref`` replaces alpha0 in E[p_i^2] (scale-free form)."""

    def __init__(self, ignore_index: Optional[int] = None, s_ref: Optional[float] = None, eps: float = 1e-12):
        super().__init__()
        self.ignore_index, self.s_ref, self.eps = _check_ignore(ignore_index), s_ref, eps
        if s_ref is not None and s_ref < 0:
            raise ValueError("s_ref must be non-negative")

    def forward(self, alpha, target):
        return _DirichletLossFn.apply(alpha, target, "brier", -1.0 if self.s_ref is None else float(self.s_ref), self.eps, self.ignore_index)


class ComplementKLUniform(nn.Module):
    """w(p_y) * KL(p_off / (1 - p_y) || uniform over the C - 1 off classes), w = (1 - p_y)^gamma * sigmoid((tau - p_y) / sigma)
    [* s_target / (alpha0 + s_target)], mean over valid pixels; 0 for C <= 2 (reference :228-314, same defaults)."""

    def __init__(self, ignore_index: Optional[int] = 0, gamma: float = 2.0, tau: float = 0.55, sigma: float = 0.12,
                 s_target: Optional[float] = None, normalize: bool = True, eps: float = 1e-8, detach_uncert: bool = True):
        super().__init__()
        self.ignore_index = _check_ignore(ignore_index)
        self.gamma, self.tau, self.sigma = float(gamma), float(tau), float(sigma)
        self.s_target, self.normalize, self.eps, self.detach_uncert = s_target, bool(normalize), eps, bool(detach_uncert)
        if s_target is not None and float(s_target) < 0:
            raise ValueError("s_target must be non-negative")

    def forward(self, alpha, target):
        if alpha.shape[1] <= 2:
            return alpha.sum() * 0.0
        params = (self.gamma, self.tau, self.sigma, -1.0 if self.s_target is None else float(self.s_target), float(self.normalize),
                  float(self.detach_uncert))
        return _DirichletLossExFn.apply(alpha, target, "complement_kl", params, self.eps, self.ignore_index, False)


class DirichletMSELoss(nn.Module):
    """Expected squared error + predictive variance (Sensoy et al. 2018, eq. 5), mean over valid pixels; 0 for C <= 2 like the reference."""

    def __init__(self, ignore_index: Optional[int] = None, eps: float = 1e-8):
        super().__init__()
        self.ignore_index, self.eps = _check_ignore(ignore_index), eps

    def forward(self, alpha, target):
        if alpha.shape[1] <= 2:
            return alpha.sum() * 0.0
        return _DirichletLossFn.apply(alpha, target, "mse", 0.0, self.eps, self.ignore_index)


# drop-in mode (this file shadows the reference's module of the same import path): names it does not define come from there
from semanticlidarunc_amd._shadow import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(__name__, __file__, globals())
