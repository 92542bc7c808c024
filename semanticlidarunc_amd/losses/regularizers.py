"""Drop-ins for two regularizers of the reference's default Dirichlet loss (``src/losses/regularizers.py``), each one fused HIP
forward / backward pass (``csrc/dirichlet_loss.hip``):
``KL_offClasses_to_uniform`` (:291-389, the ``w_kl`` term): KL(Dir(alpha~) || Dir(1, ..., 1)) with the true class's alpha replaced
by 1, mean over valid pixels, or -- ``with_conf_weighting=True`` -- weighted per pixel by the detached (1 - p_y)^gamma and averaged over
the sum of the weights;
``WrongLowEvidence`` (:218-289, the ``w_wle`` term): squared hinge on ln(alpha0) above ln(C + s_low) on confidently wrong pixels.
``LogitRegularizer`` / ``EvidenceReg(Band)`` come from the reference module in drop-in mode."""
from __future__ import annotations

from typing import Optional

import torch.nn as nn

from .dirichlet_losses import _check_ignore, _DirichletLossExFn, _DirichletLossFn


class KL_offClasses_to_uniform(nn.Module):
    def __init__(self, ignore_index: Optional[int] = None, with_conf_weighting: bool = False, gamma: float = 1.0, eps: float = 1e-8):
        super().__init__()
        self.ignore_index, self.eps, self.with_conf_weighting, self.gamma = _check_ignore(ignore_index), eps, bool(with_conf_weighting), gamma

    def forward(self, alpha, target):
        if self.with_conf_weighting:      # per-pixel weight (1 - p_y)^gamma (detached), mean over max(sum of the weights, 1): :375-385
            return _DirichletLossExFn.apply(alpha, target, "kl_off_uniform_weighted", (float(self.gamma),), self.eps, self.ignore_index, True)
        return _DirichletLossFn.apply(alpha, target, "kl_off_uniform", 0.0, self.eps, self.ignore_index)


class WrongLowEvidence(nn.Module):
    """gate * relu(ln alpha0 - ln(C + s_low + eps))^2, gate = [argmax p != y] * sigmoid((p_max - p_y - margin) / soft_margin_k)
    (soft_margin_k = 0: hard margin; margin <= 0: no margin gate), averaged over sum(gate) (reference :218-289, same defaults)."""

    def __init__(self, ignore_index=None, s_low: float = 0.0, margin: float = 0.05, soft_margin_k: float = 0.08, eps: float = 1e-8):
        super().__init__()
        self.ignore_index = _check_ignore(ignore_index)
        self.s_low, self.margin, self.k, self.eps = float(s_low), float(margin), float(soft_margin_k), float(eps)

    def forward(self, alpha, target):
        return _DirichletLossExFn.apply(alpha, target, "wrong_low_evidence", (self.s_low, self.margin, self.k), self.eps, self.ignore_index, True)


# drop-in mode (this file shadows the reference's module of the same import path): names it does not define come from there
from semanticlidarunc_amd._shadow import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(__name__, __file__, globals())
