"""Mirror of the reference's ``models.losses`` hot-path symbols (src/models/losses.py:8-73):
``classify_output_kind`` and the ``CrossEntropyLoss`` wrapper.  Same signatures, return values and
``ValueError`` behaviour.  The Lovasz loss lives in ``losses.lovasz`` as in the reference.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from semanticlidarunc_amd import ops


@torch.no_grad()
def classify_output_kind(outputs: torch.Tensor, class_dim: int = 1, sample_fraction: float = 0.1) -> str:
    """'probs' | 'log_probs' | 'logits' from a random 10 % spatial subsample (host decision made
    once on the first batch; consumes the global torch RNG like the reference does)."""
    x = outputs.movedim(class_dim, 1)
    if sample_fraction and sample_fraction < 1.0 and x.ndim > 2:
        flat = x.reshape(x.shape[0], x.shape[1], -1)
        s = flat.size(-1)
        pick = torch.randperm(s, device=x.device)[: max(1, int(s * sample_fraction))]
        x = flat[..., pick]
    lo, hi = float(x.min()), float(x.max())
    tot = x.sum(dim=1)
    one = torch.ones_like(tot)
    if lo >= -1e-6 and hi <= 1 + 1e-6 and torch.allclose(tot, one, atol=1e-3, rtol=1e-3):
        return "probs"
    if hi <= 1e-6 and torch.allclose(x.exp().sum(dim=1), one, atol=1e-3, rtol=1e-3):
        return "log_probs"
    return "logits"


class CrossEntropyLoss(nn.Module):
    """CE on logits / NLL on log(p + 1e-8) / NLL on log-probs, mean over non-ignored pixels;
    labels outside [0, C) are treated as ignored (reference models/losses.py:55-73)."""

    def __init__(self, ignore_index=255):
        super().__init__()
        self.ignore_index = ignore_index

    def forward(self, outputs, labels, num_classes=20, model_act=None):
        if model_act == "logits":
            kind, param = ops.NLL_LOGITS, 0.0
        elif model_act == "probs":
            kind, param = ops.NLL_PROBS_EPS, 1e-8       # NLL over log(p + 1e-8), as the reference
        elif model_act == "log_probs":
            kind, param = ops.NLL_LOG_PROBS, 0.0
        else:
            raise ValueError(f"Unknown model_act: {model_act}")
        from semanticlidarunc_amd.loss import NllFn
        return NllFn.apply(outputs, labels, kind, param, self.ignore_index)



class _TverskyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, outputs, labels, model_act, ignore_index, alpha, beta, smooth, reduction, num_classes):
        x = outputs.detach().float().contiguous()
        lab = labels.detach().to(device=x.device, dtype=torch.int64).contiguous()
        if x.shape[1] != num_classes:
            raise RuntimeError(f"TverskyLoss: outputs have {x.shape[1]} channels, num_classes = {num_classes}")
        loss, coef, _ = ops.tversky_fwd(x, lab, model_act, ignore_index, alpha, beta, smooth, reduction)
        ctx.save_for_backward(x, lab, coef)
        ctx.cfg = (model_act, ignore_index, alpha, beta)
        return loss if reduction == "none" else loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        x, lab, coef = ctx.saved_tensors
        model_act, ignore_index, alpha, beta = ctx.cfg
        gx = ops.tversky_bwd(x, lab, model_act, ignore_index, alpha, beta, coef, g.detach().float().contiguous().reshape(-1))
        return gx, None, None, None, None, None, None, None, None


class TverskyLoss(nn.Module):
    """1 - (TP + s) / (TP + alpha FP + beta FN + s) per class over the valid pixels, reduced over ALL classes
    (reference models/losses.py:74-128); forward and backward are one fused HIP pass each."""

    def __init__(self, alpha=0.9, beta=0.1, smooth=1.0, ignore_index=255, reduction="mean"):
        super().__init__()
        self.alpha, self.beta, self.smooth = alpha, beta, smooth
        self.ignore_index = ignore_index
        self.reduction = reduction

    def forward(self, outputs, labels, num_classes=20, model_act="logits"):
        if model_act not in ops.TVERSKY_ACTS:
            raise ValueError(f"Unknown model_act: {model_act}")
        reduction = self.reduction if self.reduction in ("mean", "sum") else "none"      # the reference returns the per-class vector otherwise
        return _TverskyFn.apply(outputs, labels, model_act, self.ignore_index, self.alpha, self.beta, self.smooth, reduction, int(num_classes))


# drop-in mode (this file shadows the reference's module of the same import path): names it does not define come from there
from semanticlidarunc_amd._shadow import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(__name__, __file__, globals())
