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
