"""Mirror of the reference's ``losses.lovasz`` (src/losses/lovasz.py:6-23): same class name, constructor and
``forward(outputs, labels, model_act)`` contract, ``ValueError`` on an unknown ``model_act``.

The arithmetic is the batched device radix sort + Jaccard scan of ``csrc/lovasz.hip``; the value equals the
reference's to fp32 rounding and the gradient w.r.t. the probabilities is exact wherever the sorted errors
are distinct (ties make the reference's own sub-gradient order-dependent).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from semanticlidarunc_amd.loss import LovaszFn, SoftmaxFn


class LovaszSoftmaxStable(nn.Module):
    def __init__(self, ignore_index=None, classes="present"):
        super().__init__()
        if not (classes in ("present", "all") or isinstance(classes, (list, tuple, range))):
            raise ValueError(f"classes must be 'present', 'all' or a list of class ids, got {classes!r}")
        self.ignore_index = ignore_index
        self.classes = classes

    def forward(self, outputs, labels, model_act=None):
        if model_act == "logits":
            probs = SoftmaxFn.apply(outputs)
        elif model_act == "probs":
            probs = outputs
        elif model_act == "log_probs":
            probs = outputs.exp()
        else:
            raise ValueError(f"Unknown model_act: {model_act}")
        if probs.dim() != 4:
            raise ValueError("probas dim must be 4 ([B,C,H,W]) on the HIP path")
        return LovaszFn.apply(probs, labels, self.ignore_index, self.classes)


# drop-in mode (this file shadows the reference's module of the same import path): names it does not define come from there
from semanticlidarunc_amd._shadow import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(__name__, __file__, globals())
