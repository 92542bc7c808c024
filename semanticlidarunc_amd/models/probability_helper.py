"""Drop-in for the Dirichlet-head part of the reference's ``models/probability_helper.py`` (:12-36, :89-136, :148-153):
same function names, arguments and module-level eps / temperature switches; the arithmetic runs in one fused HIP kernel
per call (``csrc/dirichlet.hip``).  ``dirichlet_head`` is the one-launch form of ``trainer.py:533-538`` + ``:1193-1194``.
Plotting helpers, Dirichlet losses and the remaining uncertainty decompositions of the reference module are not mirrored."""
from __future__ import annotations

import math
from typing import Optional

import torch

from semanticlidarunc_amd import _lib
from semanticlidarunc_amd._lib import check
from semanticlidarunc_amd.ops import _ptr, _req, _stream

_EPS: float = 1e-8
_T: float = 1.0


def set_eps_value(eps: float):
    global _EPS
    _EPS = eps


def get_eps_value() -> float:
    return _EPS


def set_alpha_temperature(T: float):
    global _T
    _T = T


def get_alpha_temperature() -> float:
    return _T


def _inner_contiguous(t: torch.Tensor, name: str):
    """[B, C, H, W] fp32 on the GPU whose (C, H, W) block is dense; the batch stride may be larger (a channel slice)."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor (no CPU fallback)")
    if t.dtype != torch.float32 or t.dim() != 4:
        raise RuntimeError(f"{name}: expected fp32 [B, C, H, W], got {t.dtype} {tuple(t.shape)}")
    b, c, h, w = t.shape
    if t.stride(3) != 1 or t.stride(2) != w or t.stride(1) != h * w or (b > 1 and t.stride(0) < c * h * w):
        t = t.contiguous()
    return t, (t.stride(0) if b > 1 else c * h * w)


def _launch_head(shape_logits, scale_logits, T, eps, want_alpha, want_p, want_h, want_au, want_preds):
    shape_logits, sbs = _inner_contiguous(shape_logits, "shape_logits")
    scale_logits, cbs = _inner_contiguous(scale_logits, "scale_logits")
    b, c, h, w = shape_logits.shape
    if tuple(scale_logits.shape) != (b, 1, h, w):
        raise RuntimeError(f"scale_logits: expected {(b, 1, h, w)}, got {tuple(scale_logits.shape)}")
    if not T > 0:
        raise ValueError("temperature must be positive")
    dev = shape_logits.device
    alpha = torch.empty((b, c, h, w), dtype=torch.float32, device=dev) if want_alpha else None
    p_hat = torch.empty((b, c, h, w), dtype=torch.float32, device=dev) if want_p else None
    ent = torch.empty((b, h, w), dtype=torch.float32, device=dev) if want_h else None
    au = torch.empty((b, h, w), dtype=torch.float32, device=dev) if want_au else None
    preds = torch.empty((b, h, w), dtype=torch.int64, device=dev) if want_preds else None
    check(_lib.load().slu_dirichlet_head(shape_logits.data_ptr(), sbs, scale_logits.data_ptr(), cbs, b, c, h * w, float(T), float(eps),
                                         _ptr(alpha), _ptr(p_hat), _ptr(ent), _ptr(au), _ptr(preds), _stream()), "slu_dirichlet_head")
    return alpha, p_hat, ent, au, preds


def to_alpha_concentrations_from_shape_and_scale(shape_logits, scale_logits, T: Optional[float] = None, eps: Optional[float] = None):
    """alpha = 1 + softplus(scale / T) * softmax(shape) + eps  (reference :89-105)."""
    return _launch_head(shape_logits, scale_logits, get_alpha_temperature() if T is None else T, get_eps_value() if eps is None else eps,
                        True, False, False, False, False)[0]


def _uncertainty(alpha, eps, want_h, want_au):
    _req(alpha, "alpha")
    if alpha.dim() != 4:
        raise RuntimeError(f"alpha: expected [B, C, H, W], got {tuple(alpha.shape)}")
    b, c, h, w = alpha.shape
    ent = torch.empty((b, h, w), dtype=torch.float32, device=alpha.device) if want_h else None
    au = torch.empty((b, h, w), dtype=torch.float32, device=alpha.device) if want_au else None
    check(_lib.load().slu_dirichlet_uncertainty(alpha.data_ptr(), b, c, h * w, float(get_eps_value() if eps is None else eps), None, _ptr(ent),
                                                _ptr(au), None, _stream()), "slu_dirichlet_uncertainty")
    return ent, au


def get_predictive_entropy(alpha, eps: Optional[float] = None):
    return _uncertainty(alpha, eps, True, False)[0]


def get_aleatoric_uncertainty(alpha, eps: Optional[float] = None):
    return _uncertainty(alpha, eps, False, True)[1]


def get_epistemic_uncertainty(alpha, eps: Optional[float] = None):
    h, au = _uncertainty(alpha, eps, True, True)
    return h - au


class _MeanAggregated:
    """The call surface ``utils.agg.mean_aggregator`` gives the reference's function (add / accumulate / mean / reset)."""

    def __init__(self, fn):
        self._fn, self._sum, self._count = fn, 0.0, 0
        self.__name__, self.__doc__ = fn.__name__, fn.__doc__

    def __call__(self, *args, **kwargs):
        return self._fn(*args, **kwargs)

    def add(self, x, mask=None):
        if torch.is_tensor(x):
            x = x.detach()
            if mask is not None:
                m = mask if mask.shape == x.shape else torch.broadcast_to(mask, x.shape)
                self._sum += float(x[m].float().sum().item())
                self._count += int(m.sum().item())
            else:
                self._sum += float(x.float().sum().item())
                self._count += x.numel()
        else:
            self._sum += float(x)
            self._count += 1

    def accumulate(self, *args, mask=None, **kwargs):
        out = self._fn(*args, **kwargs)
        self.add(out, mask=mask)
        return out

    def mean(self, reset: bool = False) -> float:
        m = self._sum / max(1, self._count)
        if reset:
            self.reset()
        return m

    def reset(self):
        self._sum, self._count = 0.0, 0


def _entropy_norm(alpha, eps: Optional[float] = None):
    """Predictive entropy divided by ln C (reference :148-153)."""
    return get_predictive_entropy(alpha, eps) / math.log(alpha.shape[1])


_entropy_norm.__name__ = "get_predictive_entropy_norm"
get_predictive_entropy_norm = _MeanAggregated(_entropy_norm)


def dirichlet_head(outputs: torch.Tensor, num_classes: int, T: Optional[float] = None, eps: Optional[float] = None):
    """One launch for trainer.py:533-538 + :1193-1194: outputs [B, C+1, H, W] -> (alpha, p_hat, H_norm, preds)."""
    if outputs.dim() != 4 or outputs.shape[1] < num_classes + 1:
        raise RuntimeError(f"outputs: expected [B, >= {num_classes + 1}, H, W], got {tuple(outputs.shape)}")
    alpha, p_hat, ent, _, preds = _launch_head(outputs[:, :num_classes], outputs[:, num_classes:num_classes + 1],
                                               get_alpha_temperature() if T is None else T, get_eps_value() if eps is None else eps,
                                               True, True, True, False, True)
    return alpha, p_hat, ent / math.log(num_classes), preds


# drop-in mode (this file shadows the reference's module of the same import path): names it does not define come from there
from semanticlidarunc_amd._shadow import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(__name__, __file__, globals())
