"""Mirror of the reference's ``utils.mc_dropout`` (src/utils/mc_dropout.py:13-133).

Same names and argument meaning.  ``mc_forward`` / ``mc_dropout_probs`` keep BatchNorm frozen
(``model.eval()``) and flip only the dropout layers to train mode, exactly like the reference; the
difference is scheduling: instead of T sequential forwards, the T passes are stacked along the
batch axis (``Dropout2d`` draws an independent mask per (sample, channel) and eval-BatchNorm is
per-sample, so the T*B samples are exactly T independent passes) and run as ONE sequence of
launches, which keeps the small 4x128 ... 16x512 feature maps from under-filling 256 CUs.
"""
from __future__ import annotations

import contextlib
import os

import torch

import torch.nn as nn

from semanticlidarunc_amd import ops

EPS = 1e-12
_DROPOUT_TYPES = (nn.Dropout, nn.Dropout2d, nn.Dropout3d, nn.AlphaDropout, nn.FeatureAlphaDropout)
MAX_STACK = int(os.environ.get("SLU_MC_MAX_STACK", "16"))   # at most this many passes of one batch are stacked per launch sequence


def set_dropout_mode(module: nn.Module, train: bool) -> None:
    """Toggle only dropout layers; BatchNorm etc. keep their mode."""
    for m in module.modules():
        if isinstance(m, _DROPOUT_TYPES):
            m.train(train)


@contextlib.contextmanager
def dropout_sampling(module: nn.Module, enable: bool = True):
    if enable:
        set_dropout_mode(module, True)
    try:
        yield
    finally:
        if enable:
            set_dropout_mode(module, False)


def _stacked_passes(model, inputs, T: int, share_prefix: bool = False):
    """Yield logits [t,B,C,H,W] for groups of passes that together cover T passes."""
    inputs = list(inputs)
    b = inputs[0].shape[0]
    done = 0
    shared = share_prefix and len(inputs) == 1 and hasattr(model, "forward_mc")
    while done < T:
        t = min(MAX_STACK, T - done)
        if shared:
            out = model.forward_mc(inputs[0], t)
            yield out.reshape(t, b, *out.shape[1:])
            done += t
            continue
        stacked = [x.repeat(t, *([1] * (x.dim() - 1))) for x in inputs]
        out = model(*stacked)
        if isinstance(out, tuple):
            if len(out) > 2:
                raise AssertionError("Model returned/generated unexpectedly too many outputs")
            out = out[0]
        yield out.reshape(t, b, *out.shape[1:])
        done += t


@torch.no_grad()
def mc_forward(model: nn.Module, inputs, T: int = 30, share_prefix: bool = False):
    """[T,B,C,H,W] raw model outputs of T stochastic passes (reference mc_dropout.py:98-119).
    share_prefix=True lets a model that implements `forward_mc` compute the layers no active dropout can reach
    once instead of T times (identical results; see SalsaNext.forward_mc)."""
    model.eval()
    with dropout_sampling(model, enable=True):
        parts = list(_stacked_passes(model, inputs, T, share_prefix))
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)


@torch.no_grad()
def mc_dropout_probs(model: nn.Module, inputs, T: int = 30, temperature: float | None = None):
    """[T,B,C,H,W] probabilities (reference mc_dropout.py:55-96); output kind detected once."""
    from semanticlidarunc_amd.models.losses import classify_output_kind
    logits = mc_forward(model, inputs, T)
    kind = classify_output_kind(logits[0], class_dim=1)
    if kind == "probs":
        logits = logits.clamp_min(EPS).log()
    elif kind not in ("logits", "log_probs"):
        raise ValueError(f"Unknown output kind: {kind}")
    if temperature is not None:
        logits = logits / max(1e-3, float(temperature))
    t, b = logits.shape[:2]
    probs, _, _ = ops.softmax_entropy(logits.reshape(t * b, *logits.shape[2:]).contiguous())
    return probs.reshape(logits.shape)


@torch.no_grad()
def predictive_entropy_mc(mc_probs: torch.Tensor, eps: float = 1e-12, normalize: bool = True):
    """[T,B,C,H,W] probabilities -> entropy of the mean [B,H,W] (reference mc_dropout.py:121-133)."""
    mean_p = mc_probs.mean(dim=0).clamp_min(eps)
    ent = -(mean_p * torch.log(mean_p)).sum(dim=1)
    if not normalize:
        return ent
    return ent / float(torch.log(torch.tensor(float(mean_p.shape[1]))).item())


@torch.no_grad()
def mc_predict(model: nn.Module, inputs, T: int = 30, eps: float = 1e-12, share_prefix: bool = False):
    """The whole MC evaluation step of trainer.py:1138-1154 in one call:
    (p_bar[B,C,H,W], H_norm[B,H,W], MI_norm[B,H,W], preds[B,H,W])."""
    xs = list(inputs) if isinstance(inputs, (list, tuple)) else [inputs]
    if len(xs) == 1 and T <= MAX_STACK and hasattr(model, "mc_predict_fused"):
        model.eval()
        if model.mc_fused_ok(xs[0], T):                 # half-precision SalsaNext: head conv + reduction in one launch, no logit maps
            with dropout_sampling(model, enable=True):
                return model.mc_predict_fused(xs[0], T, eps, share_prefix)
    return ops.mc_reduce(mc_forward(model, inputs, T, share_prefix).contiguous(), eps)


# drop-in mode (this file shadows the reference's module of the same import path): names it does not define come from there
from semanticlidarunc_amd._shadow import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(__name__, __file__, globals())
