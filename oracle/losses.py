"""Oracle: the "SalsaNext" training loss = NLL on clamped softmax + Lovasz-Softmax.
TEST INFRASTRUCTURE ONLY.

Follows ``src/models/trainer.py:508-516`` (loss branch), ``src/losses/lovasz.py:12-88`` and
``src/models/losses.py:50-73`` (CE/NLL wrapper).  Pinned by ``tools/gen_golden.py`` against the
imported ``LovaszSoftmaxStable`` / ``CrossEntropyLoss`` classes and the known answers of
SURVEY section 4 (0.55833334 / 0.45 / 0.74694097).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def jaccard_steps(fg_sorted: torch.Tensor) -> torch.Tensor:
    """First difference of the Jaccard loss along the sorted order (lovasz.py:25-37)."""
    total = fg_sorted.sum()
    inter = total - fg_sorted.cumsum(0)
    union = total + (1.0 - fg_sorted).cumsum(0)
    jac = 1.0 - inter / union
    out = jac.clone()
    out[1:] = jac[1:] - jac[:-1]
    return out


def lovasz_softmax(probs: torch.Tensor, labels: torch.Tensor, ignore_index=None, classes="present") -> torch.Tensor:
    """probs [B,C,H,W] (rows sum to 1), labels [B,H,W] -> scalar, mean over the summed classes (lovasz.py:56-88): classes='present' skips the
    classes without a valid pixel; 'all' / a list sums them regardless (an absent class then contributes its largest probability)."""
    c = probs.size(1)
    flat = probs.movedim(1, -1).reshape(-1, c)
    y = labels.reshape(-1).long()
    if ignore_index is not None:
        keep = y != ignore_index
        flat, y = flat[keep], y[keep]
    if flat.numel() == 0:
        return probs.new_tensor(0.0)
    per_class = []
    for k in (range(c) if classes in ("all", "present") else classes):
        fg = (y == k).to(flat.dtype)
        if classes == "present" and fg.sum() == 0:
            continue
        err = (fg - flat[:, k]).abs()
        err_sorted, order = torch.sort(err, descending=True)
        per_class.append(torch.dot(err_sorted, jaccard_steps(fg[order])))
    if not per_class:
        return probs.new_tensor(0.0)
    return torch.stack(per_class).mean()


def nll_on_probs(probs: torch.Tensor, labels: torch.Tensor, clamp: float = 1e-8, ignore_index: int = -100):
    """trainer.py:514: NLLLoss()(log(probs.clamp(min=1e-8)), labels) -- mean over ALL pixels
    (default ignore_index -100, i.e. class 0 is counted)."""
    return F.nll_loss(torch.log(probs.clamp(min=clamp)), labels.long(), ignore_index=ignore_index)


def salsanext_loss(logits, labels, w_nll=1.0, w_ls=1.0, lovasz_ignore=0):
    """trainer.py:511-516."""
    probs = F.softmax(logits, dim=1)
    nll = nll_on_probs(probs, labels)
    ls = lovasz_softmax(probs, labels, ignore_index=lovasz_ignore)
    return w_nll * nll + w_ls * ls, nll, ls


def cross_entropy(outputs, labels, ignore_index=255, model_act="logits"):
    """models/losses.py:55-73: out-of-range labels are remapped to ignore_index first."""
    labels = labels.long()
    c = outputs.shape[1]
    labels = torch.where((labels < 0) | (labels >= c), torch.full_like(labels, ignore_index), labels)
    if model_act == "logits":
        return F.cross_entropy(outputs, labels, ignore_index=ignore_index)
    if model_act == "probs":
        return F.nll_loss(torch.log(outputs + 1e-8), labels, ignore_index=ignore_index)
    if model_act == "log_probs":
        return F.nll_loss(outputs, labels, ignore_index=ignore_index)
    raise ValueError(f"Unknown model_act: {model_act}")



def tversky(outputs, labels, num_classes=20, model_act="logits", alpha=0.9, beta=0.1, smooth=1.0, ignore_index=255, reduction="mean"):
    """models/losses.py:74-128: per-class Tversky index over the valid pixels (0 <= y < C, y != ignore), loss = reduce(1 - index)."""
    if model_act == "logits":
        probs = F.softmax(outputs, dim=1)
    elif model_act == "probs":
        probs = outputs
    elif model_act == "log_probs":
        probs = outputs.exp()
    else:
        raise ValueError(f"Unknown model_act: {model_act}")
    labels = labels.long()
    valid = (labels >= 0) & (labels < num_classes)
    if ignore_index is not None:
        valid = valid & (labels != ignore_index)
    if not valid.any():
        return probs.new_tensor(0.0, requires_grad=True)
    one_hot = F.one_hot(torch.where(valid, labels, torch.zeros_like(labels)), num_classes=num_classes).permute(0, 3, 1, 2).float()
    vm = valid.unsqueeze(1).float()
    probs, one_hot = probs * vm, one_hot * vm
    tp = (probs * one_hot).sum((0, 2, 3))
    fp = ((1 - one_hot) * probs).sum((0, 2, 3))
    fn = (one_hot * (1 - probs)).sum((0, 2, 3))
    loss = 1 - (tp + smooth) / (tp + alpha * fp + beta * fn + smooth)
    return loss.mean() if reduction == "mean" else (loss.sum() if reduction == "sum" else loss)
