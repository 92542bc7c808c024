"""Autograd glue for the loss kernels (one `torch.autograd.Function` per fused op).

Backward functions only read tensors saved in forward, so `torch.autograd.grad(..., retain_graph=True)`
can be called repeatedly on the same graph (the reference's GradNorm probes do, utils/grad_norm.py:52).
"""
from __future__ import annotations

import torch

from . import ops


def _labels(labels: torch.Tensor, device) -> torch.Tensor:
    return labels.to(device=device, dtype=torch.int64).contiguous()


class SoftmaxFn(torch.autograd.Function):
    """probs = softmax(logits, dim=1) on the HIP path."""

    @staticmethod
    def forward(ctx, logits):
        probs, _, _ = ops.softmax_entropy(logits.detach().float().contiguous())
        ctx.save_for_backward(probs)
        return probs

    @staticmethod
    def backward(ctx, g):
        (probs,) = ctx.saved_tensors
        return ops.softmax_loss_bwd(probs, None, g.float().contiguous(), 1.0, 0.0, 0.0, None)


class LovaszFn(torch.autograd.Function):
    """Lovasz-Softmax of probabilities; d loss / d probs is produced by the forward kernels."""

    @staticmethod
    def forward(ctx, probs, labels, ignore_index, classes="present"):
        need = probs.requires_grad
        loss, _, grad = ops.lovasz_fwd(probs.detach().float().contiguous(), _labels(labels, probs.device), ignore_index, need, classes)
        ctx.save_for_backward(grad if need else torch.empty(0, device=probs.device))
        ctx.need = need
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        if not ctx.need:
            return None, None, None, None
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None


class NllFn(torch.autograd.Function):
    """mean over counted pixels of the per-pixel NLL (kinds: ops.NLL_*)."""

    @staticmethod
    def forward(ctx, x, labels, kind, param, ignore_index):
        xd = x.detach().float().contiguous()
        lab = _labels(labels, x.device)
        acc, cnt = ops.nll_fwd(xd, lab, kind, param, ignore_index)
        ctx.save_for_backward(xd, lab, cnt)
        ctx.cfg = (kind, param, ignore_index)
        return (acc / cnt.to(torch.float64)).to(torch.float32).reshape(())

    @staticmethod
    def backward(ctx, g):
        xd, lab, cnt = ctx.saved_tensors
        kind, param, ignore_index = ctx.cfg
        gscale = (g.reshape(1).to(torch.float64) / cnt.to(torch.float64)).to(torch.float32)
        return ops.nll_bwd(xd, lab, kind, param, ignore_index, gscale), None, None, None, None


class SalsaNextLossFn(torch.autograd.Function):
    """w_nll * NLL(log clamp(softmax(z), 1e-8), y)  +  w_ls * Lovasz(softmax(z), y, ignore)   (trainer.py:508-516),
    softmax / NLL / Lovasz / backward fused over one read of the logits."""

    @staticmethod
    def forward(ctx, logits, labels, w_nll, w_ls, lovasz_ignore):
        z = logits.detach().float().contiguous()
        lab = _labels(labels, z.device)
        probs, nll_sum = ops.softmax_nll(z, lab, clamp=1e-8, want_probs=True)
        ls, _, grad_ls = ops.lovasz_fwd(probs, lab, lovasz_ignore, want_grad=True)
        n = lab.numel()
        nll = (nll_sum / n).to(torch.float32)
        ctx.save_for_backward(probs, lab, grad_ls)
        ctx.cfg = (float(w_nll), float(w_ls), n)
        total = w_nll * nll + w_ls * ls
        return total.reshape(()), nll.reshape(()), ls.reshape(())

    @staticmethod
    def backward(ctx, g, _g_nll, _g_ls):
        probs, lab, grad_ls = ctx.saved_tensors
        w_nll, w_ls, n = ctx.cfg
        gout = g.reshape(1).float().contiguous()
        return ops.softmax_loss_bwd(probs, lab, grad_ls, w_ls, w_nll / n, 1e-8, gout), None, None, None, None


def salsanext_loss(logits, labels, w_nll: float = 1.0, w_ls: float = 1.0, lovasz_ignore=0):
    """(loss, nll, lovasz) of the reference's "SalsaNext" loss branch, differentiable w.r.t. logits."""
    return SalsaNextLossFn.apply(logits, labels, w_nll, w_ls, lovasz_ignore)
