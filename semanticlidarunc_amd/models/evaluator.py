"""Mirror of the reference's ``IoUEvaluator`` (src/models/evaluator.py:29-105).

``update`` accumulates the [C,C] int64 confusion matrix (rows = ground truth) with an LDS-histogram
HIP kernel on the device the predictions live on -- the reference first copies both int64 maps to
the CPU and calls ``bincount`` there.  ``compute`` is 20x20 host arithmetic in float64 and returns
the same ``(mIoU, {class_name: IoU, ..., 'mIoU': mIoU})``.
"""
from __future__ import annotations

import numpy as np
import torch

from semanticlidarunc_amd import ops


class IoUEvaluator:
    def __init__(self, num_classes: int, device="cpu"):
        # `device` is accepted for signature parity; counts are kept where the predictions are.
        self.C = int(num_classes)
        self.device = device
        self.reset()

    def reset(self):
        self.confmat = None

    def _ensure(self, device):
        if self.confmat is None or self.confmat.device != device:
            old = self.confmat
            self.confmat = torch.zeros((self.C, self.C), dtype=torch.int64, device=device)
            if old is not None:
                self.confmat += old.to(device)

    @torch.no_grad()
    def update(self, preds: torch.Tensor, targets: torch.Tensor):
        """preds / targets: integer maps of equal size; out-of-range pairs are dropped."""
        if not preds.is_cuda:
            raise RuntimeError("IoUEvaluator.update: predictions must be on the GPU (no CPU fallback)")
        self._ensure(preds.device)
        p = preds.reshape(-1).to(torch.int64).contiguous()
        t = targets.to(preds.device).reshape(-1).to(torch.int64).contiguous()
        ops.confusion_update(self.confmat, p, t)

    def compute(self, class_names, test_mask=None, ignore_gt=None, reduce="mean", ignore_th=None):
        c = self.C
        cm = np.zeros((c, c)) if self.confmat is None else self.confmat.cpu().numpy().astype(np.float64)
        for r in ignore_gt or ():
            if 0 <= int(r) < c:
                cm[int(r), :] = 0.0
        tp = np.diag(cm)
        union = cm.sum(axis=0) + cm.sum(axis=1) - tp
        iou = np.full(c, np.nan)
        np.divide(tp, union, out=iou, where=union > 0)
        if test_mask is None:
            mask = np.ones(c, dtype=bool)
        else:
            mask = np.asarray(torch.as_tensor(test_mask, dtype=torch.bool).cpu())
            if mask.size != c:
                raise ValueError("test_mask length != num_classes")
        sel = mask & np.isfinite(iou)
        if ignore_th is not None:
            sel &= np.nan_to_num(iou, nan=-np.inf) >= ignore_th
        out = {}
        for k in range(c):
            name = class_names[k] if isinstance(class_names, (list, dict)) else class_names[str(k)]
            out[name] = float(iou[k])
        if sel.any():
            miou = float(np.mean(iou[sel])) if reduce == "mean" else float(np.median(iou[sel]))
        else:
            miou = float("nan")
        out["mIoU"] = miou
        return miou, out
