"""Drop-in for ``metrics.auroc.AUROCAggregator`` of the reference (``src/metrics/auroc.py``): same constructor, ``update`` /
``compute`` / ``reset`` and the same sample selection -- including the numpy-seeded reservoir cap, whose index draws are made
on the host exactly as the reference makes them and applied to the DEVICE-resident sample buffers -- but probabilities,
scores, error flags, the sort and the ROC integral run in HIP kernels (``csrc/lovasz.hip``) and nothing is copied to the host
per batch.  ``compute`` returns ``(auroc, curves, fig)`` with ``fig`` None unless ``save_plot_path`` is given (the reference
raises UnboundLocalError in that case, auroc.py:147-164)."""
from __future__ import annotations

import numpy as np
import torch

from semanticlidarunc_amd import ops
from semanticlidarunc_amd._reservoir import CappedColumns


class AUROCAggregator:
    def __init__(self, mode="alpha", score="entropy_norm", ignore_index=None, max_samples=None, seed=0, eps=1e-12):
        if mode not in ops.AUROC_MODES or score not in ops.AUROC_SCORES:
            raise AssertionError(f"mode must be one of {sorted(ops.AUROC_MODES)}, score one of {sorted(ops.AUROC_SCORES)}")
        self.mode, self.score = mode, score
        self.ignore_index = ignore_index
        self.max_samples = max_samples
        self.eps = float(eps)
        self._buf = CappedColumns(max_samples, seed)       # columns: score fp32, is_error uint8 (device)

    # the reference's attribute names, for code that inspects the aggregator
    @property
    def rng(self):
        return self._buf.rng

    @property
    def _scores(self):
        return None if self._buf.columns is None else self._buf.columns[0]

    @property
    def _is_error(self):
        return None if self._buf.columns is None else self._buf.columns[1]

    @property
    def _seen(self):
        return self._buf.seen

    def reset(self):
        self._buf.clear()

    @torch.no_grad()
    def update(self, preds: torch.Tensor, labels: torch.Tensor, score_override: torch.Tensor | None = None):
        if preds.dim() != 4 or not (labels.dim() == 3 or (labels.dim() == 4 and labels.size(1) == 1)):
            raise AssertionError("labels must be [B,H,W] or [B,1,H,W]")
        lab = (labels[:, 0] if labels.dim() == 4 else labels).long().contiguous()
        override = None if score_override is None else score_override.to(torch.float32).contiguous()
        score_map, flags = ops.auroc_scores(preds.contiguous().float(), lab, self.mode, self.score, self.ignore_index, self.eps, override)
        keep = flags != 2                                    # boolean-mask order == the reference's NCHW scan order
        self._buf.push(score_map[keep], flags[keep])

    def compute(self, save_plot_path: str | None = None, title: str = "ROC: error detection", dpi: int = 200):
        if len(self._buf) == 0:
            return float("nan"), {}
        auroc, pos, neg, ss, se = ops.auroc_from_samples(self._scores.contiguous(), self._is_error.contiguous(), want_sorted=True)
        if np.isnan(auroc):
            return auroc, {}
        y = se.to(torch.float64)
        tpr = np.concatenate(([0.0], (torch.cumsum(y, 0) / pos).cpu().numpy(), [1.0]))
        fpr = np.concatenate(([0.0], (torch.cumsum(1.0 - y, 0) / neg).cpu().numpy(), [1.0]))
        thr = np.concatenate(([np.inf], ss.double().cpu().numpy(), [-np.inf]))
        fig = None
        if save_plot_path is not None:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            fig, ax = plt.subplots(figsize=(6.0, 5.0), dpi=dpi)
            ax.plot([0, 1], [0, 1])
            ax.plot(fpr, tpr)
            ax.set_xlim(0, 1); ax.set_ylim(0, 1)
            ax.set_xlabel("FPR"); ax.set_ylabel("TPR")
            ax.set_title(f"{title}\nAUROC = {auroc:.4f}")
            ax.grid(True, alpha=0.3)
            fig.tight_layout(); fig.savefig(save_plot_path, bbox_inches="tight", dpi=dpi); plt.close(fig)
        return auroc, {"fpr": fpr, "tpr": tpr, "thresholds": thr}, fig


# drop-in mode (this file shadows the reference's module of the same import path): names it does not define come from there
from semanticlidarunc_amd._shadow import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(__name__, __file__, globals())
