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


class AUROCAggregator:
    def __init__(self, mode="alpha", score="entropy_norm", ignore_index=None, max_samples=None, seed=0, eps=1e-12):
        assert mode in {"alpha", "logits", "probs"}
        assert score in {"entropy", "entropy_norm", "mi", "mi_norm", "1-maxprob"}
        self.mode, self.score = mode, score
        self.ignore_index = ignore_index
        self.max_samples = max_samples
        self.rng = np.random.default_rng(seed)
        self.eps = float(eps)
        self.reset()

    def reset(self):
        self._scores = None      # 1-D fp32 device tensor
        self._is_error = None    # 1-D uint8 device tensor
        self._seen = 0

    def _count(self) -> int:
        return 0 if self._scores is None else self._scores.numel()

    def _append(self, score, is_err):
        self._scores = score if self._scores is None else torch.cat([self._scores, score])
        self._is_error = is_err if self._is_error is None else torch.cat([self._is_error, is_err])

    @torch.no_grad()
    def update(self, preds: torch.Tensor, labels: torch.Tensor, score_override: torch.Tensor | None = None):
        assert preds.dim() == 4 and (labels.dim() == 3 or (labels.dim() == 4 and labels.size(1) == 1)), "labels must be [B,H,W] or [B,1,H,W]"
        if labels.dim() == 4:
            labels = labels[:, 0]
        so = None if score_override is None else score_override.to(torch.float32).contiguous()
        smap, flags = ops.auroc_scores(preds.contiguous().float(), labels.long().contiguous(), self.mode, self.score, self.ignore_index,
                                       self.eps, so)
        valid = flags != 2
        score, is_err = smap[valid], flags[valid]            # NCHW scan order, as the reference's boolean-mask indexing
        n_new = score.numel()
        if n_new == 0:
            return
        if self.max_samples is None:
            self._append(score, is_err)
            self._seen += n_new
            return
        # reservoir-style cap (auroc.py:125-141): the same numpy draws, applied to the device buffers
        self._seen += n_new
        if self._count() < self.max_samples:
            take = min(self.max_samples - self._count(), n_new)
            if take < n_new:
                idx = torch.from_numpy(self.rng.choice(n_new, size=take, replace=False)).to(score.device)
                score, is_err = score[idx], is_err[idx]
            self._append(score, is_err)
        else:
            p_keep = min(1.0, float(self.max_samples) / float(self._seen + 1e-9))
            keep = torch.from_numpy(self.rng.random(n_new) < p_keep)
            if keep.any():
                keep = keep.to(score.device)
                score, is_err = score[keep], is_err[keep]
                replace_idx = torch.from_numpy(self.rng.choice(self.max_samples, size=score.numel(), replace=False)).to(score.device)
                self._scores[replace_idx] = score
                self._is_error[replace_idx] = is_err

    def compute(self, save_plot_path: str | None = None, title: str = "ROC: error detection", dpi: int = 200):
        if self._count() == 0:
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
