"""Mirror of the reference's ``ECEAggregator`` (src/metrics/ece.py:13-212), top-label ECE / MCE.

Same constructor, ``update(preds, labels)``, ``compute(save_plot_path, title, dpi)`` and ``reset``.
The per-pixel work (re-normalise, max over classes, correctness, 15-bin histogram with the
reference's float32 ``linspace`` edges) is one HIP kernel that adds into three per-bin device
accumulators, so nothing is copied to the host per batch.  Consequences, both stated in DESIGN.md:
the bins are EXACT over all pixels seen (the reference's reservoir sub-sample once more than
``max_samples`` pixels were seen is not reproduced; below the cap the results coincide), and only
``binning='uniform'`` is supported on the device path.
"""
from __future__ import annotations

import numpy as np
import pandas as pd
import torch

from semanticlidarunc_amd import ops


class ECEAggregator:
    def __init__(self, n_bins=15, mode="alpha", ignore_index=None, max_samples=None, seed=0, eps=1e-12,
                 binning: str = "uniform", plot_style: str = "classic"):
        assert binning in {"uniform", "adaptive"}
        assert plot_style in {"classic", "classic+hist", "gap"}
        assert mode in {"alpha", "logits", "probs"}
        assert n_bins >= 2
        if binning != "uniform":
            raise NotImplementedError("adaptive (equal-mass) binning needs the raw confidences; only 'uniform' runs on the device")
        self.n_bins, self.mode, self.ignore_index = int(n_bins), mode, ignore_index
        self.max_samples, self.eps = max_samples, float(eps)
        self.binning, self.plot_style = binning, plot_style
        self.reset()

    def reset(self):
        self._count = self._sum_correct = self._sum_conf = None
        self._seen = 0

    def _ensure(self, device):
        if self._count is None:
            self._count = torch.zeros(self.n_bins, dtype=torch.int64, device=device)
            self._sum_correct = torch.zeros(self.n_bins, dtype=torch.float64, device=device)
            self._sum_conf = torch.zeros(self.n_bins, dtype=torch.float64, device=device)

    @torch.no_grad()
    def update(self, preds: torch.Tensor, labels: torch.Tensor):
        """preds [B,C,H,W] (alpha / logits / probs by mode), labels [B,H,W] integer class ids."""
        assert preds.dim() == 4 and labels.dim() == 3
        if not preds.is_cuda:
            raise RuntimeError("ECEAggregator.update: predictions must be on the GPU (no CPU fallback)")
        p = preds.detach().float().contiguous()
        if self.mode == "logits":
            p, _, _ = ops.softmax_entropy(p)
        # 'alpha' and 'probs' are both "non-negative scores normalised by their sum"
        self._ensure(p.device)
        lab = labels.to(p.device).to(torch.int64).contiguous()
        ops.ece_update(p, lab, self._count, self._sum_correct, self._sum_conf, self.ignore_index)

    def _stats_df(self) -> pd.DataFrame:
        cols = ["low", "high", "center", "width", "n", "pct", "acc", "conf"]
        if self._count is None:
            return pd.DataFrame(columns=cols)
        n = self._count.cpu().numpy().astype(int)
        if n.sum() == 0:
            return pd.DataFrame(columns=cols)
        acc_s, conf_s = self._sum_correct.cpu().numpy(), self._sum_conf.cpu().numpy()
        edges = np.linspace(0.0, 1.0, self.n_bins + 1, dtype=np.float32)
        edges[0], edges[-1] = 0.0, 1.0
        acc = np.divide(acc_s, n, out=np.full(self.n_bins, np.nan), where=n > 0)
        conf = np.divide(conf_s, n, out=np.full(self.n_bins, np.nan), where=n > 0)
        lows, highs = edges[:-1], edges[1:]
        return pd.DataFrame({"low": lows, "high": highs, "center": 0.5 * (lows + highs), "width": highs - lows,
                             "n": n, "pct": 100.0 * n / max(1, int(n.sum())), "acc": acc, "conf": conf})

    def compute(self, save_plot_path: str | None = None, title: str = "Reliability Diagram", dpi: int = 200):
        """((ece, mce), stats_df, fig) -- fig is None unless save_plot_path is given; an empty
        aggregator returns the reference's 2-tuple ((nan, nan), stats)."""
        stats = self._stats_df()
        if stats.empty or stats["n"].sum() == 0:
            return (float("nan"), float("nan")), stats
        w = stats["n"].to_numpy().astype(np.float64)
        acc = np.nan_to_num(stats["acc"].to_numpy(), nan=0.0)
        conf = np.nan_to_num(stats["conf"].to_numpy(), nan=0.0)
        gap = np.abs(acc - conf)
        ece = float(np.sum(w / max(1.0, w.sum()) * gap))
        mce = float(np.max(gap[w > 0])) if np.any(w > 0) else float("nan")
        fig = None
        if save_plot_path is not None:
            fig = self._plot(stats, acc, conf, ece, mce, title, dpi)
            fig.savefig(save_plot_path, bbox_inches="tight", dpi=dpi)
        return (ece, mce), stats, fig

    def _plot(self, stats, acc, conf, ece, mce, title, dpi):
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        fig, ax = plt.subplots(figsize=(6.8, 5.0), dpi=dpi)
        x, widths = stats["center"].to_numpy(), stats["width"].to_numpy()
        if self.plot_style == "gap":
            signed = conf - acc
            ax.axhline(0.0, color="k", linewidth=1)
            ax.bar(x, signed, width=widths * 0.9, color=np.where(signed >= 0, "tab:red", "tab:green"))
            ax.set_ylabel("conf - acc  (positive = over-confident)")
        else:
            ax.plot([0, 1], [0, 1], label="perfect calibration", linewidth=2)
            ax.plot(x, acc, marker="o", label="accuracy")
            ax.plot(x, conf, marker="x", linestyle="--", label="avg. confidence")
            ax.set_ylabel("Accuracy / Avg. Confidence")
            ax.set_ylim(0, 1)
            if self.plot_style == "classic+hist":
                ax2 = ax.twinx()
                ax2.bar(x, stats["n"].to_numpy() / max(1, int(stats["n"].sum())), width=widths * 0.9, alpha=0.25)
                ax2.set_ylim(0, 1)
                ax2.set_ylabel("Bin mass")
            ax.legend(loc="lower right", frameon=True)
        ax.set_xlim(0, 1)
        ax.set_xlabel("Confidence (bin center)")
        ax.set_title(f"{title}\nECE={ece:.4f}  |  MCE={mce:.4f}")
        ax.grid(True, alpha=0.3)
        fig.tight_layout()
        return fig


# drop-in mode (this file shadows the reference's module of the same import path): names it does not define come from there
from semanticlidarunc_amd._shadow import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(__name__, __file__, globals())
