"""Mirrors of the reference's ``IoUEvaluator`` (src/models/evaluator.py:29-105), ``UncertaintyPerClassAggregator`` (:191-281) and
``UncertaintyAccuracyAggregator`` (:640-869).

``update`` accumulates the [C,C] int64 confusion matrix (rows = ground truth) with an LDS-histogram
HIP kernel on the device the predictions live on -- the reference first copies both int64 maps to
the CPU and calls ``bincount`` there.  ``compute`` is 20x20 host arithmetic in float64 and returns
the same ``(mIoU, {class_name: IoU, ..., 'mIoU': mIoU})``.
"""
from __future__ import annotations

import numpy as np
import torch

from semanticlidarunc_amd import ops
from semanticlidarunc_amd._reservoir import CappedColumns


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



class UncertaintyAccuracyAggregator:
    """Accuracy per uncertainty bin (reference evaluator.py:640-869), with the (uncertainty, correct) samples kept on the
    DEVICE: ``update`` runs one HIP kernel (clamp, comparison, ignore mask) instead of three device-to-host copies per batch,
    ``binned_accuracy`` is a device histogram with ``np.histogram`` semantics, and the numpy-seeded reservoir cap draws the same
    indices as the reference and applies them to the device buffers."""

    def __init__(self, max_samples=None, seed: int = 0):
        self.max_samples = max_samples
        self._buf = CappedColumns(max_samples, seed)       # columns: uncertainty fp32 in [0, 1], correct uint8 (device)

    @property
    def rng(self):
        return self._buf.rng

    @property
    def _uncert(self):
        return None if self._buf.columns is None else self._buf.columns[0]

    @property
    def _correct(self):
        return None if self._buf.columns is None else self._buf.columns[1]

    @property
    def _seen(self):
        return self._buf.seen

    def reset(self):
        self._buf.clear()

    def _count(self) -> int:
        return len(self._buf)

    @torch.no_grad()
    def update(self, labels: torch.Tensor, preds: torch.Tensor, uncertainty: torch.Tensor, ignore_ids=()):
        if not (labels.shape == preds.shape == uncertainty.shape):
            raise AssertionError("shapes must match")
        ids = tuple(ignore_ids)
        u, flag = ops.ua_samples(labels.detach().to(torch.int64).contiguous(), preds.detach().to(torch.int64).contiguous(),
                                 uncertainty.detach().to(torch.float32).contiguous(), ids)
        if ids:
            keep = flag != 2
            u, flag = u[keep], flag[keep]
        self._buf.push(u, flag)

    def make_bins(self, num_bins=None, bin_width=None, bin_edges=None) -> np.ndarray:
        """Strictly increasing float32 edges covering [0, 1]; priority bin_edges > bin_width > num_bins (evaluator.py:708-724)."""
        if bin_edges is not None:
            edges = np.asarray(bin_edges, dtype=np.float32).copy()
        elif bin_width is not None:
            edges = np.linspace(0.0, 1.0, max(1, int(round(1.0 / float(bin_width)))) + 1, dtype=np.float32)
        else:
            edges = np.linspace(0.0, 1.0, (int(num_bins) if num_bins is not None else 10) + 1, dtype=np.float32)
        edges[0] = 0.0
        edges[-1] = 1.0
        assert np.all(np.diff(edges) > 0), "bin edges must be strictly increasing"
        return edges

    def binned_accuracy(self, num_bins: int = 10, bin_width=None, bin_edges=None):
        """DataFrame [low, high, label, n, pct, accuracy]; empty bins: n = 0, accuracy = NaN, pct = 0 (evaluator.py:726-749)."""
        import pandas as pd
        if self._count() == 0:
            return pd.DataFrame(columns=["low", "high", "label", "n", "pct", "accuracy"])
        edges = self.make_bins(num_bins=num_bins, bin_width=bin_width, bin_edges=bin_edges)
        cnt, ok = ops.binned_counts(self._uncert.contiguous(), self._correct.contiguous(), torch.from_numpy(edges).to(self._uncert.device))
        n = cnt.cpu().numpy().astype(int)
        csum = ok.cpu().numpy().astype(np.float64)
        acc = np.divide(csum, n, out=np.full_like(csum, np.nan, dtype=float), where=n > 0)
        pct = 100.0 * n / max(1, self._count())
        lows, highs = edges[:-1], edges[1:]
        labels = [f"[{lo:.2f}, {hi:.2f})" if i < len(lows) - 1 else f"[{lo:.2f}, {hi:.2f}]" for i, (lo, hi) in enumerate(zip(lows, highs))]
        return pd.DataFrame({"low": lows, "high": highs, "label": labels, "n": n, "pct": pct, "accuracy": acc})

    def plot_accuracy_vs_uncertainty_bins(self, num_bins: int = 10, bin_width=None, bin_edges=None, figsize=(14, 5),
                                          title="Pixel Accuracy vs Predictive-Uncertainty (binned)", x_label="Normalized predictive-entropy bin",
                                          y_label="Accuracy", show_percent_on_bars: bool = True, annotate_min_pct: float = 0.1,
                                          annotate_every: int = 1, percent_fmt: str = "{:.1f}%", save_path=None, show: bool = False,
                                          close_fig: bool = True, dpi: int = 200, cmap_name: str = "viridis", color_norm: str = "linear"):
        """Bar chart of the binned accuracy coloured by the share of points per bin; returns (fig, ax) like the reference (a plain
        rendering: bars, overall-accuracy line, percentage labels, colour bar)."""
        stats = self.binned_accuracy(num_bins=num_bins, bin_width=bin_width, bin_edges=bin_edges)
        if stats.empty or stats["n"].sum() == 0:
            print("No data to plot.")
            return
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        from matplotlib.cm import ScalarMappable
        from matplotlib.colors import LogNorm, Normalize
        pct = stats["pct"].to_numpy()
        if color_norm == "log":
            vmin = max(1e-3, float(pct[pct > 0].min()) if (pct > 0).any() else 1e-3)
            norm = LogNorm(vmin=vmin, vmax=max(vmin * 10, float(pct.max() or 1.0)))
        else:
            norm = Normalize(vmin=0.0, vmax=100.0 if color_norm == "linear" else max(1.0, float(pct.max())))
        cmap = matplotlib.colormaps[cmap_name]
        colors = cmap(norm(pct))
        colors[stats["n"].to_numpy() == 0, 3] = 0.25
        fig, ax = plt.subplots(figsize=figsize, dpi=dpi)
        bars = ax.bar(stats["label"].to_list(), np.nan_to_num(stats["accuracy"].to_numpy(), nan=0.0), color=colors, edgecolor="black", linewidth=0.8)
        ax.set_ylim(0.0, 1.0)
        ax.set_title(title, fontsize=18, weight="bold", pad=10)
        ax.set_xlabel(x_label, fontsize=12)
        ax.set_ylabel(y_label, fontsize=12)
        ax.tick_params(axis="x", rotation=45)
        overall = float(self._correct.float().mean()) if self._count() else float("nan")
        if np.isfinite(overall):
            ax.axhline(overall, ls="--", lw=2, color="black", alpha=0.85)
            ax.text(len(stats) - 0.5, overall + 0.012, f"overall = {overall:.3f}", ha="right", va="bottom", fontsize=12, fontweight="bold")
        if show_percent_on_bars:
            for i, (bar, pct_i, n_i) in enumerate(zip(bars, stats["pct"].to_list(), stats["n"].to_list())):
                if n_i == 0:
                    continue
                small = pct_i < float(annotate_min_pct) or i % max(1, int(annotate_every)) != 0
                ax.text(bar.get_x() + bar.get_width() / 2.0, 0.015, f"<{float(annotate_min_pct):.1f}%" if small else percent_fmt.format(pct_i),
                        ha="center", va="bottom", fontsize=9)
        sm = ScalarMappable(norm=norm, cmap=cmap)
        sm.set_array([])
        fig.colorbar(sm, ax=ax, pad=0.01).set_label("Percentage of points (%)", rotation=90)
        fig.tight_layout()
        if save_path is not None:
            fig.savefig(save_path, dpi=dpi, bbox_inches="tight")
        return fig, ax


class UncertaintyPerClassAggregator:
    """Per-class uncertainty samples across batches (reference evaluator.py:191-281) kept on the DEVICE: ``update`` is one stable
    group-by-class pass (``csrc/metrics.hip``) instead of two device-to-host copies and C boolean masks on the host.  Without a cap
    nothing synchronises until the lists are read; with ``max_per_class`` the numpy-seeded (approximate) reservoir makes the same
    draws, in the same order, as the reference (C sample counts cross to the host per batch).  ``_values`` / ``_seen_counts`` keep
    the reference's meaning (``tester.py:356-357,648-649`` caches them); the plotting helpers are the reference's, run on host copies."""

    def __init__(self, num_classes: int, max_per_class=None, seed: int = 0):
        self.num_classes = int(num_classes)
        self.max_per_class = max_per_class
        self.rng = np.random.default_rng(seed)
        self.reset()

    def reset(self):
        self._lists = [torch.empty(0, dtype=torch.float32) for _ in range(self.num_classes)]
        self._seen = [0 for _ in range(self.num_classes)]
        self._pending = []                                   # (grouped values, counts) of batches not yet split per class

    def _absorb(self, grouped: torch.Tensor, cnt) -> None:
        off = 0
        for c in range(self.num_classes):
            k = int(cnt[c])
            if k == 0:
                continue
            vals = grouped[off:off + k]
            off += k
            self._seen[c] += k
            cur = self._lists[c]
            if cur.device != vals.device:
                cur = cur.to(vals.device)
            cap = self.max_per_class
            if cap is None:
                self._lists[c] = torch.cat([cur, vals])
            elif cur.numel() < cap:
                take = min(cap - cur.numel(), k)
                if take < k:
                    vals = vals[torch.from_numpy(self.rng.choice(k, size=take, replace=False)).to(vals.device)]
                self._lists[c] = torch.cat([cur, vals])
            else:
                accept = self.rng.random(k) < min(1.0, float(cap) / float(self._seen[c] + 1e-9))
                if accept.any():
                    chosen = vals[torch.from_numpy(accept).to(vals.device)]
                    slots = torch.from_numpy(self.rng.choice(cap, size=int(chosen.numel()), replace=False)).to(vals.device)
                    cur[slots] = chosen
                    self._lists[c] = cur

    def _flush(self) -> None:
        if self._pending:
            counts = torch.stack([c for _, c in self._pending]).cpu().tolist()      # one copy for all deferred batches
            pending, self._pending = self._pending, []
            for (grouped, _), cnt in zip(pending, counts):
                self._absorb(grouped, cnt)

    @property
    def _values(self):
        self._flush()
        return self._lists

    @_values.setter
    def _values(self, vals):
        self._pending = []
        self._lists = list(vals)

    @property
    def _seen_counts(self):
        self._flush()
        return self._seen

    @_seen_counts.setter
    def _seen_counts(self, counts):
        self._flush()
        self._seen = list(counts)

    @torch.no_grad()
    def update(self, labels: torch.Tensor, uncertainty: torch.Tensor):
        assert labels.shape == uncertainty.shape, "labels and uncertainty must have same shape"
        if not uncertainty.is_cuda:
            raise RuntimeError("UncertaintyPerClassAggregator.update: the uncertainty map must be on the GPU (no CPU fallback)")
        if labels.numel() == 0:
            return
        lab = labels.detach().to(device=uncertainty.device, dtype=torch.int64).reshape(-1).contiguous()
        unc = uncertainty.detach().to(torch.float32).reshape(-1).contiguous()
        grouped, counts = ops.group_by_class(lab, unc, self.num_classes)
        if self.max_per_class is None:
            self._pending.append((grouped, counts))
        else:
            self._flush()
            self._absorb(grouped, counts.cpu().tolist())

    def as_dataframe(self, class_names: list, ignore_ids=()):
        """Long DataFrame [class_id, class, uncertainty] (evaluator.py:264-281)."""
        import pandas as pd
        skip = set(ignore_ids)
        rows = [pd.DataFrame({"class_id": c, "class": class_names[c], "uncertainty": v.cpu().numpy()})
                for c, v in enumerate(self._values) if c not in skip and v.numel() > 0]
        return pd.concat(rows, ignore_index=True) if rows else pd.DataFrame(columns=["class_id", "class", "uncertainty"])

    def __getattr__(self, name):
        # plot_boxplot / plot_ridgeline / plot_ridgeline_fast: the reference's matplotlib code, bound to a host copy of the samples
        if name.startswith("plot_"):
            from semanticlidarunc_amd._shadow import shadowed_module
            ref = getattr(shadowed_module(__name__, __file__), "UncertaintyPerClassAggregator", None)
            if ref is not None and hasattr(ref, name):
                host = ref.__new__(ref)
                host.num_classes, host.max_per_class, host.rng = self.num_classes, self.max_per_class, self.rng
                host._values, host._seen_counts = [v.cpu() for v in self._values], list(self._seen_counts)
                return getattr(host, name)
        raise AttributeError(f"{type(self).__name__!s} has no attribute {name!r}")


# drop-in mode (this file shadows the reference's module of the same import path): names it does not define come from there
from semanticlidarunc_amd._shadow import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(__name__, __file__, globals())
