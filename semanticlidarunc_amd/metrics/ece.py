"""Mirror of the reference's ``ECEAggregator`` (src/metrics/ece.py:13-212), top-label ECE / MCE.

Same constructor, ``update(preds, labels)``, ``compute(save_plot_path, title, dpi)`` and ``reset``; nothing is copied to the host
per batch.  Two accumulation forms, chosen by the constructor arguments exactly as the reference's semantics require:

* ``max_samples=None`` and ``binning='uniform'`` (the reference keeps EVERY valid pixel, ece.py:88-91): the per-pixel work
  (re-normalise, max over classes, correctness, histogram over the reference's float32 ``linspace`` edges) is one HIP kernel that
  adds into three per-bin device accumulators -- identical bins without storing the samples.
* ``max_samples=N`` (the Trainer's 500 000, trainer.py:215-222) or ``binning='adaptive'``: the (confidence, correct) samples stay in
  device buffers under the reference's reservoir policy (ece.py:93-111) -- the numpy ``default_rng(seed)`` draws are made on the host
  in the reference's order (``_reservoir.CappedColumns``) and applied to the device columns -- and are binned at ``compute()`` over
  uniform or equal-mass edges (ece.py:115-128; the quantiles are taken by numpy on a host copy of the <= N confidences).
"""
from __future__ import annotations

import numpy as np
import pandas as pd
import torch

from semanticlidarunc_amd import ops
from semanticlidarunc_amd._reservoir import CappedColumns


class ECEAggregator:
    def __init__(self, n_bins=15, mode="alpha", ignore_index=None, max_samples=None, seed=0, eps=1e-12,
                 binning: str = "uniform", plot_style: str = "classic"):
        assert binning in {"uniform", "adaptive"}
        assert plot_style in {"classic", "classic+hist", "gap"}
        assert mode in {"alpha", "logits", "probs"}
        assert n_bins >= 2
        self.n_bins, self.mode, self.ignore_index = int(n_bins), mode, ignore_index
        self.max_samples, self.eps = max_samples, float(eps)
        self.binning, self.plot_style = binning, plot_style
        self._keeps_samples = max_samples is not None or binning != "uniform"
        self._buf = CappedColumns(max_samples, seed)       # columns: confidence fp32, correct uint8 (device); sample form only
        self._merge_seed = int(seed)
        self.reset()

    # the reference's attribute names, for code that inspects the aggregator
    @property
    def rng(self):
        return self._buf.rng

    @property
    def _conf(self):
        return torch.empty(0, dtype=torch.float32) if self._buf.columns is None else self._buf.columns[0]

    @property
    def _correct(self):
        return torch.empty(0, dtype=torch.bool) if self._buf.columns is None else self._buf.columns[1].bool()

    def reset(self):
        self._count = self._sum_correct = self._sum_conf = None
        self._seen = 0
        self._buf.clear()

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
        lab = labels.to(p.device).to(torch.int64).contiguous()
        if self._keeps_samples:
            conf, flag = ops.ece_samples(p, lab, self.mode, self.ignore_index, self.eps)
            keep = flag != 2                                  # boolean-mask order == the reference's NCHW scan order
            self._buf.push(conf[keep], flag[keep])
            self._seen = self._buf.seen
            return
        if self.mode == "logits":
            p, _, _ = ops.softmax_entropy(p)
        # 'alpha' and 'probs' are both "non-negative scores normalised by their sum"
        self._ensure(p.device)
        ops.ece_update(p, lab, self._count, self._sum_correct, self._sum_conf, self.ignore_index)

    def _bin_edges(self) -> np.ndarray:
        """ece.py:115-128: float32 linspace, or equal-mass edges from the empirical quantiles of the stored confidences."""
        uniform = np.linspace(0.0, 1.0, self.n_bins + 1, dtype=np.float32)
        edges = uniform
        if self.binning == "adaptive" and len(self._buf) > 0:
            q = np.linspace(0.0, 1.0, self.n_bins + 1, dtype=np.float32)
            edges = np.quantile(self._buf.columns[0].cpu().numpy(), q)
            edges[0], edges[-1] = 0.0, 1.0
            edges = np.unique(edges)
            if edges.size < self.n_bins + 1:
                edges = uniform
        edges[0], edges[-1] = 0.0, 1.0
        return edges

    def _bins(self):
        """(edges, n int, sum_correct f64, sum_conf f64) or None when nothing was seen."""
        if self._keeps_samples:
            if len(self._buf) == 0:
                return None
            edges = self._bin_edges()
            conf, ok = self._buf.columns
            dev_edges = torch.from_numpy(np.ascontiguousarray(edges, dtype=np.float32)).to(conf.device)
            n, n_ok, s_conf = ops.binned_stats(conf.contiguous(), ok.contiguous(), dev_edges)
            return edges, n.cpu().numpy().astype(int), n_ok.cpu().numpy().astype(np.float64), s_conf.cpu().numpy()
        if self._count is None:
            return None
        edges = np.linspace(0.0, 1.0, self.n_bins + 1, dtype=np.float32)
        edges[0], edges[-1] = 0.0, 1.0
        return edges, self._count.cpu().numpy().astype(int), self._sum_correct.cpu().numpy(), self._sum_conf.cpu().numpy()

    def _stats_df(self) -> pd.DataFrame:
        cols = ["low", "high", "center", "width", "n", "pct", "acc", "conf"]
        bins = self._bins()
        if bins is None or bins[1].sum() == 0:
            return pd.DataFrame(columns=cols)
        edges, n, acc_s, conf_s = bins
        k = n.size
        acc = np.divide(acc_s, n, out=np.full(k, np.nan), where=n > 0)
        conf = np.divide(conf_s, n, out=np.full(k, np.nan), where=n > 0)
        lows, highs = edges[:-1], edges[1:]
        total = len(self._buf) if self._keeps_samples else int(n.sum())
        return pd.DataFrame({"low": lows, "high": highs, "center": 0.5 * (lows + highs), "width": highs - lows,
                             "n": n, "pct": 100.0 * n / max(1, total), "acc": acc, "conf": conf})

    @torch.no_grad()
    def merge_across_ranks(self, group=None) -> None:
        """Data-parallel evaluation: make every rank hold the union of all ranks' evidence.  Bin form: three
        all-reduces.  Sample form: the columns are all-gathered in rank order; if their union exceeds ``max_samples`` a uniformly
        drawn subset is kept (the reference has no multi-process form to follow here)."""
        import torch.distributed as dist
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            return
        world = dist.get_world_size(group)
        if not self._keeps_samples:
            if self._count is None:
                raise RuntimeError("merge_across_ranks: call on a device (see _ensure) -- a rank that saw no batch must still join the collective")
            for t in (self._count, self._sum_correct, self._sum_conf):
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return
        dev = self._merge_device
        mine = len(self._buf)
        rank = dist.get_rank(group)
        meta = torch.zeros(2, world, dtype=torch.int64, device=dev)      # row 0: reservoir sizes, row 1: pixels seen, per rank
        meta[0, rank], meta[1, rank] = mine, self._buf.seen
        dist.all_reduce(meta, op=dist.ReduceOp.SUM, group=group)
        sizes, seens = meta[0].tolist(), meta[1].tolist()
        cap = max(sizes)
        if cap == 0:
            return
        parts = []
        for ci, dt in ((0, torch.float32), (1, torch.uint8)):
            pad = torch.zeros(cap, dtype=dt, device=dev)
            if mine:
                pad[:mine] = self._buf.columns[ci]
            got = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(got, pad, group=group)
            parts.append([g[:k] for g, k in zip(got, sizes)])
        total = sum(sizes)
        if self.max_samples is not None and total > self.max_samples:
            # Every rank must keep the SAME subset, and a rank's reservoir stands for the `seen` pixels it was drawn from: quotas
            # proportional to seen (largest remainder, never more than the rank holds), indices from a generator seeded by state all
            # ranks share -- not from the per-rank generators, which rank-specific reservoir draws have advanced differently.
            quota = _proportional_quota(self.max_samples, seens, sizes)
            rng = np.random.default_rng([self._merge_seed, sum(seens)] + list(sizes))
            picks = [torch.from_numpy(np.sort(rng.choice(k, size=q, replace=False))).to(dev) for k, q in zip(sizes, quota)]
            cols = [torch.cat([p[i] for p, i in zip(col, picks)]) for col in parts]
        else:
            cols = [torch.cat(col) for col in parts]
        self._buf.columns, self._buf.seen = cols, int(sum(seens))
        self._seen = self._buf.seen

    _merge_seed = 0

    @property
    def _merge_device(self):
        if self._buf.columns is not None:
            return self._buf.columns[0].device
        return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")

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


def _proportional_quota(total: int, weights, limits):
    """Split `total` over ranks in proportion to `weights`, never above `limits`; largest remainders first, spill-over to ranks with room."""
    n = len(weights)
    quota = [0] * n
    left = min(total, sum(limits))
    active = [i for i in range(n) if limits[i] > 0 and weights[i] > 0] or [i for i in range(n) if limits[i] > 0]
    while left > 0 and active:
        wsum = float(sum(max(weights[i], 1) for i in active))
        share = [left * max(weights[i], 1) / wsum for i in active]
        give = [min(int(sh), limits[i] - quota[i]) for sh, i in zip(share, active)]
        if sum(give) == 0:      # less than one sample each: hand out single samples by largest share
            order = sorted(range(len(active)), key=lambda k: (-share[k], active[k]))
            for k in order[:left]:
                give[k] = min(1, limits[active[k]] - quota[active[k]])
        for g, i in zip(give, active):
            quota[i] += g
        left -= sum(give)
        active = [i for i in active if quota[i] < limits[i]]
        if sum(give) == 0:
            break
    return quota


# drop-in mode (this file shadows the reference's module of the same import path): names it does not define come from there
from semanticlidarunc_amd._shadow import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(__name__, __file__, globals())
