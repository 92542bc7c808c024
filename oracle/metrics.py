"""Oracle: confusion-matrix IoU, top-label ECE and error-detection AUROC in numpy / torch CPU.  TEST INFRASTRUCTURE ONLY.

Follows ``src/models/evaluator.py:29-105`` (IoUEvaluator), ``src/metrics/ece.py:54-168``
(ECEAggregator: ``_to_probs``, ``update`` without the reservoir cap, ``_stats_df``, ``compute``) and
``src/metrics/auroc.py:36-78,96-117`` (AUROCAggregator scores, sample selection and the ROC integral).
Integer work (confusion matrix, bin counts) is exact; pinned by ``tools/gen_golden.py`` against the
imported reference classes.
"""
from __future__ import annotations

import numpy as np

_trapz = getattr(np, "trapezoid", None) or np.trapz      # numpy >= 2 renamed it


def confusion_matrix(preds, targets, num_classes: int) -> np.ndarray:
    """int64 [C,C], rows = ground truth, cols = prediction; out-of-range pairs dropped
    (evaluator.py:45-53)."""
    p = np.asarray(preds).reshape(-1).astype(np.int64)
    t = np.asarray(targets).reshape(-1).astype(np.int64)
    ok = (t >= 0) & (t < num_classes) & (p >= 0) & (p < num_classes)
    idx = t[ok] * num_classes + p[ok]
    return np.bincount(idx, minlength=num_classes * num_classes).reshape(num_classes, num_classes).astype(np.int64)


def iou_from_confusion(cm, test_mask=None, ignore_gt=None):
    """(mIoU, iou[C] float64 with NaN where TP+FP+FN == 0)  (evaluator.py:63-103)."""
    cm = np.array(cm, dtype=np.float64)
    c = cm.shape[0]
    for r in ignore_gt or ():
        if 0 <= r < c:
            cm[r, :] = 0.0
    tp = np.diag(cm)
    denom = cm.sum(0) + cm.sum(1) - tp
    iou = np.full(c, np.nan)
    np.divide(tp, denom, out=iou, where=denom > 0)
    mask = np.ones(c, bool) if test_mask is None else np.asarray(test_mask, bool)
    sel = mask & np.isfinite(iou)
    miou = float(np.mean(iou[sel])) if sel.any() else float("nan")
    return miou, iou


def top_label(probs, labels, ignore_index=None, mode="probs", eps=1e-12):
    """(conf float32[n], correct bool[n]) over valid pixels in NCHW scan order (ece.py:55-84), in torch CPU fp32 ops so that the
    class sums round exactly as the reference's do."""
    import torch
    x = torch.as_tensor(np.asarray(probs, dtype=np.float32))
    if mode == "probs":
        p = x.clamp_min(0)
        p = p / p.sum(dim=1, keepdim=True).clamp_min(eps)
    elif mode == "alpha":
        p = x / (x.sum(dim=1, keepdim=True) + eps)
    elif mode == "logits":
        p = x.softmax(dim=1)
    else:
        raise ValueError(mode)
    conf, pred = p.max(dim=1)
    lab = torch.as_tensor(np.asarray(labels)).long()
    valid = torch.ones_like(lab, dtype=torch.bool) if ignore_index is None else lab != ignore_index
    conf = conf[valid].to(torch.float32).view(-1).clamp_(0, 1)
    return conf.numpy(), (pred[valid].view(-1) == lab[valid].view(-1)).numpy()


def ece_bins(conf, correct, n_bins: int = 15):
    """(n int64[n_bins], sum_correct float64, sum_conf float64): uniform float32 edges, last bin
    right-inclusive -- the np.histogram call of ece.py:136-140."""
    edges = np.linspace(0.0, 1.0, n_bins + 1, dtype=np.float32)
    edges[0], edges[-1] = 0.0, 1.0
    conf = np.asarray(conf, np.float32)
    n = np.histogram(conf, bins=edges)[0].astype(np.int64)
    acc_s = np.histogram(conf, bins=edges, weights=np.asarray(correct, np.float32))[0]
    conf_s = np.histogram(conf, bins=edges, weights=conf)[0]
    return n, acc_s.astype(np.float64), conf_s.astype(np.float64)


def ece_from_bins(n, acc_s, conf_s):
    """(ece, mce)  (ece.py:161-169)."""
    n = np.asarray(n, np.float64)
    if n.sum() == 0:
        return float("nan"), float("nan")
    acc = np.divide(acc_s, n, out=np.zeros_like(n), where=n > 0)
    conf = np.divide(conf_s, n, out=np.zeros_like(n), where=n > 0)
    gap = np.abs(acc - conf)
    return float(np.sum(n / max(1.0, n.sum()) * gap)), float(np.max(gap[n > 0]))


class ECESamples:
    """ece.py:86-111 (ECEAggregator.update after the top-label step): the (confidence, correct) sample buffers with the optional
    reservoir cap -- fill up to `max_samples` (a uniformly drawn subset of a batch that does not fit), afterwards keep each new sample
    with probability max_samples / seen and overwrite uniformly drawn slots.  One numpy Generator(seed), draws in the reference's order."""

    def __init__(self, max_samples=None, seed=0):
        self.max_samples, self.rng = max_samples, np.random.default_rng(seed)
        self.conf, self.correct, self.seen = np.empty(0, np.float32), np.empty(0, bool), 0

    def update(self, conf, correct):
        conf, correct = np.asarray(conf, np.float32), np.asarray(correct, bool)
        n_new = conf.size
        if n_new == 0:                       # ece.py:80-81: a batch without valid pixels changes nothing
            return
        self.seen += n_new
        if self.max_samples is None:
            self.conf, self.correct = np.concatenate([self.conf, conf]), np.concatenate([self.correct, correct])
        elif self.conf.size < self.max_samples:
            take = min(self.max_samples - self.conf.size, n_new)
            if take < n_new:
                idx = self.rng.choice(n_new, size=take, replace=False)
                conf, correct = conf[idx], correct[idx]
            self.conf, self.correct = np.concatenate([self.conf, conf]), np.concatenate([self.correct, correct])
        else:
            keep = self.rng.random(n_new) < min(1.0, float(self.max_samples) / float(self.seen + 1e-9))
            if keep.any():
                conf, correct = conf[keep], correct[keep]
                slots = self.rng.choice(self.max_samples, size=conf.size, replace=False)
                self.conf[slots], self.correct[slots] = conf, correct


def ece_edges(conf, n_bins: int = 15, binning: str = "uniform"):
    """ece.py:114-128 (_bin_edges): float32 linspace; 'adaptive' = empirical quantiles of the stored confidences, ends pinned to
    0 / 1, duplicates removed, falling back to the uniform edges when fewer than n_bins + 1 distinct edges remain."""
    uniform = np.linspace(0.0, 1.0, n_bins + 1, dtype=np.float32)
    conf = np.asarray(conf, np.float32)
    edges = uniform
    if binning == "adaptive" and conf.size:
        edges = np.quantile(conf, np.linspace(0.0, 1.0, n_bins + 1, dtype=np.float32))
        edges[0], edges[-1] = 0.0, 1.0
        edges = np.unique(edges)
        if edges.size < n_bins + 1:
            edges = uniform
    edges[0], edges[-1] = 0.0, 1.0
    return edges


def ece_bins_over(conf, correct, edges):
    """(n int64, sum_correct f64, sum_conf f64) over explicit edges: the three np.histogram calls of ece.py:136-140."""
    conf = np.asarray(conf, np.float32)
    n = np.histogram(conf, bins=edges)[0].astype(np.int64)
    acc_s = np.histogram(conf, bins=edges, weights=np.asarray(correct, np.float32))[0]
    conf_s = np.histogram(conf, bins=edges, weights=conf)[0]
    return n, acc_s.astype(np.float64), conf_s.astype(np.float64)


# ---- AUROC of error detection (src/metrics/auroc.py:36-78), restated with torch CPU ops / numpy ------------------------------
def auroc_samples(preds, labels, mode="alpha", score="entropy_norm", ignore_index=None, eps=1e-12, score_override=None):
    """(score float32[n], is_error uint8[n]) over valid pixels in NCHW scan order: auroc.py:36-63 (probabilities and score) and
    :96-117 (prediction, validity mask, error flag)."""
    import math

    import torch
    from torch.special import digamma
    preds = torch.as_tensor(preds)
    labels = torch.as_tensor(labels)
    if labels.dim() == 4:
        labels = labels[:, 0]
    if mode == "alpha":
        p = preds / (preds.sum(dim=1, keepdim=True) + eps)
    elif mode == "logits":
        p = preds.softmax(dim=1)
    elif mode == "probs":
        p = preds.clamp_min(0)
        p = p / p.sum(dim=1, keepdim=True).clamp_min(eps)
    else:
        raise ValueError(mode)
    pred = p.argmax(dim=1)
    lab = labels.long()
    valid = torch.ones_like(lab, dtype=torch.bool) if ignore_index is None else lab != ignore_index
    if score_override is not None:
        smap = torch.as_tensor(score_override)
    elif score in ("entropy", "entropy_norm", "1-maxprob") or mode != "alpha":
        if score == "1-maxprob":
            smap = 1.0 - p.max(dim=1).values
        else:
            h = -(p.clamp_min(eps) * p.clamp_min(eps).log()).sum(dim=1)
            smap = h / math.log(p.size(1)) if score == "entropy_norm" else h
    else:
        a0 = preds.sum(dim=1, keepdim=True) + eps
        q = preds / a0
        h = -(q.clamp_min(eps) * q.clamp_min(eps).log()).sum(dim=1)
        eh = -(q * (digamma(preds + 1.0) - digamma(a0 + 1.0))).sum(dim=1)
        mi = h - eh
        smap = mi / math.log(preds.size(1)) if score == "mi_norm" else mi
    return smap[valid].reshape(-1).to(torch.float32).numpy(), (pred != lab)[valid].reshape(-1).to(torch.uint8).numpy()


def auroc_from_samples(scores, is_error):
    """auroc.py:65-78: sort by score descending, cumulative sums, trapezoid of TPR over FPR; NaN when a class is empty."""
    scores = np.asarray(scores)
    is_error = np.asarray(is_error)
    order = np.argsort(-scores)
    y = is_error[order].astype(np.float64)
    pos, neg = y.sum(), y.size - y.sum()
    if pos == 0 or neg == 0:
        return float("nan")
    tpr = np.concatenate(([0.0], np.cumsum(y) / pos, [1.0]))
    fpr = np.concatenate(([0.0], np.cumsum(1.0 - y) / neg, [1.0]))
    return float(_trapz(tpr, fpr))


# ---- accuracy vs uncertainty bins (src/models/evaluator.py:640-749 UncertaintyAccuracyAggregator) --------------------------------
def ua_samples(labels, preds, uncertainty, ignore_ids=()):
    """(u float32[n] clamped to [0,1], correct uint8[n]) in scan order, labels in ignore_ids dropped (evaluator.py:659-673)."""
    lab = np.asarray(labels).astype(np.int64).reshape(-1)
    prd = np.asarray(preds).astype(np.int64).reshape(-1)
    unc = np.clip(np.asarray(uncertainty).astype(np.float32).reshape(-1), 0.0, 1.0)
    if len(ignore_ids):
        mask = ~np.isin(lab, np.asarray(list(ignore_ids), dtype=np.int64))
        lab, prd, unc = lab[mask], prd[mask], unc[mask]
    return unc, (lab == prd).astype(np.uint8)


def ua_make_bins(num_bins=None, bin_width=None, bin_edges=None):
    """evaluator.py:708-724: float32 edges covering [0, 1]; priority bin_edges > bin_width > num_bins (default 10)."""
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


def ua_binned(u, correct, edges):
    """(n int[K], accuracy float[K] with NaN for empty bins, pct float[K]) -- evaluator.py:733-740."""
    u = np.asarray(u, dtype=np.float32)
    c = np.asarray(correct).astype(np.float32)
    n = np.histogram(u, bins=edges)[0].astype(int)
    csum = np.histogram(u, bins=edges, weights=c)[0]
    acc = np.divide(csum, n, out=np.full_like(csum, np.nan, dtype=float), where=n > 0)
    return n, acc, 100.0 * n / max(1, u.size)


class PerClassSamples:
    """models/evaluator.py:191-262 (UncertaintyPerClassAggregator): per class the uncertainty values of the pixels labelled with it,
    in scan order; with `cap` an approximate reservoir -- fill (a random subset of a batch that does not fit), then accept each new
    sample with probability cap / seen and overwrite random slots.  One numpy Generator, classes visited in ascending order."""

    def __init__(self, num_classes, cap=None, seed=0):
        self.c, self.cap, self.rng = int(num_classes), cap, np.random.default_rng(seed)
        self.values = [np.empty(0, np.float32) for _ in range(self.c)]
        self.seen = [0] * self.c

    def update(self, labels, uncertainty):
        lab = np.asarray(labels).astype(np.int64).reshape(-1)
        unc = np.asarray(uncertainty).astype(np.float32).reshape(-1)
        for k in range(self.c):
            new = unc[lab == k]
            if new.size == 0:
                continue
            self.seen[k] += int(new.size)
            have = self.values[k]
            if self.cap is None:
                self.values[k] = np.concatenate([have, new])
            elif have.size < self.cap:
                room = min(self.cap - have.size, new.size)
                if room < new.size:
                    new = new[self.rng.choice(new.size, size=room, replace=False)]
                self.values[k] = np.concatenate([have, new])
            else:
                keep = self.rng.random(new.size) < min(1.0, float(self.cap) / float(self.seen[k] + 1e-9))
                if keep.any():
                    chosen = new[keep]
                    have[self.rng.choice(self.cap, size=chosen.size, replace=False)] = chosen
