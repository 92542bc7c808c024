"""GPU: error-detection AUROC -- device scores / sort / ROC integral and the drop-in AUROCAggregator against the
reference's golden values and the oracle.  The integral is exact integer arithmetic; what differs from the CPU is the last bit of
expf / logf in the scores, which can swap neighbours in the ranking: 1e-5 absolute on AUROC."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import metrics as ometrics
from semanticlidarunc_amd import ops
from semanticlidarunc_amd.metrics.auroc import AUROCAggregator

pytestmark = pytest.mark.gpu
_trapz = getattr(np, "trapezoid", None) or np.trapz


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_aggregator_against_reference_golden(cuda):
    g = golden("auroc_2x20x16x64")
    labs = _t(g["labels"]).to(cuda)
    inputs = {"logits": _t(g["logits"]).to(cuda), "alpha": _t(g["alpha"]).to(cuda), "probs": _t(g["logits"]).softmax(1).to(cuda)}
    for key in g.files:
        if not key.startswith("auroc:") or key == "auroc:capped1500":
            continue
        mode, score, src = key[len("auroc:"):].split("|")
        ov = _t(g["override"]).to(cuda) if src == "override" else None
        agg = AUROCAggregator(mode=mode, score=score, ignore_index=0)
        agg.update(inputs[mode], labs, score_override=ov)
        agg.update(inputs[mode].flip(0), labs.flip(0)[:, None], score_override=None if ov is None else ov.flip(0))
        auroc, curves, fig = agg.compute()
        assert fig is None and abs(auroc - float(g[key])) <= 1e-5, (key, auroc, float(g[key]))
        n = int(g["nsamples:" + key[len("auroc:"):]])
        assert curves["fpr"].shape == (n + 2,) and curves["tpr"][-1] == 1.0 and curves["thresholds"][0] == np.inf
        assert np.all(np.diff(curves["thresholds"]) <= 0) and abs(float(_trapz(curves["tpr"], curves["fpr"])) - auroc) <= 1e-9
    # the numpy-seeded reservoir cap reproduces the reference's sample set
    agg = AUROCAggregator(mode="logits", score="entropy_norm", ignore_index=0, max_samples=1500, seed=0)
    for k in range(3):
        agg.update(inputs["logits"].roll(k, 0) + 0.1 * k, labs.roll(k, 0))
    assert agg._scores.numel() == 1500
    assert float((torch.sort(agg._scores).values.cpu() - _t(g["capped_scores_sorted"])).abs().max()) <= 2e-6
    assert abs(agg.compute()[0] - float(g["auroc:capped1500"])) <= 1e-5
    agg.reset()
    assert agg.compute() == (float("nan"), {}) or np.isnan(agg.compute()[0])


def test_sort_and_integral_exact_on_large_tie_free_input(cuda):
    n = 3_000_017                                     # not a multiple of the sort tile
    gen = torch.Generator().manual_seed(4)
    scores = torch.randperm(n, generator=gen).float() / n - 0.3          # distinct values of both signs (< 2^24, exact in fp32)
    err = (torch.rand(n, generator=gen) < (0.2 + 0.5 * (scores + 0.3))).to(torch.uint8)
    want = ometrics.auroc_from_samples(scores.numpy(), err.numpy())
    a, pos, neg, ss, se = ops.auroc_from_samples(scores.to(cuda), err.to(cuda), want_sorted=True)
    assert pos == int(err.sum()) and neg == n - pos
    assert abs(a - want) <= 1e-12                      # tie-free: the ranking is unique and the sum is exact
    order = torch.argsort(scores, descending=True)
    assert torch.equal(ss.cpu(), scores[order]) and torch.equal(se.cpu(), err[order])
    # degenerate inputs
    assert np.isnan(ops.auroc_from_samples(scores[:100].to(cuda), torch.ones(100, dtype=torch.uint8, device=cuda))[0])
    a_inf = ops.auroc_from_samples(torch.tensor([float("inf"), 1.0, -1.0, float("-inf")], device=cuda), torch.tensor([1, 0, 1, 0], dtype=torch.uint8, device=cuda))[0]
    assert a_inf == 0.75


def test_scores_full_size_and_argument_checks(cuda):
    gen = torch.Generator().manual_seed(8)
    labs = torch.randint(0, 20, (2, 64, 2048), generator=gen)
    logits = torch.randn(2, 20, 64, 2048, generator=gen) * 3
    alpha = torch.nn.functional.softplus(logits) + 1.0
    for mode, score, inp in (("logits", "entropy_norm", logits), ("alpha", "mi_norm", alpha), ("probs", "1-maxprob", logits.softmax(1))):
        s, f = ops.auroc_scores(inp.to(cuda), labs.to(cuda), mode, score, ignore_index=0)
        ws, we = ometrics.auroc_samples(inp, labs, mode, score, 0)
        valid = (f != 2).cpu()
        assert torch.equal(valid, labs != 0)
        assert float((s.cpu()[valid] - torch.from_numpy(ws)).abs().max()) <= (3e-5 if score == "mi_norm" else 2e-6)
        assert float((f.cpu()[valid] != torch.from_numpy(we)).float().mean()) < 1e-5
    with pytest.raises(RuntimeError):
        ops.auroc_scores(logits, labs, "logits", "entropy")                       # CPU tensors: no fallback
    with pytest.raises(ValueError):
        ops.auroc_scores(logits.to(cuda), labs.to(cuda), "logits", "variance")
    with pytest.raises(RuntimeError):
        ops.auroc_from_samples(torch.zeros(4, device=cuda), torch.zeros(5, dtype=torch.uint8, device=cuda))
