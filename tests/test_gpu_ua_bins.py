"""GPU: accuracy-vs-uncertainty aggregator: device sample selection and np.histogram-equivalent binning against
the reference's golden counts (integer-exact) and the oracle at full size."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import metrics as ometrics
from semanticlidarunc_amd import ops
from semanticlidarunc_amd.models.evaluator import UncertaintyAccuracyAggregator

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_aggregator_against_reference_golden(cuda, tmp_path):
    g = golden("ua_bins_2x16x64")
    lab, prd, unc = (_t(g[k]).to(cuda) for k in ("labels", "preds", "uncertainty"))
    agg = UncertaintyAccuracyAggregator()
    agg.update(lab, prd, unc, ignore_ids=(0, 7))
    agg.update(lab.flip(0), prd.flip(0), unc.flip(0))
    assert agg._uncert.is_cuda                                  # samples stay on the device
    for tag, kw in (("bins10", {}), ("width0.05", {"bin_width": 0.05}), ("bins64", {"num_bins": 64}),
                    ("custom", {"bin_edges": np.array([0.0, 0.05, 0.3, 0.31, 0.8, 1.0], dtype=np.float32)})):
        df = agg.binned_accuracy(**kw)
        assert np.array_equal(df["n"].to_numpy(), g["n:" + tag])
        assert np.allclose(df["accuracy"].to_numpy(), g["accuracy:" + tag], equal_nan=True, rtol=0, atol=1e-15)
        assert np.array_equal(df["low"].to_numpy(), g["edges:" + tag][:-1]) and df["label"].iloc[-1].endswith("]")
    # reservoir cap with the reference's numpy draws
    agg = UncertaintyAccuracyAggregator(max_samples=900, seed=0)
    for k in range(3):
        agg.update(lab.roll(k, 0), prd.roll(k, 1), unc.roll(k, 2), ignore_ids=(0,))
    df = agg.binned_accuracy()
    assert agg._uncert.numel() == 900 and np.array_equal(df["n"].to_numpy(), g["capped_n:bins10"])
    assert np.allclose(df["accuracy"].to_numpy(), g["capped_accuracy:bins10"], equal_nan=True, rtol=0, atol=1e-15)
    fig, ax = agg.plot_accuracy_vs_uncertainty_bins(save_path=str(tmp_path / "ua.png"), dpi=50)
    assert (tmp_path / "ua.png").stat().st_size > 0 and len(ax.patches) >= 10
    agg.reset()
    assert agg.binned_accuracy().empty and agg.plot_accuracy_vs_uncertainty_bins() is None


def test_full_size_counts_and_argument_checks(cuda):
    gen = torch.Generator().manual_seed(31)
    lab = torch.randint(0, 20, (4, 64, 2048), generator=gen)
    prd = torch.where(torch.rand(4, 64, 2048, generator=gen) < 0.8, lab, torch.randint(0, 20, (4, 64, 2048), generator=gen))
    unc = torch.rand(4, 64, 2048, generator=gen) ** 2 * 1.1
    u, f = ops.ua_samples(lab.to(cuda), prd.to(cuda), unc.to(cuda), (0, 3))
    wu, wc = ometrics.ua_samples(lab, prd, unc, (0, 3))
    keep = (f != 2).cpu()
    assert torch.equal(u.cpu()[keep], torch.from_numpy(wu)) and torch.equal(f.cpu()[keep], torch.from_numpy(wc))
    for edges in (ometrics.ua_make_bins(10), ometrics.ua_make_bins(None, 0.004), np.array([0.0, 1e-6, 0.2, 0.99999, 1.0], np.float32)):
        cnt, ok = ops.binned_counts(u[keep.to(cuda)].contiguous(), f[keep.to(cuda)].contiguous(), torch.from_numpy(edges).to(cuda))
        n, acc, _ = ometrics.ua_binned(wu, wc, edges)
        assert np.array_equal(cnt.cpu().numpy(), n)
        assert np.array_equal(ok.cpu().numpy(), np.histogram(wu, bins=edges, weights=wc.astype(np.float64))[0].astype(np.int64))
    with pytest.raises(RuntimeError):
        ops.ua_samples(lab, prd, unc)                                               # CPU tensors: no fallback
    with pytest.raises(RuntimeError):
        ops.binned_counts(u, f, torch.linspace(0, 1, 300, device=cuda))             # more than 256 bins
    with pytest.raises(AssertionError):
        UncertaintyAccuracyAggregator().update(lab.to(cuda), prd.to(cuda), unc[:1].to(cuda))
