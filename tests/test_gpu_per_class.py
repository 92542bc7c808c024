"""GPU: per-class uncertainty samples: the stable group-by-class pass and the ``UncertaintyPerClassAggregator``
mirror against the lists the reference's aggregator held (golden; unlimited and capped -- same numpy draws) and, at full size,
against the oracle.  Bar: identical values in identical order."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import metrics as ometrics
from semanticlidarunc_amd import ops
from semanticlidarunc_amd.models.evaluator import UncertaintyPerClassAggregator

pytestmark = pytest.mark.gpu


def test_mirror_against_reference_golden(cuda):
    g = golden("per_class_uncertainty_3x2x16x64")
    for tag, cap in (("all", None), ("cap300", 300)):
        agg = UncertaintyPerClassAggregator(6, max_per_class=cap, seed=5)
        for b in range(3):
            agg.update(labels=torch.from_numpy(g["labels"][b]).to(cuda), uncertainty=torch.from_numpy(g["uncertainty"][b]).to(cuda))
        vals = agg._values
        assert all(v.is_cuda for v in vals)
        assert np.array_equal(torch.cat(vals).cpu().numpy(), g["values:" + tag]), tag
        assert [v.numel() for v in vals] == g["sizes:" + tag].tolist() and list(agg._seen_counts) == g["seen:" + tag].tolist()
        df = agg.as_dataframe([f"c{i}" for i in range(6)], ignore_ids=(0,))
        assert len(df) == int(g["sizes:" + tag][1:].sum()) and set(df["class_id"]) == {1, 2, 3, 4, 5}
        # the cache round trip of tester.py:356-357,648-649
        saved, seen = [v.detach().cpu() for v in agg._values], list(agg._seen_counts)
        agg.reset()
        assert all(v.numel() == 0 for v in agg._values) and agg.as_dataframe(["x"] * 6).empty
        agg._values, agg._seen_counts = [v.clone() for v in saved], seen
        assert np.array_equal(torch.cat(agg._values).numpy(), g["values:" + tag])
    with pytest.raises(RuntimeError):
        UncertaintyPerClassAggregator(6).update(torch.zeros(2, 2, dtype=torch.int64), torch.zeros(2, 2))      # CPU tensors: no fallback
    with pytest.raises(AttributeError):
        UncertaintyPerClassAggregator(6).plot_boxplot                        # the reference's plotting code is only there in drop-in mode


def test_group_by_class_full_size_and_ragged(cuda):
    gen = torch.Generator().manual_seed(11)
    for n, c in ((4 * 64 * 2048, 20), (1, 3), (63, 32), (1025, 5), (70001, 20)):
        lab = torch.randint(-1, c + 2, (n,), generator=gen)                   # includes labels outside [0, C): dropped
        val = torch.rand(n, generator=gen)
        grouped, counts = ops.group_by_class(lab.to(cuda), val.to(cuda), c)
        want = ometrics.PerClassSamples(c)
        want.update(lab.numpy(), val.numpy())
        assert counts.cpu().tolist() == [v.size for v in want.values]
        k = int(counts.sum())
        assert np.array_equal(grouped[:k].cpu().numpy(), np.concatenate(want.values))
    with pytest.raises(Exception):
        ops.group_by_class(torch.zeros(4, dtype=torch.int64, device=cuda), torch.zeros(4, device=cuda), 33)     # more than 32 classes
