#!/usr/bin/env python
"""Round-2 additions to the pinned oracle: same rules as tools/gen_golden.py (runs ONLY in the build container, imports the
reference read-only from /root/reference, asserts oracle == reference, writes small data-only fixtures under tests/golden/).

    python tools/gen_golden_r02.py [ece]
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
sys.path.insert(0, ROOT)
sys.path.insert(1, REF)
for stub in ("seaborn", "cv2"):
    sys.modules.setdefault(stub, types.ModuleType(stub))
sys.modules["cv2"].COLORMAP_TURBO = 20

from oracle import metrics as ometrics          # noqa: E402
from semanticlidarunc_amd.testing import synthetic_scan  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"  wrote {name}.npz  ({', '.join(f'{k}{tuple(np.asarray(v).shape)}' for k, v in arrs.items())})")


def gen_ece():
    """ECEAggregator beyond the cap (metrics/ece.py:93-111) and with binning='adaptive' (:115-128)."""
    from metrics.ece import ECEAggregator as RefECE                     # reference
    g = torch.Generator().manual_seed(41)
    pm = torch.softmax(torch.randn(2, 20, 16, 64, generator=g) * 2.0, dim=1)
    _, lab = synthetic_scan(2, 16, 64, seed=42)
    agree = torch.rand(2, 16, 64, generator=g) < 0.6
    boost = torch.zeros_like(pm).scatter_(1, lab.unsqueeze(1), 1.0)
    pe = torch.softmax(torch.log(pm) + 2.5 * boost * agree.unsqueeze(1), dim=1)
    logits = torch.log(pe) + 0.3
    alpha = torch.nn.functional.softplus(logits * 2.0) + 1.0
    out = {"probs": pe.numpy(), "logits": logits.numpy(), "alpha": alpha.numpy(), "labels": lab.numpy()}
    inputs = {"probs": pe, "logits": logits, "alpha": alpha}

    def batches(x):      # four updates of ~1840 valid pixels each: fill, overflow inside a batch, then replacement
        return [(x, lab), (x.flip(0), lab.flip(0)), (x.roll(1, 3), lab.roll(1, 2)), (x.flip(3), lab.flip(2))]

    for mode in ("probs", "logits", "alpha"):
        for cap, binning in ((3000, "uniform"), (None, "adaptive"), (2500, "adaptive")):
            ref = RefECE(n_bins=15, mode=mode, ignore_index=0, max_samples=cap, seed=0, binning=binning)
            orc = ometrics.ECESamples(cap, seed=0)
            for xb, lb in batches(inputs[mode]):
                ref.update(xb, lb)
                c, k = ometrics.top_label(xb.numpy(), lb.numpy(), 0, mode)
                orc.update(c, k)
            assert np.array_equal(orc.conf, ref._conf.numpy()) and np.array_equal(orc.correct, ref._correct.numpy()), (mode, cap, binning)
            assert orc.seen == ref._seen
            with tempfile.TemporaryDirectory() as td:
                (e_ref, m_ref), stats = ref.compute(save_plot_path=os.path.join(td, "r.png"))[:2]
            edges = ometrics.ece_edges(orc.conf, 15, binning)
            n, acc_s, conf_s = ometrics.ece_bins_over(orc.conf, orc.correct, edges)
            e_or, m_or = ometrics.ece_from_bins(n, acc_s, conf_s)
            assert np.array_equal(n, stats["n"].to_numpy()) and np.array_equal(edges[:-1], stats["low"].to_numpy())
            assert abs(e_or - e_ref) < 1e-7 and abs(m_or - m_ref) < 1e-7, (mode, cap, binning, e_or, e_ref)
            tag = f"{mode}|{cap}|{binning}"
            out["ece:" + tag], out["mce:" + tag] = np.float64(e_ref), np.float64(m_ref)
            out["n:" + tag], out["edges:" + tag] = n, edges
            out["seen:" + tag], out["kept:" + tag] = np.int64(ref._seen), np.int64(ref._conf.numel())
            out["conf_sorted:" + tag] = np.sort(ref._conf.numpy())
            out["ncorrect:" + tag] = np.int64(int(ref._correct.sum()))
            print(f"  ECE {tag}: reference={e_ref:.6f} oracle={e_or:.6f} kept {ref._conf.numel()} of {ref._seen}")
    # degenerate adaptive case: many identical confidences -> duplicate quantiles -> uniform fallback (ece.py:124-126)
    onehot = torch.zeros(1, 20, 4, 64).scatter_(1, torch.randint(1, 20, (1, 1, 4, 64), generator=g), 1.0)
    lab1 = torch.randint(1, 20, (1, 4, 64), generator=g)
    ref = RefECE(n_bins=15, mode="probs", ignore_index=0, binning="adaptive")
    ref.update(onehot, lab1)
    with tempfile.TemporaryDirectory() as td:
        (e_ref, _), stats = ref.compute(save_plot_path=os.path.join(td, "r.png"))[:2]
    c, k = ometrics.top_label(onehot.numpy(), lab1.numpy(), 0, "probs")
    edges = ometrics.ece_edges(c, 15, "adaptive")
    assert np.array_equal(edges[:-1], stats["low"].to_numpy())
    out["onehot_probs"], out["onehot_labels"], out["ece:onehot_adaptive"] = onehot.numpy(), lab1.numpy(), np.float64(e_ref)
    save("ece_capped_adaptive_2x20x16x64", **out)
    print("  oracle.metrics.ECESamples / ece_edges == reference ECEAggregator (buffers, edges, bins identical)")


GENERATORS = {"ece": gen_ece}

if __name__ == "__main__":
    for name in (sys.argv[1:] or list(GENERATORS)):
        print(f"[{name}]")
        GENERATORS[name]()
    print("round-2 oracle additions pinned against the reference")
