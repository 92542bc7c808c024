"""GPU: the north-star parity claim, end to end.  Four 64x2048 scans, T = 8 MC passes: (GPU network in the benchmark's fp16-storage
precision -> fused head + MC reduction -> device IoU / ECE accumulators) against (oracle fp32 network with the SAME Dropout2d
multipliers -> oracle reduction -> oracle IoU / ECE).  Nothing of the GPU side is fed to the oracle.  Reference: trainer.py:1138-1168,
models/evaluator.py:29-105, metrics/ece.py:67-168.  Bar (BASELINE.json north_star): |d mIoU|, |d ECE|, entropy <= 1e-3."""
import numpy as np
import pytest
import torch

from oracle import metrics as ometrics
from oracle import salsanext as osalsa
from oracle import uncertainty as ounc
from semanticlidarunc_amd import salsanext as sn
from semanticlidarunc_amd.metrics.ece import ECEAggregator
from semanticlidarunc_amd.models.evaluator import IoUEvaluator
from semanticlidarunc_amd.testing import seeded_model, synthetic_scan
from semanticlidarunc_amd.utils.mc_dropout import dropout_sampling

pytestmark = pytest.mark.gpu
T, NCLS, BAR = 8, 20, 1e-3


def _oracle_scan(sd, x1, seed):
    """(stacked multipliers, oracle MC reduction, labels).  Labels come from the oracle too, but from its deterministic eval pass
    (argmax + 30 % seeded noise, empty returns -> class 0): labels tied to the MC mean's OWN argmax would make every near-tie pixel
    "correct" for the oracle path by construction and "wrong" for any other path, which no real label does."""
    g = torch.Generator().manual_seed(seed)
    scales = [osalsa.draw_dropout_scales(1, 0.2, g) for _ in range(T)]
    with torch.no_grad():
        outs = [osalsa.salsanext_forward(sd, x1, s) for s in scales]
        det = osalsa.salsanext_forward(sd, x1).argmax(1)
    gl = torch.Generator().manual_seed(900 + seed)
    labels = torch.where(torch.rand(det.shape, generator=gl) < 0.30, torch.randint(1, NCLS, det.shape, generator=gl), det)
    labels = labels.masked_fill(x1[:, 0] == 0, 0)
    return {k: torch.cat([s[k] for s in scales], 0) for k in scales[0]}, ounc.mc_reduce(torch.stack(outs, 0)), labels


@pytest.mark.parametrize("precision,share_prefix", [("f16", False), ("f16", True), ("fp32", False)])
def test_miou_ece_entropy_of_the_gpu_path_match_the_oracle_path(cuda, precision, share_prefix):
    model = seeded_model(sn.SalsaNext).to(cuda)
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    x, _ = synthetic_scan(4, 64, 2048, seed=77)
    iou = IoUEvaluator(NCLS)
    ece = ECEAggregator(n_bins=15, mode="probs", ignore_index=0, max_samples=500000)      # the Trainer's construction (trainer.py:215-222)
    ece_all = ECEAggregator(n_bins=15, mode="probs", ignore_index=0, max_samples=None)
    cm = np.zeros((NCLS, NCLS), dtype=np.int64)
    o_ece, confs, oks = ometrics.ECESamples(500000, seed=0), [], []
    worst = {"p_bar": 0.0, "h": 0.0, "mi": 0.0, "flips": 0.0}
    sn.set_conv_precision(precision)
    try:
        for b in range(4):
            x1 = x[b:b + 1].contiguous()
            stacked, (p_w, h_w, mi_w, pred_w), labels = _oracle_scan(sd, x1, seed=b)
            model.eval()
            with torch.no_grad(), dropout_sampling(model, True):
                xg = x1.to(cuda)
                if model.mc_fused_ok(xg, T):
                    p_g, h_g, mi_g, pred_g = model.mc_predict_fused(xg, T, share_prefix=share_prefix, scales=stacked)
                else:
                    from semanticlidarunc_amd import ops
                    lg = model.forward_with_dropout_scales(xg.repeat(T, 1, 1, 1), stacked)
                    p_g, h_g, mi_g, pred_g = ops.mc_reduce(lg.reshape(T, 1, *lg.shape[1:]).contiguous())
            lab_g = labels.to(cuda)
            iou.update(pred_g, lab_g)
            ece.update(p_g, lab_g)
            ece_all.update(p_g, lab_g)
            cm += ometrics.confusion_matrix(pred_w.numpy(), labels.numpy(), NCLS)
            c, k = ometrics.top_label(p_w.numpy(), labels.numpy(), 0, "probs")
            o_ece.update(c, k)
            confs.append(c); oks.append(k)
            worst["p_bar"] = max(worst["p_bar"], float((p_g.cpu() - p_w).abs().max()))
            worst["h"] = max(worst["h"], float((h_g.cpu() - h_w).abs().max()))
            worst["mi"] = max(worst["mi"], float((mi_g.cpu() - mi_w).abs().max()))
            worst["flips"] = max(worst["flips"], float((pred_g.cpu() != pred_w).float().mean()))
    finally:
        sn.set_conv_precision("fp32")
    mask = [0] + [1] * (NCLS - 1)
    miou_g, _ = iou.compute([str(i) for i in range(NCLS)], test_mask=mask, ignore_gt=[0])
    miou_w, _ = ometrics.iou_from_confusion(cm, mask, [0])
    (ece_g, mce_g), _ = ece.compute()[:2]
    ece_w, mce_w = ometrics.ece_from_bins(*ometrics.ece_bins_over(o_ece.conf, o_ece.correct, ometrics.ece_edges(o_ece.conf, 15)))
    (ece_ag, _), _ = ece_all.compute()[:2]
    ece_aw, _ = ometrics.ece_from_bins(*ometrics.ece_bins(np.concatenate(confs), np.concatenate(oks), 15))
    assert 0.05 < miou_w < 0.95 and ece_w > 0.01                                # a non-trivial operating point (random labels give 0.0075)
    assert o_ece.seen == ece._seen and o_ece.conf.size == ece._conf.numel()     # 4 x ~118 k valid pixels: under the cap, nothing dropped
    assert abs(miou_g - miou_w) <= BAR, (miou_g, miou_w)
    assert abs(ece_g - ece_w) <= BAR and abs(ece_ag - ece_aw) <= BAR, (ece_g, ece_w, ece_ag, ece_aw)
    assert worst["p_bar"] <= BAR and worst["h"] <= BAR and worst["mi"] <= BAR, worst
    assert worst["flips"] <= 5e-3, worst
