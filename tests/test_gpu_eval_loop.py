"""GPU: the evaluation step of the reference's ``Trainer.test_one_epoch`` MC branch (trainer.py:1138-1168) assembled from this
repo's drop-in classes -- MC forward + fused reduction, IoU, accuracy-vs-uncertainty bins, ECE, AUROC -- and checked end to end
against the oracle evaluating the SAME stacked logits on the CPU (so the comparison isolates the reduction / metric chain; the
network itself is covered by test_gpu_model / test_gpu_h8)."""
import math

import numpy as np
import pytest
import torch

from oracle import metrics as ometrics
from oracle import uncertainty as ounc
from semanticlidarunc_amd import salsanext as sn
from semanticlidarunc_amd.metrics.auroc import AUROCAggregator
from semanticlidarunc_amd.metrics.ece import ECEAggregator
from semanticlidarunc_amd.models.evaluator import IoUEvaluator, UncertaintyAccuracyAggregator
from semanticlidarunc_amd.testing import seeded_model, synthetic_scan
from semanticlidarunc_amd.utils.mc_dropout import mc_forward, mc_predict

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["fp32", "f16"])
def test_mc_evaluation_step_matches_oracle_metrics(cuda, precision):
    model = seeded_model(sn.SalsaNext).to(cuda)
    sn.set_conv_precision(precision)
    try:
        T, names = 4, [str(i) for i in range(20)]
        iou, ua = IoUEvaluator(20), UncertaintyAccuracyAggregator()
        ece = ECEAggregator(n_bins=15, mode="probs", ignore_index=0, max_samples=None)
        auroc = AUROCAggregator(mode="probs", score="entropy_norm", ignore_index=0)
        auroc_mi = AUROCAggregator(mode="probs", score="entropy_norm", ignore_index=0)
        stacks, labels_all = [], []
        for step in range(2):                                        # two "batches" of two scans
            x, labels = synthetic_scan(2, 64, 512, seed=40 + step)
            x, labels = x.to(cuda), labels.to(cuda)
            torch.manual_seed(7 + step)
            mc_logits = mc_forward(model, [x], T=T)                  # [T,B,C,H,W]; the same seed reproduces the passes below
            torch.manual_seed(7 + step)
            p_bar, h_norm, mi_norm, preds = mc_predict(model, [x], T=T)
            iou.update(preds, labels)
            ua.update(labels=labels, preds=preds, uncertainty=h_norm, ignore_ids=(0,))
            ece.update(p_bar, labels)
            auroc.update(p_bar, labels)
            auroc_mi.update(p_bar, labels, score_override=mi_norm)
            stacks.append(mc_logits.cpu())
            labels_all.append(labels.cpu())
        # ---- oracle on the same logits ----
        cm = np.zeros((20, 20), dtype=np.int64)
        us, cs, confs, oks, a_s, a_e, m_s = [], [], [], [], [], [], []
        for mc_logits, labels in zip(stacks, labels_all):
            p_bar, h_norm, mi_norm, preds = ounc.mc_reduce(mc_logits)
            cm += ometrics.confusion_matrix(preds.numpy(), labels.numpy(), 20)
            u, c = ometrics.ua_samples(labels, preds, h_norm, (0,))
            us.append(u); cs.append(c)
            conf, ok = ometrics.top_label(p_bar.numpy(), labels.numpy(), 0, "probs")
            confs.append(conf); oks.append(ok)
            s, e = ometrics.auroc_samples(p_bar, labels, "probs", "entropy_norm", 0)
            a_s.append(s); a_e.append(e)
            m_s.append(ometrics.auroc_samples(p_bar, labels, "probs", "entropy_norm", 0, score_override=mi_norm)[0])
        want_miou, _ = ometrics.iou_from_confusion(cm, [0] + [1] * 19, [0])
        got_miou, _ = iou.compute(names, test_mask=[0] + [1] * 19, ignore_gt=[0])
        # argmax near-ties between the device and the CPU softmax can move a handful of pixels
        assert int(np.abs(iou.confmat.cpu().numpy() - cm).sum()) <= 32 and abs(got_miou - want_miou) <= 5e-4
        n_w, acc_w, _ = ometrics.ua_binned(np.concatenate(us), np.concatenate(cs), ometrics.ua_make_bins(10))
        df = ua.binned_accuracy()
        assert int(np.abs(df["n"].to_numpy() - n_w).sum()) <= 32
        big = n_w >= 1000                                            # accuracy of a well-filled bin moves by <= 32 / 1000 pixels at worst
        assert np.all(np.abs(df["accuracy"].to_numpy()[big] - acc_w[big]) <= 5e-3)
        n_b, acc_b, conf_b = ometrics.ece_bins(np.concatenate(confs), np.concatenate(oks), 15)
        want_ece, _ = ometrics.ece_from_bins(n_b, acc_b, conf_b)
        (got_ece, _), _, _ = ece.compute()
        assert abs(got_ece - want_ece) <= 2e-4
        assert abs(auroc.compute()[0] - ometrics.auroc_from_samples(np.concatenate(a_s), np.concatenate(a_e))) <= 5e-4
        assert abs(auroc_mi.compute()[0] - ometrics.auroc_from_samples(np.concatenate(m_s), np.concatenate(a_e))) <= 2e-3     # MI ~ 1e-3: rank noise
        assert 0.0 <= got_ece <= 1.0 and 0.0 <= got_miou <= 1.0 and math.isfinite(auroc.compute()[0])
    finally:
        sn.set_conv_precision("fp32")
