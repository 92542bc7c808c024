#!/usr/bin/env python
"""Pin the oracle against the reference and (re)generate tests/golden/*.npz.

Runs ONLY in the build container (needs /root/reference; CPU is enough).  It imports the
reference's own Python modules read-only, checks that every oracle/ function agrees with them on
seeded inputs, and stores small input/expected-output vectors.  The reference's source never
enters the repo: fixtures are data.

    python tools/gen_golden.py            # asserts + writes tests/golden/
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
sys.path.insert(0, ROOT)
sys.path.insert(1, REF)
for stub in ("seaborn", "cv2"):     # used only by plotting helpers of models.evaluator / models.probability_helper
    sys.modules.setdefault(stub, types.ModuleType(stub))
sys.modules["cv2"].COLORMAP_TURBO = 20    # default argument of a plotting helper, evaluated at import (probability_helper.py:251)

from baselines.SalsaNext.SalsaNext import SalsaNext as RefSalsaNext   # noqa: E402  (reference)
from losses.lovasz import LovaszSoftmaxStable as RefLovasz             # noqa: E402
from metrics.ece import ECEAggregator as RefECE                        # noqa: E402
from models.evaluator import IoUEvaluator as RefIoU                    # noqa: E402
from models.losses import CrossEntropyLoss as RefCE                    # noqa: E402
from utils.mc_dropout import predictive_entropy_mc as ref_pred_entropy # noqa: E402

from oracle import losses as olosses, metrics as ometrics, salsanext as osalsa, uncertainty as ounc  # noqa: E402
from semanticlidarunc_amd.testing import seeded_model, synthetic_scan  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"  wrote {name}.npz  ({', '.join(f'{k}{tuple(np.asarray(v).shape)}' for k, v in arrs.items())})")


def maxdiff(a, b):
    return float((a - b).abs().max())


def sd_digest(sd):
    """Cheap fingerprint of a state_dict: (sum, abs-sum) in float64."""
    s = sum(float(v.double().sum()) for v in sd.values())
    a = sum(float(v.double().abs().sum()) for v in sd.values())
    return np.array([s, a])


def main():
    # ---------------- SalsaNext, eval mode ----------------
    ref = seeded_model(RefSalsaNext)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    x, labels = synthetic_scan(1, 16, 64, seed=11)
    with torch.no_grad():
        y_ref = ref(x)
        y_or = osalsa.salsanext_forward(sd, x)
    d = maxdiff(y_ref, y_or)
    print(f"SalsaNext eval 1x5x16x64: |oracle - reference| = {d:.3e}")
    assert d <= 1e-5
    save("salsanext_eval_1x5x16x64", x=x, logits=y_ref, sd_digest=sd_digest(sd))
    import json
    with open(os.path.join(OUT, "salsanext_state_dict_keys.json"), "w") as f:
        json.dump({k: [str(v.dtype).replace("torch.", "")] + list(v.shape) for k, v in ref.state_dict().items()}, f, indent=0)
    drops = [n for n, m in ref.named_modules() if isinstance(m, torch.nn.Dropout2d)]
    with open(os.path.join(OUT, "salsanext_dropout_modules.json"), "w") as f:
        json.dump(drops, f)

    # ---------------- SalsaNext with explicit dropout multipliers (MC pass) ----------------
    x2, _ = synthetic_scan(2, 32, 64, seed=12)
    scales = osalsa.draw_dropout_scales(2, 0.2, torch.Generator().manual_seed(5))
    mods = dict(ref.named_modules())
    saved = {}
    for name, s in scales.items():          # instrument the reference's Dropout2d instances
        saved[name] = mods[name].forward
        mods[name].forward = (lambda t, s=s: t * s)
    with torch.no_grad():
        y_ref2 = ref(x2)
    for name, f in saved.items():
        mods[name].forward = f
    with torch.no_grad():
        y_or2 = osalsa.salsanext_forward(sd, x2, scales)
    d = maxdiff(y_ref2, y_or2)
    print(f"SalsaNext + dropout multipliers 2x5x32x64: |oracle - reference| = {d:.3e}")
    assert d <= 1e-5
    save("salsanext_mc_2x5x32x64", x=x2, logits=y_ref2,
         **{"scale:" + k: v.reshape(v.shape[0], v.shape[1]) for k, v in scales.items()})

    # train-mode BatchNorm (batch statistics) -- oracle only pinned, used by later backward work
    ref.train()
    from utils.mc_dropout import set_dropout_mode
    set_dropout_mode(ref, False)
    ref_t = RefSalsaNext(20, 5)
    ref_t.load_state_dict(sd)
    ref_t.train()
    set_dropout_mode(ref_t, False)
    y_ref3 = ref_t(x2)
    y_or3 = osalsa.salsanext_forward(sd, x2, None, bn_train=True)
    d = maxdiff(y_ref3.detach(), y_or3)
    print(f"SalsaNext train-BN 2x5x32x64: |oracle - reference| = {d:.3e}")
    assert d <= 2e-4
    ref.eval()

    # ---------------- training step: train-mode BatchNorm + dropout multipliers, forward + backward ----------------
    x3, _ = synthetic_scan(2, 64, 128, seed=13)
    x3 = x3.requires_grad_(True)
    scales3 = osalsa.draw_dropout_scales(2, 0.2, torch.Generator().manual_seed(6))
    ref_b = RefSalsaNext(20, 5)
    ref_b.load_state_dict(sd)
    ref_b.train()
    mods_b = dict(ref_b.named_modules())
    for name, s in scales3.items():
        mods_b[name].forward = (lambda t, s=s: t * s)
    proj = torch.randn(2, 20, 64, 128, generator=torch.Generator().manual_seed(7))
    out_b = ref_b(x3)
    ((out_b * proj).sum() / 64.0).backward()
    pnames = ["downCntx.conv1.weight", "downCntx.conv1.bias", "downCntx.bn1.weight", "downCntx.bn1.bias", "downCntx2.conv3.weight",
              "resBlock1.conv5.weight", "resBlock2.conv4.weight", "resBlock5.bn4.weight", "upBlock3.conv1.weight",
              "upBlock4.conv1.weight", "upBlock4.bn4.bias", "logits.weight", "logits.bias"]
    params_b = dict(ref_b.named_parameters())
    sd_o = {k: v.clone().requires_grad_(v.is_floating_point() and k in params_b) for k, v in sd.items()}
    x3o = x3.detach().clone().requires_grad_(True)
    out_o, bn_stats_o = osalsa.salsanext_forward(sd_o, x3o, scales3, bn_train=True, return_bn_stats=True)
    ((out_o * proj).sum() / 64.0).backward()
    worst = max(maxdiff(sd_o[k].grad, params_b[k].grad) / (1e-6 + float(params_b[k].grad.abs().max())) for k in params_b)
    print(f"train step: logits |d|={maxdiff(out_o.detach(), out_b.detach()):.2e}  dx |d|={maxdiff(x3o.grad, x3.grad):.2e}  "
          f"worst relative param-grad diff={worst:.2e}")
    assert maxdiff(out_o.detach(), out_b.detach()) <= 2e-4 and worst <= 2e-3
    gnorm = {k: float(params_b[k].grad.norm()) for k in params_b}
    save("train_step_2x5x64x128", x=x3.detach(), proj_seed=7, logits=out_b.detach()[:, :, ::2, ::4], grad_x=x3.grad,
         grad_norms=np.array([gnorm[k] for k in sorted(gnorm)]),
         running_mean_downCntx_bn1=ref_b.state_dict()["downCntx.bn1.running_mean"],
         running_var_downCntx_bn1=ref_b.state_dict()["downCntx.bn1.running_var"],
         running_var_resBlock5_bn4=ref_b.state_dict()["resBlock5.bn4.running_var"],
         **{"scale:" + k: v.reshape(v.shape[0], v.shape[1]) for k, v in scales3.items()},
         **{"grad:" + k: params_b[k].grad for k in pnames})

    # ---------------- MC reduction ----------------
    g = torch.Generator().manual_seed(21)
    mc_logits = torch.randn(4, 1, 20, 4, 64, generator=g) * 3.0
    p_bar, h_norm, mi_norm, preds = ounc.mc_reduce(mc_logits)
    probs = torch.softmax(mc_logits, dim=2)
    d = maxdiff(ref_pred_entropy(probs), h_norm)
    print(f"predictive entropy: |oracle - reference predictive_entropy_mc| = {d:.3e}")
    assert d <= 1e-6
    save("mc_reduce_T4_1x20x4x64", logits=mc_logits, p_bar=p_bar, h_norm=h_norm, mi_norm=mi_norm, preds=preds)
    probs1, h1, pr1 = ounc.single_pass(mc_logits[0])
    save("single_pass_1x20x4x64", logits=mc_logits[0], probs=probs1, h_norm=h1, preds=pr1)

    # ---------------- loss: NLL + Lovasz ----------------
    g = torch.Generator().manual_seed(31)
    lg = (torch.randn(2, 20, 8, 64, generator=g) * 2.0).requires_grad_(True)
    lab = torch.randint(0, 20, (2, 8, 64), generator=g)
    lab[(lab == 7) | (lab == 13)] = 3                      # two absent classes
    lab[torch.rand(2, 8, 64, generator=g) < 0.2] = 0       # ~20 % ignored by Lovasz
    pr = torch.softmax(lg, dim=1)
    ls_ref = RefLovasz(ignore_index=0)(pr, lab, "probs")
    nll_ref = torch.nn.NLLLoss()(torch.log(pr.clamp(min=1e-8)), lab)
    (ls_ref + nll_ref).backward()
    grad_ref = lg.grad.clone()
    lg2 = lg.detach().clone().requires_grad_(True)
    tot, nll_or, ls_or = olosses.salsanext_loss(lg2, lab)
    tot.backward()
    print(f"NLL |d|={abs(float((nll_or - nll_ref).detach())):.2e}  Lovasz |d|={abs(float((ls_or - ls_ref).detach())):.2e}  "
          f"grad |d|={maxdiff(lg2.grad, grad_ref):.2e}")
    assert abs(float(nll_or - nll_ref)) <= 1e-6 and abs(float(ls_or - ls_ref)) <= 1e-6
    assert maxdiff(lg2.grad, grad_ref) <= 1e-6
    ls_none = RefLovasz(ignore_index=None)(pr, lab, "probs")
    assert abs(float(olosses.lovasz_softmax(pr, lab, None) - ls_none)) <= 1e-6
    ce_ref = RefCE(ignore_index=0)(lg.detach(), lab, 20, "logits")
    assert abs(float(olosses.cross_entropy(lg.detach(), lab, 0, "logits") - ce_ref)) <= 1e-6
    save("loss_2x20x8x64", logits=lg.detach(), labels=lab, nll=nll_ref.detach(), lovasz=ls_ref.detach(),
         lovasz_noignore=ls_none.detach(), ce_ignore0=ce_ref, grad_logits=grad_ref)

    # tiny hand-checkable Lovasz / NLL cases (4 pixels, 2 classes)
    p4 = torch.tensor([[0.9, 0.1], [0.4, 0.6], [0.3, 0.7], [0.8, 0.2]]).t().reshape(1, 2, 1, 4).contiguous()
    y4 = torch.tensor([[[0, 1, 0, 1]]])
    kat = dict(probs=p4, labels=y4,
               lovasz_none=RefLovasz(ignore_index=None)(p4, y4, "probs"),
               lovasz_ign0=RefLovasz(ignore_index=0)(p4, y4, "probs"),
               nll=torch.nn.NLLLoss()(torch.log(p4.clamp(min=1e-8)), y4))
    assert abs(float(olosses.lovasz_softmax(p4, y4, None) - kat["lovasz_none"])) < 1e-7
    assert abs(float(olosses.lovasz_softmax(p4, y4, 0) - kat["lovasz_ign0"])) < 1e-7
    assert abs(float(olosses.nll_on_probs(p4, y4) - kat["nll"])) < 1e-7
    print("KAT 4px/2cls: lovasz(none)=%.8f lovasz(ign0)=%.8f nll=%.8f" % (kat["lovasz_none"], kat["lovasz_ign0"], kat["nll"]))
    save("kat_4px_2cls", **kat)

    # ---------------- metrics: IoU + ECE ----------------
    g = torch.Generator().manual_seed(41)
    pm = torch.softmax(torch.randn(2, 20, 16, 64, generator=g) * 2.0, dim=1)
    _, lab_m = synthetic_scan(2, 16, 64, seed=42)
    agree = torch.rand(2, 16, 64, generator=g) < 0.6
    pred_m = torch.where(agree, lab_m, pm.argmax(1))
    ev = RefIoU(20)
    ev.update(pred_m, lab_m)
    names = [f"c{i}" for i in range(20)]
    test_mask = [0] + [1] * 19
    miou_ref, dict_ref = ev.compute(names, test_mask=test_mask, ignore_gt=[0])
    cm_or = ometrics.confusion_matrix(pred_m.numpy(), lab_m.numpy(), 20)
    assert np.array_equal(cm_or, ev.confmat.numpy())
    miou_or, iou_or = ometrics.iou_from_confusion(cm_or, test_mask, [0])
    assert abs(miou_or - miou_ref) < 1e-12
    print(f"mIoU reference={miou_ref:.6f} oracle={miou_or:.6f}")
    save("iou_2x16x64", preds=pred_m, labels=lab_m, confmat=ev.confmat.numpy(), miou=miou_ref,
         iou=np.array([dict_ref[n] for n in names]))

    # make the probabilities agree with the labels often enough for a non-trivial reliability curve
    boost = torch.zeros_like(pm).scatter_(1, lab_m.unsqueeze(1), 1.0)
    pe = torch.softmax(torch.log(pm) + 2.5 * boost * agree.unsqueeze(1), dim=1)
    ece_ref = RefECE(n_bins=15, mode="probs", ignore_index=0, max_samples=None)
    ece_ref.update(pe, lab_m)
    # the reference's compute() only works with a plot path (its `fig` is unbound otherwise, ece.py:171,212)
    (e_ref, m_ref), stats_ref = ece_ref.compute(save_plot_path="/tmp/_ref_ece.png")[:2]
    conf, corr = ometrics.top_label(pe.numpy(), lab_m.numpy(), ignore_index=0)
    n, acc_s, conf_s = ometrics.ece_bins(conf, corr, 15)
    e_or, m_or = ometrics.ece_from_bins(n, acc_s, conf_s)
    assert np.array_equal(n, stats_ref["n"].to_numpy())
    assert abs(e_or - e_ref) < 1e-7 and abs(m_or - m_ref) < 1e-7
    print(f"ECE reference={e_ref:.6f} oracle={e_or:.6f}; MCE {m_ref:.6f}/{m_or:.6f}")
    save("ece_2x20x16x64", probs=pe, labels=lab_m, n=n, ece=e_ref, mce=m_ref,
         acc=np.nan_to_num(stats_ref["acc"].to_numpy()), conf=np.nan_to_num(stats_ref["conf"].to_numpy()))
    # ---------------- Dirichlet head (models/probability_helper.py) ----------------
    from models import probability_helper as ref_ph            # reference (cv2 stubbed above; only plotting helpers use it)
    from oracle import dirichlet as odir
    g = torch.Generator().manual_seed(77)
    outs = torch.randn(2, 21, 8, 64, generator=g) * 3.0
    outs[0, 20] += 25.0                                        # exercises softplus' linear branch (x > 20)
    shape_l, scale_l = outs[:, :20], outs[:, 20:21]
    a_ref = ref_ph.to_alpha_concentrations_from_shape_and_scale(shape_l, scale_l)
    h_ref, au_ref, eu_ref = ref_ph.get_predictive_entropy(a_ref), ref_ph.get_aleatoric_uncertainty(a_ref), ref_ph.get_epistemic_uncertainty(a_ref)
    hn_ref = ref_ph.get_predictive_entropy_norm(a_ref)
    a0 = a_ref.sum(dim=1, keepdim=True) + ref_ph.get_eps_value()           # trainer.py:537-538
    p_ref = a_ref / a0
    a_or, p_or, hn_or, pr_or = odir.head(outs, 20)
    for nme, x, y in (("alpha", a_ref, a_or), ("p_hat", p_ref, p_or), ("H_norm", hn_ref, hn_or), ("H", h_ref, odir.predictive_entropy(a_or)),
                      ("AU", au_ref, odir.aleatoric(a_or)), ("EU", eu_ref, odir.epistemic(a_or))):
        assert maxdiff(x, y) == 0.0, (nme, maxdiff(x, y))
    a_t2 = ref_ph.to_alpha_concentrations_from_shape_and_scale(shape_l, scale_l, T=2.5, eps=1e-6)
    assert maxdiff(a_t2, odir.alpha_from_shape_and_scale(shape_l, scale_l, 2.5, 1e-6)) == 0.0
    save("dirichlet_head_2x21x8x64", outputs=outs, alpha=a_ref, p_hat=p_ref, H=h_ref, H_norm=hn_ref, AU=au_ref, EU=eu_ref, alpha_T2p5_eps1em6=a_t2,
         preds=a_ref.argmax(dim=1))
    print("  oracle.dirichlet == reference probability_helper (0.0)")

    # ---------------- AUROC of error detection (metrics/auroc.py) ----------------
    import tempfile
    from metrics.auroc import AUROCAggregator as RefAUROC              # reference
    g = torch.Generator().manual_seed(91)
    labs = torch.randint(0, 20, (2, 16, 64), generator=g)
    labs[torch.rand(2, 16, 64, generator=g) < 0.15] = 0                 # ignored pixels
    logit = torch.randn(2, 20, 16, 64, generator=g) * 2.0
    logit.scatter_add_(1, labs[:, None], torch.full((2, 1, 16, 64), 2.5))   # mostly-right predictions
    alpha_in = torch.nn.functional.softplus(logit) + 1.0
    probs_in = logit.softmax(1)
    override = torch.rand(2, 16, 64, generator=g)
    out = {"labels": labs.numpy(), "logits": logit.numpy(), "alpha": alpha_in.numpy(), "override": override.numpy()}
    cases = [("alpha", "entropy_norm", alpha_in, None), ("alpha", "mi_norm", alpha_in, None), ("alpha", "mi", alpha_in, None),
             ("alpha", "1-maxprob", alpha_in, None), ("logits", "entropy", logit, None), ("logits", "mi_norm", logit, None),
             ("probs", "entropy_norm", probs_in, None), ("probs", "1-maxprob", probs_in, None), ("logits", "entropy_norm", logit, override)]
    for mode, score, inp, ov in cases:
        agg = RefAUROC(mode=mode, score=score, ignore_index=0)
        agg.update(inp, labs, score_override=ov)
        agg.update(inp.flip(0), labs.flip(0)[:, None], score_override=None if ov is None else ov.flip(0))      # [B,1,H,W] labels
        with tempfile.TemporaryDirectory() as td:
            a_ref = agg.compute(save_plot_path=os.path.join(td, "roc.png"))[0]
        s1, e1 = ometrics.auroc_samples(inp, labs, mode, score, 0, 1e-12, ov)
        s2, e2 = ometrics.auroc_samples(inp.flip(0), labs.flip(0), mode, score, 0, 1e-12, None if ov is None else ov.flip(0))
        so, eo = np.concatenate([s1, s2]), np.concatenate([e1, e2])
        assert np.array_equal(so, agg._scores.numpy()) and np.array_equal(eo, agg._is_error.numpy()), (mode, score)
        a_or = ometrics.auroc_from_samples(so, eo)
        assert a_or == a_ref, (mode, score, a_or, a_ref)
        tag = f"{mode}|{score}|{'override' if ov is not None else 'own'}"
        out["auroc:" + tag] = np.float64(a_ref)
        out["nsamples:" + tag] = np.int64(so.size)
    # reservoir cap: three updates against max_samples = 1500 (fill, then probabilistic replacement), numpy seed 0
    agg = RefAUROC(mode="logits", score="entropy_norm", ignore_index=0, max_samples=1500, seed=0)
    for k in range(3):
        agg.update(logit.roll(k, 0) + 0.1 * k, labs.roll(k, 0))
    with tempfile.TemporaryDirectory() as td:
        out["auroc:capped1500"] = np.float64(agg.compute(save_plot_path=os.path.join(td, "roc.png"))[0])
    out["capped_scores_sorted"] = np.sort(agg._scores.numpy())
    save("auroc_2x20x16x64", **out)
    print("  oracle.metrics.auroc_* == reference AUROCAggregator (samples and AUROC identical)")

    # ---------------- accuracy vs uncertainty bins (models/evaluator.py UncertaintyAccuracyAggregator) ----------------
    from models.evaluator import UncertaintyAccuracyAggregator as RefUA      # reference (seaborn / cv2 stubbed above)
    g = torch.Generator().manual_seed(123)
    ua_lab = torch.randint(0, 20, (2, 16, 64), generator=g)
    ua_prd = torch.where(torch.rand(2, 16, 64, generator=g) < 0.7, ua_lab, torch.randint(0, 20, (2, 16, 64), generator=g))
    ua_unc = torch.rand(2, 16, 64, generator=g) * 1.2 - 0.1                   # some values outside [0, 1]: clamped by update()
    ua_unc[0, 0, :8] = torch.tensor([0.0, 0.1, 0.2, 0.3, 0.5, 0.9, 1.0, 1.0])  # values on bin edges
    out = {"labels": ua_lab.numpy(), "preds": ua_prd.numpy(), "uncertainty": ua_unc.numpy()}
    agg = RefUA()
    agg.update(ua_lab, ua_prd, ua_unc, ignore_ids=(0, 7))
    agg.update(ua_lab.flip(0), ua_prd.flip(0), ua_unc.flip(0))
    u1, c1 = ometrics.ua_samples(ua_lab, ua_prd, ua_unc, (0, 7))
    u2, c2 = ometrics.ua_samples(ua_lab.flip(0), ua_prd.flip(0), ua_unc.flip(0))
    uo, co = np.concatenate([u1, u2]), np.concatenate([c1, c2])
    assert np.array_equal(uo, agg._uncert.numpy()) and np.array_equal(co, agg._correct.numpy())
    custom = np.array([0.0, 0.05, 0.3, 0.31, 0.8, 1.0], dtype=np.float32)
    for tag, kw in (("bins10", {}), ("width0.05", {"bin_width": 0.05}), ("custom", {"bin_edges": custom}), ("bins64", {"num_bins": 64})):
        df = agg.binned_accuracy(**kw)
        edges = ometrics.ua_make_bins(kw.get("num_bins", 10), kw.get("bin_width"), kw.get("bin_edges"))
        n_o, acc_o, pct_o = ometrics.ua_binned(uo, co, edges)
        assert np.array_equal(df["n"].to_numpy(), n_o) and np.allclose(df["accuracy"].to_numpy(), acc_o, equal_nan=True, rtol=0, atol=0)
        assert np.array_equal(df["low"].to_numpy(), edges[:-1]) and np.allclose(df["pct"].to_numpy(), pct_o, rtol=0, atol=0)
        out["n:" + tag], out["accuracy:" + tag], out["edges:" + tag] = n_o, acc_o, edges
    agg = RefUA(max_samples=900, seed=0)
    for k in range(3):
        agg.update(ua_lab.roll(k, 0), ua_prd.roll(k, 1), ua_unc.roll(k, 2), ignore_ids=(0,))
    out["capped_n:bins10"] = agg.binned_accuracy()["n"].to_numpy()
    out["capped_accuracy:bins10"] = agg.binned_accuracy()["accuracy"].to_numpy()
    save("ua_bins_2x16x64", **out)
    print("  oracle.metrics.ua_* == reference UncertaintyAccuracyAggregator (samples, counts, accuracies identical)")

    # ---------------- Tversky loss (models/losses.py) ----------------
    from models.losses import TverskyLoss as RefTversky                 # reference
    g = torch.Generator().manual_seed(55)
    tv_lab = torch.randint(0, 20, (2, 8, 64), generator=g)
    tv_lab[torch.rand(2, 8, 64, generator=g) < 0.1] = 255               # ignored
    tv_lab[0, 0, :3] = torch.tensor([-1, 20, 31])                       # out-of-range labels are invalid too
    tv_lab[tv_lab == 5] = 6                                             # class 5 absent
    tv_logits = torch.randn(2, 20, 8, 64, generator=g) * 2.0
    out = {"labels": tv_lab.numpy(), "logits": tv_logits.numpy()}
    for act, inp in (("logits", tv_logits), ("probs", tv_logits.softmax(1)), ("log_probs", tv_logits.log_softmax(1))):
        for red in ("mean", "sum", "none"):
            xr = inp.clone().requires_grad_(True)
            lr = RefTversky(alpha=0.7, beta=0.3, smooth=1.0, ignore_index=255, reduction=red)(xr, tv_lab, 20, act)
            wgt = torch.linspace(0.5, 1.5, 20)
            (lr * wgt).sum().backward() if red == "none" else lr.backward()
            xo = inp.clone().requires_grad_(True)
            lo = olosses.tversky(xo, tv_lab, 20, act, 0.7, 0.3, 1.0, 255, red)
            (lo * wgt).sum().backward() if red == "none" else lo.backward()
            assert maxdiff(lr.detach(), lo.detach()) == 0.0 and maxdiff(xr.grad, xo.grad) == 0.0, (act, red)
            out[f"loss:{act}|{red}"] = lr.detach().numpy()
            out[f"grad:{act}|{red}"] = xr.grad.numpy()
    out["loss:all_ignored"] = RefTversky()(tv_logits, torch.full((2, 8, 64), 255), 20, "logits").detach().numpy()
    save("tversky_2x20x8x64", **out)
    print("  oracle.losses.tversky == reference TverskyLoss (value and gradient identical)")

    # ---------------- per-class uncertainty samples (models/evaluator.py UncertaintyPerClassAggregator) ----------------
    from models.evaluator import UncertaintyPerClassAggregator as RefPC      # reference
    g = torch.Generator().manual_seed(909)
    pc_lab = torch.randint(0, 6, (3, 2, 16, 64), generator=g)                  # three batches
    pc_lab[torch.rand(3, 2, 16, 64, generator=g) < 0.5] = 1                    # one dominant class, so a cap of 300 bites early
    pc_unc = torch.rand(3, 2, 16, 64, generator=g)
    out = {"labels": pc_lab.numpy(), "uncertainty": pc_unc.numpy()}
    for tag, cap in (("all", None), ("cap300", 300)):
        ref_pc, my_pc = RefPC(6, max_per_class=cap, seed=5), ometrics.PerClassSamples(6, cap, seed=5)
        for b in range(3):
            ref_pc.update(pc_lab[b], pc_unc[b])
            my_pc.update(pc_lab[b].numpy(), pc_unc[b].numpy())
        for c in range(6):
            assert np.array_equal(ref_pc._values[c].numpy(), my_pc.values[c]), (tag, c)
        assert list(ref_pc._seen_counts) == my_pc.seen
        out["values:" + tag] = np.concatenate([v.numpy() for v in ref_pc._values])
        out["sizes:" + tag] = np.asarray([v.numel() for v in ref_pc._values], dtype=np.int64)
        out["seen:" + tag] = np.asarray(ref_pc._seen_counts, dtype=np.int64)
    save("per_class_uncertainty_3x2x16x64", **out)
    print("  oracle.metrics.PerClassSamples == reference UncertaintyPerClassAggregator (lists and seen counts identical)")

    # ---------------- per-pixel Dirichlet losses (losses/dirichlet_losses.py, losses/regularizers.py) ----------------
    from losses import dirichlet_losses as ref_dl, regularizers as ref_reg     # reference
    from oracle import dirichlet as odir2
    g = torch.Generator().manual_seed(808)
    dl_lab = torch.randint(0, 20, (2, 8, 64), generator=g)
    dl_lab[torch.rand(2, 8, 64, generator=g) < 0.12] = 0                       # ignored
    dl_alpha = 1.0 + torch.nn.functional.softplus(torch.randn(2, 20, 8, 64, generator=g) * 2.0) * torch.rand(2, 1, 8, 64, generator=g) * 30
    out = {"labels": dl_lab.numpy(), "alpha": dl_alpha.numpy()}
    pairs = [("nll_dircat", ref_dl.NLLDirichletCategorical(ignore_index=0), lambda a: odir2.loss_nll_dircat(a, dl_lab, 0)),
             ("digamma_ce", ref_dl.DigammaDirichletCE(ignore_index=0), lambda a: odir2.loss_digamma_ce(a, dl_lab, 0)),
             ("brier", ref_dl.BrierDirichlet(ignore_index=0), lambda a: odir2.loss_brier(a, dl_lab, 0)),
             ("brier_sref40", ref_dl.BrierDirichlet(ignore_index=0, s_ref=40.0), lambda a: odir2.loss_brier(a, dl_lab, 0, 40.0)),
             ("mse", ref_dl.DirichletMSELoss(ignore_index=0), lambda a: odir2.loss_mse(a, dl_lab, 0)),
             ("kl_off_uniform", ref_reg.KL_offClasses_to_uniform(ignore_index=0), lambda a: odir2.loss_kl_off_uniform(a, dl_lab, 0))]
    pairs += [("complement_kl", ref_dl.ComplementKLUniform(ignore_index=0, gamma=1.25, tau=0.65, sigma=0.15),        # the Trainer's settings (trainer.py:339)
               lambda a: odir2.loss_complement_kl(a, dl_lab, 0, 1.25, 0.65, 0.15)),
              ("complement_kl_gated", ref_dl.ComplementKLUniform(ignore_index=0, s_target=30.0, normalize=False, detach_uncert=False),
               lambda a: odir2.loss_complement_kl(a, dl_lab, 0, s_target=30.0, normalize=False, detach_uncert=False)),
              ("wrong_low_evidence", ref_reg.WrongLowEvidence(ignore_index=0), lambda a: odir2.loss_wrong_low_evidence(a, dl_lab, 0)),
              ("wrong_low_evidence_hard", ref_reg.WrongLowEvidence(ignore_index=0, s_low=4.0, margin=0.1, soft_margin_k=0.0),
               lambda a: odir2.loss_wrong_low_evidence(a, dl_lab, 0, 4.0, 0.1, 0.0)),
              ("wrong_low_evidence_nomargin", ref_reg.WrongLowEvidence(ignore_index=None, margin=0.0),
               lambda a: odir2.loss_wrong_low_evidence(a, dl_lab, None, margin=0.0))]
    for name, ref_mod, ofn in pairs:
        ar = dl_alpha.clone().requires_grad_(True)
        lr = ref_mod(ar, dl_lab[:, None] if name == "mse" else dl_lab)          # [B,1,H,W] labels are accepted too
        lr.backward()
        ao = dl_alpha.clone().requires_grad_(True)
        lo = ofn(ao)
        lo.backward()
        assert maxdiff(lr.detach(), lo.detach()) == 0.0 and maxdiff(ar.grad, ao.grad) == 0.0, name
        out["loss:" + name], out["grad:" + name] = lr.detach().numpy(), ar.grad.numpy()
    save("dirichlet_losses_2x20x8x64", **out)
    print("  oracle.dirichlet.loss_* == reference Dirichlet losses (value and gradient identical)")

    # ---------------- spherical projection (dataset/utils.py; cv2 / seaborn stubbed above, used by its plotting helpers only) ----------------
    from dataset.utils import spherical_projection as ref_projection      # reference
    from oracle import projection as oproj
    rs = np.random.default_rng(2024)
    npts = 30000
    az, el = rs.uniform(-np.pi, np.pi, npts), rs.uniform(-0.43, 0.05, npts)          # a 64-beam-like vertical field of view
    rng_m = rs.uniform(2.0, 80.0, npts)
    xyz = np.stack([rng_m * np.cos(el) * np.cos(az), rng_m * np.cos(el) * np.sin(az), rng_m * np.sin(el)], 1).astype(np.float32)
    cloud = np.concatenate([xyz, rs.uniform(0, 1, (npts, 1)).astype(np.float32), rs.integers(0, 20, (npts, 1))], axis=-1)   # float64, as the dataloader
    out = {"cloud": cloud}
    for tag, tr in (("data_range", None), ("fixed_range", [-np.pi / 8, np.pi / 8])):
        img_r, alpha_r, th_r, ph_r = ref_projection(cloud, 32, 256, theta_range=tr)
        img_o, alpha_o, th_o, ph_o = oproj.spherical_projection(cloud, 32, 256, theta_range=tr)
        assert np.array_equal(img_r, img_o) and np.array_equal(alpha_r, alpha_o) and tuple(th_r) == tuple(th_o), tag
        out["img:" + tag], out["theta_range:" + tag] = img_r, np.asarray(th_r, dtype=np.float64)
    save("spherical_projection_30000x5_32x256", **out)
    print("  oracle.projection.spherical_projection == reference dataset.utils.spherical_projection (bit-identical image)")

    # ---------------- ResNet-FPN (models/semanticFCN.py) through a stub torchvision serving oracle.fpn.ResNetRef ----------------
    from oracle import fpn as ofpn
    from semanticlidarunc_amd.fpn import SemanticNetworkWithFPN as MyFPN
    from semanticlidarunc_amd.testing import randomize_bn_
    tv = types.ModuleType("torchvision")
    tv.models = ofpn.torchvision_models_stub()
    sys.modules["torchvision"], sys.modules["torchvision.models"] = tv, tv.models
    from models.semanticFCN import SemanticNetworkWithFPN as RefFPN        # the reference's own wiring
    for tag, kw, shape in (("resnet18_m6_c20", dict(backbone="resnet18", input_channels=2, meta_channel_dim=6, num_classes=20), (1, 32, 128)),
                           ("resnet34_m3_c3_noatt", dict(backbone="resnet34", input_channels=2, meta_channel_dim=3, num_classes=3,
                                                         attention=False, multi_scale_meta=False), (2, 16, 64))):
        torch.manual_seed(0)
        mine = randomize_bn_(MyFPN(**kw), 3).eval()
        ref_f = RefFPN(**kw)
        sdf = mine.state_dict()
        assert list(sdf.keys()) == list(ref_f.state_dict().keys())
        ref_f.load_state_dict(sdf)
        ref_f.eval()
        g = torch.Generator().manual_seed(51)
        xf = torch.randn(shape[0], 2, shape[1], shape[2], generator=g) * torch.tensor([20.0, 0.3]).view(1, 2, 1, 1)
        mf = torch.randn(shape[0], kw["meta_channel_dim"], shape[1], shape[2], generator=g) * 5.0
        with torch.no_grad():
            yr = ref_f(xf, mf)
            yo = ofpn.fpn_forward(sdf, xf, mf, kw["backbone"], kw.get("attention", True), kw.get("multi_scale_meta", True))
        print(f"FPN {tag}: |oracle - reference| = {maxdiff(yr, yo):.3e}  (out min {float(yr.min()):.3f})")
        assert maxdiff(yr, yo) <= 1e-5
        save("fpn_" + tag, x=xf, meta=mf, out=yr, sd_digest=sd_digest({k: v for k, v in sdf.items() if v.is_floating_point()}))
    with open(os.path.join(OUT, "fpn_resnet18_state_dict_keys.json"), "w") as f:
        torch.manual_seed(0)
        json.dump({k: list(v.shape) for k, v in RefFPN("resnet18", 2, 6, num_classes=20).state_dict().items()}, f, indent=0)
    print("all oracle functions pinned against the reference")


if __name__ == "__main__":
    main()
