"""GPU: loss kernels (Lovasz radix sort + Jaccard scan, NLL/CE, fused SalsaNext loss and its backward) against
the reference's golden values and the oracle.  Loss values 1e-5 abs; gradients 1e-5 abs (tie-free inputs)."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import losses as olosses
from semanticlidarunc_amd import ops
from semanticlidarunc_amd.loss import salsanext_loss
from semanticlidarunc_amd.losses.lovasz import LovaszSoftmaxStable
from semanticlidarunc_amd.models.losses import CrossEntropyLoss

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_lovasz_golden_value_and_gradient(cuda):
    g = golden("loss_2x20x8x64")
    lab = _t(g["labels"]).to(cuda)
    probs = torch.softmax(_t(g["logits"]), 1).to(cuda).requires_grad_(True)
    loss = LovaszSoftmaxStable(ignore_index=0)(probs, lab, "probs")
    assert abs(float(loss) - float(g["lovasz"])) <= 1e-5
    assert abs(float(LovaszSoftmaxStable(None)(probs.detach(), lab, "probs")) - float(g["lovasz_noignore"])) <= 1e-5
    loss.backward()
    pc = torch.softmax(_t(g["logits"]), 1).requires_grad_(True)
    olosses.lovasz_softmax(pc, _t(g["labels"]), 0).backward()
    assert float((probs.grad.cpu() - pc.grad).abs().max()) <= 1e-6
    # logits mode: the device softmax and its backward are in the graph
    lg = _t(g["logits"]).to(cuda).requires_grad_(True)
    l2 = LovaszSoftmaxStable(ignore_index=0)(lg, lab, "logits")
    assert abs(float(l2) - float(g["lovasz"])) <= 1e-5
    l2.backward()
    lc = _t(g["logits"]).clone().requires_grad_(True)
    olosses.lovasz_softmax(torch.softmax(lc, 1), _t(g["labels"]), 0).backward()
    assert float((lg.grad.cpu() - lc.grad).abs().max()) <= 1e-6
    with pytest.raises(ValueError):
        LovaszSoftmaxStable(0)(probs, lab, "nope")


def test_lovasz_known_answers_and_edge_cases(cuda):
    k = golden("kat_4px_2cls")
    p, y = _t(k["probs"]).to(cuda), _t(k["labels"]).to(cuda)
    assert abs(float(LovaszSoftmaxStable(None)(p, y, "probs")) - float(k["lovasz_none"])) <= 1e-6
    assert abs(float(LovaszSoftmaxStable(0)(p, y, "probs")) - float(k["lovasz_ign0"])) <= 1e-6
    # every pixel ignored -> 0 ; a single pixel ; a class that never occurs is skipped
    assert float(LovaszSoftmaxStable(0)(p, torch.zeros_like(y), "probs")) == 0.0
    one = torch.tensor([0.25, 0.75], device=cuda).reshape(1, 2, 1, 1)
    want = olosses.lovasz_softmax(one.cpu(), torch.tensor([[[1]]]), None)
    assert abs(float(LovaszSoftmaxStable(None)(one, torch.tensor([[[1]]], device=cuda), "probs")) - float(want)) <= 1e-6


@pytest.mark.parametrize("tag,ign,classes", [("all_ign0", 0, "all"), ("all_none", None, "all"), ("list_ign0", 0, [1, 3, 19]), ("present_ign0", 0, "present")])
def test_lovasz_all_classes_and_class_lists(cuda, tag, ign, classes):
    """LovaszSoftmaxStable(classes='all' / [ids]) (lovasz.py:7,66-69): value and gradient of the reference's own class; absent classes (3, 4, 6,
    ... never occur in these labels; 3 is in the list) contribute their largest probability."""
    g = golden("lovasz_classes_2x20x8x64")
    probs = torch.softmax(_t(g["logits"]), 1).to(cuda).requires_grad_(True)
    loss = LovaszSoftmaxStable(ign, classes)(probs, _t(g["labels"]).to(cuda), "probs")
    assert abs(float(loss) - float(g["loss_" + tag])) <= 1e-5
    loss.backward()
    assert float((probs.grad.cpu() - _t(g["grad_" + tag])).abs().max()) <= 1e-6
    with pytest.raises(ValueError):
        LovaszSoftmaxStable(0, "some")
    with pytest.raises(IndexError):
        LovaszSoftmaxStable(0, [1, 20])(probs.detach(), _t(g["labels"]).to(cuda), "probs")


def test_lovasz_training_size_against_oracle(cuda):
    # BASELINE configs[1] shape: B=4, 64x2048, 20 classes, ~10 % ignored (label 0), classes 7 and 13 absent
    gen = torch.Generator().manual_seed(3)
    probs = torch.softmax(torch.randn(4, 20, 64, 2048, generator=gen) * 2.0, 1)
    lab = torch.randint(0, 20, (4, 64, 2048), generator=gen)
    lab[(lab == 7) | (lab == 13)] = 2
    lab[torch.rand(4, 64, 2048, generator=gen) < 0.1] = 0
    pd = probs.to(cuda).requires_grad_(True)
    loss = LovaszSoftmaxStable(0)(pd, lab.to(cuda), "probs")
    pc = probs.clone().requires_grad_(True)
    want = olosses.lovasz_softmax(pc, lab, 0)
    assert abs(float(loss) - float(want)) <= 2e-5
    loss.backward()
    want.backward()
    diff = (pd.grad.cpu() - pc.grad).abs()
    # ties between equal errors make the reference's own sub-gradient order-dependent: allow a few
    assert float((diff > 1e-7).float().mean()) < 1e-4 and float(diff.max()) < 1e-4
    assert float(pd.grad[:, 7].abs().max()) == 0.0 and float(pd.grad.cpu()[lab.unsqueeze(1).expand(-1, 20, -1, -1) == 0].abs().max()) == 0.0


def test_cross_entropy_wrapper(cuda):
    g = golden("loss_2x20x8x64")
    lg = _t(g["logits"]).to(cuda).requires_grad_(True)
    lab = _t(g["labels"]).to(cuda)
    ce = CrossEntropyLoss(ignore_index=0)(lg, lab, 20, "logits")
    assert abs(float(ce) - float(g["ce_ignore0"])) <= 1e-5
    ce.backward()
    lc = _t(g["logits"]).clone().requires_grad_(True)
    olosses.cross_entropy(lc, _t(g["labels"]), 0, "logits").backward()
    assert float((lg.grad.cpu() - lc.grad).abs().max()) <= 1e-7
    bad = _t(g["labels"]).clone()
    bad[0, 0, :5] = 99                                             # out-of-range labels are ignored
    want = olosses.cross_entropy(_t(g["logits"]), bad, 255, "logits")
    assert abs(float(CrossEntropyLoss(255)(lg.detach(), bad.to(cuda), 20, "logits")) - float(want)) <= 1e-5
    pr = torch.softmax(_t(g["logits"]), 1)
    for act, x in (("probs", pr), ("log_probs", pr.log())):
        xd = x.to(cuda).requires_grad_(True)
        got = CrossEntropyLoss(0)(xd, lab, 20, act)
        xc = x.clone().requires_grad_(True)
        want = olosses.cross_entropy(xc, _t(g["labels"]), 0, act)
        assert abs(float(got) - float(want)) <= 1e-5
        got.backward(); want.backward()
        assert float((xd.grad.cpu() - xc.grad).abs().max()) <= 1e-5 * max(1.0, float(xc.grad.abs().max()))
    with pytest.raises(ValueError):
        CrossEntropyLoss(0)(lg, lab, 20, "nope")


def test_fused_salsanext_loss_and_backward(cuda):
    g = golden("loss_2x20x8x64")
    lg = _t(g["logits"]).to(cuda).requires_grad_(True)
    lab = _t(g["labels"]).to(cuda)
    total, nll, ls = salsanext_loss(lg, lab, 1.0, 1.0, 0)
    assert abs(float(nll) - float(g["nll"])) <= 1e-5 and abs(float(ls) - float(g["lovasz"])) <= 1e-5
    assert abs(float(total) - float(g["nll"]) - float(g["lovasz"])) <= 2e-5
    (g1,) = torch.autograd.grad(total, lg, retain_graph=True)
    (g2,) = torch.autograd.grad(total, lg, retain_graph=True)         # re-entrant (GradNorm probes)
    assert torch.equal(g1, g2)
    assert float((g1.cpu() - _t(g["grad_logits"])).abs().max()) <= 1e-6
    (2.0 * total).backward()
    assert float((lg.grad.cpu() - 2.0 * _t(g["grad_logits"])).abs().max()) <= 2e-6
    # weights
    t2, _, _ = salsanext_loss(lg.detach(), lab, 0.5, 2.0, 0)
    assert abs(float(t2) - (0.5 * float(g["nll"]) + 2.0 * float(g["lovasz"]))) <= 2e-5
