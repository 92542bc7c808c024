"""GPU: a training step replayed as one HIP graph (semanticlidarunc_amd/graph_step.py) follows the eager step: same losses and
same parameters after several SGD steps (dropout off so that both runs are functions of the inputs; SGD because its update is
proportional to the gradient -- the weight-gradient kernels accumulate with fp32 atomics, and AdamW's sign-like first steps turn
that last-bit noise into lr-sized parameter differences), BatchNorm running statistics included; with dropout and AdamW on, the
graphed run stays finite, keeps drawing fresh masks and the loss decreases."""
import copy

import pytest
import torch

from semanticlidarunc_amd import salsanext as sn
from semanticlidarunc_amd.graph_step import GraphedTrainStep
from semanticlidarunc_amd.loss import salsanext_loss
from semanticlidarunc_amd.testing import seeded_model, synthetic_scan

pytestmark = pytest.mark.gpu


def _loss(out, y):
    return salsanext_loss(out, y, 1.0, 1.0, 0)[0]


def test_graphed_step_matches_eager(cuda):
    base = seeded_model(sn.SalsaNext).to(cuda).train()
    for m in base.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    batches = [tuple(t.to(cuda) for t in synthetic_scan(2, 32, 256, seed=60 + i)) for i in range(4)]

    def run_eager():
        model = copy.deepcopy(base)
        opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
        losses = []
        for x, y in batches[:1] * 2 + batches:        # the two warm-up steps GraphedTrainStep runs on the example batch, then the data
            opt.zero_grad(set_to_none=True)
            loss = _loss(model(x), y)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        return model, losses[2:]

    def distance(m1, m2):
        with torch.no_grad():
            return max(float((p - q).abs().max()) for p, q in zip(list(m1.parameters()) + [b for b in m1.buffers() if b.dtype.is_floating_point],
                                                                  list(m2.parameters()) + [b for b in m2.buffers() if b.dtype.is_floating_point]))

    eager_a, want = run_eager()
    eager_b, want_b = run_eager()                      # the weight-gradient kernels add with fp32 atomics: two eager runs already differ
    graphed = copy.deepcopy(base)
    opt_g = torch.optim.SGD(graphed.parameters(), lr=0.01, momentum=0.9)
    step = GraphedTrainStep(graphed, opt_g, _loss, batches[0][0], batches[0][1], warmup=2)
    got = [float(step(x, y)) for x, y in batches]
    noise_l = max(abs(a - b) for a, b in zip(want, want_b))
    assert got[0] == pytest.approx(want[0], rel=5e-5)
    assert max(abs(a - b) for a, b in zip(got, want)) <= 5.0 * noise_l + 2e-3 * max(want)
    assert distance(graphed, eager_a) <= 5.0 * distance(eager_b, eager_a) + 1e-4
    with pytest.raises(RuntimeError):
        step(batches[0][0][:1], batches[0][1][:1])                         # shape change
    with pytest.raises(RuntimeError):
        GraphedTrainStep(graphed, torch.optim.AdamW(graphed.parameters()), _loss, batches[0][0], batches[0][1])   # not capturable


def test_graphed_step_with_dropout_trains(cuda):
    model = seeded_model(sn.SalsaNext).to(cuda).train()
    x, y = (t.to(cuda) for t in synthetic_scan(2, 32, 256, seed=70))
    opt = torch.optim.AdamW(model.parameters(), lr=2e-3, capturable=True)
    step = GraphedTrainStep(model, opt, _loss, x, y)
    losses = [float(step(x, y)) for _ in range(12)]
    assert all(l == l and abs(l) < 1e4 for l in losses) and min(losses[-3:]) < losses[0]
    assert len({round(l, 6) for l in losses}) > 6                            # replays draw fresh dropout masks / keep updating
