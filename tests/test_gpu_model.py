"""GPU: the drop-in SalsaNext module against the reference's logits (golden fixtures produced by
the reference itself) and against the oracle at the BASELINE size.  Bar: <= 1e-3 abs (north star)."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import salsanext as osalsa
from semanticlidarunc_amd.salsanext import SalsaNext
from semanticlidarunc_amd.testing import seeded_model, synthetic_scan
from semanticlidarunc_amd.utils.mc_dropout import mc_forward, set_dropout_mode

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def model(cuda):
    return seeded_model(SalsaNext).to(cuda)


def test_eval_forward_matches_reference_golden(cuda, model):
    g = golden("salsanext_eval_1x5x16x64")
    with torch.no_grad():
        y = model(_t(g["x"]).to(cuda)).cpu()
    err = float((y - _t(g["logits"])).abs().max())
    assert y.shape == (1, 20, 16, 64) and err <= TOL, err


def test_dropout_multipliers_match_reference_golden(cuda, model):
    g = golden("salsanext_mc_2x5x32x64")
    scales = {k[len("scale:"):]: _t(g[k]) for k in g.files if k.startswith("scale:")}
    with torch.no_grad():
        y = model.forward_with_dropout_scales(_t(g["x"]).to(cuda), scales).cpu()
    err = float((y - _t(g["logits"])).abs().max())
    assert err <= TOL, err


def test_full_size_scan_matches_oracle(cuda, model):
    # BASELINE config 1 shape: one 64x2048x5 scan, HIP vs the oracle (CPU, ~1 s)
    x, _ = synthetic_scan(1, 64, 2048)
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        want = osalsa.salsanext_forward(sd, x)
        got = model(x.to(cuda)).cpu()
    err = float((got - want).abs().max())
    assert err <= TOL, err
    assert float((got.argmax(1) != want.argmax(1)).float().mean()) < 1e-3


def test_batch_independence_and_determinism(cuda, model):
    x, _ = synthetic_scan(3, 32, 128, seed=7)
    x = x.to(cuda)
    with torch.no_grad():
        y = model(x)
        y1 = model(x[1:2].contiguous())
        y_again = model(x)
    assert torch.equal(y, y_again)
    assert float((y[1:2] - y1).abs().max()) <= 1e-5


def test_mc_forward_draws_real_dropout_and_restores_modes(cuda, model):
    x, _ = synthetic_scan(1, 32, 128, seed=8)
    x = x.to(cuda)
    torch.manual_seed(3)
    out = mc_forward(model, [x], T=6)
    assert out.shape == (6, 1, 20, 32, 128)
    assert float(out.std(dim=0).mean()) > 1e-3             # passes differ
    assert not any(m.training for m in model.modules())    # everything back in eval
    with torch.no_grad():
        e = model(x)
    assert float((out.mean(0) - e).abs().mean()) < float(e.abs().mean())   # same scale as the eval output
    # statistical parity of the folded masks: keep-rate of one site ~ 0.8, values in {0, 1.25}
    set_dropout_mode(model, True)
    from semanticlidarunc_amd.salsanext import _draw
    s = _draw(model.resBlock3.dropout, 64, 256, cuda, None, "")
    set_dropout_mode(model, False)
    vals = torch.unique(s).cpu().tolist()
    assert vals == [0.0, 1.25] and abs(float((s > 0).float().mean()) - 0.8) < 0.02


def test_weight_update_invalidates_packed_cache(cuda):
    m = seeded_model(SalsaNext).to(cuda)
    x, _ = synthetic_scan(1, 16, 64, seed=9)
    x = x.to(cuda)
    with torch.no_grad():
        y0 = m(x)
        m.logits.weight.mul_(2.0)
        m.logits.bias.mul_(2.0)
        y1 = m(x)
    assert float((y1 - 2.0 * y0).abs().max()) <= 1e-4
    m.train()                                    # train-mode BatchNorm: batch statistics, running stats move
    before = m.downCntx.bn1.running_mean.clone()
    with torch.no_grad():
        m(x)
    assert not torch.equal(before, m.downCntx.bn1.running_mean)


def test_shared_prefix_mc_forward_equals_full_passes(cuda, model):
    """forward_mc (deterministic prefix computed once) == T stacked full forwards with the same multipliers,
    in both conv precisions; and through mc_forward / mc_predict."""
    from oracle import salsanext as osalsa_
    from semanticlidarunc_amd import salsanext as sn
    from semanticlidarunc_amd.utils.mc_dropout import mc_predict
    x, _ = synthetic_scan(2, 32, 128, seed=17)
    x = x.to(cuda)
    T = 3
    scales = osalsa_.draw_dropout_scales(T * 2, 0.2, torch.Generator().manual_seed(11))
    for prec in ("fp32", "f16x3"):
        sn.set_conv_precision(prec)
        try:
            with torch.no_grad():
                full = model.forward_with_dropout_scales(x.repeat(T, 1, 1, 1), scales)
                shared = model.forward_mc(x, T, scales)
        finally:
            sn.set_conv_precision("fp32")
        assert shared.shape == full.shape
        assert float((shared - full).abs().max()) <= 1e-6, prec
    torch.manual_seed(5)
    a = mc_forward(model, [x], T=4, share_prefix=True)
    torch.manual_seed(5)
    b = mc_forward(model, [x], T=4, share_prefix=False)
    assert a.shape == b.shape == (4, 2, 20, 32, 128)
    # same seed -> the same Dropout2d draws in the same order -> the two schedules agree to the last bit
    assert float((a - b).abs().max()) <= 1e-6 and float(a.std(0).mean()) > 1e-3
    p_bar, h, mi, preds = mc_predict(model, [x], T=4, share_prefix=True)
    assert p_bar.shape == (2, 20, 32, 128) and float((p_bar.sum(1) - 1).abs().max()) < 1e-5
    model.train()
    with pytest.raises(RuntimeError):
        model.forward_mc(x, 2)
    model.eval()


@pytest.mark.parametrize("precision", ["fp32", "f16"])
def test_eval_after_running_stat_update_with_frozen_affine_matches_oracle(cuda, precision):
    """BatchNorm re-calibration: train-mode forwards under no_grad change ONLY the running statistics (no optimizer step bumps a
    parameter version).  The next eval forward must fold the NEW statistics (the folded-BN cache is keyed on tensor versions, which
    the raw-pointer update inside slu_bn_coeffs_fwd now bumps), i.e. agree with the oracle evaluated on the updated state_dict."""
    from semanticlidarunc_amd import salsanext as sn
    m = seeded_model(SalsaNext).to(cuda)
    x, _ = synthetic_scan(2, 32, 128, seed=21)
    xg = x.to(cuda)
    sn.set_conv_precision(precision)
    try:
        with torch.no_grad():
            before = m(xg).cpu()                          # fills the folded-BN caches
            m.train()
            set_dropout_mode(m, False)                    # statistics only: keep the forward deterministic
            m(xg)
            m.eval()
            got = m(xg).cpu()
            sd = {k: v.cpu() for k, v in m.state_dict().items()}
            want = osalsa.salsanext_forward(sd, x)
    finally:
        sn.set_conv_precision("fp32")
    assert float((before - want).abs().max()) > 10 * TOL          # the statistics really moved
    assert float((got - want).abs().max()) <= TOL
    # distributed.average_buffers / broadcast_parameters write in place on the tensors themselves; same invalidation path
    with torch.no_grad():
        for b in m.buffers():
            if b.is_floating_point():
                b.mul_(1.0)
        assert float((m(xg).cpu() - want).abs().max()) <= TOL


def test_one_launch_dropout_draw(cuda, model):
    """csrc/dropout_draw.hip: every multiplier of an MC evaluation from one launch -- Bernoulli(1 - p) / (1 - p) per (sample, channel), products
    composed like UpBlock composes them, reproducible from torch's generator state and advancing it."""
    from semanticlidarunc_amd import ops
    n = 64
    plan = ops.DropoutPlan(n, [(8, 0.2, True), (2, 0.2, True), (5, 0.2, True), (6, 0.5, False)],
                           [("A", 8, [(0, 0)], False), ("B", 2, [(1, 0)], False), ("D2", 5, [(2, 0)], False),
                            ("sx", 8, [(0, 0), (1, 0), (2, 0)], True), ("ss", 3, [(2, 2)], False), ("off", 6, [(3, 0)], False),
                            ("sx_noprod", 8, [(-1, 0), (1, 0), (2, 0)], True)], cuda)
    torch.manual_seed(11)
    a = plan.run()
    assert torch.equal(a["sx"], a["A"] * (a["B"] * a["D2"][:, :2]).repeat_interleave(4, dim=1))
    assert torch.equal(a["sx_noprod"], (a["B"] * a["D2"][:, :2]).repeat_interleave(4, dim=1))
    assert torch.equal(a["ss"], a["D2"][:, 2:]) and torch.equal(a["off"], torch.ones_like(a["off"]))      # an inactive site multiplies by 1
    b = plan.run()
    assert not torch.equal(a["A"], b["A"])                                                                 # the generator moved on
    torch.manual_seed(11)
    c = plan.run()
    assert all(torch.equal(a[k], c[k]) for k in a)                                                         # and is reproducible
    # the model's own plan: values, keep rate, structure of a composed product
    set_dropout_mode(model, True)
    try:
        torch.manual_seed(5)
        s = model._predraw_dropout(512, cuda)
    finally:
        set_dropout_mode(model, False)
    m = s["resBlock3.dropout"]
    assert m.shape == (512, 256) and torch.unique(m).cpu().tolist() == [0.0, 1.25] and abs(float((m > 0).float().mean()) - 0.8) < 0.01
    means = [float((s[k] > 0).float().mean()) for k in ("resBlock2.dropout", "resBlock4.dropout", "resBlock5.dropout", "upBlock1.dropout3", "upBlock3.dropout3")]
    assert all(abs(v - 0.8) < 0.01 for v in means), means
    sx, prod = s["upBlock1._sx"], s["resBlock5.dropout"]
    assert sx.shape == (512, 256) and bool(((prod == 0) <= (sx == 0)).all())                                # the producer's zeros survive
    assert abs(float((sx > 0).float().mean()) - 0.8 ** 3) < 0.01                                            # three independent sites
    assert torch.equal(sx[:, 0::4] > 0, sx[:, 0::4] > 0) and s["upBlock1._ss"].shape == (512, 256) and s["upBlock4._sx"].shape == (512, 64)
    vals = sorted(set(torch.unique(sx).cpu().tolist()))
    assert vals == [0.0, 1.25 ** 3]
    # channels of different sites / samples are not correlated through the shared counter stream
    x = (s["resBlock3.dropout"] > 0).float()
    assert abs(float(((x[:, :-1] * x[:, 1:]).mean() - 0.64))) < 0.01 and abs(float((x[:-1] * x[1:]).mean() - 0.64)) < 0.01
