"""GPU: fp16 storage at REAL-data magnitudes (BASELINE configs[2] / [4] name bf16 storage; the h8 path stores fp16 because bf16's 8-bit mantissa
misses the 1e-3 bar -- DESIGN 3.1e -- so fp16's 65504 ceiling has to be shown to be out of reach).

Input: SemanticKITTI-like magnitudes -- range up to 120 m with a heavy tail, xyz = range x direction, remission in [0, 1], 10 % empty returns.
BatchNorm statistics: ADAPTED to that data (one train-mode pass with momentum 1 sets running mean / var to the batch statistics, what a trained
network's normalisation looks like), and separately the suite's randomised statistics with the input scaled by another 4x.
Checks: (i) every tensor the fp32 oracle stores between layers stays below 65504 / 16 (4 bits of head-room; measured peak: 467); (ii) the fp16
path's logits have no inf / nan and stay within per-case bars of the fp32 oracle (BARS below, with the measured values: 1e-3 holds on the suite's
statistics, NOT where the folded BatchNorm gains reach 89), while conv precision 'f16x3' (fp32 storage, split-fp16 products) stays inside 1e-3 in
every case; (iii) an input blown up until the oracle DOES cross 65504 makes the fp16 path fail loudly (non-finite logits): the check has teeth."""
import pytest
import torch
import torch.nn.functional as F

from oracle import salsanext as osalsa
from semanticlidarunc_amd import salsanext as sn
from semanticlidarunc_amd.salsanext import SalsaNext
from semanticlidarunc_amd.testing import seeded_model

pytestmark = pytest.mark.gpu


def kitti_like_scan(batch, h, w, seed, max_range=120.0):
    g = torch.Generator().manual_seed(seed)
    rng = torch.exp(torch.randn(batch, 1, h, w, generator=g) * 0.9 + 2.3).clamp(0.5, max_range)        # median 10 m, tail to max_range
    az = torch.linspace(-3.1416, 3.1416, w).view(1, 1, 1, w).expand(batch, 1, h, w)
    el = torch.linspace(0.05, -0.43, h).view(1, 1, h, 1).expand(batch, 1, h, w)
    xyz = torch.cat([rng * torch.cos(el) * torch.cos(az), rng * torch.cos(el) * torch.sin(az), rng * torch.sin(el)], 1)
    refl = torch.rand(batch, 1, h, w, generator=g)
    x = torch.cat([rng, refl, xyz], 1)
    return x.masked_fill(torch.rand(batch, 1, h, w, generator=g) < 0.10, 0.0).contiguous()


def oracle_with_stored_maxima(sd, x):
    """(logits, largest |value| of any tensor a layer stores): conv + activation + BatchNorm outputs of the CPU oracle"""
    peak = [0.0]
    real_bn, real_act = F.batch_norm, F.leaky_relu

    def bn(*a, **k):
        y = real_bn(*a, **k)
        peak[0] = max(peak[0], float(y.abs().max()))
        return y

    def act(*a, **k):
        y = real_act(*a, **k)
        peak[0] = max(peak[0], float(y.abs().max()))
        return y

    osalsa.F.batch_norm, osalsa.F.leaky_relu = bn, act
    try:
        with torch.no_grad():
            y = osalsa.salsanext_forward(sd, x)
    finally:
        osalsa.F.batch_norm, osalsa.F.leaky_relu = real_bn, real_act
    return y, peak[0]


def adapt_bn_(model, x):
    """running statistics := the batch statistics of x (momentum 1, one train-mode pass on the exact-fp32 HIP path)"""
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    old = [m.momentum for m in bns]
    for m in bns:
        m.momentum = 1.0
    model.train()
    with torch.no_grad():
        model(x)
    model.eval()
    for m, mo in zip(bns, old):
        m.momentum = mo
    return model


# case -> (logit bar as a fraction of max(1, logit scale), bar on softmax probabilities and normalised entropy, argmax-flip bar) for fp16 storage;
# measured on MI355X (tools/debug_range.py): random_bn 4.7e-4 / 2.4e-5 / 1.8e-3; random_bn_x4 2.2e-3 (1.3e-3 of the scale) / 2.2e-4 / 7.6e-4;
# adapted_bn 3.7e-2 (5.8e-3 of the scale 6.3) / 5.2e-3 / 7.7e-3 -- there the folded BatchNorm gains gamma / sigma reach 89 (randomly initialised
# convs leave some channels almost constant), and every fp16 rounding of a stored activation is multiplied by them
BARS = {"random_bn": (1e-3, 1e-3, 5e-3), "random_bn_x4": (3e-3, 1e-3, 5e-3), "adapted_bn": (1.2e-2, 1e-2, 1.5e-2)}


@pytest.mark.parametrize("case", ["random_bn", "random_bn_x4", "adapted_bn"])
def test_fp16_storage_is_far_from_saturation_at_real_magnitudes(cuda, case):
    model = seeded_model(SalsaNext).to(cuda)
    x = kitti_like_scan(2, 64, 512, seed=21)
    if case == "adapted_bn":
        adapt_bn_(model, kitti_like_scan(2, 64, 512, seed=22).to(cuda))
    elif case == "random_bn_x4":
        x = x * 4.0                                           # 480 m "ranges": 4x beyond the sensor, on the suite's randomised statistics
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    want, peak = oracle_with_stored_maxima(sd, x)
    assert peak < 65504.0 / 16.0, f"largest stored activation {peak:.0f}"
    got = {}
    for prec in ("f16", "f16x3"):
        sn.set_conv_precision(prec)
        try:
            with torch.no_grad():
                got[prec] = model(x.to(cuda)).cpu()
        finally:
            sn.set_conv_precision("fp32")
    assert bool(torch.isfinite(got["f16"]).all())
    ent = lambda p: -(p * torch.log(p.clamp_min(1e-8))).sum(1) / torch.log(torch.tensor(20.0))
    pw = torch.softmax(want, 1)
    # fp32 storage with split-fp16 products: inside the north star's 1e-3 in every case (measured <= 2.7e-4)
    assert float((got["f16x3"] - want).abs().max()) <= 1e-3
    # fp16 storage: PRECISION (11-bit mantissa x the BatchNorm gains), not range, is what it costs -- bars per case above
    lb, pb, fb = BARS[case]
    pg = torch.softmax(got["f16"], 1)
    assert float((got["f16"] - want).abs().max()) <= lb * max(1.0, float(want.abs().max())), (float((got["f16"] - want).abs().max()), peak)
    assert float((pg - pw).abs().max()) <= pb and float((ent(pg) - ent(pw)).abs().max()) <= pb
    assert float((got["f16"].argmax(1) != want.argmax(1)).float().mean()) <= fb


def test_the_saturation_check_has_teeth(cuda):
    model = seeded_model(SalsaNext).to(cuda)
    x = kitti_like_scan(1, 32, 256, seed=23) * 3000.0        # 360 km: absurd on purpose
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    _, peak = oracle_with_stored_maxima(sd, x)
    assert peak > 65504.0
    sn.set_conv_precision("f16")
    try:
        with torch.no_grad():
            got = model(x.to(cuda)).cpu()
    finally:
        sn.set_conv_precision("fp32")
    assert not bool(torch.isfinite(got).all())               # fp16 overflow is visible as inf / nan in the logits, never silently clipped
