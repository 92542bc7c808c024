"""GPU: the split-fp16 conv path (three f16 MFMAs per K-step, fp32 accumulate) against the fp32 oracle and the
exact-fp32 kernel.  Same bars as the exact path: single convs 1e-4 abs on O(1) outputs, whole network 1e-3."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import salsanext as osalsa
from semanticlidarunc_amd import ops, salsanext as sn
from semanticlidarunc_amd.ops import ConvSource
from semanticlidarunc_amd.testing import seeded_model, synthetic_scan

pytestmark = pytest.mark.gpu
FAMILIES = [(1, 1, 0), (3, 1, 1), (3, 2, 2), (2, 2, 1)]


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _run(dev, n, cin_parts, cout, h, w, fam, seed, scales=False, ps_first=False, big=False):
    k, dil, pad = fam
    g = torch.Generator().manual_seed(seed)
    srcs, cin = [], 0
    for i, c in enumerate(cin_parts):
        ps = ps_first and i == 0
        t = torch.randn(n, c, h // 2 if ps else h, w // 2 if ps else w, generator=g) * (30.0 if big else 1.0)
        s = (torch.rand(n, c, generator=g) > 0.2).float() * 1.25 if scales else None
        srcs.append((t, s, ps))
        cin += c // 4 if ps else c
    wgt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias, bn_a, bn_b = torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    res = torch.randn(n, cout, h, w, generator=g)
    want = osalsa.fused_conv(srcs, wgt, bias, pad, dil, 0.01, bn_a, bn_b, res)
    d = lambda t: None if t is None else t.to(dev).contiguous()
    dsrc = [ConvSource(d(t), d(s), ps) for t, s, ps in srcs]
    got = ops.conv2d_fused(dsrc, ops.pack_conv_weight_f16x3(d(wgt)), cout, k, dil, pad, bias=d(bias), slope=0.01, bn_a=d(bn_a),
                           bn_b=d(bn_b), resid=d(res), precision="f16x3")
    exact = ops.conv2d_fused(dsrc, ops.pack_conv_weight(d(wgt)), cout, k, dil, pad, bias=d(bias), slope=0.01, bn_a=d(bn_a),
                             bn_b=d(bn_b), resid=d(res))
    scale = max(1.0, float(want.abs().max()))
    assert float((got.cpu() - want).abs().max()) <= 1e-4 * scale, (fam, cin_parts, cout, h, w)
    assert float((got - exact).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize("fam", FAMILIES)
@pytest.mark.parametrize("cout,hw", [(32, (16, 128)), (64, (8, 64)), (128, (8, 64)), (256, (4, 64)), (20, (16, 64))])
def test_family_by_channel_tile(cuda, fam, cout, hw):
    _run(cuda, 2, [32], cout, hw[0], hw[1], fam, seed=cout + fam[0] * 7 + fam[1])


@pytest.mark.parametrize("fam", FAMILIES)
def test_ragged_sizes_odd_channels_and_large_values(cuda, fam):
    _run(cuda, 1, [5], 32, 13, 70, fam, seed=3)
    _run(cuda, 3, [21], 40, 5, 33, fam, seed=4)
    _run(cuda, 1, [7], 70, 9, 132, fam, seed=5, big=True)


def test_big_tiles_concat_and_pixel_shuffle(cuda):
    _run(cuda, 1, [32], 32, 64, 1024, (3, 2, 2), seed=6)
    _run(cuda, 1, [32], 64, 64, 1024, (3, 1, 1), seed=7)
    _run(cuda, 4, [64], 128, 32, 512, (2, 2, 1), seed=8)
    _run(cuda, 2, [64, 64, 64], 64, 64, 512, (1, 1, 0), seed=9)
    _run(cuda, 2, [32, 16, 48], 32, 8, 64, (3, 1, 1), seed=11, scales=True)
    _run(cuda, 2, [256, 256], 128, 8, 64, (3, 1, 1), seed=12, scales=True, ps_first=True)
    _run(cuda, 2, [64, 64], 32, 64, 256, (3, 1, 1), seed=13, scales=True, ps_first=True)


def test_whole_network_in_f16x3(cuda):
    model = seeded_model(sn.SalsaNext).to(cuda)
    sn.set_conv_precision("f16x3")
    try:
        g = golden("salsanext_eval_1x5x16x64")
        with torch.no_grad():
            y = model(_t(g["x"]).to(cuda)).cpu()
        assert float((y - _t(g["logits"])).abs().max()) <= 1e-3
        g = golden("salsanext_mc_2x5x32x64")
        scales = {k[len("scale:"):]: _t(g[k]) for k in g.files if k.startswith("scale:")}
        with torch.no_grad():
            y = model.forward_with_dropout_scales(_t(g["x"]).to(cuda), scales).cpu()
        assert float((y - _t(g["logits"])).abs().max()) <= 1e-3
        x, _ = synthetic_scan(1, 64, 2048)
        sd = {k: v.cpu() for k, v in model.state_dict().items()}
        with torch.no_grad():
            want = osalsa.salsanext_forward(sd, x)
            got = model(x.to(cuda)).cpu()
            sn.set_conv_precision("fp32")
            exact = model(x.to(cuda)).cpu()
        err, err_exact = float((got - want).abs().max()), float((exact - want).abs().max())
        assert err <= 1e-3, (err, err_exact)
        assert float((got.argmax(1) != want.argmax(1)).float().mean()) < 1e-3
    finally:
        sn.set_conv_precision("fp32")
    with pytest.raises(ValueError):
        sn.set_conv_precision("bf16")
