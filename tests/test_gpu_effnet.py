"""GPU: EfficientNetV2 encoder of `semanticFCN_opt` (SURVEY 8(f-4) second half: the shipped YAML's `model_type: efficientnet_v2_l`) -- the new
kernels against torch, and the whole model against the fixtures the reference's OWN class produced through the torchvision stub
(tools/gen_golden_r03.py effnet).  Bar: 1e-3 of the logit scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden
from oracle import fpn_opt as ofpo
from semanticlidarunc_amd import ops, salsanext as sn
from semanticlidarunc_amd.fpn_opt import SemanticNetworkWithFPN
from semanticlidarunc_amd.ops import ConvSource
from semanticlidarunc_amd.testing import randomize_bn_

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_depthwise_se_and_silu_kernels(cuda):
    g = torch.Generator().manual_seed(1)
    for (n, c, h, w) in ((2, 24, 13, 40), (1, 384, 8, 32), (3, 7, 5, 9)):
        x, wt, b = torch.randn(n, c, h, w, generator=g), torch.randn(c, 1, 3, 3, generator=g) * 0.3, torch.randn(c, generator=g) * 0.1
        for stride in (1, 2):
            want = F.silu(F.conv2d(x, wt, b, stride=stride, padding=1, groups=c))
            got = ops.dwconv3x3(x.to(cuda), wt.reshape(c, 9).contiguous().to(cuda), b.to(cuda), stride, "silu").cpu()
            assert got.shape == want.shape and float((got - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))
            got = ops.dwconv3x3(x.to(cuda), wt.reshape(c, 9).contiguous().to(cuda), None, stride, "none").cpu()
            assert float((got - F.conv2d(x, wt, None, stride=stride, padding=1, groups=c)).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))
    x = torch.randn(3, 384, 6, 20, generator=g)
    w1, b1, w2, b2 = torch.randn(24, 384, generator=g) * 0.1, torch.randn(24, generator=g) * 0.1, torch.randn(384, 24, generator=g) * 0.3, torch.randn(384, generator=g)
    want = torch.sigmoid(F.linear(F.silu(F.linear(x.mean((2, 3)), w1, b1)), w2, b2))
    got = ops.se_scale(x.to(cuda), w1.to(cuda), b1.to(cuda), w2.to(cuda), b2.to(cuda)).cpu()
    assert float((got - want).abs().max()) <= 2e-6
    # SiLU in the fused conv's epilogue, with an SE-style per-(sample, channel) input multiplier and a residual
    xin, wt = torch.randn(2, 32, 8, 64, generator=g), torch.randn(48, 32, 1, 1, generator=g) / 6
    sc, r = torch.rand(2, 32, generator=g), torch.randn(2, 48, 8, 64, generator=g)
    for prec in ("fp32", "f16x3"):
        pack = ops.pack_conv_weight_f16x3(wt.to(cuda)) if prec == "f16x3" else ops.pack_conv_weight(wt.to(cuda))
        got = ops.conv2d_fused([ConvSource(xin.to(cuda), sc.to(cuda))], pack, 48, 1, 1, 0, resid=r.to(cuda), precision=prec, act="silu").cpu()
        assert float((got - (F.silu(F.conv2d(xin * sc.view(2, 32, 1, 1), wt)) + r)).abs().max()) <= 1e-4


@pytest.mark.parametrize("prec", ["fp32", "f16x3"])
@pytest.mark.parametrize("tag,kw", [
    ("efficientnet_v2_l_m3_c20", dict(backbone="efficientnet_v2_l", input_channels=2, meta_channel_dim=3, num_classes=20)),
    ("efficientnet_v2_s_m6_c5_noatt", dict(backbone="efficientnet_v2_s", input_channels=2, meta_channel_dim=6, num_classes=5, attention=False))])
def test_efficientnet_fpn_opt_matches_reference_golden(cuda, prec, tag, kw):
    g = golden("fpn_opt_" + tag)
    torch.manual_seed(0)
    model = randomize_bn_(SemanticNetworkWithFPN(**kw), 3).eval()
    with torch.no_grad():
        gg = torch.Generator().manual_seed(9)
        for mod in model.modules():
            if isinstance(mod, torch.nn.GroupNorm):
                mod.weight.copy_(torch.rand(mod.num_channels, generator=gg) + 0.5)
                mod.bias.copy_(torch.randn(mod.num_channels, generator=gg) * 0.1)
    sd = model.state_dict()
    s = sum(float(v.double().sum()) for v in sd.values() if v.is_floating_point())
    a = sum(float(v.double().abs().sum()) for v in sd.values() if v.is_floating_point())
    assert np.allclose([s, a], g["sd_digest"], rtol=1e-10)
    model.to(cuda)
    sn.set_conv_precision(prec)
    try:
        with torch.no_grad():
            y = model(_t(g["x"]).to(cuda), _t(g["meta"]).to(cuda)).cpu()
    finally:
        sn.set_conv_precision("fp32")
    want = _t(g["out"])
    assert y.shape == want.shape
    assert float((y - want).abs().max()) <= 1e-3 * max(1.0, float(want.abs().max()))


def test_efficientnet_fpn_opt_vs_oracle_at_another_size_and_dropout(cuda):
    kw = dict(backbone="efficientnet_v2_s", input_channels=2, meta_channel_dim=3, num_classes=20)
    torch.manual_seed(2)
    model = randomize_bn_(SemanticNetworkWithFPN(**kw), 5).eval()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 2, 64, 256, generator=g) * torch.tensor([20.0, 0.3]).view(1, 2, 1, 1)
    meta = torch.randn(2, 3, 64, 256, generator=g) * 5.0
    cpyr = model.decoder_semantic[0].in_channels
    scale = (torch.rand(2, cpyr, 1, 1, generator=g) > 0.1).float() / 0.9
    with torch.no_grad():
        want = ofpo.fpn_opt_forward(sd, x, meta, kw["backbone"], True, True, scale)
        model.to(cuda)
        got = model.forward_with_dropout_scale(x.to(cuda), meta.to(cuda), scale.to(cuda)).cpu()
    assert float((got - want).abs().max()) <= 1e-3 * max(1.0, float(want.abs().max()))
    # MC-dropout mode (eval + live Dropout2d) works on the folded inference path (the training path: tests/test_gpu_fpn_train.py)
    from semanticlidarunc_amd.utils.mc_dropout import set_dropout_mode
    set_dropout_mode(model, True)
    with torch.no_grad():
        a, b = model(x.to(cuda), meta.to(cuda)), model(x.to(cuda), meta.to(cuda))
    assert not torch.equal(a, b)
