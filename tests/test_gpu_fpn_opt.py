"""GPU: the `semanticFCN_opt` variant (baselines/Reichert/semanticFCN_opt.py) -- bilinear UpsampleBlock + GroupNorm,
SpatialAttention (softmax over H*W), Dropout2d on the pyramid, GroupNorm decoder, raw logits -- against the golden the reference's own
class produced (through the torchvision stub) and against the oracle at a larger size.  Bar: 1e-3 abs on the logits."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden
from oracle import fpn_opt as ofpo
from semanticlidarunc_amd import ops, salsanext as sn
from semanticlidarunc_amd.fpn_opt import SemanticNetworkWithFPN
from semanticlidarunc_amd.testing import randomize_bn_
from semanticlidarunc_amd.utils.mc_dropout import mc_forward, set_dropout_mode

pytestmark = pytest.mark.gpu
CASES = {"resnet18_m6_c20": dict(backbone="resnet18", input_channels=2, meta_channel_dim=6, num_classes=20),
         "resnet34_m3_c21_noatt": dict(backbone="resnet34", input_channels=2, meta_channel_dim=3, num_classes=21, attention=False,
                                       multi_scale_meta=False),
         "resnet50_m3_c5": dict(backbone="resnet50", input_channels=2, meta_channel_dim=3, num_classes=5)}      # Bottleneck backbone


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _model(kw, cuda):
    torch.manual_seed(0)
    m = randomize_bn_(SemanticNetworkWithFPN(**kw), 3).eval()
    with torch.no_grad():
        g = torch.Generator().manual_seed(9)
        for mod in m.modules():
            if isinstance(mod, torch.nn.GroupNorm):
                mod.weight.copy_(torch.rand(mod.num_channels, generator=g) + 0.5)
                mod.bias.copy_(torch.randn(mod.num_channels, generator=g) * 0.1)
    return m.to(cuda)


def test_new_kernels_against_torch(cuda):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 5, 6, 20, generator=g)
    for s in (2, 4, 8):
        want = F.interpolate(x, scale_factor=s, mode="bilinear", align_corners=False)
        assert float((ops.bilinear_upsample(x.to(cuda), s).cpu() - want).abs().max()) <= 1e-6
    y = torch.randn(3, 32, 9, 40, generator=g) * 3 + 1
    for groups in (8, 32, 4):
        gam, bet = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g)
        want = F.relu(F.group_norm(y, groups, gam, bet, 1e-5))
        got = ops.groupnorm(y.to(cuda), groups, gam.to(cuda), bet.to(cuda), 1e-5, relu=True).cpu()
        assert float((got - want).abs().max()) <= 2e-5
    sc, v = torch.randn(2, 1, 12, 300, generator=g) * 4, torch.randn(2, 7, 12, 300, generator=g)
    w = torch.softmax(sc.view(2, 1, -1), -1).view(2, 1, 12, 300)
    assert float((ops.spatial_softmax_gate(v.to(cuda), sc.to(cuda)).cpu() - (v * w + v)).abs().max()) <= 1e-6


@pytest.mark.parametrize("tag", list(CASES))
@pytest.mark.parametrize("prec", ["fp32", "f16x3"])
def test_against_the_reference_golden(cuda, tag, prec):
    g = golden("fpn_opt_" + tag)
    m = _model(CASES[tag], cuda)
    sn.set_conv_precision(prec)
    try:
        with torch.no_grad():
            y = m(_t(g["x"]).to(cuda), _t(g["meta"]).to(cuda)).cpu()
            yd = m.forward_with_dropout_scale(_t(g["x"]).to(cuda), _t(g["meta"]).to(cuda), _t(g["dropout_scale"]).to(cuda)).cpu()
    finally:
        sn.set_conv_precision("fp32")
    assert y.shape == g["out"].shape
    assert float((y - _t(g["out"])).abs().max()) <= 1e-3 and float((yd - _t(g["out_dropout"])).abs().max()) <= 1e-3
    assert float((_t(g["out"]) - _t(g["out_dropout"])).abs().max()) > 1e-2        # the multipliers matter


def test_larger_scan_against_the_oracle_and_mc_dropout(cuda):
    kw = CASES["resnet18_m6_c20"]
    m = _model(kw, cuda)
    g = torch.Generator().manual_seed(13)
    x = torch.randn(2, 2, 64, 512, generator=g) * torch.tensor([20.0, 0.3]).view(1, 2, 1, 1)
    meta = torch.randn(2, 6, 64, 512, generator=g) * 5.0
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        want = ofpo.fpn_opt_forward(sd, x, meta, "resnet18")
        got = m(x.to(cuda), meta.to(cuda)).cpu()
    assert float((got - want).abs().max()) <= 1e-3 and float((got.argmax(1) != want.argmax(1)).float().mean()) < 2e-3
    # utils.mc_dropout toggles the genuine nn.Dropout2d child: passes differ, BatchNorm stays frozen, eval is restored
    torch.manual_seed(5)
    mc = mc_forward(m, [x.to(cuda), meta.to(cuda)], T=4)
    assert mc.shape == (4, 2, 20, 64, 512) and float((mc[0] - mc[1]).abs().max()) > 1e-3
    assert not m.dropout_pyramid.training
    with torch.no_grad():
        assert float((m(x.to(cuda), meta.to(cuda)).cpu() - got).abs().max()) == 0.0
    set_dropout_mode(m, False)


def test_contract_errors(cuda):
    m = SemanticNetworkWithFPN("resnet18", 2, 3, num_classes=5).to(cuda)
    with pytest.raises(ValueError):
        SemanticNetworkWithFPN("vgg")
    with pytest.raises(NotImplementedError):
        SemanticNetworkWithFPN("regnet_y_400mf")
    with pytest.raises(RuntimeError):
        m.eval()(torch.zeros(1, 2, 16, 64), torch.zeros(1, 3, 16, 64))            # CPU tensors
