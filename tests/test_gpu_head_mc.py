"""GPU: the fused segmentation head + MC-dropout reduction (csrc/head_mc_h8.hip) against the two-launch form it replaces -- the
1x1 head conv writing fp32 logits + slu_mc_reduce -- on the same fp16 features, and end to end through mc_predict with the same
dropout masks, against the CPU oracle reduction of the logits.  Same fp32 operations per pixel, so the bars are tight:
p_bar 2e-6, entropies 2e-5, argmax identical except exact-tie pixels."""
import pytest
import torch

from oracle import uncertainty as ounc
from semanticlidarunc_amd import h8, ops
from semanticlidarunc_amd import salsanext as sn
from semanticlidarunc_amd.testing import seeded_model, synthetic_scan
from semanticlidarunc_amd.utils.mc_dropout import mc_forward, mc_predict

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cin,classes,t,b,hh,ww", [(32, 20, 8, 2, 64, 512), (32, 20, 3, 1, 16, 64), (16, 7, 2, 3, 8, 32), (64, 32, 5, 2, 16, 128)])
def test_head_mc_matches_head_then_reduce(cuda, cin, classes, t, b, hh, ww):
    g = torch.Generator(device=cuda).manual_seed(cin + classes)
    feats = h8.to_h8(torch.randn(t * b, cin, hh, ww, device=cuda, generator=g) * 2.0)
    w = torch.randn(classes, cin, 1, 1, device=cuda, generator=g) * 0.4
    bias = torch.randn(classes, device=cuda, generator=g)
    wp = h8.pack_conv_weight_h8(w)
    logits = h8.conv2d_h8([h8.H8Source(feats)], wp, cin, classes, 1, 1, 0, bias=bias, out_f32_nchw=True)
    want = ops.mc_reduce(logits.reshape(t, b, classes, hh, ww).contiguous(), 1e-12)
    got = h8.head_mc_h8(feats, wp, bias, classes, t, b, 1e-12)
    assert float((got[0] - want[0]).abs().max()) <= 2e-6
    assert float((got[1] - want[1]).abs().max()) <= 2e-5 and float((got[2] - want[2]).abs().max()) <= 2e-5
    assert int((got[3] != want[3]).sum()) <= 2
    cpu = ounc.mc_reduce(logits.reshape(t, b, classes, hh, ww).cpu())
    assert float((got[0].cpu() - cpu[0]).abs().max()) <= 1e-5 and float((got[2].cpu() - cpu[2]).abs().max()) <= 1e-4
    with pytest.raises(RuntimeError):
        h8.head_mc_h8(feats, wp, bias, classes, t + 1, b)


def test_mc_predict_uses_the_fused_head_and_agrees(cuda):
    model = seeded_model(sn.SalsaNext).to(cuda)
    x, _ = synthetic_scan(2, 64, 512, seed=81)
    x = x.to(cuda)
    sn.set_conv_precision("f16")
    try:
        outs = {}
        for fused in (True, False):
            sn._FUSE_HEAD_MC = fused
            for share in (False, True):
                torch.manual_seed(5)
                outs[(fused, share)] = mc_predict(model, [x], T=4, share_prefix=share)
        sn._FUSE_HEAD_MC = True
        with torch.no_grad():
            assert model.mc_fused_ok(x, 4)
        for share in (False, True):
            a, bb = outs[(True, share)], outs[(False, share)]
            assert float((a[0] - bb[0]).abs().max()) <= 2e-6 and float((a[1] - bb[1]).abs().max()) <= 2e-5
            assert float((a[2] - bb[2]).abs().max()) <= 2e-5 and int((a[3] != bb[3]).sum()) <= 2
        torch.manual_seed(5)
        cpu = ounc.mc_reduce(mc_forward(model, [x], T=4).cpu())
        assert float((outs[(True, False)][0].cpu() - cpu[0]).abs().max()) <= 1e-5
        sn.set_conv_precision("fp32")                      # exact-fp32 inference keeps the two-launch form
        with torch.no_grad():
            assert not model.mc_fused_ok(x, 4)
    finally:
        sn._FUSE_HEAD_MC = True
        sn.set_conv_precision("fp32")
