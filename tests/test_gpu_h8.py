"""GPU: the half-precision storage path (fp16 activations in channel blocks of 8, one f16 MFMA per K-step, fp32
accumulate) -- BASELINE.json configs[2],[4].

Bars.  Single kernels: against the fp32 oracle fed the SAME fp16-rounded operands, the only differences are the fp32
summation order and the final rounding of the result to fp16, so |err| <= 2^-10 |y| + 1e-4 (one fp16 ulp is 2^-11
relative).  Data-movement kernels: exact.  Whole network: logits / entropy within the 1e-3 the north star states, against
the fp32 oracle and the committed golden vectors of the reference."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden
from oracle import salsanext as osalsa
from oracle import uncertainty as ounc
from semanticlidarunc_amd import h8, ops, salsanext as sn
from semanticlidarunc_amd.h8 import H8Source
from semanticlidarunc_amd.testing import seeded_model, synthetic_scan

pytestmark = pytest.mark.gpu
FAMILIES = [(1, 1, 0), (3, 1, 1), (3, 2, 2), (2, 2, 1)]


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _r16(t):
    return t.half().float()


def test_layout_round_trip_and_padding(cuda):
    g = torch.Generator().manual_seed(0)
    for n, c, h, w in [(2, 5, 7, 33), (1, 8, 4, 64), (3, 20, 5, 16), (1, 37, 3, 9)]:
        x = _r16(torch.randn(n, c, h, w, generator=g) * 10)
        y = h8.to_h8(x.to(cuda))
        assert y.shape == (n, (c + 7) // 8, h, w, 8) and y.dtype == torch.float16
        assert torch.equal(h8.from_h8(y, c).cpu(), x)
        full = h8.from_h8(y).cpu()                      # pad channels read back as zeros
        assert torch.equal(full[:, :c], x) and float(full[:, c:].abs().sum()) == 0.0
        # layout definition: y[n, g, h, w, k] == x[n, 8 g + k, h, w]
        ref = torch.zeros(n, 8 * ((c + 7) // 8), h, w)
        ref[:, :c] = x
        assert torch.equal(y.cpu().float(), ref.view(n, -1, 8, h, w).permute(0, 1, 3, 4, 2))
        s = (torch.rand(n, c, generator=g) > 0.3).float() * 1.25
        assert torch.equal(h8.from_h8(h8.to_h8(x.to(cuda), s.to(cuda)), c).cpu(), _r16(x * s[:, :, None, None]))
    with pytest.raises(RuntimeError):
        h8.to_h8(torch.zeros(1, 5, 4, 4))               # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        h8.from_h8(torch.zeros(1, 1, 4, 4, 8, device=cuda))   # wrong dtype


def _conv_case(dev, n, parts, cout, h, w, fam, seed, scales=False, resid=True, out_f32=False, nbatch_last=0, big=False):
    """parts: real channel counts per source (multiples of 8 except the last)."""
    k, dil, pad = fam
    g = torch.Generator().manual_seed(seed)
    srcs, osrcs, cin = [], [], 0
    for i, c in enumerate(parts):
        nimg = nbatch_last if (nbatch_last and i == len(parts) - 1 and i > 0) else n
        t = _r16(torch.randn(nimg, c, h, w, generator=g) * (30.0 if big else 1.0))
        s = (torch.rand(n, 8 * ((c + 7) // 8), generator=g) > 0.2).float() * 1.25 if scales else None
        srcs.append(H8Source(h8.to_h8(t.to(dev)), None if s is None else s.to(dev), nimg if nimg != n else 0))
        tt = t if nimg == n else t.repeat(n // nimg, 1, 1, 1)
        # the kernel multiplies fp16 * fp16(scale) and rounds to fp16: 1.25 and 0 are exact, so is the product rounding
        osrcs.append((_r16(tt * s[:, :c, None, None]) if s is not None else tt, None, False))
        cin += c
    wgt = _r16(torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5)
    bias, bn_a, bn_b = torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    res = _r16(torch.randn(n, cout, h, w, generator=g)) if (resid and not out_f32) else None
    want = osalsa.fused_conv(osrcs, wgt, bias, pad, dil, 0.01, bn_a, bn_b, res)
    d = lambda t: None if t is None else t.to(dev).contiguous()
    got = h8.conv2d_h8(srcs, h8.pack_conv_weight_h8(d(wgt)), cin, cout, k, dil, pad, bias=d(bias), slope=0.01, bn_a=d(bn_a), bn_b=d(bn_b),
                       resid=None if res is None else h8.to_h8(d(res)), out_f32_nchw=out_f32)
    if out_f32:
        assert got.dtype == torch.float32 and got.shape == (n, cout, h, w)
        y = got.cpu()
        tol = 2e-5 * max(1.0, float(want.abs().max()))
        assert float((y - want).abs().max()) <= tol, (fam, parts, cout, h, w)
    else:
        assert got.shape == (n, (cout + 7) // 8, h, w, 8)
        y = h8.from_h8(got, cout).cpu()
        err = (y - want).abs()
        assert bool((err <= 2.0 ** -10 * want.abs() + 1e-4 * max(1.0, float(want.abs().max()) / 30)).all()), (fam, parts, cout, h, w, float(err.max()))
        pad_part = h8.from_h8(got).cpu()[:, cout:]
        assert float(pad_part.abs().sum()) == 0.0       # pad channels stay zero for the next layer


@pytest.mark.parametrize("fam", FAMILIES)
@pytest.mark.parametrize("cout,hw", [(32, (16, 128)), (64, (8, 64)), (128, (8, 64)), (256, (4, 64)), (20, (16, 64))])
def test_family_by_channel_tile(cuda, fam, cout, hw):
    _conv_case(cuda, 2, [32], cout, hw[0], hw[1], fam, seed=cout + fam[0] * 7 + fam[1])


@pytest.mark.parametrize("fam", FAMILIES)
def test_ragged_sizes_odd_channels_and_large_values(cuda, fam):
    _conv_case(cuda, 1, [5], 32, 13, 70, fam, seed=3)
    _conv_case(cuda, 3, [21], 40, 5, 33, fam, seed=4)
    _conv_case(cuda, 1, [7], 70, 9, 132, fam, seed=5, big=True)
    _conv_case(cuda, 2, [8, 13], 24, 6, 40, fam, seed=6)           # odd number of blocks: the last K-step is half empty


def test_concat_scales_broadcast_and_fp32_head(cuda):
    _conv_case(cuda, 1, [32], 32, 64, 1024, (3, 2, 2), seed=6)
    _conv_case(cuda, 1, [32], 64, 64, 1024, (3, 1, 1), seed=7)
    _conv_case(cuda, 4, [64], 128, 32, 512, (2, 2, 1), seed=8)
    _conv_case(cuda, 2, [64, 64, 64], 64, 64, 512, (1, 1, 0), seed=9)
    _conv_case(cuda, 2, [128, 128, 128], 128, 16, 256, (1, 1, 0), seed=10, resid=False)
    _conv_case(cuda, 2, [32, 16, 48], 32, 8, 64, (3, 1, 1), seed=11, scales=True)
    _conv_case(cuda, 2, [64, 256], 128, 8, 64, (3, 1, 1), seed=12, scales=True)            # upBlock1.conv1 shape
    _conv_case(cuda, 4, [16, 64], 32, 64, 256, (3, 1, 1), seed=13, scales=True, nbatch_last=2)   # upBlock4.conv1, shared skip
    _conv_case(cuda, 4, [32, 32], 32, 16, 64, (1, 1, 0), seed=14, nbatch_last=1)           # streaming kernel + broadcast
    _conv_case(cuda, 2, [32], 20, 64, 512, (1, 1, 0), seed=15, out_f32=True)               # logits head
    _conv_case(cuda, 2, [5], 32, 64, 512, (1, 1, 0), seed=16, resid=False)                 # first layer (5 real channels)
    _conv_case(cuda, 1, [32], 20, 5, 33, (1, 1, 0), seed=17, out_f32=True)                # odd H*W: tiled form of the head


@pytest.mark.parametrize("parts,cout,n,hw,resid", [
    ([128, 128, 128], 128, 3, (16, 64), True),       # ResBlock / UpBlock concat conv of a 128-channel block + shortcut residual
    ([256, 256, 256], 256, 2, (8, 64), True),        # ... of a 256-channel block: 12 chunks of 64 channels
    ([256, 256, 256], 256, 5, (4, 128), False),      # H W = 512: two 256-pixel tiles per image, odd image count
    ([64], 128, 2, (16, 64), False), ([128], 256, 2, (8, 64), False), ([256], 256, 1, (4, 64), True),      # the shortcut 1x1 convs (1 / 2 / 4 chunks)
    ([64, 192], 256, 2, (8, 32), False)])            # two sources of unequal (whole-chunk) widths
def test_wide_1x1_gemm_kernel(cuda, parts, cout, n, hw, resid):
    """gemm1x1_h8_kernel (both operands through the LDS double buffer): multi-chunk layers, several tiles per workgroup, tensors whose last tile
    ends at the allocation boundary (every tensor here is exactly as large as the kernel's last access)."""
    _conv_case(cuda, n, parts, cout, hw[0], hw[1], (1, 1, 0), seed=sum(parts) + cout + n, resid=resid)


def test_wide_1x1_gemm_kernel_is_the_one_that_runs(cuda):
    from semanticlidarunc_amd import ops
    ops.TIMING, ops.TIMING_TAGS = [], []
    try:
        _conv_case(cuda, 2, [256, 256, 256], 256, 8, 64, (1, 1, 0), seed=1)
        names = [t[0] for t in ops.TIMING]
    finally:
        ops.TIMING = None
    assert names == ["gemm1x1_h8_kernel<4, 4, 2>"], names


@pytest.mark.parametrize("dil", [1, 2])
@pytest.mark.parametrize("cin,cout", [(32, 32), (32, 64), (64, 32), (64, 64)])
def test_deep_ring_3x3_full_resolution_layers(cuda, cin, cout, dil):
    """ring3_h8_kernel (one plain source of 32 / 64 channels, 32 / 64 outputs, no residual, >= 256 tiles): whole tiles, ragged borders in
    both directions, more tiles than one round of the persistent grid; the kernel name the ABI reports must be the ring form."""
    fam = (3, dil, dil)
    _conv_case(cuda, 4, [cin], cout, 64, 1024, fam, seed=20 + cin + cout + dil, resid=False)
    _conv_case(cuda, 4, [cin], cout, 70, 1000, fam, seed=21 + cin + cout + dil, resid=False)     # partial tiles right and bottom
    def launched(resid):
        ops.TIMING, ops.TIMING_TAGS = [], []              # measurement mode records the instantiation slu_conv2d_h8_kernel_name reports
        try:
            _conv_case(cuda, 4, [cin], cout, 64, 1024, fam, seed=22 + cin + cout + dil, resid=resid)
            return ops.TIMING[0][0]
        finally:
            ops.TIMING, ops.TIMING_TAGS = None, None
    assert launched(False).startswith(f"ring3_h8_kernel<{dil}, {cout // 32}, {cin // 16}, ")
    assert launched(True).startswith("conv_h8_kernel<")  # outside its reach (residual): the tiled kernel


def test_deep_ring_3x3_two_plain_sources(cuda):
    """UpBlock.conv1 at full resolution: PixelShuffle output (16 channels) | skip (64 channels) -> 32, both plain: ring form, 5 K-steps"""
    ops.TIMING, ops.TIMING_TAGS = [], []
    try:
        _conv_case(cuda, 4, [16, 64], 32, 64, 1024, (3, 1, 1), seed=31, resid=False)
        assert ops.TIMING[0][0] == "ring3_h8_kernel<1, 1, 5, 1, 4>", ops.TIMING[0][0]
    finally:
        ops.TIMING, ops.TIMING_TAGS = None, None
    _conv_case(cuda, 5, [16, 64], 32, 61, 900, (3, 1, 1), seed=32, resid=False)
    _conv_case(cuda, 4, [32, 32], 64, 64, 1024, (3, 2, 2), seed=33, resid=False)     # two sources inside the 64 -> 64 form


def test_conv_argument_checks(cuda):
    x = h8.to_h8(torch.zeros(1, 8, 4, 32, device=cuda))
    w = h8.pack_conv_weight_h8(torch.zeros(8, 8, 3, 3, device=cuda))
    with pytest.raises(RuntimeError):
        h8.conv2d_h8([H8Source(x)], w, 16, 8, 3, 1, 1)             # cin does not match the blocks
    with pytest.raises(RuntimeError):
        h8.conv2d_h8([H8Source(x)], w, 8, 40, 3, 1, 1)             # packed weight of another shape
    with pytest.raises(RuntimeError):
        h8.conv2d_h8([H8Source(x.float())], w, 8, 8, 3, 1, 1)      # not an h8 tensor
    with pytest.raises(RuntimeError):
        h8.conv2d_h8([H8Source(x)], w, 8, 8, 3, 1, 1, resid=x, out_f32_nchw=True)
    with pytest.raises(Exception):
        h8.conv2d_h8([H8Source(x)], w, 8, 8, 3, 3, 3)              # unsupported family -> SLU_EUNSUPPORTED


def test_avgpool_and_pixel_shuffle(cuda):
    g = torch.Generator().manual_seed(2)
    for n, c, h, w in [(2, 16, 8, 64), (1, 8, 5, 33), (3, 64, 16, 32)]:
        x = _r16(torch.randn(n, c, h, w, generator=g))
        s = (torch.rand(n, c, generator=g) > 0.2).float() * 1.25
        want = osalsa.avgpool3s2(x, s)
        got = h8.from_h8(h8.avgpool3s2_h8(h8.to_h8(x.to(cuda)), s.to(cuda)), c).cpu()
        assert float((got - want).abs().max()) <= 2.0 ** -10 * float(want.abs().max()) + 1e-6
        got = h8.from_h8(h8.avgpool3s2_h8(h8.to_h8(x.to(cuda))), c).cpu()
        assert float((got - osalsa.avgpool3s2(x)).abs().max()) <= 2.0 ** -10 * float(want.abs().max()) + 1e-6
    # pooled broadcast: B images feed T*B outputs with their own multipliers
    x = _r16(torch.randn(2, 16, 8, 32, generator=g))
    s = (torch.rand(6, 16, generator=g) > 0.2).float() * 1.25
    want = osalsa.avgpool3s2(x.repeat(3, 1, 1, 1), s)
    got = h8.from_h8(h8.avgpool3s2_h8(h8.to_h8(x.to(cuda)), s.to(cuda), 6), 16).cpu()
    assert float((got - want).abs().max()) <= 2.0 ** -10 * float(want.abs().max()) + 1e-6
    for n, c, h, w in [(2, 64, 4, 16), (1, 32, 3, 5), (2, 256, 4, 8), (1, 16, 2, 2)]:
        x = _r16(torch.randn(n, c, h, w, generator=g))
        want = F.pixel_shuffle(x, 2)
        got = h8.from_h8(h8.pixel_shuffle_h8(h8.to_h8(x.to(cuda))), c // 4).cpu()
        assert torch.equal(got, want)
        si = (torch.rand(n, c, generator=g) > 0.2).float() * 1.25
        so = (torch.rand(n, c // 4, generator=g) > 0.2).float() * 1.25
        want = _r16(F.pixel_shuffle(x * si[:, :, None, None], 2) * so[:, :, None, None])
        got = h8.from_h8(h8.pixel_shuffle_h8(h8.to_h8(x.to(cuda)), si.to(cuda), so.to(cuda)), c // 4).cpu()
        assert torch.equal(got, want)


def _entropy(logits):
    p = logits.softmax(1).clamp_min(1e-12)
    return -(p * p.log()).sum(1) / math.log(logits.shape[1])


def test_whole_network_in_f16(cuda):
    model = seeded_model(sn.SalsaNext).to(cuda)
    sn.set_conv_precision("f16")
    try:
        g = golden("salsanext_eval_1x5x16x64")
        with torch.no_grad():
            y = model(_t(g["x"]).to(cuda)).cpu()
        assert y.dtype == torch.float32 and float((y - _t(g["logits"])).abs().max()) <= 1e-3
        g = golden("salsanext_mc_2x5x32x64")
        scales = {k[len("scale:"):]: _t(g[k]) for k in g.files if k.startswith("scale:")}
        with torch.no_grad():
            y = model.forward_with_dropout_scales(_t(g["x"]).to(cuda), scales).cpu()
        assert float((y - _t(g["logits"])).abs().max()) <= 1e-3
        x, _ = synthetic_scan(1, 64, 2048)
        sd = {k: v.cpu() for k, v in model.state_dict().items()}
        oscales = osalsa.draw_dropout_scales(1, 0.2, torch.Generator().manual_seed(5))
        with torch.no_grad():
            want = osalsa.salsanext_forward(sd, x)
            got = model(x.to(cuda)).cpu()
            want_d = osalsa.salsanext_forward(sd, x, oscales)
            got_d = model.forward_with_dropout_scales(x.to(cuda), oscales).cpu()
        for a, b in ((got, want), (got_d, want_d)):
            assert float((a - b).abs().max()) <= 1e-3, float((a - b).abs().max())
            assert float((_entropy(a) - _entropy(b)).abs().max()) <= 1e-3
            assert float((a.argmax(1) != b.argmax(1)).float().mean()) < 5e-3      # near-ties only
        # with gradients wanted the model must not take the inference-only path
        xg = _t(golden("salsanext_eval_1x5x16x64")["x"]).to(cuda)
        out = model(xg)
        assert out.requires_grad
    finally:
        sn.set_conv_precision("fp32")


def test_mc_shared_prefix_equals_stacked_passes_in_f16(cuda):
    from semanticlidarunc_amd.utils.mc_dropout import mc_predict
    model = seeded_model(sn.SalsaNext).to(cuda)
    sn.set_conv_precision("f16")
    try:
        b, t = 2, 3
        x, _ = synthetic_scan(b, 32, 128, seed=7)
        oscales = osalsa.draw_dropout_scales(t * b, 0.2, torch.Generator().manual_seed(11))
        with torch.no_grad():
            stacked = model.forward_with_dropout_scales(x.repeat(t, 1, 1, 1).to(cuda), oscales)
            shared = model.forward_mc(x.to(cuda), t, oscales)
        assert torch.equal(stacked, shared)
        sd = {k: v.cpu() for k, v in model.state_dict().items()}
        want = osalsa.salsanext_forward(sd, x.repeat(t, 1, 1, 1), oscales)
        assert float((shared.cpu() - want).abs().max()) <= 1e-3
        # uncertainty maps from the half-precision logits stay within 1e-3 of the fp32 oracle's
        p_bar, h_norm, mi_norm, preds = ounc.mc_reduce(want.view(t, b, *want.shape[1:]))
        gp, gh, gm, _ = ops.mc_reduce(shared.view(t, b, *shared.shape[1:]))
        assert float((gp.cpu() - p_bar).abs().max()) <= 1e-3
        assert float((gh.cpu() - h_norm).abs().max()) <= 1e-3
        assert float((gm.cpu() - mi_norm).abs().max()) <= 1e-3
        # the drop-in entry point runs in this mode too
        torch.manual_seed(3)
        out = mc_predict(model, [x.to(cuda)], T=t)
        assert out[0].shape == (b, 20, 32, 128) and bool(torch.isfinite(out[1]).all())
    finally:
        sn.set_conv_precision("fp32")


def test_ouster_shape_128x4096_T16(cuda):
    """BASELINE configs[4] shape: one 128x4096 scan against the fp32 oracle, and T=16 MC passes through the shared-prefix
    schedule against T stacked passes (bit-identical) with finite, normalised uncertainty maps."""
    model = seeded_model(sn.SalsaNext).to(cuda)
    sn.set_conv_precision("f16")
    try:
        x, _ = synthetic_scan(1, 128, 4096, seed=21)
        sd = {k: v.cpu() for k, v in model.state_dict().items()}
        with torch.no_grad():
            want = osalsa.salsanext_forward(sd, x)
            got = model(x.to(cuda)).cpu()
        assert float((got - want).abs().max()) <= 1e-3
        assert float((_entropy(got) - _entropy(want)).abs().max()) <= 1e-3
        t = 16
        oscales = osalsa.draw_dropout_scales(t, 0.2, torch.Generator().manual_seed(3))
        with torch.no_grad():
            stacked = model.forward_with_dropout_scales(x.repeat(t, 1, 1, 1).to(cuda), oscales)
            shared = model.forward_mc(x.to(cuda), t, oscales)
        assert torch.equal(stacked, shared)
        p_bar, h_norm, mi_norm, preds = ops.mc_reduce(shared.view(t, 1, *shared.shape[1:]))
        assert bool(torch.isfinite(p_bar).all()) and float((p_bar.sum(1) - 1).abs().max()) <= 1e-5
        assert float(h_norm.min()) >= 0 and float(h_norm.max()) <= 1 + 1e-6 and float(mi_norm.min()) >= 0
        assert float((mi_norm - h_norm).max()) <= 1e-6             # MI <= H
        # the fused head + MC reduction at this size, same multipliers, both schedules: same maps as reducing the stored logits
        for share in (False, True):
            fp, fh, fm, fa = model.mc_predict_fused(x.to(cuda), t, share_prefix=share, scales=oscales)
            assert float((fp - p_bar).abs().max()) <= 2e-6 and float((fh - h_norm).abs().max()) <= 2e-5
            assert float((fm - mi_norm).abs().max()) <= 2e-5 and int((fa != preds).sum()) <= 4
    finally:
        sn.set_conv_precision("fp32")
