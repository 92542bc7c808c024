"""GPU: every conv kernel family / tile configuration of slu_conv2d_fwd against the torch-fp32 CPU
oracle (oracle.salsanext.fused_conv), incl. borders, ragged tiles, concat, PixelShuffle and folded
dropout multipliers.  Tolerance: 1e-4 abs on O(1) outputs (fp32 MFMA k-ordered fmaf chain vs
oneDNN summation order; north-star bar is 1e-3)."""
import pytest
import torch

from oracle import salsanext as osalsa
from semanticlidarunc_amd import ops
from semanticlidarunc_amd.ops import ConvSource

pytestmark = pytest.mark.gpu
FAMILIES = [(1, 1, 0), (3, 1, 1), (3, 2, 2), (2, 2, 1)]


def _run(dev, n, cin_parts, cout, h, w, fam, seed, act=True, bn=True, resid=True, scales=False, ps_first=False):
    k, dil, pad = fam
    g = torch.Generator().manual_seed(seed)
    srcs_cpu = []
    cin = 0
    for i, c in enumerate(cin_parts):
        ps = ps_first and i == 0
        t = torch.randn(n, c, h // 2 if ps else h, w // 2 if ps else w, generator=g)
        s = None
        if scales:
            s = (torch.rand(n, c, generator=g) > 0.2).float() * 1.25
        srcs_cpu.append((t, s, ps))
        cin += c // 4 if ps else c
    wgt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    bn_a = torch.rand(cout, generator=g) + 0.5 if bn else None
    bn_b = torch.randn(cout, generator=g) * 0.1 if bn else None
    res = torch.randn(n, cout, h, w, generator=g) if resid else None
    want = osalsa.fused_conv(srcs_cpu, wgt, bias, pad, dil, 0.01 if act else None, bn_a, bn_b, res)
    d = lambda t: None if t is None else t.to(dev).contiguous()
    wpack = ops.pack_conv_weight(d(wgt))
    got = ops.conv2d_fused([ConvSource(d(t), d(s), ps) for t, s, ps in srcs_cpu], wpack, cout, k, dil, pad,
                           bias=d(bias), slope=0.01 if act else None, bn_a=d(bn_a), bn_b=d(bn_b), resid=d(res))
    torch.cuda.synchronize()
    err = float((got.cpu() - want).abs().max())
    assert err <= 1e-4, f"fam={fam} cin={cin_parts} cout={cout} {h}x{w}: max abs err {err}"


@pytest.mark.parametrize("fam", FAMILIES)
@pytest.mark.parametrize("cout,hw", [(32, (16, 128)), (64, (8, 64)), (128, (8, 64)), (256, (4, 64)), (20, (16, 64))])
def test_family_by_channel_tile(cuda, fam, cout, hw):
    _run(cuda, 2, [32], cout, hw[0], hw[1], fam, seed=cout + fam[0] * 7 + fam[1])


@pytest.mark.parametrize("fam", FAMILIES)
def test_ragged_sizes_and_odd_channels(cuda, fam):
    # H, W not multiples of the tile; Cin not a multiple of the K-chunk; Cout not a multiple of 32
    _run(cuda, 1, [5], 32, 13, 70, fam, seed=3)
    _run(cuda, 3, [21], 40, 5, 33, fam, seed=4, bn=False, resid=False)
    _run(cuda, 1, [7], 70, 9, 129, fam, seed=5, act=False)


def test_big_tiles_full_resolution_row(cuda):
    # enough workgroups that choose_cfg picks the TH=8 / M64 / M128 tiles
    _run(cuda, 1, [32], 32, 64, 1024, (3, 2, 2), seed=6)
    _run(cuda, 1, [32], 64, 64, 1024, (3, 1, 1), seed=7)
    _run(cuda, 4, [64], 128, 32, 512, (2, 2, 1), seed=8)
    _run(cuda, 2, [64, 64, 64], 64, 64, 512, (1, 1, 0), seed=9)


def test_concat_of_three_sources_with_scales(cuda):
    _run(cuda, 2, [64, 64, 64], 64, 8, 64, (1, 1, 0), seed=10, scales=True)
    _run(cuda, 2, [32, 16, 48], 32, 8, 64, (3, 1, 1), seed=11, scales=True)


@pytest.mark.parametrize("cparts,cout", [([256, 256], 128), ([64, 64], 32), ([128, 128], 64)])
def test_pixel_shuffle_source_plus_skip(cuda, cparts, cout):
    # UpBlock.conv1: cat(PixelShuffle(x), skip) with folded dropout multipliers
    _run(cuda, 2, cparts, cout, 8, 64, (3, 1, 1), seed=12, scales=True, ps_first=True)


def test_linearity_property_full_size(cuda):
    # conv(a*x1 + x2) == a*conv(x1) + conv(x2) at 64x2048 (no activation): size-independent check
    g = torch.Generator().manual_seed(13)
    x1 = torch.randn(1, 32, 64, 2048, generator=g).to(cuda)
    x2 = torch.randn(1, 32, 64, 2048, generator=g).to(cuda)
    w = (torch.randn(32, 32, 3, 3, generator=g) / 17).to(cuda)
    wp = ops.pack_conv_weight(w)
    f = lambda x: ops.conv2d_fused([ConvSource(x)], wp, 32, 3, 2, 2)
    lhs = f((2.5 * x1 + x2).contiguous())
    rhs = 2.5 * f(x1) + f(x2)
    assert float((lhs - rhs).abs().max()) <= 1e-4


def test_avgpool_matches_oracle(cuda):
    g = torch.Generator().manual_seed(14)
    for shape in [(2, 8, 16, 64), (1, 3, 7, 13), (2, 4, 4, 128)]:
        x = torch.randn(*shape, generator=g)
        s = (torch.rand(shape[0], shape[1], generator=g) > 0.2).float() * 1.25
        got = ops.avgpool3s2(x.to(cuda), s.to(cuda)).cpu()
        assert float((got - osalsa.avgpool3s2(x, s)).abs().max()) <= 1e-6
        got = ops.avgpool3s2(x.to(cuda)).cpu()
        assert float((got - osalsa.avgpool3s2(x)).abs().max()) <= 1e-6


def test_wrappers_reject_bad_input_before_launch(cuda):
    x = torch.zeros(1, 32, 8, 64, device=cuda)
    wp = ops.pack_conv_weight(torch.zeros(32, 32, 3, 3, device=cuda))
    with pytest.raises(RuntimeError):
        ops.conv2d_fused([ConvSource(x.cpu())], wp, 32, 3, 1, 1)
    with pytest.raises(RuntimeError):
        ops.conv2d_fused([ConvSource(x)], wp, 64, 3, 1, 1)          # weight image does not match Cout
    with pytest.raises(RuntimeError):
        ops.conv2d_fused([ConvSource(x), ConvSource(x[:, :, :4])], wp, 32, 3, 1, 1)
    with pytest.raises(RuntimeError):
        ops.conv2d_fused([ConvSource(x.double())], wp, 32, 3, 1, 1)
