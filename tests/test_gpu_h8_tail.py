"""GPU: the fused tail of a SalsaNext block on the h8 path (2x2 dilated conv kept on chip + the 1x1 conv over the concatenation)
against (a) the same two layers through the unfused h8 kernels -- same fp16 operands, same fp16 rounding of the intermediate, only the
fp32 accumulation order of the 1x1 differs -- and (b) the CPU oracle's fused-conv operator (oracle.salsanext.fused_conv, torch fp32 on
the HOST) on the fp16-rounded inputs.  Bars: 2e-3 of the output scale vs the unfused path (one fp16 ulp of an O(1) output is 1e-3),
3e-3 vs the fp32 CPU oracle (its output is not rounded to fp16)."""
import pytest
import torch

from oracle import salsanext as osalsa
from semanticlidarunc_amd import h8

pytestmark = pytest.mark.gpu


def _case(cuda, c, n, hh, ww, resid, seed, slope_a=0.01, slope_b=0.01, bn=True):
    g = torch.Generator(device=cuda).manual_seed(seed)
    a1 = torch.randn(n, c, hh, ww, device=cuda, generator=g)
    a2 = torch.randn(n, c, hh, ww, device=cuda, generator=g)
    r = torch.randn(n, c, hh, ww, device=cuda, generator=g) if resid else None
    w2 = torch.randn(c, c, 2, 2, device=cuda, generator=g) / (4 * c) ** 0.5
    w1 = torch.randn(c, 3 * c, 1, 1, device=cuda, generator=g) / (3 * c) ** 0.5
    ba, bb = torch.randn(c, device=cuda, generator=g) * 0.1, torch.randn(c, device=cuda, generator=g) * 0.1
    bna = (torch.rand(c, device=cuda, generator=g) + 0.5, torch.randn(c, device=cuda, generator=g) * 0.1) if bn else None
    bnb = (torch.rand(c, device=cuda, generator=g) + 0.5, torch.randn(c, device=cuda, generator=g) * 0.1) if bn else None
    h1, h2, hr = h8.to_h8(a1), h8.to_h8(a2), (None if r is None else h8.to_h8(r))
    p2, p1 = h8.pack_conv_weight_h8(w2), h8.pack_conv_weight_h8(w1)
    fused = h8.from_h8(h8.conv_tail_h8(h1, h2, p2, p1, ba, slope_a, bna, bb, slope_b, bnb, resid=hr), c)
    # unfused h8
    kw = lambda pair: dict(bn_a=None if pair is None else pair[0], bn_b=None if pair is None else pair[1])
    h3 = h8.conv2d_h8([h8.H8Source(h2)], p2, c, c, 2, 2, 1, bias=ba, slope=slope_a, **kw(bna))
    unf = h8.from_h8(h8.conv2d_h8([h8.H8Source(h1), h8.H8Source(h2), h8.H8Source(h3)], p1, 3 * c, c, 1, 1, 0, bias=bb, slope=slope_b,
                                  resid=hr, **kw(bnb)), c)
    # the CPU oracle (fp32, host) on the fp16-rounded operands; the intermediate a3 is rounded to fp16 where the device stores it
    q = lambda t: t.half().float().cpu()
    pair = lambda pr: (None, None) if pr is None else (pr[0].cpu(), pr[1].cpu())
    a3 = osalsa.fused_conv([(q(a2), None, False)], q(w2), ba.cpu(), 1, 2, slope_a, *pair(bna)).half().float()
    ref = osalsa.fused_conv([(q(a1), None, False), (q(a2), None, False), (a3, None, False)], q(w1), bb.cpu(), 0, 1, slope_b, *pair(bnb),
                            resid=None if r is None else q(r))
    fused, unf = fused.cpu(), unf.cpu()
    scale = float(ref.abs().max())
    assert float((fused - unf).abs().max()) <= 2e-3 * scale, (c, n, hh, ww, float((fused - unf).abs().max()), scale)
    assert float((fused - ref).abs().max()) <= 3e-3 * scale, (c, n, hh, ww, float((fused - ref).abs().max()), scale)


@pytest.mark.parametrize("c", [32, 64, 128])
def test_tail_matches_unfused_and_fp32(cuda, c):
    _case(cuda, c, 2, 64, 256, True, 1)
    _case(cuda, c, 3, 16, 64, False, 2)                                    # one tile column, several images
    _case(cuda, c, 1, 19, 150, True, 3)                                    # ragged: partial tiles in both directions
    _case(cuda, c, 2, 5, 37, False, 4, slope_a=None, slope_b=None, bn=False)   # no activation, no BN, tiny image
    _case(cuda, c, 70, 8, 64, True, 5)                                     # more tiles than one round of workgroups... per image 1


@pytest.mark.parametrize("n,hh,ww,seed", [(2, 64, 256, 11), (1, 19, 150, 12), (6, 64, 2048, 13)])
def test_tail_with_the_shortcut_conv_inside(cuda, n, hh, ww, seed):
    """ResBlock 32 -> 64: out = bnB(leaky(conv1x1(cat))) + leaky(conv1x1(x) + b), the shortcut computed inside the tail from x (never stored),
    == the same tail fed the shortcut tensor of a separate conv2d_h8 launch (same fp16 rounding of the shortcut: 2e-3 of the scale, fp32
    summation order of two K-steps), and the CPU oracle's fused_conv on the fp16-rounded operands (3e-3)."""
    c, cx = 64, 32
    g = torch.Generator(device=cuda).manual_seed(seed)
    rn = lambda *sh: torch.randn(*sh, device=cuda, generator=g)
    x, a1, a2 = rn(n, cx, hh, ww), rn(n, c, hh, ww), rn(n, c, hh, ww)
    w2, w1, ws = rn(c, c, 2, 2) / (4 * c) ** 0.5, rn(c, 3 * c, 1, 1) / (3 * c) ** 0.5, rn(c, cx, 1, 1) / cx ** 0.5
    ba, bb, bs = rn(c) * 0.1, rn(c) * 0.1, rn(c) * 0.1
    bna, bnb = (torch.rand(c, device=cuda, generator=g) + 0.5, rn(c) * 0.1), (torch.rand(c, device=cuda, generator=g) + 0.5, rn(c) * 0.1)
    hx, h1, h2 = h8.to_h8(x), h8.to_h8(a1), h8.to_h8(a2)
    p2, p1, ps = h8.pack_conv_weight_h8(w2), h8.pack_conv_weight_h8(w1), h8.pack_conv_weight_h8(ws)
    assert h8.conv_tail_shortcut_supported(64, 32) and not h8.conv_tail_shortcut_supported(32, 32) and not h8.conv_tail_shortcut_supported(64, 64)
    fused = h8.from_h8(h8.conv_tail_h8(h1, h2, p2, p1, ba, 0.01, bna, bb, 0.01, bnb, shortcut=(hx, ps, bs, 0.01, cx)), c).cpu()
    sc = h8.conv2d_h8([h8.H8Source(hx)], ps, cx, c, 1, 1, 0, bias=bs, slope=0.01)
    sep = h8.from_h8(h8.conv_tail_h8(h1, h2, p2, p1, ba, 0.01, bna, bb, 0.01, bnb, resid=sc), c).cpu()
    q = lambda t: t.half().float().cpu()
    pair = lambda pr: (pr[0].cpu(), pr[1].cpu())
    osc = osalsa.fused_conv([(q(x), None, False)], q(ws), bs.cpu(), 0, 1, 0.01).half().float()
    a3 = osalsa.fused_conv([(q(a2), None, False)], q(w2), ba.cpu(), 1, 2, 0.01, *pair(bna)).half().float()
    ref = osalsa.fused_conv([(q(a1), None, False), (q(a2), None, False), (a3, None, False)], q(w1), bb.cpu(), 0, 1, 0.01, *pair(bnb), resid=osc)
    scale = float(ref.abs().max())
    assert float((fused - sep).abs().max()) <= 2e-3 * scale, (n, hh, ww, float((fused - sep).abs().max()), scale)
    assert float((fused - ref).abs().max()) <= 3e-3 * scale, (n, hh, ww, float((fused - ref).abs().max()), scale)
    with pytest.raises(RuntimeError):      # exclusive with resid
        h8.conv_tail_h8(h1, h2, p2, p1, ba, 0.01, bna, bb, 0.01, bnb, resid=sc, shortcut=(hx, ps, bs, 0.01, cx))


def test_tail_many_tiles_and_argument_checks(cuda):
    _case(cuda, 64, 6, 64, 2048, True, 6)                                  # 6 x 8 x 32 = 1536 tiles: six rounds of the persistent grid
    _case(cuda, 32, 5, 64, 2048, False, 7)
    _case(cuda, 128, 9, 32, 1024, True, 8)                                  # 9 x 8 x 16 = 1152 tiles of 4 rows
    assert h8.conv_tail_supported(32, 64, 2048) and h8.conv_tail_supported(64, 32, 1024) and h8.conv_tail_supported(128, 32, 1024) and not h8.conv_tail_supported(256, 16, 512)
    x = h8.to_h8(torch.zeros(1, 256, 8, 64, device=cuda))
    with pytest.raises(RuntimeError):
        h8.conv_tail_h8(x, x, torch.zeros(1, dtype=torch.uint8, device=cuda), torch.zeros(1, dtype=torch.uint8, device=cuda), None, None, None, None, None, None)
