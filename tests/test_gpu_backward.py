"""GPU: training-side kernels and the autograd path against torch-CPU autograd (oracle) and the reference's own
training step (golden: train-mode BatchNorm, 13 dropout multipliers, backward through a fixed projection).
Tolerances: single ops 1e-4 relative to the gradient scale; whole-network gradients 2e-3 relative (fp32
round-off through ~50 layers of batch-statistics BatchNorm; the bar for logits stays 1e-3 abs)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden
from oracle import salsanext as osalsa
from semanticlidarunc_amd import ops
from semanticlidarunc_amd.autograd import AvgPoolFn, ConvLayerFn, LayerCfg
from semanticlidarunc_amd.ops import ConvSource
from semanticlidarunc_amd.salsanext import SalsaNext
from semanticlidarunc_amd.testing import seeded_model

pytestmark = pytest.mark.gpu
FAMILIES = [(1, 1, 0), (3, 1, 1), (3, 2, 2), (2, 2, 1)]


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _rel(a, b):
    return float((a - b).abs().max()) / (1e-12 + float(b.abs().max()))


def test_elementwise_training_kernels(cuda):
    g = torch.Generator().manual_seed(1)
    y = torch.randn(3, 40, 6, 66, generator=g)
    s, q = ops.bn_stats(y.to(cuda))
    assert torch.allclose(s.cpu(), y.double().sum((0, 2, 3)), rtol=1e-9, atol=1e-9)
    assert torch.allclose(q.cpu(), (y.double() ** 2).sum((0, 2, 3)), rtol=1e-9, atol=1e-9)
    a, b, r = torch.randn(40, generator=g), torch.randn(40, generator=g), torch.randn_like(y)
    z = ops.affine(y.to(cuda), a.to(cuda), b.to(cuda), r.to(cuda)).cpu()
    assert float((z - (y * a[None, :, None, None] + b[None, :, None, None] + r)).abs().max()) <= 1e-5
    assert torch.equal(ops.affine(y.to(cuda), None, None, r.to(cuda)).cpu(), y + r)
    mean, invstd = y.mean((0, 2, 3)), 1.0 / y.std((0, 2, 3))
    dz = torch.randn_like(y)
    s1, s2 = ops.bn_bwd_reduce(dz.to(cuda), y.to(cuda), mean.to(cuda), invstd.to(cuda))
    xh = (y - mean[None, :, None, None]) * invstd[None, :, None, None]
    assert torch.allclose(s1.cpu(), dz.double().sum((0, 2, 3)), rtol=1e-8, atol=1e-8)
    assert torch.allclose(s2.cpu(), (dz * xh).double().sum((0, 2, 3)), rtol=1e-5, atol=1e-5)
    k1, k2, k3 = torch.randn(40, generator=g), torch.randn(40, generator=g), torch.randn(40, generator=g)
    da, db = ops.act_affine_bwd(dz.to(cuda), y.to(cuda), k1.to(cuda), k2.to(cuda), k3.to(cuda), 0.01)
    want = (k1[None, :, None, None] * dz + k2[None, :, None, None] + k3[None, :, None, None] * y) * torch.where(y > 0, 1.0, 0.01)
    assert float((da.cpu() - want).abs().max()) <= 1e-5
    assert torch.allclose(db.cpu(), want.double().sum((0, 2, 3)), rtol=1e-6, atol=1e-5)
    t = ops.nchw_to_nhwc(y.to(cuda)).cpu()
    assert t.shape == (3, 6 * 66, 64) and torch.equal(t[:, :, :40], y.permute(0, 2, 3, 1).reshape(3, -1, 40)) and float(t[:, :, 40:].abs().max()) == 0
    x = torch.randn(2, 5, 7, 13, generator=g)
    sc = (torch.rand(2, 5, generator=g) > 0.3).float() * 1.25
    xc = x.clone().requires_grad_(True)
    osalsa.avgpool3s2(xc, sc).backward(torch.ones(2, 5, 4, 7))
    dy = torch.randn(2, 5, 4, 7, generator=g)
    xc.grad = None
    osalsa.avgpool3s2(xc, sc).backward(dy)
    assert float((ops.avgpool3s2_bwd(dy.to(cuda), sc.to(cuda), x.shape).cpu() - xc.grad).abs().max()) <= 1e-6


def test_gather_and_split_are_adjoint_views_of_the_conv_input(cuda):
    g = torch.Generator().manual_seed(2)
    xs = torch.randn(2, 64, 4, 32, generator=g)          # read through PixelShuffle -> 16 ch at 8x64
    sk = torch.randn(2, 24, 8, 64, generator=g)
    s0 = (torch.rand(2, 64, generator=g) > 0.2).float() * 1.25
    s1 = (torch.rand(2, 24, generator=g) > 0.2).float() * 1.25
    cat = torch.cat((F.pixel_shuffle(xs * s0[:, :, None, None], 2), sk * s1[:, :, None, None]), 1)
    got = ops.gather_nhwc([ConvSource(xs.to(cuda), s0.to(cuda), True), ConvSource(sk.to(cuda), s1.to(cuda), False)]).cpu()
    assert got.shape == (2, 8 * 64, 64) and torch.equal(got[:, :, :40], cat.permute(0, 2, 3, 1).reshape(2, -1, 40))
    dcat = torch.randn(2, 40, 8, 64, generator=g)
    d0 = ops.split_grad(dcat.to(cuda), 0, xs.shape, True, s0.to(cuda)).cpu()
    d1 = ops.split_grad(dcat.to(cuda), 16, sk.shape, False, s1.to(cuda)).cpu()
    assert torch.allclose(d0, F.pixel_unshuffle(dcat[:, :16], 2) * s0[:, :, None, None], atol=1e-7)
    assert torch.allclose(d1, dcat[:, 16:] * s1[:, :, None, None], atol=1e-7)


@pytest.mark.parametrize("fam", FAMILIES)
@pytest.mark.parametrize("cin,cout", [(5, 32), (48, 40), (64, 128)])
def test_wgrad_and_dgrad_match_torch_autograd(cuda, fam, cin, cout):
    k, dil, pad = fam
    g = torch.Generator().manual_seed(cin + cout + k)
    x = torch.randn(2, cin, 9, 34, generator=g, requires_grad=True)
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).requires_grad_(True)
    da = torch.randn(2, cout, 9, 34, generator=g)
    F.conv2d(x, w, None, padding=pad, dilation=dil).backward(da)
    dw = ops.conv2d_wgrad(ops.nchw_to_nhwc(da.to(cuda)), ops.nchw_to_nhwc(x.detach().to(cuda)), 2, 9, 34, cout, cin, k, dil, pad).cpu()
    assert _rel(dw, w.grad) <= 1e-4
    wd = ops.dgrad_weight(w.detach().to(cuda))
    dx = ops.conv2d_fused([ConvSource(da.to(cuda))], ops.pack_conv_weight(wd), cin, k, dil, pad).cpu()
    assert _rel(dx, x.grad) <= 1e-4


@pytest.mark.parametrize("chans,cout,n,h,w", [((5,), 32, 2, 8, 40), ((32, 32, 32), 32, 3, 16, 64), ((64, 64, 72), 80, 2, 8, 28),
                                               ((256, 256, 256), 256, 2, 4, 128), ((48,), 40, 1, 4, 8)])
def test_wgrad_1x1_from_nchw(cuda, chans, cout, n, h, w):
    """The 1x1 weight gradient read straight from NCHW da and the (concatenated) sources == autograd of F.conv2d on the CPU (1e-4 of the
    gradient's scale: fp32 MFMA sums in another order, fp32 atomics); uncovered shapes return None (then the channel-last kernel runs)."""
    g = torch.Generator().manual_seed(sum(chans) + cout)
    xs = [torch.randn(n, c, h, w, generator=g) for c in chans]
    wt = (torch.randn(cout, sum(chans), 1, 1, generator=g) / sum(chans) ** 0.5).requires_grad_(True)
    da = torch.randn(n, cout, h, w, generator=g)
    F.conv2d(torch.cat(xs, 1), wt).backward(da)
    dw = ops.conv1x1_wgrad_nchw(da.to(cuda), [ConvSource(x.to(cuda)) for x in xs])
    assert dw is not None and _rel(dw.cpu(), wt.grad) <= 1e-4
    # not covered: H*W not a multiple of 32, an unaligned concat, PixelShuffle / multiplier sources
    assert ops.conv1x1_wgrad_nchw(torch.zeros(1, 32, 3, 5, device=cuda), [ConvSource(torch.zeros(1, 32, 3, 5, device=cuda))]) is None
    assert ops.conv1x1_wgrad_nchw(torch.zeros(1, 32, 4, 8, device=cuda), [ConvSource(torch.zeros(1, 5, 4, 8, device=cuda)), ConvSource(torch.zeros(1, 32, 4, 8, device=cuda))]) is None
    assert ops.conv1x1_wgrad_nchw(torch.zeros(1, 32, 4, 8, device=cuda), [ConvSource(torch.zeros(1, 32, 4, 8, device=cuda), torch.ones(1, 32, device=cuda))]) is None


@pytest.mark.parametrize("fam", [(3, 1, 1), (3, 2, 2), (2, 2, 1)])
@pytest.mark.parametrize("chans,cout,n,h,w", [((32,), 32, 2, 8, 32), ((5,), 32, 1, 5, 16), ((32, 48), 40, 2, 7, 48), ((64, 64, 32), 64, 3, 4, 16), ((128,), 128, 2, 16, 64)])
def test_wgrad_kxk_from_nchw(cuda, fam, chans, cout, n, h, w):
    """3x3 (dil 1 / 2) and 2x2-dilated weight gradients read straight from NCHW da and the (concatenated) sources == autograd of F.conv2d on the
    CPU (1e-4 of the gradient's scale: fp32 MFMA sums in another order, fp32 atomics): one-unit rows (both row ends in one unit), image borders
    in y, channel counts that are not multiples of 32; uncovered shapes return None (then the channel-last kernel runs)."""
    k, dil, pad = fam
    g = torch.Generator().manual_seed(sum(chans) + cout + k + dil)
    xs = [torch.randn(n, c, h, w, generator=g) for c in chans]
    wt = (torch.randn(cout, sum(chans), k, k, generator=g) / (sum(chans) * k * k) ** 0.5).requires_grad_(True)
    da = torch.randn(n, cout, h, w, generator=g)
    F.conv2d(torch.cat(xs, 1), wt, None, padding=pad, dilation=dil).backward(da)
    dw = ops.conv2d_wgrad_nchw(da.to(cuda), [ConvSource(x.to(cuda)) for x in xs], k, dil, pad)
    assert dw is not None and _rel(dw.cpu(), wt.grad) <= 1e-4
    z = lambda *sh: torch.zeros(*sh, device=cuda)
    assert ops.conv2d_wgrad_nchw(z(1, 32, 4, 24), [ConvSource(z(1, 32, 4, 24))], k, dil, pad) is None                  # W % 16
    assert ops.conv2d_wgrad_nchw(z(1, 32, 4, 16), [ConvSource(z(1, 32, 4, 16))], 3, 3, 3) is None                      # not one of the families
    # UpBlock.conv1: PixelShuffle(x) with a per-stored-channel multiplier | skip with a multiplier, block boundaries inside a 32-channel block
    cx, cs = 4 * (chans[0] // 4 + 1), 24
    xs2 = [torch.randn(n, cx, h // 2 if h % 2 == 0 else h, w // 2, generator=g), torch.randn(n, cs, h, w, generator=g)]
    if h % 2 == 0:
        sc = [(torch.rand(n, cx, generator=g) > 0.3).float() * 1.25, (torch.rand(n, cs, generator=g) > 0.3).float() * 1.25]
        cat = torch.cat([F.pixel_shuffle(xs2[0] * sc[0][:, :, None, None], 2), xs2[1] * sc[1][:, :, None, None]], 1)
        w2 = (torch.randn(cout, cx // 4 + cs, k, k, generator=g) / ((cx // 4 + cs) * k * k) ** 0.5).requires_grad_(True)
        F.conv2d(cat, w2, None, padding=pad, dilation=dil).backward(da)
        dw2 = ops.conv2d_wgrad_nchw(da.to(cuda), [ConvSource(xs2[0].to(cuda), sc[0].to(cuda), True), ConvSource(xs2[1].to(cuda), sc[1].to(cuda))], k, dil, pad)
        assert dw2 is not None and _rel(dw2.cpu(), w2.grad) <= 1e-4


def test_all_weights_packed_in_one_launch(cuda):
    """ops.WeightPackPlan: forward and data-gradient MFMA images of a list of weights from ONE launch == pack_conv_weight(w) and
    pack_conv_weight(dgrad_weight(w)) bit for bit, for every kernel family and ragged channel counts; re-running after an in-place update
    refreshes the same buffers."""
    g = torch.Generator().manual_seed(3)
    ws = [torch.randn(co, ci, k, k, generator=g).to(cuda) for co, ci, k in [(32, 5, 1), (32, 32, 3), (64, 32, 3), (40, 48, 2), (128, 384, 1), (20, 32, 1), (256, 256, 3)]]
    plan = ops.WeightPackPlan(ws)
    plan.run()
    for i, w in enumerate(ws):
        assert torch.equal(plan.fwd[i], ops.pack_conv_weight(w)), i
        assert torch.equal(plan.dgrad[i], ops.pack_conv_weight(ops.dgrad_weight(w))), i
    ptrs = [t.data_ptr() for t in plan.fwd + plan.dgrad]
    for w in ws:
        w.mul_(0.5).add_(1.0)
    assert plan.matches(ws) and not plan.matches(ws[:-1])
    plan.run()
    assert ptrs == [t.data_ptr() for t in plan.fwd + plan.dgrad]
    for i, w in enumerate(ws):
        assert torch.equal(plan.fwd[i], ops.pack_conv_weight(w)) and torch.equal(plan.dgrad[i], ops.pack_conv_weight(ops.dgrad_weight(w))), i


def _layer_oracle(srcs, w, b, gamma, beta, resid, pad, dil, train, rm, rv):
    y = osalsa.fused_conv(srcs, w, b, pad, dil, 0.01)
    if train:
        z = F.batch_norm(y, None, None, gamma, beta, True, 0.0, 1e-5)
    else:
        z = F.batch_norm(y, rm, rv, gamma, beta, False, 0.0, 1e-5)
    return z + resid if resid is not None else z


@pytest.mark.parametrize("train", [True, False])
def test_conv_layer_autograd_node(cuda, train):
    g = torch.Generator().manual_seed(5)
    xs = torch.randn(2, 64, 8, 32, generator=g)
    sk = torch.randn(2, 32, 16, 64, generator=g)
    s0 = (torch.rand(2, 64, generator=g) > 0.2).float() * 1.25
    s1 = (torch.rand(2, 32, generator=g) > 0.2).float() * 1.25
    cin, cout = 48, 40
    w = torch.randn(cout, cin, 3, 3, generator=g) / 20
    b, gamma, beta = torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    resid = torch.randn(2, cout, 16, 64, generator=g)
    proj = torch.randn(2, cout, 16, 64, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (xs, sk, w, b, gamma, beta, resid)]
    rm, rv = torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5
    z = _layer_oracle([(leaves[0], s0, True), (leaves[1], s1, False)], leaves[2], leaves[3], leaves[4], leaves[5], leaves[6], 1, 1,
                      train, rm, rv)
    (z * proj).sum().backward()
    bn = torch.nn.BatchNorm2d(cout).to(cuda)
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    bn.train(train)
    d = [t.clone().to(cuda).requires_grad_(True) for t in (xs, sk, w, b, gamma, beta, resid)]
    cfg = LayerCfg(3, 1, 1, 0.01, [s0.to(cuda), s1.to(cuda)], [True, False], bn, cout, ops.pack_conv_weight(d[2].detach()), {})
    zd = ConvLayerFn.apply(cfg, d[2], d[3], d[4], d[5], d[6], d[0], d[1])
    assert float((zd.cpu() - z.detach()).abs().max()) <= 2e-4
    (zd * proj.to(cuda)).sum().backward()
    for name, a_, b_ in zip(("x_shuffled", "skip", "weight", "bias", "gamma", "beta", "resid"), d, leaves):
        assert _rel(a_.grad.cpu(), b_.grad) <= 2e-4, name
    if train:   # running statistics: momentum 0.1, unbiased variance
        y = osalsa.fused_conv([(xs, s0, True), (sk, s1, False)], w, b, 1, 1, 0.01)
        assert torch.allclose(bn.running_mean.cpu(), 0.9 * rm + 0.1 * y.mean((0, 2, 3)), atol=1e-5)
        assert torch.allclose(bn.running_var.cpu(), 0.9 * rv + 0.1 * y.var((0, 2, 3), unbiased=True), atol=1e-5)
        assert int(bn.num_batches_tracked) == 1


@pytest.fixture(params=["fp32", "f16x3"])
def train_precision(request):
    """Products of the training forward / data-gradient convs: exact fp32 MFMA or split-fp16 (same bars for both)."""
    from semanticlidarunc_amd import salsanext as sn
    sn.set_train_conv_precision(request.param)
    yield request.param
    sn.set_train_conv_precision("fp32")


def test_training_step_matches_reference_golden(cuda, train_precision):
    """Train-mode BatchNorm through ~50 randomly initialised layers amplifies fp32 round-off (two CPU fp32
    implementations of the same formulas already differ by percents), so gradients are judged against an fp64
    run of the oracle: the HIP path must be as close to it as the reference's own fp32 result is (x4 + 1e-4)."""
    g = golden("train_step_2x5x64x128")
    model = seeded_model(SalsaNext).to(cuda).train()
    scales = {k[len("scale:"):]: _t(g[k]) for k in g.files if k.startswith("scale:")}
    x = _t(g["x"]).to(cuda).requires_grad_(True)
    proj = torch.randn(2, 20, 64, 128, generator=torch.Generator().manual_seed(int(g["proj_seed"])))
    pn = {k for k, _ in model.named_parameters()}
    sd64 = {k: (v.detach().cpu().double().requires_grad_(True) if k in pn else v.detach().cpu().double()) for k, v in model.state_dict().items()}
    out = model.forward_with_dropout_scales(x, scales)
    assert float((out.detach().cpu()[:, :, ::2, ::4] - _t(g["logits"])).abs().max()) <= 1e-3
    ((out * proj.to(cuda)).sum() / 64.0).backward()
    x64 = _t(g["x"]).double().requires_grad_(True)
    o64 = osalsa.salsanext_forward(sd64, x64, {k: v.double()[:, :, None, None] for k, v in scales.items()}, bn_train=True)
    ((o64 * proj.double()).sum() / 64.0).backward()
    params = dict(model.named_parameters())
    checks = [("grad_x", x.grad.cpu(), _t(g["grad_x"]), x64.grad)]
    for k in g.files:
        if k.startswith("grad:"):
            checks.append((k, params[k[5:]].grad.cpu(), _t(g[k]), sd64[k[5:]].grad))
    for name, hip, ref32, truth in checks:
        e_hip, e_ref = _rel(hip.double(), truth), _rel(ref32.double(), truth)
        assert e_hip <= 4.0 * e_ref + 1e-4, (name, e_hip, e_ref)
    sd = model.state_dict()
    assert torch.allclose(sd["downCntx.bn1.running_mean"].cpu(), _t(g["running_mean_downCntx_bn1"]), atol=1e-5)
    assert torch.allclose(sd["downCntx.bn1.running_var"].cpu(), _t(g["running_var_downCntx_bn1"]), atol=1e-5)
    assert torch.allclose(sd["resBlock5.bn4.running_var"].cpu(), _t(g["running_var_resBlock5_bn4"]), rtol=1e-3, atol=1e-5)
    assert int(sd["upBlock2.bn3.num_batches_tracked"]) == 1


def test_retain_graph_reentrant_and_eval_mode_grads(cuda):
    model = seeded_model(SalsaNext).to(cuda)       # eval: frozen BatchNorm, still differentiable
    x = torch.randn(1, 5, 32, 64, device=cuda)
    out = model(x)
    assert out.requires_grad
    loss = out.square().mean()
    g1 = torch.autograd.grad(loss, [model.logits.weight, model.downCntx.conv1.weight], retain_graph=True)
    g2 = torch.autograd.grad(loss, [model.logits.weight, model.downCntx.conv1.weight], retain_graph=True)
    assert all(torch.allclose(a, b, rtol=1e-4, atol=1e-7) for a, b in zip(g1, g2))      # fp32 atomics: not bit-identical
    pn = {k for k, _ in model.named_parameters()}
    sd = {k: v.detach().cpu().clone().requires_grad_(k in pn) for k, v in model.state_dict().items()}
    lo = osalsa.salsanext_forward(sd, x.cpu()).square().mean()
    lo.backward()
    assert _rel(g1[0].cpu(), sd["logits.weight"].grad) <= 1e-3 and _rel(g1[1].cpu(), sd["downCntx.conv1.weight"].grad) <= 1e-3
    with torch.no_grad():
        assert not model(x).requires_grad


def test_full_size_training_step_against_oracle(cuda, train_precision):
    """BASELINE configs[1]: batch 4 of 64x2048, train-mode forward (batch-statistics BatchNorm, 13 dropout sites with given multipliers) +
    the SalsaNext loss (NLL + Lovasz, trainer.py:508-516) + backward.  Train-mode BatchNorm through ~50 layers amplifies fp32 round-off, so
    every parameter gradient is judged PER TENSOR against an fp64 run of the oracle, with the fp32 oracle's own distance from it as the
    yardstick: relative error of the HIP gradient <= 4 x the fp32 CPU oracle's + 1e-3 (same rule as the golden-size test above), or, for
    the few tensors where the amplified round-off lands differently, <= 1.5 x the fp32 oracle's own worst tensor."""
    from oracle import losses as olosses
    from semanticlidarunc_amd.loss import salsanext_loss
    from semanticlidarunc_amd.testing import synthetic_scan
    model = seeded_model(SalsaNext).to(cuda).train()
    x, y = synthetic_scan(4, 64, 2048, seed=21)
    scales = osalsa.draw_dropout_scales(4, 0.2, torch.Generator().manual_seed(9))
    loss, nll, ls = salsanext_loss(model.forward_with_dropout_scales(x.to(cuda), scales), y.to(cuda), 1.0, 1.0, 0)
    loss.backward()
    pn = {k for k, _ in model.named_parameters()}
    base = seeded_model(SalsaNext).state_dict()

    def oracle(dtype):
        sd = {k: (v.detach().clone().to(dtype).requires_grad_(True) if k in pn else v.detach().clone().to(dtype if v.is_floating_point() else v.dtype))
              for k, v in base.items()}
        sc = {k: v.to(dtype) for k, v in scales.items()}
        lo, nll_o, ls_o = olosses.salsanext_loss(osalsa.salsanext_forward(sd, x.to(dtype), sc, bn_train=True), y)
        lo.backward()
        return sd, float(nll_o), float(ls_o)

    sd32, nll32, ls32 = oracle(torch.float32)
    sd64, nll64, ls64 = oracle(torch.float64)
    assert abs(float(nll) - nll64) <= 2e-4 * nll64 + 4 * abs(nll32 - nll64) and abs(float(ls) - ls64) <= 2e-4 + 4 * abs(ls32 - ls64)
    params = dict(model.named_parameters())
    table = []
    for k in sorted(pn):
        truth = sd64[k].grad
        table.append((_rel(params[k].grad.cpu().double(), truth), _rel(sd32[k].grad.double(), truth), k))
    table.sort(reverse=True)
    print("largest per-tensor gradient errors vs fp64 (HIP, fp32 CPU oracle):", [(k, round(a, 5), round(b, 5)) for a, b, k in table[:12]])
    # Which tensor catches the amplified round-off differs between two fp32 evaluation orders (the CPU oracle itself is up to ~5 % off
    # the fp64 run on some bottom-of-the-U-Net tensors and 0.5 % on their neighbours), so a tensor passes if it is within 4 x the fp32
    # oracle's own error OR within 1.5 x the largest error that fp32 oracle shows on any tensor of this step (its noise floor).
    floor = 1.5 * max(b for _, b, _ in table)
    for e_hip, e_ref, k in table:
        assert e_hip <= max(4.0 * e_ref + 1e-3, floor), (k, e_hip, e_ref, floor)
    assert sum(1 for a, b, _ in table if a <= 4.0 * b + 1e-3) >= 0.9 * len(table)       # and most tensors meet the tight form


def test_fused_bn_statistics(cuda):
    """Training-path fusion: the conv kernel's own per-channel sum / sum of squares of what it stores == slu_bn_stats of its output."""
    g = torch.Generator().manual_seed(12)
    for cin, cout, k, dil, pad, h, w in ((32, 32, 3, 1, 1, 24, 200), (48, 80, 3, 2, 2, 16, 96), (64, 20, 1, 1, 0, 9, 70), (32, 64, 2, 2, 1, 8, 64)):
        x = torch.randn(3, cin, h, w, generator=g).to(cuda)
        wgt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(cuda)
        bias = torch.randn(cout, generator=g).to(cuda)
        stats = torch.zeros((2, cout), dtype=torch.float64, device=cuda)
        y = ops.conv2d_fused([ConvSource(x)], ops.pack_conv_weight(wgt), cout, k, dil, pad, bias=bias, slope=0.01, stats=stats)
        s, q = ops.bn_stats(y)
        assert float(((stats[0] - s).abs() / (s.abs() + 1.0)).max()) <= 1e-5 and float(((stats[1] - q).abs() / q).max()) <= 1e-5
        ref = y.double()
        assert float(((stats[0] - ref.sum((0, 2, 3))).abs() / (ref.abs().sum((0, 2, 3)) + 1.0)).max()) <= 1e-6
