"""GPU: the training path of the ResNet-FPN models (SURVEY 8(a) row a3 backward, 8(b) Autograd row).

(1) every autograd node of semanticlidarunc_amd/fpn_autograd.py against torch's own CPU autograd of the same op;
(2) one training step (train-mode BatchNorm, loss = sum(out * R), backward) against the fixture produced by the REFERENCE's own class
    (tools/gen_golden_r03.py: models/semanticFCN.py through the torchvision stub), judged with the fixture's float64 run as the yardstick:
    the HIP path may be off by 4x the reference's own fp32 error or 1e-3 of the tensor's scale, whichever is larger."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden
from semanticlidarunc_amd import fpn_autograd as fa
from semanticlidarunc_amd.fpn import SemanticNetworkWithFPN
from semanticlidarunc_amd.testing import randomize_bn_

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _check_node(fn_gpu, fn_cpu, inputs, cuda, tol=1e-5, grad_tol=None):
    """forward values and the gradients of all inputs for a random cotangent, HIP node vs torch CPU autograd"""
    cpu_in = [t.clone().requires_grad_(t.is_floating_point()) for t in inputs]
    gpu_in = [t.clone().to(cuda).requires_grad_(t.is_floating_point()) for t in inputs]
    yc, yg = fn_cpu(*cpu_in), fn_gpu(*gpu_in)
    assert yc.shape == yg.shape
    scale = max(1.0, float(yc.detach().abs().max()))
    assert float((yg.detach().cpu() - yc.detach()).abs().max()) <= tol * scale
    cot = torch.randn(yc.shape, generator=torch.Generator().manual_seed(5))
    gc = torch.autograd.grad(yc, [t for t in cpu_in if t.requires_grad], cot, allow_unused=True)
    gg = torch.autograd.grad(yg, [t for t in gpu_in if t.requires_grad], cot.to(cuda), allow_unused=True, retain_graph=True)
    gg2 = torch.autograd.grad(yg, [t for t in gpu_in if t.requires_grad], cot.to(cuda), allow_unused=True)      # re-entrant
    for a, b, b2 in zip(gc, gg, gg2):
        assert (a is None) == (b is None)
        if a is not None:
            s = max(1e-6, float(a.abs().max()))
            assert float((b.cpu() - a).abs().max()) <= (grad_tol or tol) * max(1.0, s), float((b.cpu() - a).abs().max())
            assert torch.equal(b, b2)


def test_pointwise_pool_and_data_movement_nodes(cuda):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 5, 12, 40, generator=g)
    _check_node(fa.relu, F.relu, [x], cuda)
    _check_node(fa.tanh, torch.tanh, [x], cuda)
    _check_node(fa.elu_plus_one, lambda t: F.elu(t) + 1, [x * 2], cuda)
    # max-pool with exact ties (post-ReLU zeros): the gradient must go to the same (first) maximum ATen picks
    xr = F.relu(torch.randn(2, 3, 13, 38, generator=g) - 0.5)
    _check_node(lambda t: fa.MaxPoolFn.apply(t), lambda t: F.max_pool2d(t, 3, 2, 1), [xr], cuda)
    for f in (2, 4):
        x16 = torch.randn(2, 3, 16, 48, generator=g)
        _check_node(lambda t: fa.NearestDownFn.apply(t, f), lambda t: F.interpolate(t, scale_factor=1 / f, mode="nearest"), [x16], cuda)
    meta = torch.randn(2, 3, 12, 40, generator=g)
    _check_node(lambda a, b: fa.ReplaceTailFn.apply(a, b), lambda a, b: torch.cat([a[:, :-3], b], 1), [x, meta], cuda)
    sc, v = torch.randn(2, 1, 6, 300, generator=g) * 3, torch.randn(2, 7, 6, 300, generator=g)
    _check_node(lambda s, t: fa.RowSoftmaxMulFn.apply(s, t), lambda s, t: t * torch.softmax(s, -1), [sc, v], cuda, tol=2e-6, grad_tol=2e-5)
    y2, y4 = torch.randn(2, 3 * 4, 8, 12, generator=g), torch.randn(2, 2 * 16, 4, 6, generator=g)
    _check_node(lambda a, b: fa.DepthToSpaceCatFn.apply((2, 4), a, b), lambda a, b: torch.cat([F.pixel_shuffle(a, 2), F.pixel_shuffle(b, 4)], 1),
                [y2, y4], cuda)
    for s in (2, 4):
        xs = torch.randn(2, 3, 5, 9, generator=g)
        _check_node(lambda t: fa.BilinearUpFn.apply(t, s), lambda t: F.interpolate(t, scale_factor=s, mode="bilinear", align_corners=False), [xs], cuda,
                    tol=2e-6, grad_tol=1e-5)


def test_norm_and_gate_nodes(cuda):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(3, 16, 10, 24, generator=g) * 2 + 0.5
    gn_c = torch.nn.GroupNorm(4, 16)
    with torch.no_grad():
        gn_c.weight.copy_(torch.rand(16, generator=g) + 0.5)
        gn_c.bias.copy_(torch.randn(16, generator=g) * 0.2)
    for relu_after in (False, True):
        def cpu(t, w, b):
            y = F.group_norm(t, 4, w, b, gn_c.eps)
            return F.relu(y) if relu_after else y
        _check_node(lambda t, w, b: fa.GroupNormFn.apply(t, w, b, 4, gn_c.eps, relu_after), cpu, [x, gn_c.weight.detach(), gn_c.bias.detach()], cuda,
                    tol=1e-5, grad_tol=5e-5)
    sc = torch.randn(3, 1, 10, 24, generator=g) * 2
    _check_node(lambda t, s: fa.SpatialGateFn.apply(t, s),
                lambda t, s: t * torch.softmax(s.flatten(2), -1).view_as(s) + t, [x, sc], cuda, tol=2e-6, grad_tol=2e-5)
    # BatchNorm on its own, train and eval statistics, with and without the residual
    for train in (True, False):
        for with_resid in (False, True):
            bn_c = torch.nn.BatchNorm2d(16)
            with torch.no_grad():
                bn_c.weight.copy_(torch.rand(16, generator=g) + 0.5)
                bn_c.bias.copy_(torch.randn(16, generator=g) * 0.2)
                bn_c.running_mean.copy_(torch.randn(16, generator=g) * 0.1)
                bn_c.running_var.copy_(torch.rand(16, generator=g) + 0.5)
            bn_g = torch.nn.BatchNorm2d(16)
            bn_g.load_state_dict(bn_c.state_dict())
            bn_g = bn_g.to(cuda)
            bn_c.train(train)
            bn_g.train(train)
            r = torch.randn_like(x)
            xc, xg = x.clone().requires_grad_(True), x.clone().to(cuda).requires_grad_(True)
            rc, rg = r.clone().requires_grad_(True), r.clone().to(cuda).requires_grad_(True)
            zc = bn_c(xc) + (rc if with_resid else 0)
            zg = fa.batch_norm(bn_g, xg, rg if with_resid else None)
            assert float((zg.detach().cpu() - zc.detach()).abs().max()) <= 2e-5
            cot = torch.randn(zc.shape, generator=g)
            zc.backward(cot)
            zg.backward(cot.to(cuda))
            assert float((xg.grad.cpu() - xc.grad).abs().max()) <= 2e-5 * max(1.0, float(xc.grad.abs().max()))
            assert float((bn_g.weight.grad.cpu() - bn_c.weight.grad).abs().max()) <= 2e-4 * max(1.0, float(bn_c.weight.grad.abs().max()))
            assert float((bn_g.bias.grad.cpu() - bn_c.bias.grad).abs().max()) <= 2e-4 * max(1.0, float(bn_c.bias.grad.abs().max()))
            if with_resid:
                assert float((rg.grad.cpu() - rc.grad).abs().max()) <= 1e-6
            assert float((bn_g.running_mean.cpu() - bn_c.running_mean).abs().max()) <= 1e-5
            assert float((bn_g.running_var.cpu() - bn_c.running_var).abs().max()) <= 1e-5
            assert int(bn_g.num_batches_tracked) == int(bn_c.num_batches_tracked)


def test_efficientnet_block_nodes(cuda):
    """SiLU, depthwise 3x3 (data + weight gradient), SqueezeExcitation's pooling / gating, StochasticDepth + residual."""
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 24, 9, 40, generator=g) * 2
    _check_node(fa.silu, F.silu, [x], cuda, tol=2e-6, grad_tol=1e-5)
    for (n, c, h, w) in ((2, 24, 9, 40), (3, 130, 4, 16)):
        xx, wt = torch.randn(n, c, h, w, generator=g), torch.randn(c, 1, 3, 3, generator=g) * 0.3
        _check_node(lambda a, b: fa.DepthwiseConv3x3Fn.apply(a, b), lambda a, b: F.conv2d(a, b, None, padding=1, groups=c), [xx, wt], cuda, tol=2e-6,
                    grad_tol=2e-5)
    _check_node(lambda a: fa.GlobalAvgPoolFn.apply(a), lambda a: a.mean((2, 3)), [x], cuda, tol=2e-6)
    gate = torch.rand(2, 24, generator=g)
    _check_node(lambda a, b: fa.ChannelGateFn.apply(a, b), lambda a, b: a * b.view(2, 24, 1, 1), [x, gate], cuda, tol=2e-6, grad_tol=2e-5)
    noise = (torch.tensor([0.0, 1.0 / 0.8]).view(2, 1).expand(2, 24)).contiguous()
    r = torch.randn(2, 24, 9, 40, generator=g)
    yc, rc = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
    yg, rg = x.clone().to(cuda).requires_grad_(True), r.clone().to(cuda).requires_grad_(True)
    oc = yc * noise.view(2, 24, 1, 1) + rc
    og = fa.ScaleAddFn.apply(yg, noise.to(cuda), rg)
    assert float((og.detach().cpu() - oc.detach()).abs().max()) <= 1e-6
    cot = torch.randn(oc.shape, generator=g)
    oc.backward(cot)
    og.backward(cot.to(cuda))
    assert float((yg.grad.cpu() - yc.grad).abs().max()) <= 1e-6 and float((rg.grad.cpu() - rc.grad).abs().max()) <= 1e-6
    assert float(yg.grad[0].abs().max()) == 0.0                      # the dropped sample gets no gradient through the block


def _bar(err, err_ref, scale, sens=0.0):
    """HIP error vs the float64 run: <= 4x the reference's own fp32 error, 4x the reference graph's own response to 1e-6-sized perturbations of
    its inputs and parameters (ReLU masks of near-zero pre-activations flip; tools/gen_golden_r03.py measures it), or 1e-3 of the scale"""
    return err <= max(4.0 * err_ref, 4.0 * sens, 1e-3 * scale)


def _check_against_fixture(gold, names, out, grads):
    """out, dL/dx, dL/dmeta, the norm of EVERY parameter gradient and the sampled gradients element by element, against the float64 run"""
    o64, o32 = _t(gold["out64"]).double(), _t(gold["out"]).double()
    assert _bar(float((out.detach().cpu().double() - o64).abs().max()), float((o32 - o64).abs().max()), float(o64.abs().max()), float(gold["out_sens"]))
    for key, gt in (("dx", grads[-2]), ("dmeta", grads[-1])):
        g64, g32 = _t(gold[key + "64"]).double(), _t(gold[key]).double()
        assert _bar(float((gt.cpu().double() - g64).norm()), float((g32 - g64).norm()), float(g64.norm()), float(gold[key + "_sens"])), key
    # every parameter: present / absent like the reference (backbone.bn1 / fc never get a gradient), norm of the gradient
    has, n64, e32, sens = gold["grad_has"], gold["grad_norm64"], gold["grad_err32"], gold["grad_sens"]
    floor = 1e-6 * float(n64.max())
    bad = []
    for n, gt, h, nn_, e, sn in zip(names, grads, has, n64, e32, sens):
        assert (gt is not None) == bool(h), n
        if gt is not None and not max(abs(float(gt.double().norm()) - nn_) - floor, 0.0) <= max(4.0 * e, 4.0 * sn, 1e-3 * nn_):
            bad.append((n, float(gt.double().norm()), float(nn_)))
    assert not bad, bad[:8]
    for key in gold.files:
        if key.startswith("g64:"):
            n = key[4:]
            g64, g32 = _t(gold[key]).double(), _t(gold["g:" + n]).double()
            got = grads[names.index(n)].reshape(-1)[:g64.numel()].cpu().double()
            frac = g64.numel() / max(1, grads[names.index(n)].numel())      # the sensitivity is stored for the whole tensor
            assert _bar(float((got - g64).norm()), float((g32 - g64).norm()), float(g64.norm()) + floor, float(sens[names.index(n)]) * max(frac, 0.25) ** 0.5), n


def test_eval_mode_gradients_vs_float64_autograd_of_the_oracle(cuda):
    """Frozen BatchNorm (the fine-tuning / grad-probe configuration), attention off: no train-mode BatchNorm backward amplifies a flipped ReLU
    mask, so the whole backward chain is held to 1e-4 against float64 CPU autograd of the functional oracle (measured 1e-6)."""
    from oracle import fpn as ofpn
    torch.manual_seed(0)
    kw = dict(backbone="resnet18", input_channels=2, meta_channel_dim=3, num_classes=20, attention=False)
    model = randomize_bn_(SemanticNetworkWithFPN(**kw), 3).eval()
    g = torch.Generator().manual_seed(61)
    x = torch.randn(2, 2, 32, 128, generator=g) * torch.tensor([20.0, 0.3]).view(1, 2, 1, 1)
    meta = torch.randn(2, 3, 32, 128, generator=g) * 5.0
    R = torch.randn(2, 20, 32, 128, generator=g) / (32 * 128)
    sd = {k: (v.detach().clone().double().requires_grad_("running_" not in k) if v.is_floating_point() else v) for k, v in model.state_dict().items()}
    xc, mc = x.double().requires_grad_(True), meta.double().requires_grad_(True)
    ofpn.BN_TRAIN = False
    (ofpn.fpn_forward(sd, xc, mc, "resnet18", False, True) * R.double()).sum().backward()
    m = model.to(cuda)
    xg, mg = x.to(cuda).requires_grad_(True), meta.to(cuda).requires_grad_(True)
    (m(xg, mg) * R.to(cuda)).sum().backward()
    rel = lambda a, b: float((a.cpu().double() - b).norm() / max(float(b.norm()), 1e-30))
    assert rel(xg.grad, xc.grad) <= 1e-4 and rel(mg.grad, mc.grad) <= 1e-4
    worst = max((rel(p.grad, sd[n].grad), n) for n, p in m.named_parameters() if p.grad is not None and float(sd[n].grad.norm()) > 1e-9)
    assert worst[0] <= 1e-4, worst


@pytest.mark.parametrize("tag,kw", [("fpn_train_resnet18_m3_c20", dict(backbone="resnet18", input_channels=2, meta_channel_dim=3, num_classes=20)),
                                    ("fpn_train_resnet50_m3_c5_noatt", dict(backbone="resnet50", input_channels=2, meta_channel_dim=3, num_classes=5,
                                                                             attention=False))])
def test_training_step_vs_the_reference_class(cuda, tag, kw):
    gold = golden(tag)
    torch.manual_seed(0)
    model = randomize_bn_(SemanticNetworkWithFPN(**kw), 3).to(cuda).train()
    x, meta = _t(gold["x"]).to(cuda).requires_grad_(True), _t(gold["meta"]).to(cuda).requires_grad_(True)
    R = _t(gold["R"]).to(cuda)
    out = model(x, meta)
    assert out.requires_grad and out.shape == R.shape
    loss = (out * R).sum()
    params = dict(model.named_parameters())
    names = [str(n) for n in gold["grad_names"]]
    assert names == list(params)                                    # same parameter list (aliases de-duplicated the same way) as the reference
    grads = torch.autograd.grad(loss, [params[n] for n in names] + [x, meta], retain_graph=True, allow_unused=True)
    again = torch.autograd.grad(loss, [params[n] for n in names] + [x, meta], allow_unused=True)      # grad_norm.py:52: the same graph, twice
    for a, b in zip(grads, again):      # the weight-gradient kernels add partial sums with float atomics: equal up to the summation order
        assert (a is None and b is None) or float((a - b).abs().max()) <= 1e-4 * max(float(a.abs().max()), 1e-12)
    o64, o32 = _t(gold["out64"]).double(), _t(gold["out"]).double()
    _check_against_fixture(gold, names, out, grads)
    # BatchNorm bookkeeping of the step
    for key in gold.files:
        if key.startswith("bn64:"):
            n = key[5:]
            want = _t(gold[key])
            got = dict(model.named_buffers())[n].cpu()
            if want.is_floating_point():
                assert float((got.double() - want.double()).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max())), n
            else:
                assert int(got) == int(want), n


def test_opt_training_step_vs_the_reference_class(cuda):
    """baselines/Reichert/semanticFCN_opt.py -- the model train_semantics.py:134 builds -- one training step, pyramid dropout pinned."""
    import os
    from conftest import GOLDEN
    from semanticlidarunc_amd.fpn_opt import SemanticNetworkWithFPN as OptFPN
    tag, kw = "fpn_opt_train_resnet18_m3_c20", dict(backbone="resnet18", input_channels=2, meta_channel_dim=3, num_classes=20)
    gold = golden(tag)
    scale = _t(np.load(os.path.join(GOLDEN, "fpn_opt_train_resnet18_dropout_scale.npy")))
    torch.manual_seed(0)
    model = randomize_bn_(OptFPN(**kw), 3)
    with torch.no_grad():
        g = torch.Generator().manual_seed(9)
        for mod in model.modules():
            if isinstance(mod, torch.nn.GroupNorm):
                mod.weight.copy_(torch.rand(mod.num_channels, generator=g) + 0.5)
                mod.bias.copy_(torch.randn(mod.num_channels, generator=g) * 0.1)
    model = model.to(cuda).train()
    x, meta = _t(gold["x"]).to(cuda).requires_grad_(True), _t(gold["meta"]).to(cuda).requires_grad_(True)
    R = _t(gold["R"]).to(cuda)
    out = model.forward_with_dropout_scale(x, meta, scale.to(cuda))
    loss = (out * R).sum()
    params = dict(model.named_parameters())
    names = [str(n) for n in gold["grad_names"]]
    assert names == list(params)
    grads = torch.autograd.grad(loss, [params[n] for n in names] + [x, meta], allow_unused=True)
    _check_against_fixture(gold, names, out, grads)
    # the real Dropout2d child draws when no multiplier is given, and MC-dropout mode (eval + live dropout) still takes the inference path
    out2 = model(x, meta)
    assert out2.requires_grad and not torch.equal(out2, out)


def test_efficientnet_training_step_vs_the_reference_class(cuda):
    """The shipped YAML's model family (`model_type: efficientnet_v2_l`, configs/SemanticKitti_default.yaml:14; the fixture uses _s): one training
    step of the reference's own class -- train-mode BatchNorm (eps 1e-3), depthwise convs, squeeze-excitation, SiLU, StochasticDepth on the
    residual blocks with a pinned noise that drops samples, pyramid dropout pinned."""
    import os
    from conftest import GOLDEN
    from semanticlidarunc_amd.fpn_opt import SemanticNetworkWithFPN as OptFPN
    tag, kw = "fpn_opt_train_efficientnet_v2_s_m3_c20", dict(backbone="efficientnet_v2_s", input_channels=2, meta_channel_dim=3, num_classes=20)
    gold = golden(tag)
    scale = _t(np.load(os.path.join(GOLDEN, "fpn_opt_train_efficientnet_v2_s_dropout_scale.npy")))
    noise = {k: _t(v) for k, v in np.load(os.path.join(GOLDEN, "fpn_opt_train_efficientnet_v2_s_sd_noise.npz")).items()}
    torch.manual_seed(0)
    model = randomize_bn_(OptFPN(**kw), 3)
    with torch.no_grad():
        g = torch.Generator().manual_seed(9)
        for mod in model.modules():
            if isinstance(mod, torch.nn.GroupNorm):
                mod.weight.copy_(torch.rand(mod.num_channels, generator=g) + 0.5)
                mod.bias.copy_(torch.randn(mod.num_channels, generator=g) * 0.1)
    model = model.to(cuda).train()
    model._sd_noise_override = noise
    x, meta = _t(gold["x"]).to(cuda).requires_grad_(True), _t(gold["meta"]).to(cuda).requires_grad_(True)
    R = _t(gold["R"]).to(cuda)
    out = model.forward_with_dropout_scale(x, meta, scale.to(cuda))
    loss = (out * R).sum()
    params = dict(model.named_parameters())
    names = [str(n) for n in gold["grad_names"]]
    assert names == list(params)
    grads = torch.autograd.grad(loss, [params[n] for n in names] + [x, meta], allow_unused=True)
    _check_against_fixture(gold, names, out, grads)
    for key in gold.files:
        if key.startswith("bn64:"):
            want, got = _t(gold[key]), dict(model.named_buffers())[key[5:]].cpu()
            if want.is_floating_point():
                assert float((got.double() - want.double()).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max())), key
            else:
                assert int(got) == int(want), key
    # live StochasticDepth: the model draws its own noise when none is pinned
    del model._sd_noise_override
    a = model.forward_with_dropout_scale(x, meta, scale.to(cuda))
    b = model.forward_with_dropout_scale(x, meta, scale.to(cuda))
    assert a.requires_grad and not torch.equal(a, b)
    # eval mode with gradients: StochasticDepth is the identity and the autograd path reproduces the folded inference launches
    model.eval()
    with torch.no_grad():
        want = model.forward_with_dropout_scale(x.detach(), meta.detach(), scale.to(cuda))
    got = model.forward_with_dropout_scale(x, meta, scale.to(cuda))
    assert got.requires_grad and float((got.detach() - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))


def test_eval_mode_with_gradients_equals_the_folded_inference_path(cuda):
    torch.manual_seed(1)
    model = randomize_bn_(SemanticNetworkWithFPN("resnet18", 2, 3, num_classes=7), 4).to(cuda).eval()
    g = torch.Generator().manual_seed(8)
    x, meta = (torch.randn(1, 2, 32, 64, generator=g) * 3).to(cuda), torch.randn(1, 3, 32, 64, generator=g).to(cuda)
    with torch.no_grad():
        want = model(x, meta)
    assert not want.requires_grad
    out = model(x, meta)                       # parameters require grad -> the autograd path with eval-mode BatchNorm
    assert out.requires_grad
    assert float((out.detach() - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))
    out.sum().backward()
    assert model.backbone.conv1.weight.grad is not None and model.backbone.fc.weight.grad is None


def test_opt_model_takes_optimizer_steps(cuda):
    """The trainer's loop shape (trainer.py:783-786: zero_grad, forward, loss.backward(), optimizer.step()) on the model train_semantics.py:134
    builds, twice: the second step's gradients must be those of the UPDATED weights (nothing cached from step one)."""
    from semanticlidarunc_amd.fpn_opt import SemanticNetworkWithFPN as OptFPN
    from semanticlidarunc_amd.loss import salsanext_loss
    torch.manual_seed(3)
    kw = dict(backbone="resnet18", input_channels=2, meta_channel_dim=3, num_classes=8)
    model = OptFPN(**kw).to(cuda).train()
    model.dropout_pyramid.p = 0.0                          # deterministic: the comparison below re-runs step two on a fresh copy
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(4)
    x, meta = (torch.randn(2, 2, 32, 64, generator=g) * 3).to(cuda), torch.randn(2, 3, 32, 64, generator=g).to(cuda)
    labels = torch.randint(0, 8, (2, 32, 64), generator=g).to(cuda)
    crit = lambda logits, lab: salsanext_loss(logits, lab)[0]      # softmax -> NLL + Lovasz, the trainer's 'SalsaNext' loss branch (trainer.py:508-516)
    losses = []
    for step in range(2):
        opt.zero_grad()
        loss = crit(model(x, meta), labels)
        loss.backward()
        if step == 1:
            fresh = OptFPN(**kw).to(cuda).train()
            fresh.load_state_dict(model.state_dict())
            fresh.dropout_pyramid.p = 0.0
            for bn_a, bn_b in zip([m for m in fresh.modules() if isinstance(m, torch.nn.BatchNorm2d)], [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]):
                bn_a.num_batches_tracked.copy_(bn_b.num_batches_tracked - 1)      # (momentum is fixed: only bookkeeping)
            crit(fresh(x, meta), labels).backward()
            for (n, p), (_, q) in zip(model.named_parameters(), fresh.named_parameters()):
                if p.grad is not None:
                    assert float((p.grad - q.grad).abs().max()) <= 2e-4 * max(float(q.grad.abs().max()), 1e-8), n
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[1] < losses[0]
