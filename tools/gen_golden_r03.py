#!/usr/bin/env python
"""Round-3 additions to the pinned fixtures: same rules as tools/gen_golden.py (runs ONLY in the build container, imports the reference
read-only from /root/reference, writes small data-only fixtures under tests/golden/).

    python tools/gen_golden_r03.py [fpn_train] [fpn_opt_train] [effnet] [effnet_train] [lovasz_classes]

fpn_train / fpn_opt_train: one TRAINING step of the reference's own classes (models/semanticFCN.py, baselines/Reichert/semanticFCN_opt.py) in
train mode -- batch-statistics BatchNorm, running-statistics update, loss = sum(out * R), backward -- through the stub torchvision.models that
serves oracle.fpn's restated ResNet (backbone internals unpinned, head wiring pinned by the reference code).  Stored: inputs, R, the fp32
outputs, the gradients of both inputs, a sample of parameter gradients, the L2 norm of EVERY parameter gradient, the updated running
statistics of two BatchNorm layers -- and the same quantities from a float64 run of the same reference code, the yardstick the GPU test uses
(train-mode BatchNorm amplifies fp32 round-off; two fp32 evaluation orders of identical formulas differ by more than a fixed 1e-3)."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
sys.path.insert(0, ROOT)
sys.path.insert(1, REF)
for stub in ("seaborn", "cv2"):
    sys.modules.setdefault(stub, types.ModuleType(stub))

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"  wrote {name}.npz  ({len(arrs)} arrays, {os.path.getsize(os.path.join(OUT, name + '.npz')) / 1024:.0f} KB)")


SAMPLED = 4096      # elements kept of a sampled parameter gradient (flattened prefix)
SENS_RUNS = 6


def _train_step(model, x, meta, R, dtype):
    model = model.to(dtype).train()
    xx, mm = x.detach().clone().to(dtype).requires_grad_(True), meta.detach().clone().to(dtype).requires_grad_(True)
    out = model(xx, mm)
    loss = (out * R.to(dtype)).sum()
    loss.backward()
    grads = {n: (p.grad.detach() if p.grad is not None else None) for n, p in model.named_parameters()}
    return out.detach(), xx.grad.detach(), mm.grad.detach(), grads


def _fixture(tag, ref_cls, my_cls, kw, shape, sample_names, bn_names, fix_dropout=None):
    from semanticlidarunc_amd.testing import randomize_bn_
    torch.manual_seed(0)
    mine = randomize_bn_(my_cls(**kw), 3)
    with torch.no_grad():
        g = torch.Generator().manual_seed(9)
        for mod in mine.modules():
            if isinstance(mod, torch.nn.GroupNorm):
                mod.weight.copy_(torch.rand(mod.num_channels, generator=g) + 0.5)
                mod.bias.copy_(torch.randn(mod.num_channels, generator=g) * 0.1)
    sd = {k: v.clone() for k, v in mine.state_dict().items()}
    g = torch.Generator().manual_seed(61)
    x = torch.randn(shape[0], 2, shape[1], shape[2], generator=g) * torch.tensor([20.0, 0.3]).view(1, 2, 1, 1)
    meta = torch.randn(shape[0], kw["meta_channel_dim"], shape[1], shape[2], generator=g) * 5.0
    R = torch.randn(shape[0], kw["num_classes"], shape[1], shape[2], generator=g) / (shape[1] * shape[2])
    res = {}
    for dtype, key in ((torch.float32, "f32"), (torch.float64, "f64")):
        ref = ref_cls(**kw)
        assert list(ref.state_dict().keys()) == list(sd.keys())
        ref.load_state_dict(sd)
        if fix_dropout is not None:
            fix_dropout(ref, dtype)
        out, gx, gm, grads = _train_step(ref, x, meta, R, dtype)
        res[key] = (out, gx, gm, grads, {k: v.detach().clone() for k, v in ref.state_dict().items()})
    o32, gx32, gm32, gr32, sd32 = res["f32"]
    o64, gx64, gm64, gr64, sd64 = res["f64"]
    # Sensitivity of the reference graph itself to fp32-sized differences: the same float64 step with every input and parameter multiplied by
    # (1 + 1e-6 N(0, 1)) -- what two fp32 implementations' forward passes differ by after ~20 layers.  A ReLU input within that distance of zero
    # flips its mask, which changes the gradient downstream by far more than fp32 round-off (one flip in the decoder moved dL/dx by 2e-2 through
    # the train-mode BatchNorm backward of the layers before it); the largest deviation over SENS_RUNS draws is the second yardstick of the test.
    sens = dict(out=0.0, dx=0.0, dmeta=0.0, grads={n: 0.0 for n in gr64})
    for k in range(SENS_RUNS):
        gp = torch.Generator().manual_seed(1000 + k)
        ref = ref_cls(**kw)
        ref.load_state_dict(sd)
        ref = ref.double()
        with torch.no_grad():
            for prm in ref.parameters():
                prm.mul_(1.0 + 1e-6 * torch.randn(prm.shape, generator=gp, dtype=torch.float64))
        if fix_dropout is not None:
            fix_dropout(ref, torch.float64)
        noise = lambda t: t.double() * (1.0 + 1e-6 * torch.randn(t.shape, generator=gp, dtype=torch.float64))
        o, gx, gm, grads = _train_step(ref, noise(x), noise(meta), R, torch.float64)
        sens["out"] = max(sens["out"], float((o - o64).abs().max()))
        sens["dx"] = max(sens["dx"], float((gx - gx64).norm()))
        sens["dmeta"] = max(sens["dmeta"], float((gm - gm64).norm()))
        for n in gr64:
            if gr64[n] is not None:
                sens["grads"][n] = max(sens["grads"][n], float((grads[n] - gr64[n]).norm()))
    print(f"{tag}: sensitivity to 1e-6 perturbations: out {sens['out']:.2e}, dx rel {sens['dx'] / float(gx64.norm()):.2e}, "
          f"dmeta rel {sens['dmeta'] / float(gm64.norm()):.2e}")
    names = [n for n, v in gr32.items()]
    has = np.array([gr32[n] is not None for n in names])
    norms32 = np.array([float(gr32[n].double().norm()) if gr32[n] is not None else 0.0 for n in names])
    norms64 = np.array([float(gr64[n].norm()) if gr64[n] is not None else 0.0 for n in names])
    # per-parameter relative error of the reference's own fp32 run against its fp64 run: the yardstick
    rel32 = np.array([float((gr32[n].double() - gr64[n]).norm() / max(float(gr64[n].norm()), 1e-30)) if gr32[n] is not None else 0.0 for n in names])
    print(f"{tag}: out |f32 - f64| {float((o32.double() - o64).abs().max()):.2e} (scale {float(o64.abs().max()):.2f}); "
          f"dx rel {float((gx32.double() - gx64).norm() / gx64.norm()):.2e}; worst parameter-gradient rel error of the fp32 reference among the parameters with a gradient above 1e-6 of the largest: "
          f"{max(r for r, nn_ in zip(rel32, norms64) if nn_ > 1e-6 * norms64.max()):.2e}; parameters without gradient: {[n for n, h in zip(names, has) if not h]}")
    f = lambda t: t.float().numpy()      # the float64 results are stored rounded to float32: far below the errors they are the yardstick for
    arrs = dict(x=x.numpy(), meta=meta.numpy(), R=R.numpy(), out=o32.numpy(), out64=f(o64), dx=gx32.numpy(), dx64=f(gx64),
                dmeta=gm32.numpy(), dmeta64=f(gm64), grad_names=np.array(names), grad_has=has, grad_norm=norms32, grad_norm64=norms64,
                grad_err32=np.array([float((gr32[n].double() - gr64[n]).norm()) if gr32[n] is not None else 0.0 for n in names]),
                grad_sens=np.array([sens["grads"][n] for n in names]), out_sens=np.array(sens["out"]), dx_sens=np.array(sens["dx"]),
                dmeta_sens=np.array(sens["dmeta"]))
    sample_names = [n if n in gr32 else "backbone." + n for n in sample_names]      # named_parameters() lists an aliased stage under backbone.*
    bn_names = [n if f"{n}.running_mean" in sd32 and not n.startswith("layer") else "backbone." + n for n in bn_names]
    for n in sample_names:
        arrs["g:" + n] = gr32[n].reshape(-1)[:SAMPLED].numpy()
        arrs["g64:" + n] = f(gr64[n].reshape(-1)[:SAMPLED])
    for n in bn_names:
        for stat in ("running_mean", "running_var", "num_batches_tracked"):
            arrs[f"bn:{n}.{stat}"] = sd32[f"{n}.{stat}"].numpy()
            arrs[f"bn64:{n}.{stat}"] = sd64[f"{n}.{stat}"].float().numpy() if sd64[f"{n}.{stat}"].is_floating_point() else sd64[f"{n}.{stat}"].numpy()
    save(tag, **arrs)


def _stub_torchvision():
    from oracle import fpn as ofpn
    tv = types.ModuleType("torchvision")
    tv.models = ofpn.torchvision_models_stub()
    sys.modules["torchvision"], sys.modules["torchvision.models"] = tv, tv.models


def gen_fpn_train():
    _stub_torchvision()
    from models.semanticFCN import SemanticNetworkWithFPN as RefFPN                       # the reference's own wiring
    from semanticlidarunc_amd.fpn import SemanticNetworkWithFPN as MyFPN
    _fixture("fpn_train_resnet18_m3_c20", RefFPN, MyFPN, dict(backbone="resnet18", input_channels=2, meta_channel_dim=3, num_classes=20), (2, 32, 128),
             ["backbone.conv1.weight", "layer1.0.conv1.weight", "layer2.0.conv1.weight", "layer2.0.downsample.0.weight", "layer2.0.bn1.weight",
              "layer4.1.conv2.weight", "layer4.1.bn2.bias", "fpn_block1.0.weight", "fpn_block1.0.bias", "fpn_block4.1.weight",
              "attention1.query_conv.weight", "attention1.key_conv.bias", "attention1.attention_conv.weight", "attention3.value_conv.weight",
              "upsample_layer_x2.weight", "upsample_layer_x4.weight", "upsample_layer_x4.bias", "decoder_semantic.0.weight", "decoder_semantic.4.weight",
              "decoder_semantic.6.weight", "decoder_semantic.6.bias"],
             ["layer2.0.bn1", "decoder_semantic.1"])
    _fixture("fpn_train_resnet50_m3_c5_noatt", RefFPN, MyFPN, dict(backbone="resnet50", input_channels=2, meta_channel_dim=3, num_classes=5, attention=False),
             (2, 32, 128),
             ["backbone.conv1.weight", "layer1.0.conv3.weight", "layer1.0.downsample.0.weight", "layer2.0.conv2.weight", "layer3.0.downsample.0.weight",
              "layer4.2.conv3.weight", "fpn_block2.0.weight", "upsample_layer_x3.weight", "decoder_semantic.6.weight"],
             ["layer2.0.bn2", "layer1.0.downsample.1"])


def gen_fpn_opt_train():
    _stub_torchvision()
    from baselines.Reichert.semanticFCN_opt import SemanticNetworkWithFPN as RefOpt       # what train_semantics.py:134 builds
    from semanticlidarunc_amd.fpn_opt import SemanticNetworkWithFPN as MyOpt
    kw = dict(backbone="resnet18", input_channels=2, meta_channel_dim=3, num_classes=20)
    shape = (2, 32, 128)
    gd = torch.Generator().manual_seed(77)
    probe = RefOpt(**kw)
    cpyr = probe.decoder_semantic[0].in_channels
    scale = (torch.rand(shape[0], cpyr, 1, 1, generator=gd) > 0.1).float() / 0.9

    def fix_dropout(ref, dtype):
        """The pyramid Dropout2d draws from torch's global generator: pin it with a fixed multiplier so both precisions (and the GPU) see the
        same one."""
        class Fixed(torch.nn.Module):
            def forward(self, t):
                return t * scale.to(t.dtype)
        for name, mod in list(ref.named_children()):
            if isinstance(mod, torch.nn.Dropout2d):
                setattr(ref, name, Fixed())
    names = [n for n, _ in probe.named_parameters()]
    want = [n for n in names if n in ("backbone.conv1.weight", "layer2.0.conv1.weight", "layer3.0.downsample.0.weight")]
    want += [n for n in names if n.startswith(("attention", "spatial", "upsample", "up_", "decoder_semantic", "fpn_block1"))][:40]
    bn = [n[:-len(".weight")] for n in names if n.endswith("bn1.weight")][:1]
    _fixture("fpn_opt_train_resnet18_m3_c20", RefOpt, MyOpt, kw, shape, want, bn, fix_dropout)
    np.save(os.path.join(OUT, "fpn_opt_train_resnet18_dropout_scale.npy"), scale.numpy())


def gen_effnet():
    """f-4, second half: semanticFCN_opt with the EfficientNetV2 backbones (the shipped YAML's `model_type: efficientnet_v2_l`,
    configs/SemanticKitti_default.yaml:14), eval mode: the reference's own class through the stub torchvision.models serving oracle/effnet.py's
    restated blocks.  Pins which stages the model uses, the meta injection, x4 = cat(x3[:, :-m], meta3), the [4, 4, 2] up-sampling ladder, the head
    and the state_dict layout; the block internals rest on the public torchvision definition (its published parameter counts are asserted)."""
    import json
    _stub_torchvision()
    from baselines.Reichert.semanticFCN_opt import SemanticNetworkWithFPN as RefOpt
    from oracle import effnet as oeff, fpn_opt as ofpo
    from semanticlidarunc_amd.fpn_opt import SemanticNetworkWithFPN as MyOpt
    from semanticlidarunc_amd.testing import randomize_bn_
    published = {"efficientnet_v2_s": 21458488, "efficientnet_v2_m": 54139356, "efficientnet_v2_l": 118515272}      # torchvision docs: num_params
    for name, n in published.items():
        got = sum(p.numel() for p in oeff.EfficientNetRef(name).parameters())
        assert got == n, (name, got, n)
    print("  restated efficientnet_v2_{s,m,l}: parameter counts equal torchvision's published ones")
    for tag, kw, shape in (("efficientnet_v2_l_m3_c20", dict(backbone="efficientnet_v2_l", input_channels=2, meta_channel_dim=3, num_classes=20), (1, 32, 128)),
                           ("efficientnet_v2_s_m6_c5_noatt", dict(backbone="efficientnet_v2_s", input_channels=2, meta_channel_dim=6, num_classes=5,
                                                                   attention=False), (2, 32, 64))):
        torch.manual_seed(0)
        mine = randomize_bn_(MyOpt(**kw), 3).eval()
        with torch.no_grad():
            g = torch.Generator().manual_seed(9)
            for mod in mine.modules():
                if isinstance(mod, torch.nn.GroupNorm):
                    mod.weight.copy_(torch.rand(mod.num_channels, generator=g) + 0.5)
                    mod.bias.copy_(torch.randn(mod.num_channels, generator=g) * 0.1)
        ref = RefOpt(**kw)
        sdf = mine.state_dict()
        assert list(sdf.keys()) == list(ref.state_dict().keys()), "state_dict key order differs from the reference class"
        assert all(tuple(a.shape) == tuple(b.shape) for a, b in zip(sdf.values(), ref.state_dict().values()))
        ref.load_state_dict(sdf)
        ref.eval()
        g = torch.Generator().manual_seed(51)
        xf = torch.randn(shape[0], 2, shape[1], shape[2], generator=g) * torch.tensor([20.0, 0.3]).view(1, 2, 1, 1)
        mf = torch.randn(shape[0], kw["meta_channel_dim"], shape[1], shape[2], generator=g) * 5.0
        with torch.no_grad():
            yr = ref(xf, mf)
            yo = ofpo.fpn_opt_forward(sdf, xf, mf, kw["backbone"], kw.get("attention", True), True)
        print(f"FPN-opt {tag}: |oracle - reference| = {float((yr - yo).abs().max()):.3e}  (|logit| max {float(yr.abs().max()):.2f}, {len(sdf)} state_dict entries)")
        assert float((yr - yo).abs().max()) <= 1e-5 * max(1.0, float(yr.abs().max()))
        s_ = sum(float(v.double().sum()) for v in sdf.values() if v.is_floating_point())
        a_ = sum(float(v.double().abs().sum()) for v in sdf.values() if v.is_floating_point())
        save("fpn_opt_" + tag, x=xf.numpy(), meta=mf.numpy(), out=yr.numpy(), sd_digest=np.array([s_, a_]))
    with open(os.path.join(OUT, "fpn_opt_efficientnet_v2_s_state_dict_keys.json"), "w") as f:
        json.dump({k: list(v.shape) for k, v in RefOpt("efficientnet_v2_s", 2, 3, num_classes=20).state_dict().items()}, f, indent=0)


def gen_effnet_train():
    """One TRAINING step of the reference's semanticFCN_opt with the efficientnet_v2_s backbone (train-mode BatchNorm eps 1e-3, depthwise conv,
    squeeze-excitation, SiLU, StochasticDepth on the residual blocks) through the stub torchvision.models.  The two random draws are pinned: the
    pyramid Dropout2d by a fixed multiplier, StochasticDepth by a fixed per-(block, sample) noise (stored; it contains dropped samples)."""
    _stub_torchvision()
    from baselines.Reichert.semanticFCN_opt import SemanticNetworkWithFPN as RefOpt
    from oracle import effnet as oeff
    from semanticlidarunc_amd.fpn_opt import SemanticNetworkWithFPN as MyOpt
    kw = dict(backbone="efficientnet_v2_s", input_channels=2, meta_channel_dim=3, num_classes=20)
    shape = (2, 32, 128)
    gd = torch.Generator().manual_seed(78)
    probe = RefOpt(**kw)
    cpyr = probe.decoder_semantic[0].in_channels
    scale = (torch.rand(shape[0], cpyr, 1, 1, generator=gd) > 0.1).float() / 0.9
    stages = (("layer1", 2), ("layer2", 3), ("layer3", 4))
    noise = {}
    for lname, fi in stages:
        for bi, blk in enumerate(probe.backbone.features[fi]):
            p = blk.stochastic_depth.p
            # torchvision's draw, but with an inflated drop rate for the stored pattern (p <= 0.05 here would leave every sample alive): the
            # multiplier of a kept sample stays the block's own 1 / (1 - p)
            keep = (torch.rand(shape[0], generator=gd) > 0.3).float()
            noise[f"{lname}.{bi}"] = keep / (1.0 - p) if p > 0.0 else torch.ones(shape[0])
    assert any(float(v.min()) == 0.0 for v in noise.values())

    def fix_random(ref, dtype):
        class Fixed(torch.nn.Module):
            def forward(self, t):
                return t * scale.to(t.dtype)

        class FixedSD(torch.nn.Module):
            def __init__(self, v, p):
                super().__init__()
                self.v, self.p = v, p

            def forward(self, t):
                return t * self.v.to(t.dtype).view(-1, 1, 1, 1) if (self.training and self.p > 0.0) else t
        for name, mod in list(ref.named_children()):
            if isinstance(mod, torch.nn.Dropout2d):
                setattr(ref, name, Fixed())
        for lname, fi in stages:
            for bi, blk in enumerate(ref.backbone.features[fi]):
                assert isinstance(blk.stochastic_depth, oeff.SD)
                blk.stochastic_depth = FixedSD(noise[f"{lname}.{bi}"], blk.stochastic_depth.p)
    names = [n for n, _ in probe.named_parameters()]
    want = ["backbone.features.0.0.weight", "backbone.features.0.1.weight", "backbone.features.2.0.block.0.0.weight", "backbone.features.2.3.block.1.0.weight",
            "backbone.features.3.0.block.0.0.weight", "backbone.features.3.1.block.1.1.bias", "backbone.features.4.0.block.0.0.weight",
            "backbone.features.4.0.block.1.0.weight", "backbone.features.4.0.block.1.1.weight", "backbone.features.4.0.block.2.fc1.weight",
            "backbone.features.4.0.block.2.fc1.bias", "backbone.features.4.0.block.2.fc2.weight", "backbone.features.4.0.block.2.fc2.bias",
            "backbone.features.4.0.block.3.0.weight", "backbone.features.4.2.block.1.0.weight", "backbone.features.4.5.block.3.1.weight"]
    assert all(n in names for n in want), [n for n in want if n not in names]
    want += [n for n in names if n.startswith(("attention", "upsample", "decoder_semantic", "fpn_block1"))][:10]
    _fixture("fpn_opt_train_efficientnet_v2_s_m3_c20", RefOpt, MyOpt, kw, shape, want,
             ["backbone.features.4.0.block.1.1", "backbone.features.2.0.block.0.1"], fix_random)
    np.save(os.path.join(OUT, "fpn_opt_train_efficientnet_v2_s_dropout_scale.npy"), scale.numpy())
    np.savez(os.path.join(OUT, "fpn_opt_train_efficientnet_v2_s_sd_noise.npz"), **{k: v.numpy() for k, v in noise.items()})


def gen_lovasz_classes():
    """LovaszSoftmaxStable(classes='all') and classes=[...] (lovasz.py:7,56-88): value and gradient of the reference's own class on inputs where
    several classes -- one of the listed ones included -- have no valid pixel (their term is the largest probability of that class)."""
    from losses.lovasz import LovaszSoftmaxStable as RefLovasz
    from oracle import losses as olosses
    g = torch.Generator().manual_seed(17)
    logits = torch.randn(2, 20, 8, 64, generator=g) * 2.0
    pool = torch.tensor([0, 1, 2, 5, 7, 11, 12, 19])                       # classes 3, 4, 6, ... never occur
    labels = pool[torch.randint(0, len(pool), (2, 8, 64), generator=g)]
    arrs = dict(logits=logits.numpy(), labels=labels.numpy())
    for tag, ign, classes in (("all_ign0", 0, "all"), ("all_none", None, "all"), ("list_ign0", 0, [1, 3, 19]), ("present_ign0", 0, "present")):
        p = torch.softmax(logits, 1).requires_grad_(True)
        loss = RefLovasz(ignore_index=ign, classes=classes)(p, labels, "probs")
        loss.backward()
        po = torch.softmax(logits, 1)
        assert abs(float(olosses.lovasz_softmax(po, labels, ign, classes)) - float(loss)) <= 1e-7, tag
        arrs["loss_" + tag], arrs["grad_" + tag] = loss.detach().numpy(), p.grad.numpy()
        print(f"  lovasz {tag}: {float(loss):.8f}")
    save("lovasz_classes_2x20x8x64", **arrs)


if __name__ == "__main__":
    what = sys.argv[1:] or ["fpn_train"]
    for w in what:
        {"fpn_train": gen_fpn_train, "fpn_opt_train": gen_fpn_opt_train, "effnet": gen_effnet, "effnet_train": gen_effnet_train, "lovasz_classes": gen_lovasz_classes}[w]()
