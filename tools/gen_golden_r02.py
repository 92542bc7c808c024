#!/usr/bin/env python
"""Round-2 additions to the pinned oracle: same rules as tools/gen_golden.py (runs ONLY in the build container, imports the
reference read-only from /root/reference, asserts oracle == reference, writes small data-only fixtures under tests/golden/).

    python tools/gen_golden_r02.py [ece]
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
sys.path.insert(0, ROOT)
sys.path.insert(1, REF)
for stub in ("seaborn", "cv2"):
    sys.modules.setdefault(stub, types.ModuleType(stub))
sys.modules["cv2"].COLORMAP_TURBO = 20

from oracle import metrics as ometrics          # noqa: E402
from semanticlidarunc_amd.testing import synthetic_scan  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"  wrote {name}.npz  ({', '.join(f'{k}{tuple(np.asarray(v).shape)}' for k, v in arrs.items())})")


def gen_ece():
    """ECEAggregator beyond the cap (metrics/ece.py:93-111) and with binning='adaptive' (:115-128)."""
    from metrics.ece import ECEAggregator as RefECE                     # reference
    g = torch.Generator().manual_seed(41)
    pm = torch.softmax(torch.randn(2, 20, 16, 64, generator=g) * 2.0, dim=1)
    _, lab = synthetic_scan(2, 16, 64, seed=42)
    agree = torch.rand(2, 16, 64, generator=g) < 0.6
    boost = torch.zeros_like(pm).scatter_(1, lab.unsqueeze(1), 1.0)
    pe = torch.softmax(torch.log(pm) + 2.5 * boost * agree.unsqueeze(1), dim=1)
    logits = torch.log(pe) + 0.3
    alpha = torch.nn.functional.softplus(logits * 2.0) + 1.0
    out = {"probs": pe.numpy(), "logits": logits.numpy(), "alpha": alpha.numpy(), "labels": lab.numpy()}
    inputs = {"probs": pe, "logits": logits, "alpha": alpha}

    def batches(x):      # four updates of ~1840 valid pixels each: fill, overflow inside a batch, then replacement
        return [(x, lab), (x.flip(0), lab.flip(0)), (x.roll(1, 3), lab.roll(1, 2)), (x.flip(3), lab.flip(2))]

    for mode in ("probs", "logits", "alpha"):
        for cap, binning in ((3000, "uniform"), (None, "adaptive"), (2500, "adaptive")):
            ref = RefECE(n_bins=15, mode=mode, ignore_index=0, max_samples=cap, seed=0, binning=binning)
            orc = ometrics.ECESamples(cap, seed=0)
            for xb, lb in batches(inputs[mode]):
                ref.update(xb, lb)
                c, k = ometrics.top_label(xb.numpy(), lb.numpy(), 0, mode)
                orc.update(c, k)
            assert np.array_equal(orc.conf, ref._conf.numpy()) and np.array_equal(orc.correct, ref._correct.numpy()), (mode, cap, binning)
            assert orc.seen == ref._seen
            with tempfile.TemporaryDirectory() as td:
                (e_ref, m_ref), stats = ref.compute(save_plot_path=os.path.join(td, "r.png"))[:2]
            edges = ometrics.ece_edges(orc.conf, 15, binning)
            n, acc_s, conf_s = ometrics.ece_bins_over(orc.conf, orc.correct, edges)
            e_or, m_or = ometrics.ece_from_bins(n, acc_s, conf_s)
            assert np.array_equal(n, stats["n"].to_numpy()) and np.array_equal(edges[:-1], stats["low"].to_numpy())
            assert abs(e_or - e_ref) < 1e-7 and abs(m_or - m_ref) < 1e-7, (mode, cap, binning, e_or, e_ref)
            tag = f"{mode}|{cap}|{binning}"
            out["ece:" + tag], out["mce:" + tag] = np.float64(e_ref), np.float64(m_ref)
            out["n:" + tag], out["edges:" + tag] = n, edges
            out["seen:" + tag], out["kept:" + tag] = np.int64(ref._seen), np.int64(ref._conf.numel())
            out["conf_sorted:" + tag] = np.sort(ref._conf.numpy())
            out["ncorrect:" + tag] = np.int64(int(ref._correct.sum()))
            print(f"  ECE {tag}: reference={e_ref:.6f} oracle={e_or:.6f} kept {ref._conf.numel()} of {ref._seen}")
    # degenerate adaptive case: many identical confidences -> duplicate quantiles -> uniform fallback (ece.py:124-126)
    onehot = torch.zeros(1, 20, 4, 64).scatter_(1, torch.randint(1, 20, (1, 1, 4, 64), generator=g), 1.0)
    lab1 = torch.randint(1, 20, (1, 4, 64), generator=g)
    ref = RefECE(n_bins=15, mode="probs", ignore_index=0, binning="adaptive")
    ref.update(onehot, lab1)
    with tempfile.TemporaryDirectory() as td:
        (e_ref, _), stats = ref.compute(save_plot_path=os.path.join(td, "r.png"))[:2]
    c, k = ometrics.top_label(onehot.numpy(), lab1.numpy(), 0, "probs")
    edges = ometrics.ece_edges(c, 15, "adaptive")
    assert np.array_equal(edges[:-1], stats["low"].to_numpy())
    out["onehot_probs"], out["onehot_labels"], out["ece:onehot_adaptive"] = onehot.numpy(), lab1.numpy(), np.float64(e_ref)
    save("ece_capped_adaptive_2x20x16x64", **out)
    print("  oracle.metrics.ECESamples / ece_edges == reference ECEAggregator (buffers, edges, bins identical)")


def synthetic_kitti_scan(n_points=16000, seed=5):
    """A LiDAR-like cloud as the two files of a SemanticKITTI frame would hold it: .bin float32 [N,4], .label uint32 [N]
    (low 16 bits: raw semantic ids of dataset/definitions.py id_map, high 16 bits: an instance id)."""
    from dataset.definitions import id_map
    g = np.random.default_rng(seed)
    az = g.uniform(-np.pi, np.pi, n_points)
    el = np.radians(g.uniform(-24.8, 2.0, n_points))
    rng = g.uniform(1.5, 80.0, n_points)
    xyz = np.stack([rng * np.cos(el) * np.cos(az), rng * np.cos(el) * np.sin(az), rng * np.sin(el)], -1)
    xyzi = np.concatenate([xyz, g.uniform(0, 1, (n_points, 1))], -1).astype(np.float32)
    keys = np.array(sorted(id_map), dtype=np.uint32)
    label = keys[g.integers(0, len(keys), n_points)] | (g.integers(0, 500, n_points).astype(np.uint32) << 16)
    return xyzi, label.astype(np.uint32)


def gen_kitti():
    """f-3: SemanticKitti.__getitem__ (dataloader_semantic_KITTI.py:31-99) incl. rotate / flip, and spherical_projection with
    sort_largest_first / bins_h (dataset/utils.py:288-349).  torchvision.transforms and cv2 are absent: the dataloader only CONSTRUCTS a
    transforms.Compose it never applies, and cv2 is needed for cv2.Scharr inside build_normal_xyz -- served here by oracle.normals'
    restatement of OpenCV's definition, so the NORMALS of this fixture are not reference-pinned (everything else is)."""
    import tempfile
    from oracle import kitti as okitti, normals as onormals, projection as oproj
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    tv.transforms.Compose = lambda ts: ts
    tv.transforms.ToTensor = lambda: None
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tv.transforms
    cv2 = sys.modules["cv2"]
    cv2.CV_32FC1 = 5
    cv2.Scharr = lambda img, ddepth, dx, dy, scale=1.0: onormals.scharr(img, dx, dy, scale)
    from dataset.dataloader_semantic_KITTI import SemanticKitti as RefKitti        # reference
    from dataset.definitions import id_map
    from dataset.utils import spherical_projection as ref_projection
    xyzi, label = synthetic_kitti_scan()
    out = {"xyzi": xyzi, "label": label, "id_map_keys": np.array(sorted(id_map)), "id_map_values": np.array([id_map[k] for k in sorted(id_map)])}
    with tempfile.TemporaryDirectory() as td:
        fb, fl = os.path.join(td, "000000.bin"), os.path.join(td, "000000.label")
        xyzi.tofile(fb)
        label.tofile(fl)
        for tag, rotate, flip, seed in (("plain", False, False, 0), ("rot", True, False, 3), ("flip", False, True, 1), ("rotflip", True, True, 2)):
            # the reference draws np.random.randint(-180, 180) (rotate) and np.random.rand() < 0.5 (flip) from the global numpy RNG
            np.random.seed(seed)
            angle = float(np.random.randint(-180, 180)) if rotate else None
            do_flip = bool(flip and np.random.rand() < 0.5)
            assert do_flip == flip, "pick a seed whose flip draw fires"
            np.random.seed(seed)
            ds = RefKitti([(fb, fl)], rotate=rotate, flip=flip, projection=(32, 256), resize=False)
            ref = [t.numpy() for t in ds[0]]
            mine = okitti.sample(xyzi.tobytes(), label.tobytes(), id_map, (32, 256), angle, do_flip)
            for name, a, b in zip(("range", "reflectivity", "xyz", "normals", "semantics"), ref, mine):
                assert a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b), (tag, name)
                out[f"{tag}:{name}"] = a
            out[f"{tag}:angle"] = np.float64(angle if angle is not None else np.nan)
        print("  oracle.kitti.sample == reference SemanticKitti.__getitem__ (range, reflectivity, xyz, semantics bit-identical; normals via the restated Scharr)")
    cloud = okitti.decode(xyzi.tobytes(), label.tobytes(), id_map)
    beams = np.radians(np.linspace(2.0, -24.8, 32)) + 1e-3                        # an explicit, decreasing beam-elevation table
    for tag, kw in (("farthest", dict(sort_largest_first=True)), ("bins_h", dict(bins_h=beams)), ("bins_h_increasing", dict(bins_h=beams[::-1].copy())),
                    ("farthest_bins_h_range", dict(sort_largest_first=True, bins_h=beams, theta_range=(-0.45, 0.05)))):
        img_r, alpha_r, th_r, _ = ref_projection(cloud, 32, 256, **kw)
        img_o, alpha_o, th_o, _ = oproj.spherical_projection(cloud, 32, 256, **kw)
        assert np.array_equal(img_r, img_o) and np.array_equal(alpha_r, alpha_o) and tuple(th_r) == tuple(th_o), tag
        out[f"proj:{tag}:img"], out[f"proj:{tag}:theta"] = img_r, np.array(th_r)
    out["beams"] = beams
    print("  oracle.projection.spherical_projection(sort_largest_first / bins_h) == reference (bit-identical)")
    save("kitti_sample_16000_32x256", **out)


def gen_fpn_opt():
    """f-4: baselines/Reichert/semanticFCN_opt.py (resnet18 / resnet34) through the stub torchvision.models that serves oracle.fpn's restated
    BasicBlock ResNet: the reference's OWN head wiring (SpatialAttention, UpsampleBlock + GroupNorm, dropout_pyramid, GN decoder) runs on
    the state_dict of this repo's class; the Dropout2d of the pyramid is pinned by instrumenting the reference module with a fixed
    multiplier."""
    import json
    from oracle import fpn as ofpn, fpn_opt as ofpo
    from semanticlidarunc_amd.fpn_opt import SemanticNetworkWithFPN as MyOpt
    from semanticlidarunc_amd.testing import randomize_bn_
    tv = types.ModuleType("torchvision")
    tv.models = ofpn.torchvision_models_stub()
    sys.modules["torchvision"], sys.modules["torchvision.models"] = tv, tv.models
    from baselines.Reichert.semanticFCN_opt import SemanticNetworkWithFPN as RefOpt        # the reference's own wiring
    for tag, kw, shape in (("resnet18_m6_c20", dict(backbone="resnet18", input_channels=2, meta_channel_dim=6, num_classes=20), (2, 32, 128)),
                           ("resnet34_m3_c21_noatt", dict(backbone="resnet34", input_channels=2, meta_channel_dim=3, num_classes=21,
                                                          attention=False, multi_scale_meta=False), (1, 16, 64)),
                           ("resnet50_m3_c5", dict(backbone="resnet50", input_channels=2, meta_channel_dim=3, num_classes=5), (1, 32, 64))):
        torch.manual_seed(0)
        mine = randomize_bn_(MyOpt(**kw), 3).eval()
        with torch.no_grad():                                # non-trivial GroupNorm affines
            g = torch.Generator().manual_seed(9)
            for mod in mine.modules():
                if isinstance(mod, torch.nn.GroupNorm):
                    mod.weight.copy_(torch.rand(mod.num_channels, generator=g) + 0.5)
                    mod.bias.copy_(torch.randn(mod.num_channels, generator=g) * 0.1)
        ref = RefOpt(**kw)
        sdf = mine.state_dict()
        assert list(sdf.keys()) == list(ref.state_dict().keys()), "state_dict key order differs from the reference class"
        assert all(tuple(a.shape) == tuple(b.shape) for a, b in zip(sdf.values(), ref.state_dict().values()))
        assert [type(m).__name__ for m in mine.modules() if isinstance(m, (torch.nn.Dropout2d, torch.nn.GroupNorm))] == \
               [type(m).__name__ for m in ref.modules() if isinstance(m, (torch.nn.Dropout2d, torch.nn.GroupNorm))]
        ref.load_state_dict(sdf)
        ref.eval()
        g = torch.Generator().manual_seed(51)
        xf = torch.randn(shape[0], 2, shape[1], shape[2], generator=g) * torch.tensor([20.0, 0.3]).view(1, 2, 1, 1)
        mf = torch.randn(shape[0], kw["meta_channel_dim"], shape[1], shape[2], generator=g) * 5.0
        with torch.no_grad():
            yr = ref(xf, mf)
            yo = ofpo.fpn_opt_forward(sdf, xf, mf, kw["backbone"], kw.get("attention", True), kw.get("multi_scale_meta", True))
            print(f"FPN-opt {tag}: |oracle - reference| = {float((yr - yo).abs().max()):.3e}  (|logit| max {float(yr.abs().max()):.2f})")
            assert float((yr - yo).abs().max()) <= 1e-5 * max(1.0, float(yr.abs().max()))
            # the pyramid dropout with a fixed multiplier: replace the reference's Dropout2d child by a module that applies it
            cpyr = ref.decoder_semantic[0].in_channels
            scale = (torch.rand(shape[0], cpyr, 1, 1, generator=g) > 0.1).float() / 0.9
            class Fixed(torch.nn.Module):
                def forward(self, t):
                    return t * scale
            ref.dropout_pyramid = Fixed()
            yr_d = ref(xf, mf)
            yo_d = ofpo.fpn_opt_forward(sdf, xf, mf, kw["backbone"], kw.get("attention", True), kw.get("multi_scale_meta", True), dropout_scale=scale)
            assert float((yr_d - yo_d).abs().max()) <= 1e-5 * max(1.0, float(yr.abs().max()))
        save("fpn_opt_" + tag, x=xf, meta=mf, out=yr, out_dropout=yr_d, dropout_scale=scale[:, :, 0, 0],
             sd_digest=np.array([sum(float(v.double().sum()) for v in sdf.values() if v.is_floating_point()),
                                 sum(float(v.double().abs().sum()) for v in sdf.values() if v.is_floating_point())]))
    with open(os.path.join(OUT, "fpn_opt_resnet18_state_dict_keys.json"), "w") as f:
        torch.manual_seed(0)
        json.dump({k: list(v.shape) for k, v in RefOpt("resnet18", 2, 6, num_classes=20).state_dict().items()}, f, indent=0)
    print("  oracle.fpn_opt == reference baselines.Reichert.semanticFCN_opt (head wiring; backbone internals restated)")


def gen_fpn_resnet50():
    """a3 widened: models/semanticFCN.py with the resnet50 backbone (Bottleneck blocks, channel ladder 2048..128) through the stub
    torchvision.models that serves oracle.fpn's restated ResNet: the reference's own wiring on the state_dict of this repo's class."""
    from oracle import fpn as ofpn
    from semanticlidarunc_amd.fpn import SemanticNetworkWithFPN as MyFPN
    from semanticlidarunc_amd.testing import randomize_bn_
    tv = types.ModuleType("torchvision")
    tv.models = ofpn.torchvision_models_stub()
    sys.modules["torchvision"], sys.modules["torchvision.models"] = tv, tv.models
    from models.semanticFCN import SemanticNetworkWithFPN as RefFPN        # the reference's own wiring
    tag, kw, shape = "resnet50_m3_c5", dict(backbone="resnet50", input_channels=2, meta_channel_dim=3, num_classes=5), (1, 32, 64)
    torch.manual_seed(0)
    mine = randomize_bn_(MyFPN(**kw), 3).eval()
    ref_f = RefFPN(**kw)
    sdf = mine.state_dict()
    assert list(sdf.keys()) == list(ref_f.state_dict().keys()), "state_dict key order differs from the reference class"
    assert all(tuple(a.shape) == tuple(b.shape) for a, b in zip(sdf.values(), ref_f.state_dict().values()))
    ref_f.load_state_dict(sdf)
    ref_f.eval()
    g = torch.Generator().manual_seed(52)
    xf = torch.randn(shape[0], 2, shape[1], shape[2], generator=g) * torch.tensor([20.0, 0.3]).view(1, 2, 1, 1)
    mf = torch.randn(shape[0], kw["meta_channel_dim"], shape[1], shape[2], generator=g) * 5.0
    with torch.no_grad():
        yr = ref_f(xf, mf)
        yo = ofpn.fpn_forward(sdf, xf, mf, kw["backbone"], True, True)
    d = float((yr - yo).abs().max())
    print(f"FPN {tag}: |oracle - reference| = {d:.3e}  (out min {float(yr.min()):.3f}, max {float(yr.max()):.3f})")
    assert d <= 1e-5 * max(1.0, float(yr.abs().max()))
    fl = {k: v for k, v in sdf.items() if v.is_floating_point()}
    digest = np.array([sum(float(v.double().sum()) for v in fl.values()), sum(float(v.double().abs().sum()) for v in fl.values())])
    save("fpn_" + tag, x=xf.numpy(), meta=mf.numpy(), out=yr.numpy(), sd_digest=digest)


def gen_kl_weighted():
    """KL_offClasses_to_uniform(with_conf_weighting=True) (losses/regularizers.py:375-385): value and gradient, two gammas."""
    from losses import regularizers as ref_reg                           # reference
    from oracle import dirichlet as odir
    g = torch.Generator().manual_seed(808)
    lab = torch.randint(0, 20, (2, 8, 64), generator=g)
    lab[torch.rand(2, 8, 64, generator=g) < 0.12] = 0
    alpha = 1.0 + torch.nn.functional.softplus(torch.randn(2, 20, 8, 64, generator=g) * 2.0) * torch.rand(2, 1, 8, 64, generator=g) * 30
    out = {"labels": lab.numpy(), "alpha": alpha.numpy()}
    for gamma in (1.0, 2.5):
        ar = alpha.clone().requires_grad_(True)
        lr = ref_reg.KL_offClasses_to_uniform(ignore_index=0, with_conf_weighting=True, gamma=gamma)(ar, lab)
        lr.backward()
        ao = alpha.clone().requires_grad_(True)
        lo = odir.loss_kl_off_uniform(ao, lab, 0, with_conf_weighting=True, gamma=gamma)
        lo.backward()
        assert float((lr - lo).abs()) == 0.0 and float((ar.grad - ao.grad).abs().max()) == 0.0, gamma
        out[f"loss:gamma{gamma}"], out[f"grad:gamma{gamma}"] = lr.detach().numpy(), ar.grad.numpy()
        print(f"  KL_offClasses_to_uniform(with_conf_weighting, gamma={gamma}): reference = oracle = {float(lr):.6f}")
    save("kl_off_weighted_2x20x8x64", **out)


GENERATORS = {"ece": gen_ece, "kitti": gen_kitti, "fpn_opt": gen_fpn_opt, "fpn_resnet50": gen_fpn_resnet50, "kl_weighted": gen_kl_weighted}

if __name__ == "__main__":
    for name in (sys.argv[1:] or list(GENERATORS)):
        print(f"[{name}]")
        GENERATORS[name]()
    print("round-2 oracle additions pinned against the reference")
