"""CPU: host-side contract of the drop-in classes (no kernel launches)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import GOLDEN
from semanticlidarunc_amd.metrics.ece import ECEAggregator
from semanticlidarunc_amd.models.evaluator import IoUEvaluator
from semanticlidarunc_amd.models.losses import classify_output_kind
from semanticlidarunc_amd.salsanext import SalsaNext
from semanticlidarunc_amd.utils.inputs import set_model_inputs
from semanticlidarunc_amd.utils.mc_dropout import dropout_sampling, set_dropout_mode


def test_state_dict_contract_matches_reference():
    want = json.load(open(os.path.join(GOLDEN, "salsanext_state_dict_keys.json")))
    sd = SalsaNext(20, 5).state_dict()
    assert list(sd.keys()) == list(want.keys())
    for k, v in sd.items():
        assert [str(v.dtype).replace("torch.", "")] + list(v.shape) == want[k], k


def test_dropout_children_and_mc_toggle():
    m = SalsaNext(20, 5)
    want = json.load(open(os.path.join(GOLDEN, "salsanext_dropout_modules.json")))
    have = [n for n, mod in m.named_modules() if isinstance(mod, nn.Dropout2d)]
    assert sorted(have) == sorted(want) and len(have) == 17
    m.eval()
    bns = [x for x in m.modules() if isinstance(x, nn.BatchNorm2d)]
    assert len(bns) == 42
    with dropout_sampling(m, True):
        assert all(x.training for x in m.modules() if isinstance(x, nn.Dropout2d))
        assert not any(b.training for b in bns)
    assert not any(x.training for x in m.modules() if isinstance(x, nn.Dropout2d))
    set_dropout_mode(m, True)
    assert m.resBlock2.dropout.training and not m.training


def test_forward_refuses_cpu_and_bad_shapes():
    m = SalsaNext(20, 5).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 5, 16, 64))
    with pytest.raises(RuntimeError):
        m(torch.zeros(5, 16, 64))


def test_set_model_inputs_channel_order():
    r, refl = torch.full((1, 1, 2, 2), 1.0), torch.full((1, 1, 2, 2), 2.0)
    xyz, nrm = torch.full((1, 3, 2, 2), 3.0), torch.full((1, 3, 2, 2), 4.0)
    cfg = {"model_settings": {"baseline": "SalsaNext", "reflectivity": 1, "normals": 0}}
    (x,) = set_model_inputs(r, refl, xyz, nrm, cfg)
    assert x.shape == (1, 5, 2, 2) and x[0, :, 0, 0].tolist() == [1, 2, 3, 3, 3]
    cfg = {"model_settings": {"baseline": "Reichert", "reflectivity": 1, "normals": 1}}
    a, b = set_model_inputs(r, refl, xyz, nrm, cfg)
    assert a.shape[1] == 2 and b.shape[1] == 6
    with pytest.raises(ValueError):
        set_model_inputs(r, refl, xyz, nrm, {"model_settings": {"baseline": "nope"}})


def test_classify_output_kind():
    torch.manual_seed(0)
    lg = torch.randn(2, 20, 8, 16) * 3
    assert classify_output_kind(lg) == "logits"
    assert classify_output_kind(torch.softmax(lg, 1)) == "probs"
    assert classify_output_kind(torch.log_softmax(lg, 1)) == "log_probs"


def test_iou_compute_from_injected_confusion_matrix():
    ev = IoUEvaluator(3)
    ev.confmat = torch.tensor([[5, 1, 0], [2, 6, 0], [0, 0, 0]])
    miou, d = ev.compute(["a", "b", "c"], test_mask=[1, 1, 1], ignore_gt=None)
    assert abs(d["a"] - 5 / 8) < 1e-12 and abs(d["b"] - 6 / 9) < 1e-12 and np.isnan(d["c"])
    assert abs(miou - (5 / 8 + 6 / 9) / 2) < 1e-12 and d["mIoU"] == miou
    miou0, d0 = ev.compute({0: "a", 1: "b", 2: "c"}, test_mask=[0, 1, 1], ignore_gt=[0])
    assert abs(d0["b"] - 6 / 8) < 1e-12 and abs(miou0 - 6 / 8) < 1e-12
    with pytest.raises(ValueError):
        ev.compute(["a", "b", "c"], test_mask=[1, 1])


def test_ece_compute_from_injected_bins_and_empty_case():
    agg = ECEAggregator(n_bins=4, mode="probs", ignore_index=0)
    (e, m), stats = agg.compute()
    assert np.isnan(e) and np.isnan(m) and stats.empty
    agg._count = torch.tensor([0, 10, 0, 30])
    agg._sum_correct = torch.tensor([0.0, 5.0, 0.0, 27.0], dtype=torch.float64)
    agg._sum_conf = torch.tensor([0.0, 4.0, 0.0, 28.5], dtype=torch.float64)
    (e, m), stats, fig = agg.compute()
    assert fig is None and list(stats["n"]) == [0, 10, 0, 30]
    assert abs(e - (10 / 40 * 0.1 + 30 / 40 * 0.05)) < 1e-12 and abs(m - 0.1) < 1e-12
    # the sample-keeping forms (reservoir cap / equal-mass bins, metrics/ece.py:93-128) are constructible and start empty
    for kw in ({"binning": "adaptive"}, {"max_samples": 10}):
        agg = ECEAggregator(n_bins=4, mode="probs", **kw)
        assert agg._keeps_samples and agg._conf.numel() == 0 and agg._seen == 0
        (e, m), stats = agg.compute()
        assert np.isnan(e) and stats.empty
    with pytest.raises(AssertionError):
        ECEAggregator(binning="quantile")


def test_id_map_table_and_raw_scan_loader_plumbing(tmp_path):
    """dataset/definitions.py id_map equals the reference's table (stored in the golden that the reference generated); the worker side of
    the device input pipeline only reads files and keeps ragged scans as lists through a DataLoader with workers."""
    from conftest import golden
    from semanticlidarunc_amd.dataset import gpu_pipeline
    from semanticlidarunc_amd.dataset.definitions import id_map
    g = golden("kitti_sample_16000_32x256")
    assert id_map == {int(k): int(v) for k, v in zip(g["id_map_keys"], g["id_map_values"])}
    lut = gpu_pipeline.id_map_lut(id_map)
    assert lut.dtype == torch.int32 and int(lut[259]) == 5 and int(lut[2]) == -1 and lut.numel() == 260
    paths = []
    for k, n in enumerate((300, 170, 220)):
        fb, fl = tmp_path / f"{k:06d}.bin", tmp_path / f"{k:06d}.label"
        g["xyzi"][:n].tofile(fb)
        g["label"][:n].tofile(fl)
        paths.append((str(fb), str(fl)))
    ds = gpu_pipeline.RawScanDataset(paths)
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, num_workers=2, collate_fn=gpu_pipeline.raw_collate)
    batches = list(loader)
    assert [len(b[0]) for b in batches] == [2, 1] and [t.shape[0] for t in batches[0][0]] == [300, 170]
    assert batches[0][1][1].dtype == torch.int32 and np.array_equal(batches[1][0][0].numpy(), g["xyzi"][:220])
    assert np.array_equal(batches[0][1][0].numpy().view(np.uint32), g["label"][:300])


def test_efficientnet_container_matches_the_reference_state_dict_layout():
    """semanticFCN_opt with efficientnet_v2_s: state_dict keys, order and shapes equal the reference class's (recorded by tools/gen_golden_r03.py from
    the reference's own constructor through the torchvision stub); the restated V2 configurations have torchvision's published parameter counts."""
    import json
    from conftest import GOLDEN
    from semanticlidarunc_amd import effnet
    from semanticlidarunc_amd.fpn_opt import SemanticNetworkWithFPN
    want = json.load(open(os.path.join(GOLDEN, "fpn_opt_efficientnet_v2_s_state_dict_keys.json")))
    sd = SemanticNetworkWithFPN("efficientnet_v2_s", 2, 3, num_classes=20).state_dict()
    assert list(sd.keys()) == list(want.keys())
    assert all(list(v.shape) == want[k] for k, v in sd.items())
    for name, n in (("efficientnet_v2_s", 21458488), ("efficientnet_v2_m", 54139356), ("efficientnet_v2_l", 118515272)):
        assert sum(p.numel() for p in effnet.EfficientNetContainer(name).parameters()) == n
