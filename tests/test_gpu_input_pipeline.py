"""GPU: the input side of the path -- file decode + id_map LUT + yaw rotation + spherical projection (+ flip) + range /
normals / channel split as HIP kernels -- against the golden the reference's own SemanticKitti.__getitem__ produced
(tools/gen_golden_r02.py; normals through the restated Scharr: unpinned) and the projection options sort_largest_first / bins_h."""
import numpy as np
import pytest
import torch

from conftest import golden
from semanticlidarunc_amd.dataset import gpu_pipeline
from semanticlidarunc_amd.dataset import utils as dutils
from semanticlidarunc_amd.dataset.dataloader_semantic_KITTI import SemanticKitti
from semanticlidarunc_amd.dataset.definitions import id_map

pytestmark = pytest.mark.gpu
H, W = 32, 256


def _files(tmp_path, g, counts=(16000,)):
    paths = []
    for k, n in enumerate(counts):
        fb, fl = tmp_path / f"{k:06d}.bin", tmp_path / f"{k:06d}.label"
        g["xyzi"][:n].tofile(fb)
        g["label"][:n].tofile(fl)
        paths.append((str(fb), str(fl)))
    return paths


def _close_images(got, want, int_exact=False, max_bad=2e-3):
    """A point exactly on a bin edge may land in the neighbouring pixel (last-bit atan2 difference): allow a few pixels to differ."""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape and got.dtype == want.dtype
    bad = (got != want) if int_exact else (np.abs(got - want) > 1e-5 * np.maximum(1.0, np.abs(want)))
    frac = float(bad.reshape(got.shape[0], -1).any(0).mean()) if got.ndim == 3 else float(bad.mean())
    assert frac <= max_bad, frac


@pytest.mark.parametrize("tag,flip", [("plain", False), ("rot", False), ("flip", True), ("rotflip", True)])
def test_scan_projector_matches_the_reference_sample(cuda, tag, flip):
    g = golden("kitti_sample_16000_32x256")
    angle = None if np.isnan(float(g[f"{tag}:angle"])) else float(g[f"{tag}:angle"])
    proj = gpu_pipeline.ScanProjector(id_map, (H, W), rotate=angle is not None, flip=flip, device=cuda)
    xyzi, label = torch.from_numpy(g["xyzi"].copy()), torch.from_numpy(g["label"].view(np.int32).copy())
    rng, refl, xyz, nrm, sem = proj([xyzi, xyzi], [label, label], augmentation=[(angle, flip), (None, False)])
    assert rng.shape == (2, 1, H, W) and xyz.shape == (2, 3, H, W) and sem.dtype == torch.int64 and rng.is_cuda
    _close_images(rng[0].cpu().numpy(), g[f"{tag}:range"])
    _close_images(refl[0].cpu().numpy(), g[f"{tag}:reflectivity"])
    _close_images(xyz[0].cpu().numpy(), g[f"{tag}:xyz"])
    _close_images(sem[0].cpu().numpy(), g[f"{tag}:semantics"], int_exact=True)
    # normals: a differing pixel also changes its 8 neighbours' derivatives
    _close_images(nrm[0].cpu().numpy(), g[f"{tag}:normals"], max_bad=2e-2)
    _close_images(rng[1].cpu().numpy(), g["plain:range"])                       # the second scan of the batch took its own augmentation
    # range really is |xyz| of the SAME image in float32
    want = torch.sqrt((xyz[0, 0] * xyz[0, 0] + xyz[0, 1] * xyz[0, 1]) + xyz[0, 2] * xyz[0, 2])
    assert float((rng[0, 0] - want).abs().max()) <= 1e-6 * float(want.max())


def test_unknown_label_raises_like_the_reference_dict_lookup(cuda):
    g = golden("kitti_sample_16000_32x256")
    label = g["label"].copy()
    label[7] = 2                                                            # 2 is not a key of id_map
    proj = gpu_pipeline.ScanProjector(id_map, (H, W), device=cuda)
    with pytest.raises(KeyError):
        proj([torch.from_numpy(g["xyzi"].copy())], [torch.from_numpy(label.view(np.int32))])


@pytest.mark.parametrize("tag", ["farthest", "bins_h", "bins_h_increasing", "farthest_bins_h_range"])
def test_projection_options_match_reference(cuda, tag):
    from oracle import kitti as okitti
    g = golden("kitti_sample_16000_32x256")
    cloud = okitti.decode(g["xyzi"].tobytes(), g["label"].tobytes(), id_map)
    beams = g["beams"]
    kw = {"farthest": dict(sort_largest_first=True), "bins_h": dict(bins_h=beams), "bins_h_increasing": dict(bins_h=beams[::-1].copy()),
          "farthest_bins_h_range": dict(sort_largest_first=True, bins_h=beams, theta_range=(-0.45, 0.05))}[tag]
    img, alpha, th, ph = dutils.spherical_projection(cloud, H, W, **kw)
    _close_images(img.transpose(2, 0, 1), g[f"proj:{tag}:img"].transpose(2, 0, 1))
    assert np.allclose(th, g[f"proj:{tag}:theta"], atol=1e-12) and alpha.shape == (H, W)
    with pytest.raises(ValueError):
        dutils.spherical_projection(cloud, H, W, bins_h=np.zeros(H))


def test_dataset_mirror_and_worker_loader_yield_device_batches(cuda, tmp_path):
    """The drop-in dataset class in the main process, and its gpu_loader with num_workers = 2: workers read files, the main process
    projects; both agree with the golden sample."""
    g = golden("kitti_sample_16000_32x256")
    paths = _files(tmp_path, g, (16000, 9000, 12000))
    ds = SemanticKitti(paths, rotate=False, flip=False, projection=(H, W), resize=False)
    one = ds[0]
    assert all(not t.is_cuda for t in one) and one[4].dtype == torch.int64
    _close_images(one[0].numpy(), g["plain:range"])
    loader = ds.gpu_loader(device=cuda, batch_size=2, shuffle=False, num_workers=2, pin_memory=True, persistent_workers=True, prefetch_factor=2)
    batches = list(loader)
    assert len(loader) == 2 and [b[0].shape[0] for b in batches] == [2, 1] and all(t.is_cuda for t in batches[0])
    _close_images(batches[0][0][0].cpu().numpy(), g["plain:range"])
    _close_images(batches[0][4][0].cpu().numpy(), g["plain:semantics"], int_exact=True)
    assert int((batches[0][0][1] > 0).sum()) < int((batches[0][0][0] > 0).sum())          # the 9000-point scan fills fewer pixels
    # resize=True (the reference's constructor default): always cv2.resize(..., (2048, 128), INTER_NEAREST) after the projection
    big = SemanticKitti(paths, projection=(H, W), resize=True)[0]
    assert big[0].shape == (1, 128, 2048) and big[2].shape == (3, 128, 2048) and big[4].shape == (1, 128, 2048)


@pytest.mark.parametrize("size,flip", [((64, 256), False), ((48, 384), True), ((20, 100), True)])
def test_nearest_resize_between_projection_and_flip(cuda, size, flip):
    """SemanticKitti(resize=True), dataloader_semantic_KITTI.py:61-62 + :71-73 -- OpenCV's INTER_NEAREST restated (cv2 absent: unpinned), up- and
    down-sampling, non-integer ratios; the flip comes AFTER the resize as in the reference."""
    from oracle import kitti as okitti
    g = golden("kitti_sample_16000_32x256")
    want = okitti.sample(g["xyzi"].tobytes(), g["label"].tobytes(), id_map, (H, W), None, flip, resize=size)
    proj = gpu_pipeline.ScanProjector(id_map, (H, W), flip=flip, device=cuda, resize=size)
    xyzi, label = torch.from_numpy(g["xyzi"].copy()), torch.from_numpy(g["label"].view(np.int32).copy())
    got = proj([xyzi], [label], augmentation=[(None, flip)])
    assert got[0].shape == (1, 1, size[0], size[1])
    for k, (a, b) in enumerate(zip(got, want)):
        _close_images(a[0].cpu().numpy(), b, int_exact=(k == 4), max_bad=2e-2 if k == 3 else 4e-3)
