"""The input side of the hot path on the GPU, in a form that still runs under ``DataLoader(num_workers > 0)``
(reference ``dataset/dataloader_semantic_KITTI.py:31-99``).

The reference does everything of a sample -- file decode, ``id_map`` lookup, yaw rotation, spherical projection, flip, range,
normals -- in numpy / cv2 inside forked DataLoader workers, which must not touch the GPU.  Here the workers only READ THE FILES
(``RawScanDataset``: two ``np.fromfile`` calls per sample, returned as CPU tensors the loader can pin), and the main process turns a
whole batch of raw scans into the five tensors the Trainer consumes with HIP kernels (``ScanProjector``: ``slu_kitti_decode`` ->
``slu_spherical_projection_ex`` (flip folded in) -> ``slu_build_normals`` -> ``slu_range_image_split``), written straight into the
batch tensors on the device.  ``projecting_loader_class`` packages both as a ``DataLoader`` subclass with the reference's constructor
signature, so a launcher can put it in the training script's namespace (tools/dp_launch.py does) and the script stays unchanged.

Random augmentation draws follow the reference: per sample ``np.random.randint(-180, 180)`` when ``rotate`` and then
``np.random.rand() < 0.5`` when ``flip`` (dataloader :53-54,72), from numpy's global generator -- of the main process here.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch
import torch.utils.data as tud

from semanticlidarunc_amd import ops


def id_map_lut(id_map: dict) -> torch.Tensor:
    """Dense int32 table of ``dataset.definitions.id_map`` (-1 where the dict has no key: the reference raises KeyError there)."""
    lut = np.full(max(id_map) + 1, -1, dtype=np.int32)
    for k, v in id_map.items():
        lut[int(k)] = int(v)
    return torch.from_numpy(lut)


class RawScanDataset(tud.Dataset):
    """Worker side: sample idx -> (xyzi fp32 [N, 4], label int32 [N]) exactly as the two files hold them (dataloader :35-38)."""

    def __init__(self, data_path: Sequence[Tuple[str, str]]):
        self.data_path = list(data_path)

    def __len__(self) -> int:
        return len(self.data_path)

    def __getitem__(self, idx):
        frame_path, label_path = self.data_path[idx]
        xyzi = np.fromfile(frame_path, dtype=np.float32).reshape(-1, 4)
        label = np.fromfile(label_path, dtype=np.uint32).reshape(-1).view(np.int32)
        return torch.from_numpy(xyzi), torch.from_numpy(label)


def raw_collate(samples):
    """Scans have different point counts: keep the batch as two lists (the default collate of lists of tensors would try to stack)."""
    return [s[0] for s in samples], [s[1] for s in samples]


class ScanProjector:
    """Main-process side: a batch of raw scans -> (range [B,1,H,W], reflectivity [B,1,H,W], xyz [B,3,H,W], normals [B,3,H,W],
    semantics int64 [B,1,H,W]) on `device`, the tuple the reference's default-collated dataset yields (dataloader :90-99)."""

    def __init__(self, id_map: dict, projection=(64, 2048), rotate: bool = False, flip: bool = False, device="cuda", norm_factor: float = 0.25,
                 resize=None):
        """resize: None, or (rows, columns) of the nearest-neighbour resize between projection and flip (SemanticKitti(resize=True) fixes it
        to (128, 2048), dataloader_semantic_KITTI.py:61-62); the five outputs then have that size."""
        self.height, self.width = int(projection[0]), int(projection[1])
        self.resize = None if resize is None else (int(resize[0]), int(resize[1]))
        self.rotate, self.flip, self.device, self.norm_factor = bool(rotate), bool(flip), torch.device(device), float(norm_factor)
        self._lut_cpu = id_map_lut(id_map)
        self._lut = None

    def draw_augmentation(self):
        """(yaw angle in degrees or None, flip decision) for one sample, in the reference's draw order."""
        angle = float(np.random.randint(-180, 180)) if self.rotate else None
        do_flip = bool(self.flip and np.random.rand() < 0.5)
        return angle, do_flip

    @torch.no_grad()
    def __call__(self, xyzi_list, label_list, augmentation: Optional[Sequence[Tuple[Optional[float], bool]]] = None):
        b, dev = len(xyzi_list), self.device
        h, w = self.resize if self.resize is not None else (self.height, self.width)
        if b == 0 or b != len(label_list):
            raise RuntimeError("ScanProjector: a non-empty batch of (xyzi, label) pairs expected")
        if self._lut is None or self._lut.device != dev:
            self._lut = self._lut_cpu.to(dev)
        rng = torch.empty((b, 1, h, w), dtype=torch.float32, device=dev)
        refl = torch.empty((b, 1, h, w), dtype=torch.float32, device=dev)
        xyz = torch.empty((b, 3, h, w), dtype=torch.float32, device=dev)
        nrm = torch.empty((b, 3, h, w), dtype=torch.float32, device=dev)
        sem = torch.empty((b, 1, h, w), dtype=torch.int64, device=dev)
        bad = torch.zeros(1, dtype=torch.int32, device=dev)
        for i in range(b):
            angle, do_flip = augmentation[i] if augmentation is not None else self.draw_augmentation()
            pts = xyzi_list[i].to(dev, non_blocking=True).contiguous()
            lab = label_list[i].to(dev, non_blocking=True).contiguous()
            pc = ops.kitti_decode(pts, lab, self._lut, bad, angle)
            if self.resize is None:
                img, _ = ops.spherical_projection(pc, h, w, flip=do_flip)
            else:       # the reference order: project, resize, flip
                img, _ = ops.spherical_projection(pc, self.height, self.width)
                img = ops.resize_nearest_hwc(img, h, w, flip=do_flip)
            normals = ops.build_normals(img, self.norm_factor)
            ops.range_image_split(img, normals, rng[i], refl[i], xyz[i], nrm[i], sem[i])
        if int(bad.item()):            # one sync per batch; the reference's dict lookup would have raised inside __getitem__
            raise KeyError(f"{int(bad.item())} point label(s) of this batch have no entry in id_map")
        return rng, refl, xyz, nrm, sem


def projecting_loader_class(projector_factory, raw_dataset_of):
    """A ``DataLoader`` subclass for a training script's namespace.  ``raw_dataset_of(dataset)`` returns the ``RawScanDataset`` twin of a
    dataset this pipeline can serve (or None: the loader then behaves like the stock one); ``projector_factory(dataset)`` builds its
    ``ScanProjector``.  Iterating yields device-resident batches in the reference's format; ``len`` / sampler / workers / pinning are the
    stock loader's."""

    class ProjectingDataLoader(tud.DataLoader):
        def __init__(self, dataset=None, *args, **kwargs):
            if dataset is None:
                dataset = kwargs.pop("dataset")
            raw = raw_dataset_of(dataset)
            self._slu_projector = None
            if raw is not None:
                if kwargs.get("collate_fn") is not None:
                    raise RuntimeError("ProjectingDataLoader: a custom collate_fn cannot be combined with the device-side projection")
                kwargs["collate_fn"] = raw_collate
                self._slu_projector = projector_factory(dataset)
                dataset = raw
            if kwargs.get("num_workers", 0) == 0:
                kwargs.pop("prefetch_factor", None)
                kwargs.pop("persistent_workers", None)
            super().__init__(dataset, *args, **kwargs)

        def __iter__(self):
            it = super().__iter__()
            if self._slu_projector is None:
                yield from it
                return
            for xyzi_list, label_list in it:
                yield self._slu_projector(xyzi_list, label_list)

    return ProjectingDataLoader
