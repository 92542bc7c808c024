"""Drop-in for ``dataset.dataloader_semantic_KITTI.SemanticKitti`` of the reference (``src/dataset/dataloader_semantic_KITTI.py:16-99``):
same constructor and the same five tensors per sample, computed by HIP kernels when ``__getitem__`` runs in the main process
(``num_workers=0``); inside a forked DataLoader worker (which must not touch the GPU) the call is handed to the reference class this
module shadows.  For ``num_workers > 0`` WITH the device path use ``gpu_loader`` (workers only read the files, the main process
projects whole batches: ``dataset/gpu_pipeline.py``)."""
from __future__ import annotations

import torch
import torch.utils.data
from torch.utils.data import Dataset

from semanticlidarunc_amd.dataset import gpu_pipeline
from semanticlidarunc_amd.dataset.definitions import id_map


class SemanticKitti(Dataset):
    def __init__(self, data_path, rotate=False, flip=False, resolution=(2048, 128), projection=(64, 2048), resize=True):
        self.data_path = data_path
        self.rotate, self.flip = rotate, flip
        self.resolution, self.projection, self.resize = resolution, projection, resize
        self._raw = gpu_pipeline.RawScanDataset(data_path)
        self._projector = None
        self._ref = None

    def __len__(self):
        return len(self.data_path)

    def projector(self, device="cuda") -> gpu_pipeline.ScanProjector:
        # resize=True: cv2.resize(xyzi_img, (2048, 128), INTER_NEAREST) whatever `resolution` says (the reference never reads it, :61-62)
        return gpu_pipeline.ScanProjector(id_map, self.projection, self.rotate, self.flip, device, resize=(128, 2048) if self.resize else None)

    def __getitem__(self, idx):
        if torch.utils.data.get_worker_info() is not None:
            if _shadowed is None:
                raise RuntimeError("SemanticKitti: the HIP path cannot run inside a DataLoader worker; use SemanticKitti.gpu_loader(...) "
                                   "(workers read files, the main process projects) or keep the reference's dataset package on sys.path")
            if self._ref is None:
                from .utils import _warn_delegation
                _warn_delegation("SemanticKitti.__getitem__")
                self._ref = _shadowed.SemanticKitti(self.data_path, self.rotate, self.flip, self.resolution, self.projection, self.resize)
            return self._ref[idx]
        if self._projector is None:
            self._projector = self.projector()
        xyzi, label = self._raw[idx]
        out = self._projector([xyzi], [label])
        return tuple(t[0].cpu() for t in out)          # CPU tensors like the reference (a loader with pin_memory=True pins them)

    def gpu_loader(self, device="cuda", **loader_kwargs):
        """DataLoader(num_workers > 0)-compatible device pipeline over this dataset: yields (range, reflectivity, xyz, normals, semantics)
        batches that already live on `device`."""
        cls = gpu_pipeline.projecting_loader_class(lambda ds: ds.projector(device), lambda ds: getattr(ds, "_raw", None) if hasattr(ds, "projector") else None)
        return cls(self, **loader_kwargs)


# drop-in mode (this file shadows the reference's module of the same import path): names it does not define come from there
from semanticlidarunc_amd._shadow import shadowed_module as _shadowed_module  # noqa: E402

_shadowed = _shadowed_module(__name__, __file__)
