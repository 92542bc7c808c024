"""Drop-in for ``spherical_projection`` / ``to_deflection_coordinates`` / ``build_normal_xyz`` of the reference's ``dataset/utils.py``
(:61-67, :288-349, :30-58): same arguments and return values (numpy arrays), the projection itself (angles, bin search,
nearest-point selection, gather) and the Scharr / cross-product normals running in HIP kernels (``csrc/projection.hip``).
Every other helper of the reference module (plots, rotations) is re-exported from the shadowed file in drop-in mode."""
from __future__ import annotations

import numpy as np
import torch
import torch.utils.data

from semanticlidarunc_amd import ops


def to_deflection_coordinates(x, y, z):
    p = np.sqrt(x ** 2 + y ** 2)
    phi = np.arctan2(y, x)
    theta = -np.arctan2(p, z) + np.pi / 2
    return phi, theta


_WARNED = set()


def _warn_delegation(what: str) -> None:
    """Once per process: a DataLoader WORKER cannot touch the GPU, so this call runs the shadowed reference code (numpy / cv2) instead of the HIP
    kernels -- say so, and say how to get the device path."""
    if what in _WARNED:
        return
    _WARNED.add(what)
    import warnings
    warnings.warn(f"semanticlidarunc_amd: {what} was called inside a DataLoader worker process and is handled by the reference's CPU code there "
                  "(a forked worker must not use the GPU). To run decode / projection / normals as HIP kernels keep the workers for file reads "
                  "only: SemanticKitti.gpu_loader(...), dataset.gpu_pipeline.projecting_loader_class, or tools/dp_launch.py --gpu-projection "
                  "(or num_workers=0).", RuntimeWarning, stacklevel=3)


def spherical_projection(pc, height=64, width=2048, theta_range=None, th=1.0, sort_largest_first=False, bins_h=None, max_range=None,
                         device="cuda"):
    """pc: [N, C] array or tensor (x, y, z, ...).  The nearest point of a pixel survives (the reference writes the points in
    descending range order) unless sort_largest_first (ascending order: the farthest survives); bins_h: explicit monotone row bins;
    `th` / `max_range` are accepted and unused, as in the reference."""
    if torch.utils.data.get_worker_info() is not None:
        # a forked DataLoader worker must not touch the GPU: hand the call to the module this file shadows (the reference's numpy code)
        if _shadowed is not None:
            _warn_delegation("spherical_projection")
            return _shadowed.spherical_projection(pc, height, width, theta_range, th, sort_largest_first, bins_h, max_range)
        raise RuntimeError("spherical_projection: the HIP projection cannot run inside a DataLoader worker process; "
                           "project in the main process (num_workers=0) or keep the reference's dataset/utils.py on sys.path")
    t = pc if isinstance(pc, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(pc))
    if not t.is_cuda:
        t = t.to(device)
    bins_dev, increasing = None, False
    if bins_h is not None:
        bins_np = np.ascontiguousarray(np.asarray(bins_h, dtype=np.float64)).reshape(-1)
        d = np.diff(bins_np)
        if bins_np.size != int(height) or not (np.all(d > 0) or np.all(d < 0)):
            raise ValueError("bins_h must hold `height` strictly monotone row bins")      # numpy.digitize rejects non-monotone bins too
        increasing = bool(d[0] > 0) if d.size else False
        bins_dev = torch.from_numpy(bins_np).to(t.device)
    img, tr = ops.spherical_projection(t.to(torch.float64).contiguous(), height, width, theta_range, bins_h=bins_dev, bins_increasing=increasing,
                                       keep_farthest=bool(sort_largest_first))
    theta_min, theta_max = (float(v) for v in tr.cpu())
    if theta_range is not None:
        theta_min, theta_max = theta_range
    phi_min, phi_max = -np.pi, np.pi
    if bins_h is None:
        bins_h = np.linspace(theta_min, theta_max, height)[::-1]
    bins_w = np.linspace(phi_min, phi_max, width)[::-1]
    alpha = np.sqrt(np.square(np.stack(width * [bins_h], axis=-1)) + np.square(np.stack(height * [bins_w], axis=0)))
    return img.cpu().numpy(), alpha, (theta_min, theta_max), (phi_min, phi_max)


def build_normal_xyz(xyz, norm_factor=0.25, ksize=3, device="cuda"):
    """(h, w, 3) staggered point image -> (h, w, 3) float32 unit normals (reference :30-58; ``ksize`` is unused there as well).
    A numpy image returns numpy (the reference's contract); a GPU tensor returns a GPU tensor without leaving the device."""
    if torch.utils.data.get_worker_info() is not None:
        if _shadowed is not None:                     # forked DataLoader worker: the reference's cv2 code, as for the projection
            _warn_delegation("build_normal_xyz")
            return _shadowed.build_normal_xyz(xyz, norm_factor, ksize)
        raise RuntimeError("build_normal_xyz: the HIP kernel cannot run inside a DataLoader worker process; compute the normals in the "
                           "main process (num_workers=0) or keep the reference's dataset/utils.py on sys.path")
    as_tensor = isinstance(xyz, torch.Tensor)
    t = xyz if as_tensor else torch.from_numpy(np.ascontiguousarray(xyz))
    out = ops.build_normals(t.to(device=device if not t.is_cuda else t.device, dtype=torch.float32).contiguous(), norm_factor)
    return out if as_tensor else out.cpu().numpy()


# drop-in mode (this file shadows the reference's module of the same import path): names it does not define come from there
from semanticlidarunc_amd._shadow import reexport_missing as _reexport_missing, shadowed_module as _shadowed_module  # noqa: E402

_shadowed = _shadowed_module(__name__, __file__)
_reexport_missing(__name__, __file__, globals())
