"""Mirror of the reference's ``utils.inputs`` (src/utils/inputs.py:4-34): builds the model input list.

SalsaNext-style nets take one tensor ``[range, (reflectivity), x, y, z, (normals)]``; the
Reichert FPN takes ``[range,(reflectivity)]`` plus a metadata tensor ``[xyz,(normals)]``.
The concat is plain torch (2.6 MB per 64x2048 scan) -- it defines the channel order, nothing more.
"""
import torch


def set_model_inputs(range_img, reflectivity, xyz, normals, cfg):
    settings = cfg["model_settings"]
    baseline = settings["baseline"].lower()
    head = [range_img] + ([reflectivity] if settings.get("reflectivity", 0) else [])
    with_normals = bool(settings.get("normals", 0))
    if baseline in ("salsanext", "salsanextadf"):
        tail = [xyz] + ([normals] if with_normals else [])
        return [torch.cat(head + tail, dim=1)]
    if baseline == "reichert":
        meta = torch.cat([xyz, normals], dim=1) if with_normals else xyz
        return [torch.cat(head, dim=1), meta]
    raise ValueError(f"Unknown baseline: {settings['baseline']}")


# drop-in mode (this file shadows the reference's module of the same import path): names it does not define come from there
from semanticlidarunc_amd._shadow import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(__name__, __file__, globals())
