"""Oracle: one sample of the SemanticKITTI dataloader, from the raw file contents to the five tensors the Trainer receives.
TEST INFRASTRUCTURE ONLY.

Restates ``src/dataset/dataloader_semantic_KITTI.py:31-99`` (``SemanticKitti.__getitem__``) and ``src/dataset/utils.py:4-18``
(``rotate_z``) in numpy, with the two random augmentation draws (yaw angle, flip decision) as explicit arguments; the projection and
the normals come from ``oracle.projection`` / ``oracle.normals``.  Pinned by ``tools/gen_golden_r02.py`` against the reference class
itself (``cv2.Scharr`` -- absent from this image -- served by ``oracle.normals``' restatement of OpenCV's definition, so the normals
stay "parity unpinned"; every other output is bit-identical)."""
from __future__ import annotations

import numpy as np

from oracle import normals as onormals
from oracle import projection as oproj


def rotate_z(points, angle_deg):
    a = np.radians(angle_deg)
    rot = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    return np.dot(points, rot)


def id_map_lut(id_map: dict) -> np.ndarray:
    """Dense int32 table of the dict (dataset/definitions.py:3-39), -1 where it has no key."""
    lut = np.full(max(id_map) + 1, -1, dtype=np.int32)
    for k, v in id_map.items():
        lut[k] = v
    return lut


def decode(bin_bytes, label_bytes, id_map: dict):
    """(.bin, .label) file contents -> float64 [N, 5] = (x, y, z, intensity, mapped class): dataloader :35-49."""
    xyzi = np.frombuffer(bin_bytes, dtype=np.float32).reshape(-1, 4)
    label = np.frombuffer(label_bytes, dtype=np.uint32).reshape(-1)
    sem = label & 0xFFFF
    mapped = np.array([id_map[int(l)] for l in sem])
    return np.concatenate([xyzi, mapped[..., np.newaxis]], axis=-1)


def resize_nearest(img, out_h, out_w):
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_NEAREST) restated from OpenCV's definition (resizeNN: source index
    min(floor(dst * src_size / dst_size), src_size - 1) per axis) -- cv2 is absent from this image: PARITY UNPINNED, like the normals."""
    h, w = img.shape[:2]
    ys = np.minimum(np.floor(np.arange(out_h) * (h / out_h)).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(out_w) * (w / out_w)).astype(np.int64), w - 1)
    return img[ys][:, xs]


def sample(bin_bytes, label_bytes, id_map: dict, projection=(64, 2048), rotate_angle=None, flip=False, resize=None):
    """-> (range [1,H,W] f32, reflectivity [1,H,W] f32, xyz [3,H,W] f32, normals [3,H,W] f32, semantics [1,H,W] int64): :51-99; resize: None
    or the (rows, columns) of the dataloader's nearest resize (the reference fixes (128, 2048), :61-62)."""
    xyzil = decode(bin_bytes, label_bytes, id_map)
    if rotate_angle is not None:
        xyzil[..., 0:3] = rotate_z(xyzil[..., 0:3].reshape(-1, 3), float(rotate_angle))
    img, _, _, _ = oproj.spherical_projection(xyzil, projection[0], projection[1])
    if resize is not None:
        img = resize_nearest(img, resize[0], resize[1])
    if flip:
        img = img[:, ::-1, :]
        img[..., 1] *= -1
    label_img, refl, xyz = img[..., 4:5], img[..., 3], img[..., 0:3]
    rng = np.linalg.norm(xyz, axis=-1)
    nrm = onormals.build_normal_xyz(xyz[..., 0:3])
    return (rng[..., None].transpose(2, 0, 1).astype("float32"), refl[..., None].transpose(2, 0, 1).astype("float32"),
            xyz.transpose(2, 0, 1).astype("float32"), nrm.transpose(2, 0, 1).astype("float32"), label_img.transpose(2, 0, 1).astype("int64"))
