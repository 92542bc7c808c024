"""CPU oracle (test infrastructure only) for the surface normals of the projected range image.

Restates ``build_normal_xyz`` of the reference (``src/dataset/utils.py:30-58``): six ``cv2.Scharr`` derivatives, the cross product
and the normalisation.  PARITY UNPINNED for the Scharr part: the derivative lives in a third-party dependency that is absent from
this image (opencv-python, "tested version 4.11.0.86" in the reference's ``docker/requirements.txt``), so it is restated from
OpenCV's published definition -- 3x3 taps [-1, 0, 1] (derivative) x [3, 10, 3] (smoothing), border ``BORDER_REFLECT_101``
(gfedcb|abcdefgh|gfedcba), output multiplied by ``scale`` -- and the function is anchored on the reference's call
(``scale = 1 / norm_factor``, float32 planes, ``-dstack`` of the cross product, ``/ (norm + 1e-10)``) and on known answers
(planes, a sphere) in ``tests/test_oracle.py``.  No golden fixture exists for it.
"""
from __future__ import annotations

import numpy as np


def scharr(img, dx: int, dy: int, scale: float = 1.0):
    """cv2.Scharr(img, CV_32F, dx, dy, scale) for (dx, dy) in {(1, 0), (0, 1)} on a float32 plane."""
    assert (dx, dy) in ((1, 0), (0, 1))
    a = np.pad(np.asarray(img, np.float32), 1, mode="reflect")          # numpy 'reflect' == BORDER_REFLECT_101
    k3, k10 = np.float32(3.0 * scale), np.float32(10.0 * scale)
    if dx == 1:
        r = a[:, 2:] - a[:, :-2]                                        # derivative along the columns, then smoothing down the rows
        return k10 * r[1:-1] + k3 * (r[:-2] + r[2:])
    s = k10 * a[:, 1:-1] + k3 * (a[:, :-2] + a[:, 2:])                  # smoothing along the columns, then derivative down the rows
    return s[2:] - s[:-2]


def build_normal_xyz(xyz, norm_factor: float = 0.25):
    """(H, W, >= 3) staggered point image -> (H, W, 3) float32 unit normals (dataset/utils.py:30-58)."""
    planes = [np.asarray(xyz[..., c], np.float32) for c in range(3)]
    sc = 1.0 / norm_factor
    (sxx, sxy), (syx, syy), (szx, szy) = [(scharr(p, 1, 0, sc), scharr(p, 0, 1, sc)) for p in planes]
    n = -np.dstack((syx * szy - szx * syy, szx * sxy - szy * sxx, sxx * syy - syx * sxy))
    return (n / (np.linalg.norm(n, axis=2) + np.float32(1e-10))[..., None]).astype(np.float32)
