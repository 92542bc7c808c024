"""Oracle: spherical projection of a point cloud into the range image.  TEST INFRASTRUCTURE ONLY.

Restates ``src/dataset/utils.py:61-67`` (deflection coordinates) and ``:288-349`` (``spherical_projection``) in numpy, line by
line (argsort by range, reversed linspace bins, ``np.digitize`` - 1, fancy-index assignment in descending range order).  Pinned by
``tools/gen_golden.py`` against the imported reference function (bit-identical image)."""
from __future__ import annotations

import numpy as np


def deflection(x, y, z):
    return np.arctan2(y, x), -np.arctan2(np.sqrt(x ** 2 + y ** 2), z) + np.pi / 2


def spherical_projection(pc, height=64, width=2048, theta_range=None, sort_largest_first=False, bins_h=None):
    r = np.sqrt(pc[:, 0] ** 2 + pc[:, 1] ** 2 + pc[:, 2] ** 2)
    order = r.argsort()
    # default: farthest first, so the nearest point is written last and survives; sort_largest_first=True (utils.py:301-304) writes in
    # ASCENDING range order instead -- the farthest point survives
    pc = pc[order] if sort_largest_first else pc[order[::-1]]
    phi, theta = deflection(pc[:, 0], pc[:, 1], pc[:, 2])
    theta_min, theta_max = (theta.min(), theta.max()) if theta_range is None else theta_range
    if bins_h is None:
        bins_h = np.linspace(theta_min, theta_max, height)[::-1]
    bins_w = np.linspace(-np.pi, np.pi, width)[::-1]
    idx_h = np.digitize(theta, bins_h) - 1
    idx_w = np.digitize(phi, bins_w) - 1
    img = np.zeros((height, width, pc.shape[1])).astype(np.float32)
    img[idx_h, idx_w, :] = pc
    alpha = np.sqrt(np.square(np.stack(width * [bins_h], axis=-1)) + np.square(np.stack(height * [bins_w], axis=0)))
    return img, alpha, (theta_min, theta_max), (-np.pi, np.pi)
