"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's train-time augmentation.

Follows `segdino3d/datasets/transform/point_cloud_transforms.py`: `CustomRandomFlip3D` `:36-157`,
`CustomGlobalRotScaleTrans` `:167-354` (rotation about z, scale, translation, in that order), `NormalizePointsColor`
`:357-389`, `ElasticTransfrom` `:392-473`, composed as `Scannet200Transforms('train')` does
(`wrappers_3d.py:27-44`).  Random numbers are drawn from `numpy.random` in the reference's order, so the same seed gives
the same augmentation.  `rotation_3d_in_axis` belongs to mmdet3d (1.4, not vendored by the reference); its published
algorithm for axis = 2, counter-clockwise, is restated in `rotate_z`.  Pinned against the reference itself by
tests/golden/augment.npz (tests/golden/make_golden_aug.py imports the reference's module with stubs for mmdet / mmdet3d /
torchvision).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
from __future__ import annotations

import numpy as np
import scipy.interpolate
import scipy.ndimage

COLOR_MEAN = (0.47793125906962 * 255, 0.4303257521323044 * 255, 0.3749598901421883 * 255)          # wrappers_3d.py:19-26
COLOR_STD = (0.2834475483823543 * 255, 0.27566157565723015 * 255, 0.27018971370874995 * 255)


def draw_parameters(rng_module=np.random, flip_h=0.5, flip_v=0.5, rot_range=(-3.14, 3.14), scale_range=(0.8, 1.2),
                    trans_std=(0.1, 0.1, 0.1)):
    """The random draws of flip + rot/scale/trans in the reference's order (`:139-146`, `:296`, `:339-340`, `:252-253`)."""
    fh = bool(rng_module.rand() < flip_h)
    fv = bool(rng_module.rand() < flip_v)
    angle = rng_module.uniform(rot_range[0], rot_range[1])
    scale = rng_module.uniform(scale_range[0], scale_range[1])
    trans = rng_module.normal(scale=np.array(trans_std, dtype=np.float32), size=3).T.astype(np.float32)
    return dict(flip_h=fh, flip_v=fv, angle=float(angle), scale=float(scale), trans=trans)


def rotate_z(xyz: np.ndarray, angle: float) -> np.ndarray:
    """mmdet3d `rotation_3d_in_axis(points, angle, axis=2)`: points @ [[c, s, 0], [-s, c, 0], [0, 0, 1]] in float32."""
    c, s = np.float32(np.cos(np.float32(angle))), np.float32(np.sin(np.float32(angle)))
    rot_t = np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]], dtype=np.float32)
    return (xyz.astype(np.float32) @ rot_t).astype(np.float32)


def affine(xyz: np.ndarray, prm: dict) -> np.ndarray:
    x = xyz.astype(np.float32).copy()
    if prm["flip_h"]:
        x[:, 0] = -x[:, 0]
    if prm["flip_v"]:
        x[:, 1] = -x[:, 1]
    x = rotate_z(x, prm["angle"])
    x = x * np.float32(prm["scale"])
    return x + prm["trans"]


def normalize_color(rgb: np.ndarray) -> np.ndarray:
    return (rgb.astype(np.float32) - np.array(COLOR_MEAN, dtype=np.float32)) / np.array(COLOR_STD, dtype=np.float32)


def elastic_noise(coords: np.ndarray, gran: float, rng_module=np.random):
    """Blurred noise grids + their axes for one granularity (`:446-466`)."""
    noise_dim = np.abs(coords).max(0).astype(np.int32) // gran + 3
    noise = [rng_module.randn(noise_dim[0], noise_dim[1], noise_dim[2]).astype("float32") for _ in range(3)]
    blurs = [np.ones(s, dtype="float32") / 3 for s in ((3, 1, 1), (1, 3, 1), (1, 1, 3))]
    for blur in blurs + blurs:
        noise = [scipy.ndimage.convolve(n, blur, mode="constant", cval=0) for n in noise]
    ax = [np.linspace(-(b - 1) * gran, (b - 1) * gran, b) for b in noise_dim]
    return noise, ax


def elastic_apply(coords: np.ndarray, noise, ax, mag: float) -> np.ndarray:
    interp = [scipy.interpolate.RegularGridInterpolator(ax, n, bounds_error=0, fill_value=0) for n in noise]
    return coords + np.hstack([i(coords)[:, None] for i in interp]) * mag


def elastic(xyz: np.ndarray, query2d: np.ndarray | None, voxel_size: float, gran=(6, 20), mag=(40, 160), p=0.5, rng_module=np.random):
    """`ElasticTransfrom.__call__` (`:410-431`): coordinates in voxel units, two granularities, the SAME noise for the
    2D-query centres.  -> (elastic_coords float32, elastic query coords (float64, as the reference leaves them) or None, applied)."""
    coords = xyz.astype(np.float32) / voxel_size
    q = None if query2d is None else query2d.astype(np.float32) / voxel_size
    applied = bool(rng_module.rand() < p)
    if applied:
        for g, m in zip(gran, mag):
            noise, ax = elastic_noise(coords, g, rng_module)
            coords = elastic_apply(coords, noise, ax, m)
            if q is not None:
                q = elastic_apply(q, noise, ax, m)
    return coords.astype(np.float32), q, applied


def train_transform(points: np.ndarray, query2d_pos: np.ndarray | None, voxel_size: float = 0.02, rng_module=np.random):
    """`Scannet200Transforms('train')`: -> dict(points [N,6] float32, query2d_pos, elastic_coords, elastic_coords_query2d_pos, ...)."""
    prm = draw_parameters(rng_module)
    pts = points.astype(np.float32).copy()
    pts[:, :3] = affine(pts[:, :3], prm)
    q = None if query2d_pos is None else affine(query2d_pos, prm)
    pts[:, 3:] = normalize_color(pts[:, 3:])
    ec, eq, applied = elastic(pts[:, :3], q, voxel_size, rng_module=rng_module)
    return dict(points=pts, query2d_pos=q, elastic_coords=ec, elastic_coords_query2d_pos=eq, elastic_applied=applied, **prm)
