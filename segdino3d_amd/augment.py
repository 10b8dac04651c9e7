"""Train-time augmentation on the HIP device.

Mirrors `segdino3d/datasets/transform/point_cloud_transforms.py` + `wrappers_3d.py` behind the same registry name and
class names: `Scannet200Transforms('train', voxel_size)` = `Compose3D([CustomRandomFlip3D, CustomGlobalRotScaleTrans,
NormalizePointsColor, ElasticTransfrom, ToTensor])`, each called as `t(points, targets)` and writing the same target
keys (`pcd_horizontal_flip`, `pcd_rotation`, `pcd_scale_factor`, `elastic_coords`,
`extra_features['elastic_coords_query2d_pos']`, `coords_voxel_size`, ...).

What differs is where the data lives: `points` [N, 6] and `extra_features['query2d_pos']` [M, 3] are DEVICE tensors (the
scene has already been uploaded by io_scene's prefetcher) and are transformed in place by csrc/augment.hip.  The random
draws are made on the host with `numpy.random` in exactly the reference's order, so a seeded run reproduces the
reference's augmentation (tests/golden/augment.npz); the Gaussian noise volumes of the elastic distortion (a few
hundred KB) are drawn on the host for the same reason and uploaded, their six box blurs and the interpolation at
N + M positions run on the device.  The reference spends ~0.3 s per scene in `scipy.interpolate.RegularGridInterpolator`
here; one `sd3d_scene_stats` read-back per granularity (the noise volume's extent depends on the distorted scene) is
the only synchronisation.  There is no CPU fallback: CPU tensors raise.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib, ops
from .builder import TRANSFORMS
from .io_scene import COLOR_MEAN, COLOR_STD


def _f3(v: Optional[Sequence[float]]):
    return None if v is None else (ctypes.c_float * 3)(*[float(x) for x in v])


def _dev_rows(t: torch.Tensor, name: str, min_cols: int):
    if not t.is_cuda:
        raise RuntimeError(f"augment: {name} must live on the HIP device (no CPU fallback), got {t.device}")
    if t.dtype != torch.float32 or t.dim() != 2 or t.shape[1] < min_cols or t.stride(1) != 1:
        raise ValueError(f"augment: {name} must be a 2-D fp32 tensor with >= {min_cols} contiguous columns")
    return t.data_ptr(), t.stride(0)


def affine_(xyz: torch.Tensor, flip_x=False, flip_y=False, angle=0.0, scale=1.0, trans=None, color_mean=None, color_std=None):
    """In place: flip, rotate about z, scale, translate the first three columns (and normalise columns 3..5)."""
    lib = _lib.load()
    p, ld = _dev_rows(xyz, "points", 6 if color_mean is not None else 3)
    _lib.check(lib.sd3d_augment_points(p, ld, xyz.shape[0], int(flip_x), int(flip_y), float(np.float32(angle)), float(np.float32(scale)),
                                       _f3(trans), _f3(color_mean), _f3(color_std), ops._stream()), "augment_points")
    return xyz


def _query2d(targets):
    ef = targets["extra_features"] if "extra_features" in targets else None
    if ef is None:
        return None
    q = ef["query2d_pos"] if "query2d_pos" in ef else None
    return q


class ToTensor:
    """`:19-22`: the device tensors already are tensors."""

    def __call__(self, points, target):
        return points, target


class Compose3D:
    """`:25-33`."""

    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, points, target):
        for t in self.transforms:
            points, target = t(points, target)
        return points, target


class CustomRandomFlip3D:
    """`:36-157`: x -> -x with probability `flip_ratio_bev_horizontal`, y -> -y with `flip_ratio_bev_vertical`."""

    def __init__(self, flip_ratio_bev_horizontal: float = 0.0, flip_ratio_bev_vertical: float = 0.0, **kwargs):
        for r in (flip_ratio_bev_horizontal, flip_ratio_bev_vertical):
            if r is not None:
                assert isinstance(r, (int, float)) and 0 <= r <= 1
        self.flip_ratio_bev_horizontal = flip_ratio_bev_horizontal
        self.flip_ratio_bev_vertical = flip_ratio_bev_vertical

    def __call__(self, points, targets):
        if "pcd_horizontal_flip" not in targets:
            targets["pcd_horizontal_flip"] = bool(np.random.rand() < self.flip_ratio_bev_horizontal)
        if "pcd_vertical_flip" not in targets:
            targets["pcd_vertical_flip"] = bool(np.random.rand() < self.flip_ratio_bev_vertical)
        fx, fy = targets["pcd_horizontal_flip"], targets["pcd_vertical_flip"]
        if fx or fy:
            affine_(points, flip_x=fx, flip_y=fy)
            q = _query2d(targets)
            if q is not None:
                affine_(q, flip_x=fx, flip_y=fy)
        return points, targets


class CustomGlobalRotScaleTrans:
    """`:167-354`: rotation about z by U(rot_range), scale by U(scale_ratio_range), translation by N(0, translation_std),
    applied in that order to the points and the 2D-query centres (one launch each)."""

    def __init__(self, rot_range=(-0.78539816, 0.78539816), scale_ratio_range=(0.95, 1.05), translation_std=(0, 0, 0), shift_height=False):
        if not isinstance(rot_range, (list, tuple, np.ndarray)):
            rot_range = [-rot_range, rot_range]
        if not isinstance(translation_std, (list, tuple, np.ndarray)):
            translation_std = [translation_std] * 3
        assert all(s >= 0 for s in translation_std), "translation_std should be positive"
        if shift_height:
            raise NotImplementedError                                   # as the reference (`:314-315`)
        self.rot_range, self.scale_ratio_range, self.translation_std = list(rot_range), list(scale_ratio_range), list(translation_std)

    def __call__(self, points, targets):
        angle = np.random.uniform(self.rot_range[0], self.rot_range[1])
        scale = np.random.uniform(self.scale_ratio_range[0], self.scale_ratio_range[1])
        trans = np.random.normal(scale=np.array(self.translation_std, dtype=np.float32), size=3).T.astype(np.float32)
        affine_(points, angle=angle, scale=scale, trans=trans)
        q = _query2d(targets)
        if q is not None:
            affine_(q, angle=angle, scale=scale, trans=trans)
        c, s = float(np.cos(np.float32(angle))), float(np.sin(np.float32(angle)))
        rot_t = torch.tensor([[c, s, 0.0], [-s, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
        targets["pcd_rotation"] = rot_t
        targets["pcd_rotation_angle"] = rot_t if q is not None else angle    # the reference overwrites the angle with the matrix (`:297-300`)
        targets["pcd_scale_factor"] = scale
        targets["pcd_trans"] = trans
        return points, targets


class NormalizePointsColor:
    """`:357-389`."""

    def __init__(self, color_mean, color_std=127.5):
        self.color_mean = color_mean
        self.color_std = color_std

    def __call__(self, points, targets):
        assert points.shape[1] == 6, f"points should have 6 channels (xyz rgb), but got {points.shape[1]}"
        mean = self.color_mean if self.color_mean is not None else (0.0, 0.0, 0.0)
        std = self.color_std if self.color_std is not None else 1.0
        if not isinstance(std, (list, tuple)):
            std = (std,) * 3
        affine_(points, color_mean=mean, color_std=std)
        return points, targets


def blurred_noise(dims, seed_module=np.random, device="cuda"):
    """Three Gaussian noise volumes [3, D0, D1, D2] (host draw, reference order) box-blurred twice along every axis on the device."""
    lib = _lib.load()
    d0, d1, d2 = (int(v) for v in dims)
    host = np.stack([seed_module.randn(d0, d1, d2).astype("float32") for _ in range(3)])
    a = torch.from_numpy(host).to(device)
    b = torch.empty_like(a)
    for axis in (0, 1, 2, 0, 1, 2):
        _lib.check(lib.sd3d_box_blur3(a.data_ptr(), b.data_ptr(), 3, d0, d1, d2, axis, ops._stream()), "box_blur3")
        a, b = b, a
    return a


class ElasticTransfrom:
    """`:392-473` (the reference's spelling).  Writes `targets['elastic_coords']` [N, 3] (voxel units),
    `extra_features['elastic_coords_query2d_pos']` and `coords_voxel_size`; `points` itself is not changed."""

    def __init__(self, gran, mag, voxel_size, p=1.0):
        self.gran, self.mag, self.voxel_size, self.p = gran, mag, voxel_size, p

    def _voxel_units(self, t):
        lib = _lib.load()
        p, ld = _dev_rows(t, "coordinates", 3)
        out = torch.empty(t.shape[0], 3, dtype=torch.float32, device=t.device)
        _lib.check(lib.sd3d_voxel_units(p, ld, t.shape[0], float(self.voxel_size), out.data_ptr(), ops._stream()), "voxel_units")
        return out

    def _displace(self, coords, noise, gran, mag):
        lib = _lib.load()
        _, d0, d1, d2 = noise.shape
        _lib.check(lib.sd3d_elastic_displace(coords.data_ptr(), coords.shape[0], noise.data_ptr(), d0, d1, d2, float(gran), float(mag),
                                             ops._stream()), "elastic_displace")

    def __call__(self, points, targets):
        coords = self._voxel_units(points)
        q = _query2d(targets)
        qc = self._voxel_units(q) if q is not None else None
        if np.random.rand() < self.p:
            for gran, mag in zip(self.gran, self.mag):
                st = ops.scene_stats(coords).cpu().numpy()                   # min xyz, max xyz of the (distorted) coordinates
                extent = np.maximum(np.abs(st[0:3]), np.abs(st[3:6])).astype(np.float32)
                dims = extent.astype(np.int32) // gran + 3                   # `:450`
                noise = blurred_noise(dims, device=points.device)
                self._displace(coords, noise, gran, mag)
                if qc is not None:
                    self._displace(qc, noise, gran, mag)
        targets["elastic_coords"] = coords
        if qc is not None:
            targets["extra_features"]["elastic_coords_query2d_pos"] = qc
        targets["coords_voxel_size"] = self.voxel_size
        return points, targets


@TRANSFORMS.register_module(force=True)
def Scannet200Transforms(scene_set: str, voxel_size=0.02, debug=False) -> Compose3D:
    """`wrappers_3d.py:6-57`."""
    mean, std = COLOR_MEAN, COLOR_STD
    if scene_set == "train":
        return Compose3D([
            CustomRandomFlip3D(flip_ratio_bev_horizontal=0.5, flip_ratio_bev_vertical=0.5),
            CustomGlobalRotScaleTrans(rot_range=[-3.14, 3.14], scale_ratio_range=[0.8, 1.2], translation_std=[0.1, 0.1, 0.1]),
            NormalizePointsColor(color_mean=mean, color_std=std),
            ElasticTransfrom(gran=[6, 20], mag=[40, 160], voxel_size=voxel_size, p=0.5),
            ToTensor()])
    if scene_set in ("val", "test"):
        return Compose3D([NormalizePointsColor(color_mean=mean, color_std=std), ToTensor()])
    raise ValueError(f"unknown {scene_set}")
