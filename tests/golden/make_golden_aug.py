"""Generates tests/golden/augment.npz by running the REFERENCE's train-time transforms
(/root/reference/segdino3d/datasets/transform/{point_cloud_transforms,wrappers_3d}.py: Scannet200Transforms('train') =
CustomRandomFlip3D -> CustomGlobalRotScaleTrans -> NormalizePointsColor -> ElasticTransfrom -> ToTensor) on seeded
synthetic scenes with seeded numpy.random.  Runs in the build container only; the fixture it writes is data.
Third-party imports of those files are stubbed: mmdet's RandomFlip and mmdet3d's GlobalRotScaleTrans (base classes whose
methods the reference overrides or never calls), torchvision (unused by the 3D transforms) and mmdet3d's
rotation_3d_in_axis - the latter with the published mmdet3d 1.4 algorithm for a single angle about one axis."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _RandomFlip:
    def __init__(self, prob=None, direction="horizontal", **kwargs):
        self.prob, self.direction = prob, direction


class _GlobalRotScaleTrans:
    pass


def rotation_3d_in_axis(points, angles, axis=0, return_mat=False, clockwise=False):
    """mmdet3d.structures.bbox_3d.utils.rotation_3d_in_axis (v1.4), batch of point sets [B, N, 3], angles [B] or scalar."""
    if angles.dim() == 0:
        angles = angles.reshape(1)
    rot_sin, rot_cos = torch.sin(angles), torch.cos(angles)
    ones, zeros = torch.ones_like(rot_cos), torch.zeros_like(rot_cos)
    if axis in (2, -1):
        rot_mat_T = torch.stack([torch.stack([rot_cos, rot_sin, zeros]), torch.stack([-rot_sin, rot_cos, zeros]),
                                 torch.stack([zeros, zeros, ones])])
    elif axis in (1, -2):
        rot_mat_T = torch.stack([torch.stack([rot_cos, zeros, -rot_sin]), torch.stack([zeros, ones, zeros]),
                                 torch.stack([rot_sin, zeros, rot_cos])])
    else:
        rot_mat_T = torch.stack([torch.stack([ones, zeros, zeros]), torch.stack([zeros, rot_cos, rot_sin]),
                                 torch.stack([zeros, -rot_sin, rot_cos])])
    if clockwise:
        rot_mat_T = rot_mat_T.transpose(0, 1)
    new = torch.einsum("aij,jka->aik", points, rot_mat_T)
    return (new, torch.einsum("jka->ajk", rot_mat_T)) if return_mat else new


class _Registry:
    def register_module(self, *a, **k):
        return lambda f: f


_stub("torchvision"); _stub("torchvision.transforms"); _stub("torchvision.transforms.functional")
_stub("mmdet"); _stub("mmdet.datasets"); _stub("mmdet.datasets.transforms", RandomFlip=_RandomFlip)
_stub("mmdet3d"); _stub("mmdet3d.datasets"); _stub("mmdet3d.datasets.transforms", GlobalRotScaleTrans=_GlobalRotScaleTrans)
_stub("mmdet3d.structures"); _stub("mmdet3d.structures.bbox_3d")
_stub("mmdet3d.structures.bbox_3d.utils", rotation_3d_in_axis=rotation_3d_in_axis)
if not hasattr(__import__("scipy.ndimage").ndimage, "filters"):
    import scipy.ndimage as _ndi
    _ndi.filters = _ndi                                   # scipy >= 1.10 folded ndimage.filters into ndimage (`:457`)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


T = _load("segdino3d.datasets.transform.point_cloud_transforms", "/root/reference/segdino3d/datasets/transform/point_cloud_transforms.py")
seg = _stub("segdino3d", TRANSFORMS=_Registry())
_stub("segdino3d.datasets"); tr = _stub("segdino3d.datasets.transform", point_cloud_transforms=T)
W = _load("segdino3d.datasets.transform.wrappers_3d", "/root/reference/segdino3d/datasets/transform/wrappers_3d.py")


def make_scene(seed, n, m):
    g = np.random.default_rng(seed)
    xyz = (g.random((n, 3)) * np.array([7.5, 5.5, 2.8]) - np.array([3.0, 2.0, 0.2])).astype(np.float32)
    rgb = (g.random((n, 3)) * 255).astype(np.float32)
    q = xyz[g.choice(n, m, replace=False)] + g.normal(0, 0.1, (m, 3)).astype(np.float32)
    return np.concatenate([xyz, rgb], 1), q.astype(np.float32)


def main():
    blob = {}
    cases = [(0, 11), (1, 5), (2, 23), (3, 8), (4, 2)]                # (scene seed, numpy.random seed): cover flips and elastic on / off
    for ci, (sseed, rseed) in enumerate(cases):
        pts, q = make_scene(sseed, 2400, 24)
        blob[f"c{ci}/points_in"], blob[f"c{ci}/query2d_pos_in"] = pts, q
        tf = W.Scannet200Transforms("train", voxel_size=0.02)
        np.random.seed(rseed)
        target = {"extra_features": {"query2d_pos": torch.from_numpy(q.copy())}}
        out_pts, tgt = tf(torch.from_numpy(pts.copy()), target)
        blob[f"c{ci}/seed"] = np.array(rseed)
        blob[f"c{ci}/points"] = out_pts.numpy()
        blob[f"c{ci}/query2d_pos"] = tgt["extra_features"]["query2d_pos"].numpy()
        blob[f"c{ci}/elastic_coords"] = tgt["elastic_coords"].numpy()
        blob[f"c{ci}/elastic_coords_query2d_pos"] = tgt["extra_features"]["elastic_coords_query2d_pos"].numpy()
        blob[f"c{ci}/flags"] = np.array([tgt["pcd_horizontal_flip"], tgt["pcd_vertical_flip"]])
        blob[f"c{ci}/scale"] = np.array(tgt["pcd_scale_factor"])
        blob[f"c{ci}/rotation"] = tgt["pcd_rotation"].numpy()       # rot_mat_T; `pcd_rotation_angle` ends up holding the same matrix (:300)
        moved = np.abs(tgt["elastic_coords"].numpy() - out_pts.numpy()[:, :3] / 0.02).max()
        print(ci, "flips", blob[f"c{ci}/flags"], "elastic displacement (voxels)", float(moved))
    # val transform on one scene (colour normalisation only)
    pts, q = make_scene(9, 1200, 10)
    out_pts, _ = W.Scannet200Transforms("val")(torch.from_numpy(pts.copy()), {"extra_features": {"query2d_pos": torch.from_numpy(q)}})
    blob["val/points_in"], blob["val/points"] = pts, out_pts.numpy()
    path = os.path.join(HERE, "augment.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
