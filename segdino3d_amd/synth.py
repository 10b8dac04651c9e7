"""Seeded synthetic ScanNet-like scenes (SURVEY.md 8(d) "Synthetic scene generator").

The reference ships no data and no benchmark; its input contract is defined by
`segdino3d/datasets/dataset/scannet200.py:198-289` (points [N,6] f32 = xyz + normalised rgb,
`extra_features` = points_2dfeats [N,256], query2d_feats [M,256], query2d_pos [M,3],
super_point_masks [N] i64) and `evaluation/evaluator_3d.py:80-86`.  This module synthesises
scenes with the same layout so that the forward path can be exercised and timed without
ScanNet: surface-like geometry (room shell + cuboid "objects") so that voxel occupancy per
level resembles a real scan.

All randomness comes from `torch.Generator().manual_seed(1234 + scene_idx)` on the CPU, so a
scene is bit-identical in the build container and on the GPU box.
"""
from __future__ import annotations

import torch

from .gtypes import GD3DTarget

ROOM = (8.0, 6.0, 3.0)


def _sample_box_surface(n: int, lo: torch.Tensor, hi: torch.Tensor, g: torch.Generator) -> torch.Tensor:
    """n points uniformly (area-weighted) on the six faces of the axis-aligned box [lo, hi]."""
    ext = hi - lo
    area = torch.stack([ext[1] * ext[2], ext[1] * ext[2], ext[0] * ext[2],
                        ext[0] * ext[2], ext[0] * ext[1], ext[0] * ext[1]])
    face = torch.multinomial(area / area.sum(), n, replacement=True, generator=g)
    p = lo + torch.rand(n, 3, generator=g) * ext
    axis = face // 2
    side = (face % 2).to(p.dtype)
    fixed = lo[axis] + side * ext[axis]
    p[torch.arange(n), axis] = fixed
    return p


def _lattice_box_surface(lo: torch.Tensor, hi: torch.Tensor, spacing: float, g: torch.Generator) -> torch.Tensor:
    """Jittered lattice (mesh-vertex-like, near-regular) on the six faces of the box [lo, hi]."""
    out = []
    for axis in range(3):
        a, b = [i for i in range(3) if i != axis]
        na, nb = max(1, int((hi[a] - lo[a]) / spacing)), max(1, int((hi[b] - lo[b]) / spacing))
        ga, gb = torch.meshgrid(torch.arange(na), torch.arange(nb), indexing="ij")
        for side in (lo[axis], hi[axis]):
            p = torch.zeros(na * nb, 3)
            p[:, a] = lo[a] + (ga.reshape(-1) + 0.5) * spacing
            p[:, b] = lo[b] + (gb.reshape(-1) + 0.5) * spacing
            p[:, axis] = side
            out.append(p)
    p = torch.cat(out)
    return p + 0.3 * spacing * (torch.rand(p.shape, generator=g) - 0.5)


SCAN_ROOM = (5.0, 4.0, 2.6)


def make_scene(scene_idx: int = 0, n_points: int = 150_000, n_superpoints: int = 3000,
               n_query2d: int = 300, feat2d_dim: int = 256, n_objects: int = 20,
               device: str | torch.device = "cpu", layout: str = "benchmark"):
    """Returns (points [N,6] f32, GD3DTarget) laid out like the reference dataset output.
    layout "benchmark": uniformly random points on an 8 x 6 x 3 m shell + 20 cuboids (the scene every reported number uses: at
    150 k points ~0.3 points per occupied 2 cm cell, 3 neighbours per voxel at level 0).  layout "scan": a 5 x 4 x 2.6 m room whose
    surfaces carry a jittered near-regular lattice like the vertices of a reconstructed mesh (~9 neighbours per voxel at level 0,
    voxel counts falling ~3x per level: the occupancy statistics of a real ScanNet scan)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(1234 + scene_idx)
    if layout == "scan":
        room = torch.tensor(SCAN_ROOM)
        boxes = [(torch.zeros(3), room)]
        for k in range(n_objects):
            edge = 0.3 + 0.9 * torch.rand(3, generator=g)
            lo = torch.rand(3, generator=g) * (room - edge)
            lo[2] = 0.0 if k % 2 == 0 else lo[2]
            boxes.append((lo, lo + edge))
        area = sum(float(2 * ((h - l)[0] * (h - l)[1] + (h - l)[1] * (h - l)[2] + (h - l)[0] * (h - l)[2])) for l, h in boxes)
        spacing = (area / n_points) ** 0.5
        xyz = torch.cat([_lattice_box_surface(l, h, spacing, g) for l, h in boxes])
        if xyz.shape[0] >= n_points:
            xyz = xyz[torch.randperm(xyz.shape[0], generator=g)[:n_points]]
        else:
            xyz = torch.cat([xyz, _sample_box_surface(n_points - xyz.shape[0], torch.zeros(3), room, g)])
        xyz = xyz + 0.002 * torch.randn(n_points, 3, generator=g)
    elif layout == "benchmark":
        room = torch.tensor(ROOM)
        n_room = n_points // 2
        n_obj_total = n_points - n_room
        pts = [_sample_box_surface(n_room, torch.zeros(3), room, g)]
        per = [n_obj_total // n_objects] * n_objects
        per[-1] += n_obj_total - sum(per)
        for k in range(n_objects):
            edge = 0.3 + 1.2 * torch.rand(3, generator=g)
            lo = torch.rand(3, generator=g) * (room - edge)
            lo[2] = 0.0 if k % 2 == 0 else lo[2]          # half of the objects stand on the floor
            pts.append(_sample_box_surface(per[k], lo, lo + edge, g))
        xyz = torch.cat(pts) + 0.005 * torch.randn(n_points, 3, generator=g)
    else:
        raise ValueError(f"unknown scene layout {layout!r}")
    perm = torch.randperm(n_points, generator=g)          # scans are not spatially sorted
    xyz = xyz[perm].contiguous()
    rgb = torch.randn(n_points, 3, generator=g)
    points = torch.cat([xyz, rgb], dim=1).float().contiguous()
    feats2d = torch.randn(n_points, feat2d_dim, generator=g)

    # superpoints = Voronoi cells of S seed points drawn from the cloud: every id 0..S-1 is used
    seed_idx = torch.randperm(n_points, generator=g)[:n_superpoints]
    seeds = xyz[seed_idx]
    sp = torch.empty(n_points, dtype=torch.long)
    chunk = 16384
    for s in range(0, n_points, chunk):
        d = torch.cdist(xyz[s:s + chunk], seeds)
        sp[s:s + chunk] = d.argmin(dim=1)
    sp[seed_idx] = torch.arange(n_superpoints)

    q_idx = torch.randint(0, n_points, (n_query2d,), generator=g)
    q_pos = xyz[q_idx] + 0.1 * torch.randn(n_query2d, 3, generator=g)
    q_feat = torch.randn(n_query2d, feat2d_dim, generator=g)

    masks = torch.zeros(1, n_points, 1, dtype=torch.bool)
    masks[0, : n_points // 10, 0] = True
    target = GD3DTarget(
        labels=torch.zeros(1, dtype=torch.long),
        masks=masks,
        scene_id=f"synthetic_{scene_idx:04d}",
        extra_features={
            "points_2dfeats": feats2d.contiguous(),
            "query2d_feats": q_feat.contiguous(),
            "query2d_pos": q_pos.contiguous(),
            "super_point_masks": sp,
        },
    )
    if str(device) != "cpu":
        points = points.to(device)
        target = target.to(device)
    return points, target


def add_training_targets(points: torch.Tensor, target, n_instances: int = 12, n_semantic: int = 200, n_instance_classes: int = 198,
                         seed: int = 0):
    """Superpoint-coherent instance / semantic labels for the training step (the reference's losses are NaN
    unless every superpoint carries one label): every superpoint goes to the nearest of `n_instances` random object centres
    (or to "no object" when it is far from all of them).  Adds the keys `Baseline3D.forward` and the criterion read:
    `masks` [G, N, 1] bool, `labels` [G], `sp_inst_sem_masks` [G + n_semantic + 1, S] bool (`scannet200.py:246-253`)."""
    g = torch.Generator().manual_seed(4321 + seed)
    sp = target.extra_features["super_point_masks"].cpu()
    xyz = points[:, :3].cpu()
    S = int(sp.max()) + 1
    cnt = torch.bincount(sp, minlength=S).clamp(min=1).float()
    ctr = torch.zeros(S, 3).index_add_(0, sp, xyz) / cnt[:, None]
    seeds = ctr[torch.randperm(S, generator=g)[:n_instances]]
    dist = torch.cdist(ctr, seeds)
    owner = dist.argmin(1)
    owner[dist.min(1)[0] > 1.5] = n_instances                     # background superpoints
    keep = [k for k in range(n_instances) if bool((owner == k).any())]
    inst = torch.stack([owner == k for k in keep])                 # [G, S]
    G = inst.shape[0]
    labels = torch.randint(0, n_instance_classes, (G,), generator=g)
    sem_of = torch.randint(2, n_semantic, (G + 1,), generator=g)   # classes 0 / 1 = stuff (wall, floor)
    sem_id = torch.full((S,), n_semantic, dtype=torch.long)        # unlabeled unless it belongs to something
    for j, k in enumerate(keep):
        sem_id[owner == k] = sem_of[j]
    sem_id[(owner == n_instances) & (ctr[:, 2] < 0.3)] = 1
    sem = torch.stack([sem_id == c for c in range(n_semantic + 1)])
    target.sp_inst_sem_masks = torch.cat([inst, sem]).to(points.device)
    target.masks = inst[:, sp].unsqueeze(-1).to(points.device)
    target.labels = labels.to(points.device)
    return target


def structure_scene(points: torch.Tensor, target, seed: int = 5, embed_scale: float = 2.0, noise: float = 0.1):
    """Give a `make_scene` scene per-superpoint structure, in place: iid-noise inputs and random weights give every
    superpoint the same features (and every predicted mask the same points), so post-processing would be exercised on
    empty or degenerate selections.  2D features and colours become a per-superpoint embedding plus noise."""
    g = torch.Generator().manual_seed(seed)
    ef = target.extra_features
    sp = ef["super_point_masks"].cpu()
    S = int(sp.max()) + 1
    n, c = ef["points_2dfeats"].shape
    f2d = (torch.randn(S, c, generator=g) * embed_scale)[sp] + noise * torch.randn(n, c, generator=g)
    rgb = torch.randn(S, 3, generator=g)[sp]
    ef["points_2dfeats"] = f2d.to(ef["points_2dfeats"].device)
    points[:, 3:] = rgb.to(points.device)
    return points, target


def sharpen_random_model(model, seed: int = 1, mask_gain: float = 40.0):
    """Random-init weights that yield a non-trivial operating point: non-identity BatchNorm running statistics and a
    sharpened mask branch, so that a prediction switches on a handful of superpoints.  In place; returns the model."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(0.1 * torch.randn(m.num_features, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.num_features, generator=g))
        model.decoder.x_mask[2].weight.mul_(mask_gain)
        model.decoder.x_mask[2].bias.zero_()
    return model
