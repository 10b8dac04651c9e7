"""ORACLE (test infrastructure, not product code): CPU restatement of SegDINO3D post-processing.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.

Restates (single scene, batch size 1 as the reference requires, baseline3d.py:335-338):
  - `segdino3d/models/architecture/baseline3d.py:22-141`   mask_matrix_nms (linear / gaussian)
  - `.../baseline3d.py:406-486`   predict_by_feat_instance
  - `.../baseline3d.py:348-371`   filter_outofbox_points
  - `.../baseline3d.py:488-507`   predict_by_feat_semantic
  - `.../baseline3d.py:509-556`   predict_by_feat_panoptic
  - `.../baseline3d.py:266-306`   get_extra_instance_data (scene range, GT centres/sizes)
PINNED by tests/test_oracle_golden.py against golden vectors produced by the imported reference.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import torch


@dataclass
class TestCfg:
    topk_insts: int = 600
    inst_score_thr: float = 0.0
    pan_score_thr: float = 0.5
    npoint_thr: int = 100
    obj_normalization: bool = True
    sp_score_thr: float = 0.4
    nms: bool = True
    matrix_nms_kernel: str = "linear"
    stuff_classes: List[int] = field(default_factory=lambda: [0, 1])


def matrix_nms(masks, labels, scores, kernel="linear", sigma=2.0):
    """Soft-mask matrix NMS.  Returns (scores, labels, masks, keep_inds, record) sorted by the
    decayed score; `record[i]` = index into the INPUT arrays of output row i (:60,:72,:139)."""
    n = scores.shape[0]
    area = masks.sum(1).float()
    scores, order = torch.sort(scores, descending=True)
    masks, area, labels = masks[order], area[order], labels[order]
    flat = masks.reshape(n, -1).float()
    inter = flat @ flat.t()
    iou = (inter / (area[None, :] + area[:, None] - inter)).triu(diagonal=1)
    same = (labels[None, :] == labels[:, None]).triu(diagonal=1)
    decay_iou = iou * same
    comp = decay_iou.max(0)[0][:, None].expand(n, n)      # comp[i, j] = compensate of row i
    if kernel == "gaussian":
        coef = (torch.exp(-sigma * decay_iou ** 2) / torch.exp(-sigma * comp ** 2)).min(0)[0]
    elif kernel == "linear":
        coef = ((1 - decay_iou) / (1 - comp)).min(0)[0]
    else:
        raise NotImplementedError(kernel)
    scores = scores * coef
    scores, order2 = torch.sort(scores, descending=True)
    return scores, labels[order2], masks[order2], order[order2], order[order2]


def scene_range_and_gt_boxes(xyz, gt_masks=None, mode="median"):
    """get_extra_instance_data (:266-306): (lo, hi) of raw xyz + per-GT-instance centre/size."""
    lo, hi = xyz.min(0)[0], xyz.max(0)[0]
    centers = sizes = None
    if gt_masks is not None:
        m = gt_masks[..., 0] if gt_masks.dim() == 3 else gt_masks
        n = m.shape[0]
        centers, sizes = torch.zeros(n, 3), torch.zeros(n, 3)
        for j in range(n):
            p = xyz[m[j]]
            if p.shape[0] == 0:
                continue
            pmin, pmax = p.min(0)[0], p.max(0)[0]
            centers[j] = p.mean(0) if mode == "mean" else (pmax + pmin) / 2
            sizes[j] = pmax - pmin
    return lo, hi, centers, sizes


def filter_outofbox(points, mask, centers, sizes, loose_ratio=1.5):
    s = sizes * (1 + loose_ratio)
    lo, hi = centers - s / 2, centers + s / 2
    inside = ((points[None, :, :] >= lo[:, None, :]) & (points[None, :, :] <= hi[:, None, :])).all(dim=2)
    return mask & inside


def predict_instance(cls_preds, mask_logits, superpoints, points, centers, sizes, num_classes,
                     cfg: TestCfg, score_thr: float, box_filter: bool):
    """predict_by_feat_instance (:406-486).  Returns dict with mask [n,N] bool, labels, scores,
    boxes [n,6], (topk_query_idx, score_mask, npoint_mask)."""
    Q = cls_preds.shape[0]
    prob = torch.softmax(cls_preds, dim=-1)[:, :-1]
    flat = prob.flatten()
    k = cfg.topk_insts
    scores, flat_idx = flat.topk(k, sorted=True)
    labels = flat_idx % num_classes
    qidx = torch.div(flat_idx, num_classes, rounding_mode="floor")
    logit = mask_logits[qidx]
    sig = torch.sigmoid(logit)
    if cfg.obj_normalization:
        pos = logit > 0
        scores = scores * ((sig * pos).sum(1) / (pos.sum(1) + 1e-6))
    if cfg.nms:
        scores, labels, sig, _, record = matrix_nms(sig, labels, scores, kernel=cfg.matrix_nms_kernel)
    else:
        record = torch.arange(k)
    pt_mask = sig[:, superpoints] > cfg.sp_score_thr
    score_mask = scores > score_thr
    scores, labels, pt_mask, record = scores[score_mask], labels[score_mask], pt_mask[score_mask], record[score_mask]
    npoint_mask = pt_mask.sum(1) > cfg.npoint_thr
    scores, labels, pt_mask, record = scores[npoint_mask], labels[npoint_mask], pt_mask[npoint_mask], record[npoint_mask]
    c = centers[qidx][record] if centers is not None else None
    s = sizes[qidx][record] if sizes is not None else None
    boxes = torch.cat([c, s], dim=-1) if c is not None and s is not None else None
    if box_filter:
        pt_mask = filter_outofbox(points, pt_mask, c, s)
    return dict(masks=pt_mask, labels=labels, scores=scores, boxes=boxes,
                sort_and_mask=(qidx, score_mask, npoint_mask), record=record)


def predict_semantic(sem_preds, superpoints, classes=None, query_num=-1):
    if classes is None:
        classes = list(range(sem_preds.shape[1] - 1))
    am = sem_preds[:, classes].argmax(dim=1)
    if query_num == -1:
        return am[superpoints]
    return am[torch.zeros_like(superpoints)]


def predict_panoptic(cls_preds, sem_preds, mask_logits, superpoints, points, centers, sizes, num_classes,
                     cfg: TestCfg, box_filter: bool, query_num=-1):
    sem_map = predict_semantic(sem_preds, superpoints, cfg.stuff_classes, query_num)
    r = predict_instance(cls_preds, mask_logits, superpoints, points, centers, sizes, num_classes, cfg,
                         cfg.pan_score_thr, box_filter)
    mask, labels, scores = r["masks"], r["labels"], r["scores"]
    if mask.shape[0] == 0:
        return sem_map, sem_map
    scores, idxs = scores.sort()
    labels, mask = labels[idxs], mask[idxs]
    n_stuff = len(cfg.stuff_classes)
    inst_ids = torch.arange(n_stuff, mask.shape[0] + n_stuff).view(-1, 1)
    things_inst, arg = (inst_ids * mask).max(dim=0)
    things_sem = labels[arg] + n_stuff
    ids, cnt = things_inst.unique(return_counts=True)
    for i, c in zip(ids.tolist(), cnt.tolist()):
        if c <= cfg.npoint_thr and i != 0:
            things_inst[things_inst == i] = 0
    things_sem[things_inst == 0] = 0
    sem_map = sem_map.clone()
    sem_map[things_inst != 0] = 0
    inst_map = sem_map.clone() + things_inst
    sem_map = sem_map + things_sem
    return sem_map, inst_map


def predict_by_feat(out, superpoints, points, num_classes, cfg: TestCfg, box_filter: bool, query_num=-1):
    """predict_by_feat (:373-404) -> dict laid out like the reference PointData."""
    args = (out["masks"], superpoints, points, out["centers"], out["sizes"], num_classes, cfg)
    inst = predict_instance(out["cls_preds"], *args, cfg.inst_score_thr, box_filter)
    sem = predict_semantic(out["sem_preds"], superpoints, None, query_num)
    pan = predict_panoptic(out["cls_preds"], out["sem_preds"], *args, box_filter, query_num)
    return dict(pts_semantic_mask=[sem, pan[0]], pts_instance_mask=[inst["masks"], pan[1]],
                instance_labels=inst["labels"], instance_scores=inst["scores"],
                sort_and_mask=inst["sort_and_mask"], instance_boxes=inst["boxes"])
