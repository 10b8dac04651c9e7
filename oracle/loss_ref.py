"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's training criterion.

Follows `segdino3d/models/loss/loss_3d.py`: matching costs `:63-97, 139-271`, `SparseMatcher` `:315-365`,
`HungarianMatcher` `:274-312`, the per-layer instance loss `:398-555` (auxiliary layers) and `:557-710`
(last layer - note the two differ in how the dice term is scaled with the batch size), the semantic loss
`:26-60` and `ScanNetUnifiedCriterion` `:713-780`.  Written as plain functions over torch CPU tensors (any
float dtype: the parity tests run it in float64); gradients come from torch autograd.  Pinned against the
reference itself by tests/golden/loss_criterion.npz (tests/golden/make_golden_loss.py imports the
reference's module in the build container).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this file; the product path is segdino3d_amd/criterion.py + csrc/loss.hip.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

INF_COST = 1e8                                     # loss_3d.py:326


def match_costs(cls, masks, centers, sizes, labels, gt_masks, gt_centers, gt_sizes, weights) -> torch.Tensor:
    """[Q, G] matching cost = w_cls * (-softmax(cls)[:, label]) + w_bce * BCE + w_dice * dice
    + w_ctr * L1(centres) + w_size * L1(sizes)   (loss_3d.py:63-97, 139-271).  `weights` = 5 floats; centre /
    size terms are zero when the layer predicts none (`:237-240, 265-268`)."""
    t = gt_masks.to(masks.dtype)                                               # [G, S]
    n_s = masks.shape[1]
    cost = -weights[0] * cls.softmax(-1)[:, labels]
    softplus_neg, softplus_pos = F.softplus(-masks), F.softplus(masks)         # BCE against all-ones / all-zeros
    cost = cost + weights[1] * (softplus_neg @ t.T + softplus_pos @ (1 - t).T) / n_s
    sig = masks.sigmoid()
    cost = cost + weights[2] * (1 - (2 * sig @ t.T + 1) / (sig.sum(-1)[:, None] + t.sum(-1)[None, :] + 1))
    if centers is not None and weights[3] != 0:
        cost = cost + weights[3] * (centers[:, None, :] - gt_centers[None, :, :3]).abs().sum(-1)
    if sizes is not None and weights[4] != 0:
        cost = cost + weights[4] * (sizes[:, None, :] - gt_sizes[None, :, :3]).abs().sum(-1)
    return cost


def sparse_match(cost, query_masks, topk: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Queries may only be matched to objects they lie in; per object the `topk` cheapest such queries
    (strictly cheaper than the (topk+1)-th value), listed query-major (loss_3d.py:352-365)."""
    c = torch.where(query_masks.T, cost, torch.full_like(cost, INF_COST))
    kth = torch.topk(c, topk + 1, dim=0, largest=False, sorted=True).values[-1:, :]
    ids = torch.argwhere(c < kth)
    return ids[:, 0], ids[:, 1]


def hungarian_match(cost) -> Tuple[torch.Tensor, torch.Tensor]:
    from scipy.optimize import linear_sum_assignment                           # loss_3d.py:13, 311
    q, g = linear_sum_assignment(cost.detach().cpu().numpy())
    return torch.as_tensor(q, dtype=torch.long), torch.as_tensor(g, dtype=torch.long)


def _dice(x, t):                                                               # loss_3d.py:120-137
    s = x.sigmoid()
    return (1 - (2 * (s * t).sum(-1) + 1) / (s.sum(-1) + t.sum(-1) + 1)).mean()


def _iou(x, t):                                                                # loss_3d.py:100-117
    b = (x.sigmoid() >= 0.5).to(x.dtype)
    tt = (t > 0.5).to(x.dtype)
    inter = (b * tt).sum(-1)
    return inter / (tt.sum(-1) + b.sum(-1) - inter + 1e-6)


def instance_layer_loss(layer: Dict[str, list], insts: Sequence[dict], cfg: dict, last: bool,
                        indices: Optional[list] = None):
    """One decoder layer's instance loss over a batch (lists of per-scene tensors).  `insts[i]`: labels [G],
    sp_masks [G, S] bool, query_masks [G, Q] bool, optionally instance_centers / instance_sizes [G, 3].
    Returns (loss, indices, parts) with parts = [cls, bce, dice, score, centre, size]."""
    n_b = len(insts)
    w = cfg["loss_weight"]
    cw = list(cfg["cost_weights"]) + [0.0] * (5 - len(cfg["cost_weights"]))
    if indices is None:
        indices = []
        for i, g in enumerate(insts):
            if len(g["labels"]) == 0:
                e = g["labels"].new_empty((0,))
                indices.append((e, e))
                continue
            with torch.no_grad():
                cost = match_costs(layer["cls_preds"][i], layer["masks"][i], layer["centers"][i], layer["sizes"][i],
                                   g["labels"], g["sp_masks"], g.get("instance_centers"), g.get("instance_sizes"), cw)
                if cfg["matcher"] == "sparse":
                    indices.append(sparse_match(cost, g["query_masks"], cfg["topk"]))
                else:
                    indices.append(hungarian_match(cost))
    n_cls = cfg["num_classes"]
    class_weight = torch.tensor([1.0] * n_cls + [cfg["non_object_weight"]], dtype=layer["cls_preds"][0].dtype)
    cls_losses = []
    for i, g in enumerate(insts):                                              # loss_3d.py:459-467
        cp = layer["cls_preds"][i]
        target = torch.full((cp.shape[0],), n_cls, dtype=torch.long)
        iq, ig = indices[i]
        target[iq] = g["labels"][ig]                                           # duplicates: the last (largest object id) wins on CPU
        cls_losses.append(F.cross_entropy(cp, target, class_weight))
    cls_loss = torch.stack(cls_losses).mean()
    bce, dice, score, ctr, size = [], [], [], [], []
    for i, g in enumerate(insts):
        iq, ig = indices[i]
        pm = layer["masks"][i][iq]
        tm = g["sp_masks"][ig].to(pm.dtype)
        bce.append(F.binary_cross_entropy_with_logits(pm, tm))
        dice.append(_dice(pm, tm))
        if layer["centers"][i] is not None:
            ctr.append((layer["centers"][i][iq] - g["instance_centers"][ig, :3]).abs().sum(-1).mean())
        if layer["sizes"][i] is not None:
            size.append((layer["sizes"][i][iq] - g["instance_sizes"][ig, :3]).abs().sum(-1).mean())
        if layer["scores"][i] is None:
            continue
        with torch.no_grad():
            tgt = _iou(pm, tm).unsqueeze(1)
        keep = torch.where(tgt > 0.5)[0]
        if keep.numel():
            score.append(F.mse_loss(layer["scores"][i][iq][keep], tgt[keep]))
    zero = cls_loss.new_zeros(())
    score_loss = torch.stack(score).sum() / n_b if score else zero
    bce_loss = torch.stack(bce).sum() / n_b
    # the last layer sums the dice terms, auxiliary layers divide the sum by the batch size first (loss_3d.py:502 vs :657)
    dice_loss = torch.stack(dice).sum() if last else torch.stack(dice).sum() / n_b
    if cfg["fix_dice_loss_weight"]:
        dice_loss = dice_loss / n_b * 4
    if cfg["fix_mean_loss"]:
        bce_loss = bce_loss * n_b / len(bce)
        dice_loss = dice_loss * n_b / len(dice)
    ctr_loss = torch.stack(ctr).mean() if ctr else zero
    size_loss = torch.stack(size).mean() if size else zero
    parts = [cls_loss, bce_loss, dice_loss, score_loss, ctr_loss, size_loss]
    loss = sum(wi * p for wi, p in zip(w, parts))                              # 4, 5 or 6 weights
    return loss, indices, parts


def semantic_loss(sem_preds, sem_masks, ignore_index: int, loss_weight: float):
    """sem_preds[i] [Q, n + 1]; sem_masks[i] [n + 1, Q] bool (loss_3d.py:37-60)."""
    losses = []
    for p, m in zip(sem_preds, sem_masks):
        if ignore_index >= 0:
            p = p[:, :-1]
        losses.append(F.cross_entropy(p, m.to(p.dtype).argmax(0), ignore_index=ignore_index))
    return loss_weight * torch.stack(losses).mean()


def unified_criterion(pred: dict, targets: Sequence[dict], cfg: dict) -> Dict[str, torch.Tensor]:
    """pred: cls_preds / sem_preds / masks / scores / centers / sizes (lists over scenes) + aux_outputs (list of
    the same dicts); targets[i]: sp_inst_sem_masks [G + n + 1, S], query_inst_sem_masks [G + n + 1, Q], labels [G],
    optional instance_centers / instance_sizes (loss_3d.py:726-780)."""
    n = cfg["num_semantic_classes"]
    insts, sem_masks = [], []
    for t in targets:
        sem_masks.append(t["query_inst_sem_masks"][-n - 1:, :])
        g = dict(labels=t["labels"], sp_masks=t["sp_inst_sem_masks"][:-n - 1, :], query_masks=t["query_inst_sem_masks"][:-n - 1, :])
        for k in ("instance_centers", "instance_sizes"):
            if t.get(k) is not None:
                g[k] = t[k]
        insts.append(g)
    out = dict(seg_loss=semantic_loss(pred["sem_preds"], sem_masks, cfg["sem_ignore_index"], cfg["sem_loss_weight"]))
    loss, indices, parts = instance_layer_loss(pred, insts, cfg, last=True)
    per_layer = [parts]
    for aux in pred.get("aux_outputs", []):
        l, _, p = instance_layer_loss(aux, insts, cfg, last=False, indices=None if cfg["iter_matcher"] else indices)
        loss = loss + l
        per_layer.append(p)
    out["inst_loss"] = loss
    out["_parts"] = per_layer
    out["_indices"] = indices
    return out
