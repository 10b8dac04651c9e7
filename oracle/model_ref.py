"""ORACLE (test infrastructure, not product code): whole eval-mode `Baseline3D.forward` on the CPU.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.
Chains the restated pieces in the order of `segdino3d/models/architecture/baseline3d.py:308-346`:
scene range (:266-306) -> backbone.forward_wrapper -> _select_queries (:207-264) -> decoder ->
predict_by_feat (:373-404).  The decoder / post-processing parts are pinned to the reference's golden
vectors; the backbone part is "parity unpinned" (see oracle/sparse_ref.py).
"""
from __future__ import annotations

import torch

from . import decoder_ref as D
from . import postprocess_ref as P
from . import sparse_ref as R


def forward_eval(sd, points, feats2d, superpoints, q2d_feat, q2d_pos, gt_masks=None, backbone="mink", num_classes=198,
                 dec_cfg: D.DecoderCfg = None, test_cfg: P.TestCfg = None, query_num=-1, box_filter=True,
                 mode_3d_center="median", voxel_size=0.02, return_intermediate=False):
    dec_cfg = dec_cfg or D.DecoderCfg()
    test_cfg = test_cfg or P.TestCfg()
    with torch.no_grad():
        lo, hi, centers, sizes = P.scene_range_and_gt_boxes(points[:, :3], gt_masks, mode_3d_center)
        if backbone == "mink":
            sp_feat, sp_pos, sp_pos_wo = R.mink_forward_wrapper(sd, points, feats2d, superpoints, voxel_size)
        elif backbone == "spconv":
            sp_feat, sp_pos, sp_pos_wo = R.spconv_forward_wrapper(sd, points, feats2d, superpoints, voxel_size)
        else:
            raise ValueError(backbone)
        q, qpos, ids = D.select_queries(sd, sp_feat, sp_pos, query_num)
        out = D.decoder_forward(sd, dec_cfg, sp_feat, sp_pos, sp_pos_wo, q, qpos, q2d_feat, q2d_pos, lo, hi)
        res = P.predict_by_feat(out, superpoints, points[:, :3], num_classes, test_cfg, box_filter, query_num)
    if return_intermediate:
        return res, dict(sp_feat=sp_feat, sp_pos=sp_pos, decoder=out, query_ids=ids, lo=lo, hi=hi,
                         instance_centers=centers, instance_sizes=sizes)
    return res


def forward_train(sd, points, feats2d, superpoints, q2d_feat, q2d_pos, gt_masks, labels, sp_inst_sem_masks, query_ids, loss_cfg,
                  dec_cfg: D.DecoderCfg = None, mode_3d_center="median", voxel_size=0.02):
    """Training-mode `Baseline3D.forward` (baseline3d.py:308-346) for ONE scene, differentiable: backbone with batch-statistics
    BatchNorm, the given query subset (`_select_queries` :250-264 draws it at random), decoder with auxiliary outputs, criterion.
    Test infrastructure: torch autograd of this function is the reference for the device training step."""
    from . import loss_ref as L
    dec_cfg = dec_cfg or D.DecoderCfg()
    lo, hi, centers, sizes = P.scene_range_and_gt_boxes(points[:, :3], gt_masks, mode_3d_center)
    R.BN_TRAIN = True
    try:
        sp_feat, sp_pos, sp_pos_wo = R.mink_forward_wrapper(sd, points, feats2d, superpoints, voxel_size)
    finally:
        R.BN_TRAIN = False
    sp_pos, sp_pos_wo = sp_pos.to(points.dtype), sp_pos_wo.to(points.dtype)
    q, qpos = sp_feat[query_ids], sp_pos[query_ids]
    out = D.decoder_forward(sd, dec_cfg, sp_feat, sp_pos, sp_pos_wo, q, qpos, q2d_feat, q2d_pos, lo, hi)
    layer = lambda o: dict(cls_preds=[o["cls_preds"]], sem_preds=[o.get("sem_preds")], masks=[o["masks"]], scores=[None],
                           centers=[o["centers"]], sizes=[o["sizes"]])
    pred = layer(out)
    pred["aux_outputs"] = [layer(a) for a in out["aux"]]
    target = dict(sp_inst_sem_masks=sp_inst_sem_masks, query_inst_sem_masks=sp_inst_sem_masks[:, query_ids], labels=labels,
                  instance_centers=centers, instance_sizes=sizes)
    return L.unified_criterion(pred, [target], loss_cfg), out
