"""ScanNet instance-AP protocol with the per-scene association on the GPU.

Mirrors the interface of the reference's `evaluation/utils_instance_seg_3d_eval.py`
(`instance_seg_eval` :497-565, `scannet_eval` :380-408, `assign_instances_for_scan` :305-378,
`evaluate_matches` :18-209, `compute_averages` :212-302, `get_options` :411-430, `rename_gt` :465-494):
same argument meaning, same result dictionaries - but

  * the association of one scene (`assign_scene`) takes the `[n_pred, N]` masks and the `[N]` ground-truth
    ids as DEVICE tensors - they never leave the GPU that produced them - and counts all prediction x
    ground-truth intersections in one pass over the masks (`sd3d_mask_overlaps`); the reference spends
    `n_pred x n_gt` numpy passes over N on it (seconds per scene);
  * what leaves the GPU is a compact `SceneRecord` of fixed-width integer / float rows (per prediction:
    label, vert_count, void_intersection, confidence; per ground truth: label, instance id, vert_count; per
    intersecting pair: prediction, ground truth, intersection) - the payload of the per-rank all-gather in
    `dist_eval` - and `evaluate_records` computes AP from those rows on the host (greedy matching in the
    reference's order; there is nothing to parallelise in it and it is O(records));
  * `to_reference_dicts` rebuilds the reference's `gt2pred / pred2gt` dictionaries from a record, for callers
    that keep the reference's own `evaluate_matches`.

`get_instances` restates mmdet3d's `util_3d.get_instances` (not part of the reference repository): instances =
unique ids except 0, label = id // 1000, kept when the label is a valid class.
The product path has no CPU fallback: `assign_scene` raises on CPU tensors.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib, ops


def get_options(options: Optional[dict] = None) -> dict:
    o = dict(overlaps=np.append(np.arange(0.5, 0.95, 0.05), 0.25), min_region_sizes=np.array([100]),
             distance_threshes=np.array([float("inf")]), distance_confs=np.array([-float("inf")]))
    if options is not None:
        assert isinstance(options, dict)
        o.update(options)
    return o


def scannet200_groups() -> Dict[str, List[str]]:
    """head / common / tail category names of the ScanNet200 benchmark (the reference averages over them)."""
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "scannet200_groups.json")) as f:
        return json.load(f)


@dataclass
class SceneRecord:
    """Everything AP needs from one scene, as flat arrays (host numpy)."""
    pred_label: np.ndarray       # [P] i64  label id (already filtered: valid label, vert_count >= min region)
    pred_index: np.ndarray       # [P] i64  row of the prediction in the model output (the reference's file name suffix)
    pred_vert: np.ndarray        # [P] i64
    pred_void: np.ndarray        # [P] i64
    pred_conf: np.ndarray        # [P] f64
    gt_label: np.ndarray         # [G] i64
    gt_id: np.ndarray            # [G] i64  instance id (label * 1000 + index)
    gt_vert: np.ndarray          # [G] i64
    pair_pred: np.ndarray        # [M] i64  index into the P arrays
    pair_gt: np.ndarray          # [M] i64  index into the G arrays
    pair_inter: np.ndarray       # [M] i64  > 0

    def pack(self) -> np.ndarray:
        """One float64 row vector (length-prefixed sections) for the padded all-gather of dist_eval."""
        parts = [self.pred_label, self.pred_index, self.pred_vert, self.pred_void, self.pred_conf, self.gt_label, self.gt_id,
                 self.gt_vert, self.pair_pred, self.pair_gt, self.pair_inter]
        head = np.array([len(self.pred_label), len(self.gt_label), len(self.pair_pred)], dtype=np.float64)
        return np.concatenate([head] + [np.asarray(p, dtype=np.float64) for p in parts])

    @staticmethod
    def unpack(row: np.ndarray) -> "SceneRecord":
        P, G, M = (int(v) for v in row[:3])
        sizes = [P, P, P, P, P, G, G, G, M, M, M]
        out, o = [], 3
        for i, n in enumerate(sizes):
            seg = row[o:o + n]
            out.append(seg.astype(np.float64) if i == 4 else np.rint(seg).astype(np.int64))
            o += n
        return SceneRecord(*out)


def rename_gt(gt_semantic_masks, gt_instance_masks, valid_class_ids):
    """Instance ids -> 1000 * semantic id + instance index for valid classes (:465-494).  Accepts numpy arrays or
    torch tensors (any device); returns a list of the same kind."""
    out = []
    for sem, inst in zip(gt_semantic_masks, gt_instance_masks):
        if torch.is_tensor(inst):
            sem_t, inst_t = sem.long(), inst.long().clone()
            uniq, inv = torch.unique(inst_t, return_inverse=True)
            assert uniq.numel() < 1000
            # semantic id of every instance (must be unique per instance, as the reference asserts)
            first = torch.full((uniq.numel(),), -1, dtype=torch.long, device=inst_t.device)
            first.scatter_(0, inv, sem_t)                               # any member's semantic id
            assert bool((first[inv] == sem_t).all()), "an instance spans several semantic classes"
            valid = torch.isin(first, torch.as_tensor(list(valid_class_ids), device=inst_t.device))
            new_ids = torch.where(valid, 1000 * first + uniq, uniq)
            out.append(new_ids[inv])
        else:
            inst = np.array(inst, copy=True)
            for i in np.unique(inst):
                s = np.unique(sem[inst == i])
                assert len(s) == 1
                if s[0] in valid_class_ids:
                    inst[inst == i] = 1000 * s[0] + i
            out.append(inst)
    return out


def assign_scene(masks: torch.Tensor, labels: torch.Tensor, scores: torch.Tensor, gt_ids: torch.Tensor, options: dict,
                 valid_class_ids: Sequence[int]) -> SceneRecord:
    """`assign_instances_for_scan` (:305-378) for one scene on the device.

    masks [n, N] bool / uint8 (non-zero = member), labels [n] class INDEX (into valid_class_ids, as the model emits
    them; `aggregate_predictions` :450 maps them to ids), scores [n], gt_ids [N] int64 as produced by `rename_gt`."""
    if not (masks.is_cuda and gt_ids.is_cuda):
        raise RuntimeError("assign_scene needs device-resident masks and ground truth (no CPU fallback in the product path)")
    lib = _lib.load()
    dev = masks.device
    n, N = masks.shape
    if gt_ids.numel() != N:
        raise ValueError("len(pred_mask) != len(gt_ids)")
    m8 = masks.view(torch.uint8) if masks.dtype == torch.bool else masks
    if m8.dtype != torch.uint8 or m8.stride(1) != 1:
        raise TypeError("masks must be bool / uint8 with contiguous rows")
    valid = torch.as_tensor(list(valid_class_ids), dtype=torch.long, device=dev)
    uniq, inv, cnt = torch.unique(gt_ids.long(), return_inverse=True, return_counts=True)
    is_inst = torch.isin(uniq // 1000, valid) & (uniq != 0)              # get_instances: id 0 skipped, label must be valid
    n_gt = int(is_inst.sum())                                            # one small sync; n_gt sizes the histogram
    col = torch.where(is_inst, torch.cumsum(is_inst.long(), 0) - 1, torch.full_like(uniq, n_gt))
    # bool_void (:332) = label not valid; a point of id 0 with a valid label 0 would be neither: the reference then counts it
    # nowhere, so it gets no column
    void = ~torch.isin(uniq // 1000, valid)
    col = torch.where(is_inst | void, col, torch.full_like(uniq, -1))
    gt_index = col[inv].to(torch.int32).contiguous()
    n_cols = n_gt + 1
    counts = torch.empty(max(n, 1), n_cols, dtype=torch.int32, device=dev)
    if n > 0:
        _lib.check(lib.sd3d_mask_overlaps(m8.data_ptr(), m8.stride(0), n, gt_index.data_ptr(), N, n_cols, counts.data_ptr(),
                                          ops._stream()), "mask_overlaps")
    vert = m8.ne(0).sum(dim=1) if n > 0 else torch.zeros(0, dtype=torch.long, device=dev)
    # ---- everything below is O(n x n_gt) on the host
    counts_h = counts[:n].cpu().numpy().astype(np.int64)
    vert_h = vert.cpu().numpy().astype(np.int64)
    labels_h = labels.cpu().numpy().astype(np.int64)
    scores_h = scores.detach().cpu().numpy().astype(np.float64)
    gt_id = uniq[is_inst].cpu().numpy().astype(np.int64)
    gt_vert = cnt[is_inst].cpu().numpy().astype(np.int64)
    gt_label = gt_id // 1000
    valid_np = np.asarray(list(valid_class_ids), dtype=np.int64)
    label_id = valid_np[labels_h] if n > 0 else np.zeros(0, dtype=np.int64)
    keep = vert_h >= int(options["min_region_sizes"][0])                  # every label id is valid by construction
    idx = np.nonzero(keep)[0]
    inter = counts_h[idx][:, :n_gt] if n_gt > 0 else np.zeros((len(idx), 0), dtype=np.int64)
    same = (label_id[idx][:, None] == gt_label[None, :]) & (inter > 0)    # only ground truth of the prediction's label (:352)
    pp, gg = np.nonzero(same)
    return SceneRecord(pred_label=label_id[idx], pred_index=idx.astype(np.int64), pred_vert=vert_h[idx],
                       pred_void=counts_h[idx][:, n_gt] if len(idx) else np.zeros(0, dtype=np.int64), pred_conf=scores_h[idx],
                       gt_label=gt_label, gt_id=gt_id, gt_vert=gt_vert, pair_pred=pp.astype(np.int64), pair_gt=gg.astype(np.int64),
                       pair_inter=inter[pp, gg].astype(np.int64))


def to_reference_dicts(rec: SceneRecord, scene_id, class_labels, id_to_label):
    """The `(gt2pred, pred2gt)` dictionaries `assign_instances_for_scan` returns, rebuilt from a record."""
    gt2pred = {label: [] for label in class_labels}
    pred2gt = {label: [] for label in class_labels}
    gts, preds = [], []
    for g in range(len(rec.gt_id)):
        d = dict(instance_id=int(rec.gt_id[g]), label_id=int(rec.gt_label[g]), vert_count=int(rec.gt_vert[g]), med_dist=-1,
                 dist_conf=0.0, matched_pred=[])
        gts.append(d)
        gt2pred[id_to_label[int(rec.gt_label[g])]].append(d)
    for p in range(len(rec.pred_label)):
        d = dict(filename=f"{scene_id}_{int(rec.pred_index[p])}", pred_id=p, label_id=int(rec.pred_label[p]),
                 vert_count=int(rec.pred_vert[p]), confidence=rec.pred_conf[p], void_intersection=int(rec.pred_void[p]))
        d["matched_gt"] = []
        preds.append(d)
        pred2gt[id_to_label[int(rec.pred_label[p])]].append(d)
    for p, g, i in zip(rec.pair_pred, rec.pair_gt, rec.pair_inter):      # row-major: predictions in order, their gts in order
        gc = {k: v for k, v in gts[g].items() if k != "matched_pred"}
        gc["intersection"] = int(i)
        pc = {k: v for k, v in preds[p].items() if k != "matched_gt"}
        pc["intersection"] = int(i)
        preds[p]["matched_gt"].append(gc)
        gts[g]["matched_pred"].append(pc)
    return gt2pred, pred2gt


def evaluate_records(records: Sequence[SceneRecord], class_labels, valid_class_ids, options: dict):
    """`evaluate_matches` (:18-209) on compact records: returns (ap [1, C, O], pr_rc [2, C, O])."""
    overlaps = options["overlaps"]
    min_region = options["min_region_sizes"][0]
    C, O = len(class_labels), len(overlaps)
    ap = np.zeros((1, C, O), float)
    pr_rc = np.zeros((2, C, O), float)
    label_ids = list(valid_class_ids)
    # per scene, per label: index lists (built once)
    per = []
    for rec in records:
        order_by_gt = np.lexsort((rec.pair_pred, rec.pair_gt))           # a gt's matched predictions in prediction order
        per.append(dict(rec=rec, gt_pairs=order_by_gt,
                        gt_start=np.searchsorted(rec.pair_gt[order_by_gt], np.arange(len(rec.gt_id) + 1)),
                        pred_start=np.searchsorted(rec.pair_pred, np.arange(len(rec.pred_label) + 1))))
    for oi, th in enumerate(overlaps):
        visited = [np.zeros(len(s["rec"].pred_label), dtype=bool) for s in per]
        for li in range(C):
            lid = label_ids[li]
            y_true: List[float] = []
            y_score: List[float] = []
            hard_fn, has_gt, has_pred = 0, False, False
            for si, s in enumerate(per):
                rec = s["rec"]
                gts = [g for g in np.nonzero(rec.gt_label == lid)[0] if rec.gt_vert[g] >= min_region]   # med_dist / dist_conf never filter
                preds = np.nonzero(rec.pred_label == lid)[0]
                has_gt |= len(gts) > 0
                has_pred |= len(preds) > 0
                matched_true: List[float] = []
                matched_score: List[float] = []
                extra_true: List[float] = []
                extra_score: List[float] = []
                for g in gts:
                    cur_match, cur_score, found = False, -float("inf"), False
                    for q in s["gt_pairs"][s["gt_start"][g]:s["gt_start"][g + 1]]:
                        p = rec.pair_pred[q]
                        if visited[si][p]:
                            continue
                        inter = rec.pair_inter[q]
                        if float(inter) / (rec.gt_vert[g] + rec.pred_vert[p] - inter) > th:
                            conf = rec.pred_conf[p]
                            if cur_match:                                  # a second prediction on the same gt: the weaker one is a FP
                                hi, lo = max(cur_score, conf), min(cur_score, conf)
                                cur_score = hi
                                extra_true.append(0.0)
                                extra_score.append(lo)
                            else:
                                found, cur_match, cur_score = True, True, conf
                                visited[si][p] = True
                    if not found:
                        hard_fn += 1
                    if cur_match:
                        matched_true.append(1.0)
                        matched_score.append(cur_score)
                cur_true = matched_true + extra_true
                cur_score_l = matched_score + extra_score
                for p in preds:                                            # unmatched predictions: false positives unless mostly void
                    qs = range(s["pred_start"][p], s["pred_start"][p + 1])
                    found_gt = False
                    for q in qs:
                        g = rec.pair_gt[q]
                        inter = rec.pair_inter[q]
                        if float(inter) / (rec.gt_vert[g] + rec.pred_vert[p] - inter) > th:
                            found_gt = True
                            break
                    if not found_gt:
                        ignore = int(rec.pred_void[p])
                        for q in qs:
                            g = rec.pair_gt[q]
                            if rec.gt_id[g] < 1000:
                                ignore += int(rec.pair_inter[q])
                            if rec.gt_vert[g] < min_region:
                                ignore += int(rec.pair_inter[q])
                        if float(ignore) / rec.pred_vert[p] <= th:
                            cur_true.append(0.0)
                            cur_score_l.append(rec.pred_conf[p])
                y_true += cur_true
                y_score += cur_score_l
            if has_gt and has_pred:
                yt, ys = np.asarray(y_true, dtype=float), np.asarray(y_score, dtype=float)
                order = np.argsort(ys)
                ys, yt = ys[order], yt[order]
                cum = np.cumsum(yt)
                _, uniq = np.unique(ys, return_index=True)
                n_pr, n_ex = len(uniq) + 1, len(ys)
                n_true = cum[-1] if len(cum) > 0 else 0
                prec, rec_ = np.zeros(n_pr), np.zeros(n_pr)
                cum = np.append(cum, 0)
                for ir, isc in enumerate(uniq):
                    c = cum[isc - 1]
                    tp = n_true - c
                    fp = n_ex - isc - tp
                    fn = c + hard_fn
                    prec[ir] = float(tp) / (tp + fp)
                    rec_[ir] = float(tp) / (tp + fn)
                prec[-1], rec_[-1] = 1.0, 0.0
                f1 = 2 * prec * rec_ / (prec + rec_ + 0.0001)
                best = f1.argmax()
                best_pr, best_rc = prec[best], rec_[best]
                rconv = np.append(np.append(rec_[0], rec_), 0.0)
                ap_cur = np.dot(prec, np.convolve(rconv, [-0.5, 0, 0.5], "valid"))
            elif has_gt:
                ap_cur, best_pr, best_rc = 0.0, 0, 0
            else:
                ap_cur = best_pr = best_rc = float("nan")
            ap[0, li, oi] = ap_cur
            pr_rc[0, li, oi], pr_rc[1, li, oi] = best_pr, best_rc
    return ap, pr_rc


def compute_averages(aps, pr_rc, options, class_labels, groups: Optional[Dict[str, List[str]]] = None):
    """`compute_averages` (:212-302); groups default to the ScanNet200 head / common / tail lists."""
    if groups is None:
        groups = scannet200_groups()
    o50 = np.where(np.isclose(options["overlaps"], 0.5))
    o25 = np.where(np.isclose(options["overlaps"], 0.25))
    oall = np.where(np.logical_not(np.isclose(options["overlaps"], 0.25)))
    with np.errstate(invalid="ignore"), _quiet():
        d = {"all_ap": np.nanmean(aps[0, :, oall]), "all_ap_50%": np.nanmean(aps[0, :, o50]), "all_ap_25%": np.nanmean(aps[0, :, o25]),
             "all_prec_50%": np.nanmean(pr_rc[0, :, o50]), "all_rec_50%": np.nanmean(pr_rc[1, :, o50]), "classes": {}}
        for li, label in enumerate(class_labels):
            d["classes"][label] = {"ap": np.average(aps[0, li, oall]), "ap50%": np.average(aps[0, li, o50]),
                                   "ap25%": np.average(aps[0, li, o25]), "prec50%": np.average(pr_rc[0, li, o50]),
                                   "rec50%": np.average(pr_rc[1, li, o50])}
        for gname, cats in groups.items():
            idx = [i for i, c in enumerate(class_labels) if c in cats]
            d[f"{gname}_ap"] = np.nanmean(aps[0][np.ix_(idx, oall[0])])
            d[f"{gname}_ap_50%"] = np.nanmean(aps[0][np.ix_(idx, o50[0])])
            d[f"{gname}_ap_25%"] = np.nanmean(aps[0][np.ix_(idx, o25[0])])
            d[f"{gname}_prec_50%"] = np.nanmean(pr_rc[0][np.ix_(idx, o50[0])])
            d[f"{gname}_rec_50%"] = np.nanmean(pr_rc[1][np.ix_(idx, o50[0])])
    return d


class _quiet:
    def __enter__(self):
        import warnings
        self._cm = warnings.catch_warnings()
        self._cm.__enter__()
        warnings.simplefilter("ignore", category=RuntimeWarning)         # nanmean of an empty / all-nan group, as in the reference

    def __exit__(self, *a):
        return self._cm.__exit__(*a)


def instance_seg_eval(gt_semantic_masks, gt_instance_masks, pred_instance_masks, pred_instance_labels, pred_instance_scores,
                      valid_class_ids, class_labels, options=None, logger=None, print_log_flag=False, groups=None):
    """`instance_seg_eval` (:497-565) with device tensors per scene: returns the same metrics dictionary."""
    assert len(valid_class_ids) == len(class_labels)
    opts = get_options(options)
    gts = rename_gt(gt_semantic_masks, gt_instance_masks, valid_class_ids)
    records = [assign_scene(m, l, s, g, opts, valid_class_ids)
               for m, l, s, g in zip(pred_instance_masks, pred_instance_labels, pred_instance_scores, gts)]
    ap, pr_rc = evaluate_records(records, class_labels, valid_class_ids, opts)
    return compute_averages(ap, pr_rc, opts, class_labels, groups)


def map_inst_markup(pts_semantic_mask: torch.Tensor, pts_instance_mask: torch.Tensor, valid_class_ids, num_stuff_cls: int):
    """`InstanceSeg3DEvaluator.map_inst_markup` (evaluation/evaluator_3d.py:323-349) on tensors (any device, not in place):
    instance ids shifted down by the stuff classes (stuff -> -1), semantic ids of the remaining points mapped to the dataset's
    class ids through `valid_class_ids + [-1]` (negative indices wrap as in numpy: index -1 is the appended -1)."""
    inst = pts_instance_mask.long() - int(num_stuff_cls)
    inst = torch.where(inst < 0, torch.full_like(inst, -1), inst)
    sem = pts_semantic_mask.long() - int(num_stuff_cls)
    sem = torch.where(inst == -1, torch.full_like(sem, -1), sem)
    mapping = torch.tensor(list(valid_class_ids) + [-1], dtype=torch.long, device=sem.device)
    return mapping[sem], inst


def evaluator_instance_metrics(results, classes, valid_class_ids, num_stuff_cls: int, options=None, groups=None):
    """The ScanNet branch of `InstanceSeg3DEvaluator.compute_metrics` (evaluator_3d.py:124-219) with device tensors: per scene
    `(eval_ann, pred)` as the reference's evaluator collects them - `eval_ann` = dict(pts_semantic_mask, pts_instance_mask)
    (panoptic-style), `pred` = this package's `PointData` fields (`pts_instance_mask[0]` [n, N] bool, `instance_labels`,
    `instance_scores`) - through `map_inst_markup` into `instance_seg_eval(valid_class_ids[num_stuff:], classes[num_stuff:-1])`.
    Returns the same metrics dictionary (the reference computes it and, with its last lines commented out, drops it)."""
    things = tuple(int(v) for v in valid_class_ids[num_stuff_cls:])
    labels = tuple(classes[num_stuff_cls:-1])
    as_t = lambda a, dev: a if torch.is_tensor(a) else torch.as_tensor(np.asarray(a), device=dev)     # noqa: E731
    sems, insts, masks, labs, scores = [], [], [], [], []
    for ann, pred in results:
        if not isinstance(pred, dict):                                 # a PointData straight from the model (SegMetric.process makes it a dict)
            pred = dict(pred.items())
        m = pred["pts_instance_mask"][0]
        m = m if torch.is_tensor(m) else torch.as_tensor(np.asarray(m))
        dev = m.device
        s, i = map_inst_markup(as_t(ann["pts_semantic_mask"], dev), as_t(ann["pts_instance_mask"], dev), things, num_stuff_cls)
        sems.append(s); insts.append(i); masks.append(m)
        labs.append(as_t(pred["instance_labels"], dev)); scores.append(as_t(pred["instance_scores"], dev))
    return instance_seg_eval(sems, insts, masks, labs, scores, valid_class_ids=things, class_labels=labels, options=options, groups=groups)


def eval_ann_info(target, bg_class_id: int) -> dict:
    """The per-scene ground-truth record of the reference's evaluation loop (evaluation/evaluate_3d.py:49-63) from a
    `GD3DTarget` as the model returns it - tensors stay on the target's device (the reference moves them to numpy for the
    CPU protocol; `evaluator_instance_metrics` takes either).  Instance / semantic ids are the SUM over the instance masks that
    cover a point, uncovered points -1 / `bg_class_id`, exactly as the loop computes them."""
    m = target["masks"].squeeze(-1).long()                                     # [n, N]
    n = m.shape[0]
    covered = m.sum(dim=0) != 0
    inst = (m * torch.arange(n, device=m.device)[:, None]).sum(dim=0)
    inst = torch.where(covered, inst, torch.full_like(inst, -1))
    sem = (m * target["labels"].long()[:, None]).sum(dim=0)
    sem = torch.where(covered, sem, torch.full_like(sem, int(bg_class_id)))
    return dict(pts_instance_mask=inst, pts_semantic_mask=sem, sp_pts_mask=target["extra_features"]["super_point_masks"],
                lidar_idx=target["scene_id"])
