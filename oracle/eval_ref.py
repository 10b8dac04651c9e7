"""TEST INFRASTRUCTURE - CPU restatement (numpy) of the ScanNet instance-AP protocol the reference evaluates
with.  Only tests/, __graft_entry__.smoke() and tools that time a CPU baseline may import it.

Follows /root/reference/evaluation/utils_instance_seg_3d_eval.py:
  assign_instances  <- assign_instances_for_scan :305-378
  evaluate_matches  <- :18-209
  compute_averages  <- :212-302 (the overall / per-class part; the ScanNet200 head / common / tail groups are
                       computed from the category lists the caller passes)
  get_options, rename_gt, aggregate_predictions, scannet_eval <- :411-494, :380-408

PINNING: `tests/golden/ap_protocol.npz` holds outputs of the reference's own scannet_eval, produced by importing
that file in the build container (tests/golden/make_golden_ap.py).  One callee is NOT in the reference
repository: `mmdet3d.evaluation.functional.scannet_utils.util_3d.get_instances` (mmdet3d 1.4, itself a copy
of ScanNet's BenchmarkScripts/util_3d.py).  `get_instances` below restates its published behaviour
(unique ids except 0, label = id // 1000, kept when the label is a valid class; fields instance_id, label_id,
vert_count, med_dist = -1, dist_conf = 0.0) and the golden vectors were generated with this restatement
standing in for it: PARITY UNPINNED for that one function.
"""
from copy import deepcopy

import numpy as np


def get_instances(ids, class_ids, class_labels, id2label):
    instances = {label: [] for label in class_labels}
    for i in np.unique(ids):
        if i == 0:
            continue
        label_id = int(i // 1000)
        if label_id in class_ids:
            instances[id2label[label_id]].append(dict(instance_id=int(i), label_id=label_id,
                                                      vert_count=int(np.count_nonzero(ids == i)), med_dist=-1, dist_conf=0.0))
    return instances


def get_options(options=None):
    o = dict(overlaps=np.append(np.arange(0.5, 0.95, 0.05), 0.25), min_region_sizes=np.array([100]),
             distance_threshes=np.array([float("inf")]), distance_confs=np.array([-float("inf")]))
    if options is not None:
        o.update(options)
    return o


def assign_instances(pred_info, gt_ids, options, valid_class_ids, class_labels, id_to_label):
    """pred_info: {name: {mask [N], label_id, conf}} in insertion order."""
    gt2pred = deepcopy(get_instances(gt_ids, valid_class_ids, class_labels, id_to_label))
    for label in gt2pred:
        for gt in gt2pred[label]:
            gt["matched_pred"] = []
    pred2gt = {label: [] for label in class_labels}
    n_pred = 0
    bool_void = np.logical_not(np.isin(gt_ids // 1000, valid_class_ids))
    for name, info in pred_info.items():
        label_id = int(info["label_id"])
        if label_id not in id_to_label:
            continue
        label_name = id_to_label[label_id]
        mask = np.not_equal(info["mask"], 0)
        if len(mask) != len(gt_ids):
            raise ValueError("len(pred_mask) != len(gt_ids)")
        num = int(np.count_nonzero(mask))
        if num < options["min_region_sizes"][0]:
            continue
        pred = dict(filename=name, pred_id=n_pred, label_id=label_id, vert_count=num, confidence=info["conf"],
                    void_intersection=int(np.count_nonzero(np.logical_and(bool_void, mask))))
        matched_gt = []
        for gi, gt in enumerate(gt2pred[label_name]):
            inter = int(np.count_nonzero(np.logical_and(gt_ids == gt["instance_id"], mask)))
            if inter > 0:
                g, q = gt.copy(), pred.copy()
                g["intersection"] = inter
                q["intersection"] = inter
                matched_gt.append(g)
                gt2pred[label_name][gi]["matched_pred"].append(q)
        pred["matched_gt"] = matched_gt
        n_pred += 1
        pred2gt[label_name].append(pred)
    return gt2pred, pred2gt


def evaluate_matches(matches, class_labels, options):
    overlaps = options["overlaps"]
    min_region_size = options["min_region_sizes"][0]
    distance_thresh = options["distance_threshes"][0]
    distance_conf = options["distance_confs"][0]
    ap = np.zeros((1, len(class_labels), len(overlaps)), float)
    pr_rc = np.zeros((2, len(class_labels), len(overlaps)), float)
    for oi, th in enumerate(overlaps):
        visited = {}
        for m in matches:
            for label in class_labels:
                for p in matches[m]["pred"][label]:
                    if "filename" in p:
                        visited[p["filename"]] = False
        for li, label in enumerate(class_labels):
            y_true, y_score = np.empty(0), np.empty(0)
            hard_fn, has_gt, has_pred = 0, False, False
            for m in matches:
                preds = matches[m]["pred"][label]
                gts = [g for g in matches[m]["gt"][label]
                       if g["vert_count"] >= min_region_size and g["med_dist"] <= distance_thresh and g["dist_conf"] >= distance_conf]
                has_gt |= bool(gts)
                has_pred |= bool(preds)
                cur_true = np.ones(len(gts))
                cur_score = np.ones(len(gts)) * (-float("inf"))
                cur_match = np.zeros(len(gts), dtype=bool)
                for gi, gt in enumerate(gts):
                    found = False
                    for pred in gt["matched_pred"]:
                        if visited[pred["filename"]]:
                            continue
                        ov = float(pred["intersection"]) / (gt["vert_count"] + pred["vert_count"] - pred["intersection"])
                        if ov > th:
                            conf = pred["confidence"]
                            if cur_match[gi]:
                                hi, lo = max(cur_score[gi], conf), min(cur_score[gi], conf)
                                cur_score[gi] = hi
                                cur_true = np.append(cur_true, 0)
                                cur_score = np.append(cur_score, lo)
                                cur_match = np.append(cur_match, True)
                            else:
                                found = True
                                cur_match[gi] = True
                                cur_score[gi] = conf
                                visited[pred["filename"]] = True
                    if not found:
                        hard_fn += 1
                cur_true, cur_score = cur_true[cur_match], cur_score[cur_match]
                for pred in preds:
                    found_gt = False
                    for gt in pred["matched_gt"]:
                        ov = float(gt["intersection"]) / (gt["vert_count"] + pred["vert_count"] - gt["intersection"])
                        if ov > th:
                            found_gt = True
                            break
                    if not found_gt:
                        ignore = pred["void_intersection"]
                        for gt in pred["matched_gt"]:
                            if gt["instance_id"] < 1000:
                                ignore += gt["intersection"]
                            if gt["vert_count"] < min_region_size or gt["med_dist"] > distance_thresh or gt["dist_conf"] < distance_conf:
                                ignore += gt["intersection"]
                        if float(ignore) / pred["vert_count"] <= th:
                            cur_true = np.append(cur_true, 0)
                            cur_score = np.append(cur_score, pred["confidence"])
                y_true = np.append(y_true, cur_true)
                y_score = np.append(y_score, cur_score)
            if has_gt and has_pred:
                order = np.argsort(y_score)
                ys, yt = y_score[order], y_true[order]
                cum = np.cumsum(yt)
                _, uniq = np.unique(ys, return_index=True)
                n_pr = len(uniq) + 1
                n_ex = len(ys)
                n_true = cum[-1] if len(cum) > 0 else 0
                prec, rec = np.zeros(n_pr), np.zeros(n_pr)
                cum = np.append(cum, 0)
                for ir, isc in enumerate(uniq):
                    c = cum[isc - 1]
                    tp = n_true - c
                    fp = n_ex - isc - tp
                    fn = c + hard_fn
                    prec[ir] = float(tp) / (tp + fp)
                    rec[ir] = float(tp) / (tp + fn)
                prec[-1], rec[-1] = 1.0, 0.0
                f1 = 2 * prec * rec / (prec + rec + 0.0001)
                best = f1.argmax()
                best_pr, best_rc = prec[best], rec[best]
                rconv = np.append(np.append(rec[0], rec), 0.0)
                ap_cur = np.dot(prec, np.convolve(rconv, [-0.5, 0, 0.5], "valid"))
            elif has_gt:
                ap_cur, best_pr, best_rc = 0.0, 0, 0
            else:
                ap_cur = best_pr = best_rc = float("nan")
            ap[0, li, oi] = ap_cur
            pr_rc[0, li, oi], pr_rc[1, li, oi] = best_pr, best_rc
    return ap, pr_rc


def compute_averages(aps, pr_rc, options, class_labels, groups=None):
    """groups: optional {name: [class names]} (ScanNet200 head / common / tail)."""
    o50 = np.where(np.isclose(options["overlaps"], 0.5))
    o25 = np.where(np.isclose(options["overlaps"], 0.25))
    oall = np.where(np.logical_not(np.isclose(options["overlaps"], 0.25)))
    d = {"all_ap": np.nanmean(aps[0, :, oall]), "all_ap_50%": np.nanmean(aps[0, :, o50]), "all_ap_25%": np.nanmean(aps[0, :, o25]),
         "all_prec_50%": np.nanmean(pr_rc[0, :, o50]), "all_rec_50%": np.nanmean(pr_rc[1, :, o50]), "classes": {}}
    for li, label in enumerate(class_labels):
        d["classes"][label] = {"ap": np.average(aps[0, li, oall]), "ap50%": np.average(aps[0, li, o50]),
                               "ap25%": np.average(aps[0, li, o25]), "prec50%": np.average(pr_rc[0, li, o50]),
                               "rec50%": np.average(pr_rc[1, li, o50])}
    for gname, cats in (groups or {}).items():
        idx = [i for i, c in enumerate(class_labels) if c in cats]
        d[f"{gname}_ap"] = np.nanmean(aps[0][np.ix_(idx, oall[0])])
        d[f"{gname}_ap_50%"] = np.nanmean(aps[0][np.ix_(idx, o50[0])])
        d[f"{gname}_ap_25%"] = np.nanmean(aps[0][np.ix_(idx, o25[0])])
        d[f"{gname}_prec_50%"] = np.nanmean(pr_rc[0][np.ix_(idx, o50[0])])
        d[f"{gname}_rec_50%"] = np.nanmean(pr_rc[1][np.ix_(idx, o50[0])])
    return d


def aggregate_predictions(masks, labels, scores, valid_class_ids):
    infos = []
    for sid, (mask, label, score) in enumerate(zip(masks, labels, scores)):
        info = {}
        for i in range(mask.shape[0]):
            info[f"{sid}_{i}"] = dict(mask=mask[i], label_id=valid_class_ids[label[i]], conf=score[i])
        infos.append(info)
    return infos


def rename_gt(gt_semantic_masks, gt_instance_masks, valid_class_ids):
    out = []
    for sem, inst in zip(gt_semantic_masks, gt_instance_masks):
        inst = inst.copy()
        uniq = np.unique(inst)
        assert len(uniq) < 1000
        for i in uniq:
            s = np.unique(sem[inst == i])
            assert len(s) == 1
            if s[0] in valid_class_ids:
                inst[inst == i] = 1000 * s[0] + i
        out.append(inst)
    return out


def scannet_eval(preds, gts, options, valid_class_ids, class_labels, id_to_label, groups=None):
    options = get_options(options)
    matches = {}
    for i, (pred, gt) in enumerate(zip(preds, gts)):
        g2p, p2g = assign_instances(pred, gt, options, valid_class_ids, class_labels, id_to_label)
        matches[i] = {"gt": g2p, "pred": p2g}
    ap, pr_rc = evaluate_matches(matches, class_labels, options)
    return compute_averages(ap, pr_rc, options, class_labels, groups), ap, pr_rc


def map_inst_markup(pts_semantic_mask, pts_instance_mask, valid_class_ids, num_stuff_cls):
    """`InstanceSeg3DEvaluator.map_inst_markup` (evaluation/evaluator_3d.py:323-349): panoptic-style annotation -> the instance
    task's ground truth.  valid_class_ids = the THING ids (the caller passes `valid_class_ids[num_stuff_cls:]`, :171);
    numpy's negative-index wrap of `mapping[...]` is part of the behaviour (index -1 = the appended -1)."""
    inst = np.array(pts_instance_mask, copy=True)
    sem = np.array(pts_semantic_mask, copy=True)
    inst -= num_stuff_cls
    inst[inst < 0] = -1
    sem -= num_stuff_cls
    sem[inst == -1] = -1
    mapping = np.array(list(valid_class_ids) + [-1])
    return mapping[sem], inst


def evaluator_instance_metrics(results, classes, valid_class_ids, num_stuff_cls, options=None, groups=None):
    """The ScanNet branch of `InstanceSeg3DEvaluator.compute_metrics` (evaluator_3d.py:124-219): per scene
    (eval_ann, pred) -> map_inst_markup -> instance_seg_eval(valid_class_ids[num_stuff:], classes[num_stuff:-1]).
    results: [(dict(pts_semantic_mask, pts_instance_mask), dict(pts_instance_mask=[masks, ...], instance_labels, instance_scores))]."""
    things = tuple(valid_class_ids[num_stuff_cls:])
    labels = tuple(classes[num_stuff_cls:-1])
    sems, insts, masks, labs, scores = [], [], [], [], []
    for ann, pred in results:
        s, i = map_inst_markup(ann["pts_semantic_mask"], ann["pts_instance_mask"], things, num_stuff_cls)
        sems.append(s); insts.append(i)
        masks.append(np.asarray(pred["pts_instance_mask"][0])); labs.append(np.asarray(pred["instance_labels"]))
        scores.append(np.asarray(pred["instance_scores"]))
    id_to_label = {things[i]: labels[i] for i in range(len(things))}
    preds = aggregate_predictions(masks, labs, scores, things)
    gts = rename_gt(sems, insts, things)
    metrics, _, _ = scannet_eval(preds, gts, options, things, labels, id_to_label, groups)
    return metrics, sems, insts


def eval_ann_from_target(masks, labels, bg_class_id):
    """The ground-truth record the evaluation loop builds per scene (evaluation/evaluate_3d.py:49-57): masks [n, N] bool,
    labels [n] -> (pts_instance_mask [N], pts_semantic_mask [N]).  Ids are SUMMED over the instances that cover a point
    (overlapping masks add up, as in the reference); uncovered points get -1 / bg_class_id."""
    masks = np.asarray(masks).astype(np.int64)
    n = masks.shape[0]
    covered = masks.sum(axis=0) != 0
    inst = (masks * np.arange(n, dtype=np.int64)[:, None]).sum(axis=0)
    inst[~covered] = -1
    sem = (masks * np.asarray(labels, dtype=np.int64)[:, None]).sum(axis=0)
    sem[~covered] = bg_class_id
    return inst, sem
