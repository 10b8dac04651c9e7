"""Generates tests/golden/evaluator_scannet.npz by running the REFERENCE's evaluator-level metric code
(/root/reference/evaluation/evaluator_3d.py: InstanceSeg3DEvaluator.compute_metrics, ScanNet branch :160-219, and
map_inst_markup :323-349) on seeded synthetic per-scene results: panoptic-style ground truth (two stuff classes + things)
-> map_inst_markup -> instance_seg_eval with valid_class_ids[num_stuff_cls:] / classes[num_stuff_cls:-1].
Runs in the build container only.  Third-party imports are stubbed like in make_golden_ap.py (mmengine, terminaltables,
mmdet3d: SegMetric is a plain base class here, get_instances the restated helper of oracle/eval_ref.py)."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import eval_ref  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Table:
    def __init__(self, rows):
        self.table = "\n".join(" | ".join(map(str, r)) for r in rows)
        self.inner_footing_row_border = False


class _Logger:
    @classmethod
    def get_current_instance(cls):
        return cls()


class _SegMetric:
    def __init__(self, **kw):
        pass


_stub("mmengine")
_stub("mmengine.logging", print_log=lambda *a, **k: None, MMLogger=_Logger)
_stub("terminaltables", AsciiTable=_Table)
_stub("mmdet3d")
_stub("mmdet3d.evaluation", InstanceSegMetric=object, panoptic_seg_eval=None, seg_eval=None)
_stub("mmdet3d.evaluation.metrics", SegMetric=_SegMetric)
_stub("mmdet3d.registry", METRICS=None)
_stub("mmdet3d.evaluation.functional")
util = _stub("mmdet3d.evaluation.functional.scannet_utils.util_3d", get_instances=eval_ref.get_instances)
_stub("mmdet3d.evaluation.functional.scannet_utils", util_3d=util)
# `evaluation/__init__.py` pulls in the whole evaluation driver (datasets, mmdet3d ...): load the two files it needs as a package by hand
pkg = types.ModuleType("evaluation")
pkg.__path__ = ["/root/reference/evaluation"]
sys.modules["evaluation"] = pkg
import importlib  # noqa: E402
ev = importlib.import_module("evaluation.evaluator_3d")

# ScanNet-like meta: classes = 2 stuff + 8 things + "unlabeled"; valid ids of all 10 real classes
CLASSES = ("wall", "floor", "chair", "table", "door", "cushion", "laptop", "bottle", "paper", "cup", "unlabeled")
VALID_IDS = (1, 2, 3, 5, 8, 13, 21, 34, 55, 89)
N_STUFF = 2


def make_scene(seed, n=5000):
    """Panoptic-style annotation as the reference's dataset hands it over: semantic ids 0..9 (index into CLASSES, 10 = unlabeled
    is expressed as -1 + num_stuff after the shift), instance ids with the stuff classes as instances 0 and 1."""
    g = np.random.default_rng(seed)
    bounds = np.sort(g.choice(np.arange(40, n - 40), size=14, replace=False))
    edges = [0] + bounds.tolist() + [n]
    sem = np.zeros(n, dtype=np.int64)
    inst = np.zeros(n, dtype=np.int64)
    next_inst = N_STUFF
    for i in range(len(edges) - 1):
        lo, hi = edges[i], edges[i + 1]
        c = int(g.integers(0, len(CLASSES) - 1))
        sem[lo:hi] = c
        if c < N_STUFF:
            inst[lo:hi] = c                                   # stuff: instance id = class id
        else:
            inst[lo:hi] = next_inst
            next_inst += 1
    masks, labels, scores = [], [], []
    for i in range(len(edges) - 1):
        lo, hi = edges[i], edges[i + 1]
        if sem[lo] < N_STUFF and g.random() > 0.3:
            continue
        for _ in range(int(g.integers(1, 3))):
            a = int(np.clip(lo + g.integers(-80, 80), 0, n - 1))
            b = int(np.clip(hi + g.integers(-80, 80), a + 1, n))
            m = np.zeros(n, dtype=bool)
            m[a:b] = True
            m &= g.random(n) > 0.04
            lab = int(sem[lo]) - N_STUFF if (sem[lo] >= N_STUFF and g.random() > 0.25) else int(g.integers(len(VALID_IDS) - N_STUFF))
            masks.append(m); labels.append(lab); scores.append(float(np.round(g.random(), 2)))
    return sem, inst, np.stack(masks), np.array(labels, dtype=np.int64), np.array(scores, dtype=np.float32)


def main():
    scenes = [make_scene(300 + s) for s in range(4)]
    e = object.__new__(ev.InstanceSeg3DEvaluator)
    e.debug = False
    e.dataset_meta = dict(seg_valid_class_ids=list(VALID_IDS))
    e.metric_meta = dict(label2cat={i: c for i, c in enumerate(CLASSES)}, ignore_index=[len(CLASSES) - 1], classes=list(CLASSES),
                         dataset_name="ScanNet")
    e.thing_class_inds = list(range(N_STUFF, len(CLASSES) - 1))
    e.stuff_class_inds = list(range(N_STUFF))
    e.submission_prefix_instance = e.submission_prefix_semantic = None
    results = []
    for sem, inst, masks, labels, scores in scenes:
        ann = dict(pts_semantic_mask=sem.copy(), pts_instance_mask=inst.copy())
        pred = dict(pts_semantic_mask=[np.zeros_like(sem), np.zeros_like(sem)], pts_instance_mask=[masks, np.zeros_like(sem)],
                    instance_labels=labels, instance_scores=scores)
        results.append((ann, pred))
    # the reference's compute_metrics builds ret_inst and (its last lines are commented out) returns None: catch the dict
    caught = {}
    real = ev.instance_seg_eval

    def spy(*a, **k):
        caught["gt_sem"] = [np.asarray(x).copy() for x in a[0]]      # before the call: rename_gt works in place
        caught["gt_inst"] = [np.asarray(x).copy() for x in a[1]]
        caught["ret"] = real(*a, **k)
        caught["kw"] = {kk: k[kk] for kk in ("valid_class_ids", "class_labels")}
        return caught["ret"]

    ev.instance_seg_eval = spy
    e.compute_metrics(results)
    ev.instance_seg_eval = real
    ret = caught["ret"]
    assert tuple(caught["kw"]["valid_class_ids"]) == VALID_IDS[N_STUFF:] and tuple(caught["kw"]["class_labels"]) == CLASSES[N_STUFF:-1]
    out = {"classes": np.array(CLASSES), "valid_class_ids": np.array(VALID_IDS), "num_stuff_cls": np.array(N_STUFF)}
    keys = sorted(k for k in ret if k != "classes")
    out["keys"] = np.array(keys)
    out["vals"] = np.array([ret[k] for k in keys], dtype=np.float64)
    out["class_ap"] = np.array([[ret["classes"][c][f] for f in ("ap", "ap50%", "ap25%")] for c in CLASSES[N_STUFF:-1]], dtype=np.float64)
    for si, s in enumerate(scenes):
        out[f"s{si}_sem"], out[f"s{si}_inst"] = s[0], s[1]
        out[f"s{si}_masks"] = np.packbits(s[2], axis=1)
        out[f"s{si}_n"] = np.array(s[2].shape[1])
        out[f"s{si}_labels"], out[f"s{si}_scores"] = s[3], s[4]
        out[f"s{si}_mapped_sem"], out[f"s{si}_mapped_inst"] = caught["gt_sem"][si], caught["gt_inst"][si]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "evaluator_scannet.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    print({k: round(float(v), 5) for k, v in zip(keys, out["vals"])})


if __name__ == "__main__":
    main()
