"""Generates tests/golden/ap_protocol.npz by running the REFERENCE's ScanNet AP code
(/root/reference/evaluation/utils_instance_seg_3d_eval.py: instance_seg_eval -> scannet_eval ->
assign_instances_for_scan / evaluate_matches / compute_averages) on seeded synthetic scenes.
Runs in the build container only (needs /root/reference); the fixture it writes is data.

Third-party imports of that file are stubbed: mmengine.logging.print_log, terminaltables.AsciiTable, and
mmdet3d's util_3d.get_instances - the latter with oracle.eval_ref.get_instances (a restatement of the
published ScanNet / mmdet3d helper; mmdet3d is not installed here and not vendored by the reference)."""
import importlib.util
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


_stub("mmengine")
_stub("mmengine.logging", print_log=lambda *a, **k: None)
_stub("terminaltables", AsciiTable=_Table)
_stub("mmdet3d")
_stub("mmdet3d.evaluation")
_stub("mmdet3d.evaluation.functional")
util = _stub("mmdet3d.evaluation.functional.scannet_utils.util_3d", get_instances=eval_ref.get_instances)
_stub("mmdet3d.evaluation.functional.scannet_utils", util_3d=util)

spec = importlib.util.spec_from_file_location("ref_ap", "/root/reference/evaluation/utils_instance_seg_3d_eval.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

# class names drawn from the three ScanNet200 frequency groups the reference averages over, plus one outside them
CLASS_LABELS = ("chair", "table", "door", "cushion", "laptop", "bottle", "paper", "cup", "clock", "wall-thing")
VALID_IDS = (2, 4, 7, 11, 13, 17, 19, 23, 29, 31)


def make_scene(seed, n=6000):
    g = np.random.default_rng(seed)
    # ground truth: contiguous runs of points = instances; semantic ids include two classes that are NOT valid (void)
    sem_pool = list(VALID_IDS) + [1, 3]
    bounds = np.sort(g.choice(np.arange(50, n - 50), size=17, replace=False))
    inst = np.zeros(n, dtype=np.int64)
    sem = np.zeros(n, dtype=np.int64)
    edges = [0] + bounds.tolist() + [n]
    for i in range(len(edges) - 1):
        inst[edges[i]:edges[i + 1]] = i
        sem[edges[i]:edges[i + 1]] = sem_pool[int(g.integers(len(sem_pool)))]
    # predictions: perturbed copies of GT instances (shifted / grown / shrunk), duplicates, and random blobs
    masks, labels, scores = [], [], []
    for i in range(len(edges) - 1):
        lo, hi = edges[i], edges[i + 1]
        for _ in range(int(g.integers(1, 4))):
            a = int(np.clip(lo + g.integers(-120, 120), 0, n - 1))
            b = int(np.clip(hi + g.integers(-120, 120), a + 1, n))
            m = np.zeros(n, dtype=bool)
            m[a:b] = True
            m &= g.random(n) > 0.05
            s = int(sem[lo])
            lab = VALID_IDS.index(s) if (s in VALID_IDS and g.random() > 0.2) else int(g.integers(len(VALID_IDS)))
            masks.append(m); labels.append(lab); scores.append(float(np.round(g.random(), 2)))   # rounded: tied scores occur
    for _ in range(6):
        m = g.random(n) > 0.97 if g.random() > 0.5 else np.zeros(n, dtype=bool)                # tiny / empty masks
        masks.append(m); labels.append(int(g.integers(len(VALID_IDS)))); scores.append(float(g.random()))
    return sem, inst, np.stack(masks), np.array(labels, dtype=np.int64), np.array(scores, dtype=np.float32)


def main():
    import torch
    scenes = [make_scene(100 + s) for s in range(4)]
    out = {"class_labels": np.array(CLASS_LABELS), "valid_class_ids": np.array(VALID_IDS), "n_scenes": np.array(len(scenes))}
    for opt_name, options in (("default", None), ("min30", dict(min_region_sizes=np.array([30])))):
        metrics = ref.instance_seg_eval(
            [s[0].copy() for s in scenes], [s[1].copy() for s in scenes], [torch.from_numpy(s[2]) for s in scenes],
            [torch.from_numpy(s[3]) for s in scenes], [torch.from_numpy(s[4]) for s in scenes],
            valid_class_ids=VALID_IDS, class_labels=CLASS_LABELS, options=options, print_log_flag=False)
        keys = sorted(k for k in metrics if k != "classes")
        out[f"{opt_name}_keys"] = np.array(keys)
        out[f"{opt_name}_vals"] = np.array([metrics[k] for k in keys], dtype=np.float64)
        out[f"{opt_name}_class_ap"] = np.array([[metrics["classes"][c][f] for f in ("ap", "ap50%", "ap25%", "prec50%", "rec50%")]
                                                for c in CLASS_LABELS], dtype=np.float64)
        # the per-scene association the reference builds (compact form)
        opts = ref.get_options(options)
        id_to_label = {VALID_IDS[i]: CLASS_LABELS[i] for i in range(len(VALID_IDS))}
        preds = ref.aggregate_predictions([torch.from_numpy(s[2]) for s in scenes], [torch.from_numpy(s[3]) for s in scenes],
                                          [torch.from_numpy(s[4]) for s in scenes], VALID_IDS)
        gts = ref.rename_gt([s[0].copy() for s in scenes], [s[1].copy() for s in scenes], VALID_IDS)
        rows = []
        for si, (p, gt) in enumerate(zip(preds, gts)):
            g2p, p2g = ref.assign_instances_for_scan(p, gt, opts, VALID_IDS, CLASS_LABELS, id_to_label)
            for label in CLASS_LABELS:
                for pr in p2g[label]:
                    for m in pr["matched_gt"]:
                        rows.append([si, int(pr["filename"].split("_")[1]), pr["label_id"], pr["vert_count"], pr["void_intersection"],
                                     m["instance_id"], m["vert_count"], m["intersection"]])
                    if not pr["matched_gt"]:
                        rows.append([si, int(pr["filename"].split("_")[1]), pr["label_id"], pr["vert_count"], pr["void_intersection"], -1, 0, 0])
        out[f"{opt_name}_assoc"] = np.array(sorted(rows), dtype=np.int64)
    for si, s in enumerate(scenes):
        out[f"s{si}_sem"], out[f"s{si}_inst"] = s[0], s[1]
        out[f"s{si}_masks"] = np.packbits(s[2], axis=1)
        out[f"s{si}_n"] = np.array(s[2].shape[1])
        out[f"s{si}_labels"], out[f"s{si}_scores"] = s[3], s[4]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ap_protocol.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    print({k: float(v) for k, v in zip(out["default_keys"], out["default_vals"])})


if __name__ == "__main__":
    main()
