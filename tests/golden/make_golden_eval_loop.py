"""Generates tests/golden/eval_loop.npz by running the REFERENCE's evaluation loop body
(/root/reference/evaluation/evaluate_3d.py: evaluate_3d, :44-71 - how a forward's output list becomes the evaluator's
`eval_ann_info` / `pred_pts_seg` records) on seeded synthetic targets.  The module itself imports the whole dataset stack, so the
function is taken out of the file with `ast` AT GENERATION TIME and executed with its three free names (torch, tqdm, print);
nothing of its text is stored.  Runs in the build container only."""
import ast
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from segdino3d_amd.gtypes import GD3DTarget  # noqa: E402

SRC = "/root/reference/evaluation/evaluate_3d.py"


def reference_loop():
    tree = ast.parse(open(SRC).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "evaluate_3d")
    ns = {"torch": torch, "tqdm": lambda it: it, "print": lambda *a, **k: None}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), SRC, "exec"), ns)
    return ns["evaluate_3d"]


def make_target(seed, n_points=3000, n_inst=9):
    g = np.random.default_rng(seed)
    masks = np.zeros((n_inst, n_points, 1), dtype=bool)
    bounds = np.sort(g.choice(np.arange(30, n_points - 30), size=n_inst, replace=False))
    edges = [0] + bounds.tolist()
    for i in range(n_inst):
        lo, hi = edges[i], edges[i + 1]
        masks[i, lo:hi, 0] = True
    masks[2, edges[1]:edges[1] + 40, 0] = True                 # instances 1 and 2 overlap on 40 points: the loop SUMS ids there
    labels = g.integers(0, 200, size=n_inst)
    labels[0], labels[1] = 0, 1                                # the two stuff classes come first, as in the dataset
    return GD3DTarget(masks=torch.from_numpy(masks), labels=torch.from_numpy(labels), scene_id=f"scene{seed:04d}_00",
                      extra_features=dict(super_point_masks=torch.from_numpy(g.integers(0, 50, size=n_points))))


class _Loader(list):
    class dataset:                                              # noqa: N801 - `loader.dataset.bg_class_id` (:57)
        bg_class_id = 200


class _Evaluator:
    def __init__(self, targets):
        self.targets, self.records = targets, []

    def inference_single(self, samples, targets, device):
        for t in targets:
            t.pred_pts_seg = dict(tag=t.scene_id)
        return targets

    def process(self, _, results):
        self.records += results

    def evaluate(self, n):
        self.n = n


def main():
    targets = [make_target(500 + s) for s in range(3)]
    ev = _Evaluator(targets)
    loader = _Loader([(None, [t]) for t in targets])
    reference_loop()(ev, loader, cfg=None, current_iter=0, device="cpu")
    assert ev.n == 3 and len(ev.records) == 3
    out = {"bg_class_id": np.array(200)}
    for i, (t, r) in enumerate(zip(targets, ev.records)):
        assert r["pred_pts_seg"] == dict(tag=t.scene_id) and r["eval_ann_info"]["lidar_idx"] == t.scene_id
        out[f"s{i}_masks"] = np.packbits(t.masks.numpy()[:, :, 0], axis=1)
        out[f"s{i}_n"] = np.array(t.masks.shape[1])
        out[f"s{i}_labels"] = t.labels.numpy()
        out[f"s{i}_sp"] = t.extra_features["super_point_masks"].numpy()
        out[f"s{i}_inst"] = r["eval_ann_info"]["pts_instance_mask"]
        out[f"s{i}_sem"] = r["eval_ann_info"]["pts_semantic_mask"]
        assert np.array_equal(r["eval_ann_info"]["sp_pts_mask"], out[f"s{i}_sp"])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "eval_loop.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; instance ids", np.unique(out["s0_inst"]))


if __name__ == "__main__":
    main()
