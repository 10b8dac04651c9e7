"""Loads tests/golden/loss_criterion.npz (reference outputs, see tests/golden/make_golden_loss.py) into the
structures the oracle and the device criterion take."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss_criterion.npz")

CASE_CFG = {
    "s200": dict(matcher="sparse", topk=1, cost_weights=[0.5, 1.0, 1.0, 0.5, 0.5], loss_weight=[0.5, 1.0, 1.0, 0.5, 0.5, 0.5]),
    "base": dict(matcher="sparse", topk=2, cost_weights=[0.5, 1.0, 1.0], loss_weight=[0.5, 1.0, 1.0, 0.5]),
    "hung": dict(matcher="hungarian", topk=0, cost_weights=[0.5, 1.0, 1.0, 0.5, 0.5], loss_weight=[0.5, 1.0, 1.0, 0.5, 0.5, 0.5]),
}
KEYS = ("cls_preds", "sem_preds", "masks", "scores", "centers", "sizes")


def load_case(name, dtype=torch.float32, device="cpu"):
    z = np.load(GOLDEN)
    n_scenes, n_layers, n_cls, n_sem = (int(v) for v in z[f"{name}/shape"])
    cfg = dict(CASE_CFG[name], num_classes=n_cls, num_semantic_classes=n_sem, sem_ignore_index=n_sem, sem_loss_weight=0.5,
               non_object_weight=0.1, fix_dice_loss_weight=True, iter_matcher=True, fix_mean_loss=True)
    targets = []
    for b in range(n_scenes):
        t = {}
        for k in ("sp_inst_sem_masks", "query_inst_sem_masks", "labels", "instance_centers", "instance_sizes"):
            key = f"{name}/target{b}/{k}"
            if key in z:
                v = torch.from_numpy(z[key])
                t[k] = (v.to(dtype) if v.is_floating_point() else v).to(device)
        targets.append(t)
    layers, grads = [], []
    for l in range(n_layers):
        layer, grad = {k: [] for k in KEYS}, {k: [] for k in KEYS}
        for k in KEYS:
            for b in range(n_scenes):
                key = f"{name}/layer{l}/{k}{b}"
                if key in z:
                    layer[k].append(torch.from_numpy(z[key]).to(dtype).to(device))
                    grad[k].append(torch.from_numpy(z[f"{name}/layer{l}/grad_{k}{b}"]))
                else:
                    layer[k].append(None)
                    grad[k].append(None)
        layers.append(layer)
        grads.append(grad)
    expected = dict(seg_loss=float(z[f"{name}/seg_loss"]), inst_loss=float(z[f"{name}/inst_loss"]), grads=grads)
    return cfg, targets, layers, expected


def as_pred(layers):
    pred = dict(layers[-1])
    pred["aux_outputs"] = layers[:-1]
    return pred
