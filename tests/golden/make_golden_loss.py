"""Generates tests/golden/loss_criterion.npz by running the REFERENCE's training criterion
(/root/reference/segdino3d/models/loss/loss_3d.py: ScanNetUnifiedCriterion -> InstanceCriterion / SparseMatcher /
HungarianMatcher / ScanNetSemanticCriterion) on seeded synthetic predictions, with torch autograd for the
gradients.  Runs in the build container only (needs /root/reference); the fixture it writes is data.
The one import of that file that is not torch / scipy (`from segdino3d import LOSSES`, a registry decorator) is
stubbed with a pass-through."""
import copy
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


class _Registry:
    def register_module(self, *a, **k):
        return lambda cls: cls


pkg = types.ModuleType("segdino3d")
pkg.LOSSES = _Registry()
sys.modules["segdino3d"] = pkg
spec = importlib.util.spec_from_file_location("ref_loss", "/root/reference/segdino3d/models/loss/loss_3d.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


class Target(dict):
    """Stands in for the reference's GD3DTarget: item AND attribute access (loss_3d.py:751-763 uses both)."""
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def make_case(seed, n_scenes, n_layers, Q, S, G, n_cls, n_sem, with_box, with_score):
    g = torch.Generator().manual_seed(seed)
    targets, pred_layers = [], []
    for b in range(n_scenes):
        Qb, Sb, Gb = Q - 3 * b, S + 8 * b, G + 2 * b
        owner = torch.randint(0, Gb + 1, (Sb,), generator=g)               # object of each superpoint (Gb = none)
        inst = torch.stack([owner == k for k in range(Gb)])
        sem_of = torch.randint(0, n_sem, (Gb + 1,), generator=g)
        sem_id = sem_of[owner]
        sem_id[torch.rand(Sb, generator=g) < 0.1] = n_sem                  # unlabeled superpoints
        sem = torch.stack([sem_id == k for k in range(n_sem + 1)])
        sp_masks = torch.cat([inst, sem])                                  # [G + n + 1, S]
        ids = torch.randperm(Sb, generator=g)[:Qb]                         # queries = a subset of the superpoints
        t = Target(sp_inst_sem_masks=sp_masks, query_inst_sem_masks=sp_masks[:, ids],
                   labels=torch.randint(0, n_cls, (Gb,), generator=g))
        if with_box:
            t["instance_centers"] = torch.rand(Gb, 3, generator=g) * 4
            t["instance_sizes"] = torch.rand(Gb, 3, generator=g) * 2
        targets.append(t)
    for l in range(n_layers):
        layer = dict(cls_preds=[], sem_preds=[], masks=[], scores=[], centers=[], sizes=[])
        for b in range(n_scenes):
            Qb, Sb = Q - 3 * b, S + 8 * b
            t = targets[b]
            qm = t["query_inst_sem_masks"][:-n_sem - 1].float()            # [G, Q]
            gt_of_q = t["sp_inst_sem_masks"][:-n_sem - 1].float().T @ qm   # [S, Q]: superpoint s and query q share an object
            layer["cls_preds"].append(torch.randn(Qb, n_cls + 1, generator=g))
            layer["sem_preds"].append(torch.randn(Qb, n_sem + 1, generator=g))
            layer["masks"].append(2.5 * (gt_of_q.T * 2 - 1) * (0.3 + 0.2 * l) + 1.5 * torch.randn(Qb, Sb, generator=g))
            layer["scores"].append(torch.rand(Qb, 1, generator=g) if with_score else None)
            has_box = with_box and l > 0                                   # the first prediction set has no boxes
            layer["centers"].append(torch.rand(Qb, 3, generator=g) * 4 if has_box else None)
            layer["sizes"].append(torch.rand(Qb, 3, generator=g) * 2 if has_box else None)
        pred_layers.append(layer)
    return targets, pred_layers


def run_reference(targets, pred_layers, cfg):
    leaves = []
    layers = copy.deepcopy(pred_layers)
    for layer in layers:
        for k, lst in layer.items():
            for i, v in enumerate(lst):
                if v is not None:
                    lst[i] = v.clone().requires_grad_(True)
                    leaves.append(lst[i])
    pred = dict(layers[-1])
    pred["aux_outputs"] = layers[:-1]
    crit = ref.ScanNetUnifiedCriterion(**copy.deepcopy(cfg))
    out = crit(pred, copy.deepcopy(targets))
    total = out["seg_loss"] + out["inst_loss"]
    total.backward()
    return out, layers


def main():
    cases = {
        # ScanNet200 prototype: sparse matcher with centre / size costs, 6 loss weights, no objectness scores
        "s200": dict(shape=dict(seed=1, n_scenes=2, n_layers=3, Q=40, S=72, G=6, n_cls=11, n_sem=13, with_box=True, with_score=False),
                     matcher=dict(type="SparseMatcher", topk=1,
                                  costs=[dict(type="QueryClassificationCost", weight=0.5), dict(type="MaskBCECost", weight=1.0),
                                         dict(type="MaskDiceCost", weight=1.0), dict(type="CenterL1Cost", weight=0.5),
                                         dict(type="SizeL1Cost", weight=0.5)]),
                     loss_weight=[0.5, 1.0, 1.0, 0.5, 0.5, 0.5]),
        # base config: 3 costs, 4 loss weights, objectness scores present, top-2 matches
        "base": dict(shape=dict(seed=2, n_scenes=1, n_layers=2, Q=56, S=64, G=5, n_cls=7, n_sem=9, with_box=False, with_score=True),
                     matcher=dict(type="SparseMatcher", topk=2,
                                  costs=[dict(type="QueryClassificationCost", weight=0.5), dict(type="MaskBCECost", weight=1.0),
                                         dict(type="MaskDiceCost", weight=1.0)]),
                     loss_weight=[0.5, 1.0, 1.0, 0.5]),
        # Hungarian matcher, three scenes (exercises the batch-size dependent dice scaling)
        "hung": dict(shape=dict(seed=3, n_scenes=3, n_layers=2, Q=32, S=48, G=4, n_cls=5, n_sem=6, with_box=True, with_score=True),
                     matcher=dict(type="HungarianMatcher",
                                  costs=[dict(type="QueryClassificationCost", weight=0.5), dict(type="MaskBCECost", weight=1.0),
                                         dict(type="MaskDiceCost", weight=1.0), dict(type="CenterL1Cost", weight=0.5),
                                         dict(type="SizeL1Cost", weight=0.5)]),
                     loss_weight=[0.5, 1.0, 1.0, 0.5, 0.5, 0.5]),
    }
    blob = {}
    for name, c in cases.items():
        sh = c["shape"]
        targets, layers = make_case(**sh)
        cfg = dict(num_semantic_classes=sh["n_sem"],
                   sem_criterion=dict(type="ScanNetSemanticCriterion", ignore_index=sh["n_sem"], loss_weight=0.5),
                   inst_criterion=dict(type="InstanceCriterion", matcher=c["matcher"], loss_weight=c["loss_weight"],
                                       num_classes=sh["n_cls"], non_object_weight=0.1, fix_dice_loss_weight=True,
                                       iter_matcher=True, fix_mean_loss=True))
        out, leaves = run_reference(targets, layers, cfg)
        blob[f"{name}/seg_loss"] = out["seg_loss"].detach().numpy()
        blob[f"{name}/inst_loss"] = out["inst_loss"].detach().numpy()
        blob[f"{name}/shape"] = np.array([sh[k] for k in ("n_scenes", "n_layers", "n_cls", "n_sem")])
        for b, t in enumerate(targets):
            for k, v in t.items():
                blob[f"{name}/target{b}/{k}"] = v.numpy()
        for l, layer in enumerate(leaves):
            for k, lst in layer.items():
                for b, v in enumerate(lst):
                    if v is not None:
                        blob[f"{name}/layer{l}/{k}{b}"] = v.detach().numpy()
                        blob[f"{name}/layer{l}/grad_{k}{b}"] = (v.grad if v.grad is not None else torch.zeros_like(v)).numpy()
        print(name, float(out["seg_loss"]), float(out["inst_loss"]))
    np.savez_compressed(os.path.join(HERE, "loss_criterion.npz"), **blob)
    print("wrote", os.path.join(HERE, "loss_criterion.npz"), os.path.getsize(os.path.join(HERE, "loss_criterion.npz")))


if __name__ == "__main__":
    main()
