"""Training criterion on the HIP device.

Mirrors the reference's `segdino3d/models/loss/loss_3d.py` behind the same registry name and config surface:
`ScanNetUnifiedCriterion(num_semantic_classes, sem_criterion, inst_criterion)` (`:713-780`) built from the shipped
`criterion_cfg` (`configs/models/base_3d.py:37-63`, `configs/prototypes/SegDINO3D_ScanNet200.py:19-28`), called as
`criterion(pred, targets)` with the decoder's output dict (`instance_seg_3d_decoder.py:773-797`) and returning
`{"seg_loss", "inst_loss"}`.

What differs from the reference is where the work happens.  The reference builds the losses from ~40 torch ops per
layer and scene and lets autograd derive the gradients; here each (layer, scene) is three C-ABI calls
(`sd3d_match_costs`, `sd3d_sparse_match`, `sd3d_instance_loss`; csrc/loss.hip) that produce the loss terms AND the
gradients with respect to every prediction in the same pass over the `[Q, S]` mask logits.  The returned losses are
attached to those gradients through one `torch.autograd.Function`, so `loss.backward()` hands them to whatever
produced the predictions.  Only the Hungarian assignment itself stays on the host (scipy, as in the reference `:311`).
There is no CPU fallback: CPU tensors raise.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Sequence

import torch

from . import _lib, ops
from .builder import LOSSES

PRED_KEYS = ("cls_preds", "masks", "scores", "centers", "sizes")
_COST_SLOT = {"QueryClassificationCost": 0, "MaskBCECost": 1, "MaskDiceCost": 2, "CenterL1Cost": 3, "SizeL1Cost": 4}


def _get(obj, key, default=None):
    """Targets are GD3DTarget-like: attribute or item access (the reference uses both, loss_3d.py:751-763)."""
    if isinstance(obj, dict):
        return obj.get(key, default)
    v = getattr(obj, key, None)
    if v is None and hasattr(obj, "__getitem__"):
        try:
            v = obj[key]
        except (KeyError, IndexError, TypeError):
            v = None
    return default if v is None else v


def _f32(t, name):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"criterion: {name} must live on the HIP device (no CPU fallback), got {t.device}")
    return t.detach().float().contiguous()


class _SceneTruth:
    """Device-side ground truth of one scene in the layout the kernels take: bit rows of the object masks over the
    superpoints, their sizes, byte rows of the query / semantic masks."""

    def __init__(self, target, n_sem: int):
        lib = _lib.load()
        sp = _get(target, "sp_inst_sem_masks")
        qm = _get(target, "query_inst_sem_masks")
        if sp is None or qm is None:
            raise KeyError("criterion: targets need sp_inst_sem_masks and query_inst_sem_masks (baseline3d.py:246-262)")
        if not sp.is_cuda:
            raise RuntimeError("criterion: targets must live on the HIP device (no CPU fallback)")
        dev = sp.device
        self.labels = _get(target, "labels").to(torch.int64).contiguous()
        self.G = int(self.labels.shape[0])
        if sp.shape[0] != self.G + n_sem + 1:
            raise ValueError(f"criterion: sp_inst_sem_masks has {sp.shape[0]} rows, expected {self.G} objects + {n_sem + 1} classes")
        self.S = int(sp.shape[1])
        self.Q = int(qm.shape[1])
        inst = sp[: self.G].to(torch.uint8).contiguous()
        self.words = (self.S + 31) // 32
        self.bits = torch.empty(max(self.G, 1), self.words, dtype=torch.int32, device=dev)
        self.count = torch.empty(max(self.G, 1), dtype=torch.int32, device=dev)
        if self.G:
            _lib.check(lib.sd3d_pack_mask_bits(inst.data_ptr(), inst.stride(0), self.G, self.S, self.bits.data_ptr(), self.words,
                                               self.count.data_ptr(), ops._stream()), "pack_mask_bits")
        self.query_masks = qm[: self.G].to(torch.uint8).contiguous()              # [G, Q]
        self.sem_masks = qm[qm.shape[0] - n_sem - 1:].to(torch.uint8).contiguous()  # [n + 1, Q]
        c, s = _get(target, "instance_centers"), _get(target, "instance_sizes")
        self.centers = None if c is None else c.float().contiguous()
        self.sizes = None if s is None else s.float().contiguous()


class InstanceCriterion:
    """`loss_3d.py:368-710`: same constructor arguments; `layer_terms` evaluates one prediction set of one scene."""

    def __init__(self, matcher, loss_weight, non_object_weight, num_classes, fix_dice_loss_weight, iter_matcher, fix_mean_loss=False):
        matcher = dict(matcher)
        kind = matcher.pop("type", None)
        if kind not in ("SparseMatcher", "HungarianMatcher"):
            raise AssertionError("Matcher type must be 'SparseMatcher' or 'HungarianMatcher'.")    # loss_3d.py:389-390
        self.matcher_kind = kind
        self.topk = int(matcher.get("topk", 0))
        self.cost_weights = [0.0] * 5
        for c in matcher.get("costs", []):
            c = dict(c)
            self.cost_weights[_COST_SLOT[c.pop("type")]] += float(c["weight"])
        self.loss_weight = [float(w) for w in loss_weight]
        if len(self.loss_weight) not in (4, 5, 6):
            raise ValueError("loss_weight must hold 4, 5 or 6 weights (loss_3d.py:532-553)")
        self.class_weight = [1.0] * num_classes + [float(non_object_weight)]
        self.num_classes = num_classes
        self.fix_dice_loss_weight = fix_dice_loss_weight
        self.iter_matcher = iter_matcher
        self.fix_mean_loss = fix_mean_loss
        self._cw = {}

    def scene_coefficients(self, n_scenes: int, last: bool) -> List[float]:
        """d(layer loss) / d(per-scene term) for [cls, bce, dice, score, centre, size]: the loss weights times the
        batch-size factors of loss_3d.py:505-521 (auxiliary layers) and :665-679 (last layer)."""
        b = float(n_scenes)
        w = self.loss_weight + [0.0] * (6 - len(self.loss_weight))
        bce = 1.0 / b                                   # sum / B; `fix_mean_loss` multiplies by B / B
        dice = 1.0 if last else 1.0 / b                 # the last layer does not divide the sum (:657 vs :502)
        if self.fix_dice_loss_weight:
            dice = dice / b * 4.0
        return [w[0] / b, w[1] * bce, w[2] * dice, w[3] / b, w[4] / b, w[5] / b]

    def _class_weight(self, device):
        key = str(device)
        if key not in self._cw:
            self._cw[key] = torch.tensor(self.class_weight, dtype=torch.float32, device=device)
        return self._cw[key]

    def match(self, layer, i, truth: _SceneTruth) -> torch.Tensor:
        """[Q, G] byte matrix of matched (query, object) pairs for scene i of a prediction set."""
        lib = _lib.load()
        cls, masks = _f32(layer["cls_preds"][i], "cls_preds"), _f32(layer["masks"][i], "masks")
        ctr, size = _f32(layer["centers"][i], "centers"), _f32(layer["sizes"][i], "sizes")
        Q, G = masks.shape[0], truth.G
        dev = masks.device
        cost = torch.empty(Q, G, dtype=torch.float32, device=dev)
        w5 = (ctypes.c_float * 5)(*self.cost_weights)
        sparse = self.matcher_kind == "SparseMatcher"
        _lib.check(lib.sd3d_match_costs(cls.data_ptr(), cls.stride(0), cls.shape[1], masks.data_ptr(), masks.stride(0), Q, masks.shape[1],
                                        ops._ptr(ctr), ops._ptr(size), truth.labels.data_ptr(), truth.bits.data_ptr(), truth.words,
                                        truth.count.data_ptr(), G, ops._ptr(truth.centers), truth.centers.stride(0) if truth.centers is not None else 0,
                                        ops._ptr(truth.sizes), truth.sizes.stride(0) if truth.sizes is not None else 0,
                                        truth.query_masks.data_ptr() if sparse else None, w5, cost.data_ptr(), ops._stream()), "match_costs")
        match = torch.zeros(Q, G, dtype=torch.uint8, device=dev)
        if sparse:
            _lib.check(lib.sd3d_sparse_match(cost.data_ptr(), Q, G, self.topk, match.data_ptr(), ops._stream()), "sparse_match")
        else:
            from scipy.optimize import linear_sum_assignment                      # the assignment itself: host, as loss_3d.py:311
            q_ids, g_ids = linear_sum_assignment(cost.cpu().numpy())
            match[torch.as_tensor(q_ids, device=dev), torch.as_tensor(g_ids, device=dev)] = 1
        self.last_cost = cost
        return match

    def layer_terms(self, layer, i, truth: _SceneTruth, match: torch.Tensor, coef: Sequence[float]):
        """-> (parts [8] device tensor, dict of gradients of the total loss w.r.t. this scene's predictions)."""
        lib = _lib.load()
        cls, masks = _f32(layer["cls_preds"][i], "cls_preds"), _f32(layer["masks"][i], "masks")
        score = _f32(layer["scores"][i], "scores")
        ctr, size = _f32(layer["centers"][i], "centers"), _f32(layer["sizes"][i], "sizes")
        Q, S = masks.shape
        dev = masks.device
        if S != truth.S or Q != truth.Q:
            raise ValueError(f"criterion: predictions are [{Q}, {S}], targets describe [{truth.Q}, {truth.S}]")
        grads = dict(cls_preds=torch.empty_like(cls), masks=torch.empty(Q, S, dtype=torch.float32, device=dev),
                     scores=None if score is None else torch.empty(Q, dtype=torch.float32, device=dev),
                     centers=None if ctr is None else torch.empty_like(ctr), sizes=None if size is None else torch.empty_like(size))
        parts = torch.empty(8, dtype=torch.float32, device=dev)
        ws_bytes = lib.sd3d_instance_loss_ws_bytes(Q)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        c6 = (ctypes.c_float * 6)(*coef)
        _lib.check(lib.sd3d_instance_loss(
            cls.data_ptr(), cls.stride(0), cls.shape[1], masks.data_ptr(), masks.stride(0), Q, S,
            None if score is None else score.reshape(-1).data_ptr(), ops._ptr(ctr), ops._ptr(size),
            truth.labels.data_ptr(), truth.bits.data_ptr(), truth.words, truth.count.data_ptr(), truth.G,
            ops._ptr(truth.centers), truth.centers.stride(0) if truth.centers is not None else 0,
            ops._ptr(truth.sizes), truth.sizes.stride(0) if truth.sizes is not None else 0,
            match.data_ptr(), self._class_weight(dev).data_ptr(), c6, grads["cls_preds"].data_ptr(), grads["masks"].data_ptr(),
            ops._ptr(grads["scores"]), ops._ptr(grads["centers"]), ops._ptr(grads["sizes"]), parts.data_ptr(),
            ws.data_ptr(), ws_bytes, ops._stream()), "instance_loss")
        if grads["scores"] is not None:
            grads["scores"] = grads["scores"].reshape(layer["scores"][i].shape)
        return parts, grads


class ScanNetSemanticCriterion:
    """`loss_3d.py:26-60`."""

    def __init__(self, ignore_index, loss_weight):
        self.ignore_index = ignore_index
        self.loss_weight = loss_weight

    def scene_terms(self, sem_pred, truth: _SceneTruth, n_scenes: int):
        lib = _lib.load()
        sem = _f32(sem_pred, "sem_preds")
        Q, C = sem.shape
        n_logits = C - 1 if self.ignore_index >= 0 else C
        grad = torch.empty_like(sem)
        loss = torch.empty(1, dtype=torch.float32, device=sem.device)
        ws_bytes = lib.sd3d_semantic_loss_ws_bytes(Q)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=sem.device)
        _lib.check(lib.sd3d_semantic_loss(sem.data_ptr(), sem.stride(0), Q, truth.sem_masks.shape[0], n_logits, truth.sem_masks.data_ptr(),
                                          int(self.ignore_index), float(self.loss_weight) / n_scenes, grad.data_ptr(), grad.stride(0),
                                          loss.data_ptr(), ws.data_ptr(), ws_bytes, ops._stream()), "semantic_loss")
        return loss, grad


class _AttachGradients(torch.autograd.Function):
    """losses [2] = (seg_loss, inst_loss) with precomputed d(seg_loss + inst_loss-weighted sum)/d(prediction): backward
    scales the stored gradients by the incoming ones (seg and inst gradients are kept apart)."""

    @staticmethod
    def forward(ctx, losses, n_seg, *preds):
        ctx.n_seg = n_seg
        ctx.grads = _AttachGradients._pending
        _AttachGradients._pending = None
        return losses.clone()

    @staticmethod
    def backward(ctx, g):
        out = []
        for k, gr in enumerate(ctx.grads):
            out.append(None if gr is None else gr * (g[0] if k < ctx.n_seg else g[1]))
        return (None, None, *out)

    _pending = None


@LOSSES.register_module(force=True)
class ScanNetUnifiedCriterion:
    """`loss_3d.py:713-780`: the semantic criterion on the last layer's `sem_preds` plus the instance criterion on the
    last layer and every auxiliary prediction set."""

    def __init__(self, num_semantic_classes, sem_criterion, inst_criterion):
        sem_criterion, inst_criterion = dict(sem_criterion), dict(inst_criterion)
        self.num_semantic_classes = num_semantic_classes
        assert sem_criterion.pop("type", None) == "ScanNetSemanticCriterion", \
            "Semantic criterion only support 'ScanNetSemanticCriterion' type currently."
        assert inst_criterion.pop("type", None) == "InstanceCriterion", \
            "Instance criterion only support 'InstanceCriterion' type currently."
        self.sem_criterion = ScanNetSemanticCriterion(**sem_criterion)
        self.inst_criterion = InstanceCriterion(**inst_criterion)
        self.last_parts = None
        self.last_matches = None

    def __call__(self, pred: Dict, insts: Sequence) -> Dict[str, torch.Tensor]:
        ic = self.inst_criterion
        n_b = len(pred["masks"])
        truths = [_SceneTruth(t, self.num_semantic_classes) for t in insts]
        dev = pred["masks"][0].device
        leaves, grads = [], []
        # ---- semantic loss (last layer only, :37-60)
        seg_terms = []
        for i in range(n_b):
            loss, g = self.sem_criterion.scene_terms(pred["sem_preds"][i], truths[i], n_b)
            seg_terms.append(loss)
            leaves.append(pred["sem_preds"][i]); grads.append(g)
        seg_loss = self.sem_criterion.loss_weight * torch.cat(seg_terms).mean()
        n_seg = len(leaves)
        # ---- instance loss: last layer, then the auxiliary prediction sets (:557-710, :398-555)
        layers = [(pred, True)] + [(aux, False) for aux in pred.get("aux_outputs", [])]
        inst_loss = torch.zeros((), dtype=torch.float32, device=dev)
        self.last_parts, self.last_matches = [], []
        w = ic.loss_weight + [0.0] * (6 - len(ic.loss_weight))
        matches_last = None
        for layer, last in layers:
            coef = ic.scene_coefficients(n_b, last)
            parts_b, matches = [], []
            for i in range(n_b):
                if last or ic.iter_matcher:
                    m = ic.match(layer, i, truths[i])
                else:
                    m = matches_last[i]
                matches.append(m)
                parts, g = ic.layer_terms(layer, i, truths[i], m, coef)
                parts_b.append(parts)
                for k in PRED_KEYS:
                    if layer[k][i] is not None:
                        leaves.append(layer[k][i]); grads.append(g[k])
            if last:
                matches_last = matches
            P = torch.stack(parts_b)                                           # [B, 8]
            has_box = [layer["centers"][i] is not None for i in range(n_b)]
            has_size = [layer["sizes"][i] is not None for i in range(n_b)]
            cls = P[:, 0].mean()
            bce = P[:, 1].sum() * (coef[1] / w[1] if w[1] else 0.0)
            dice = P[:, 2].sum() * (coef[2] / w[2] if w[2] else 0.0)
            # scores: sum over the scenes that kept any / B (:505-507); no scores -> 0
            score = (P[:, 3].sum() / n_b) if any(s is not None for s in layer["scores"]) else P.new_zeros(())
            ctr = P[has_box, 4].mean() if any(has_box) else P.new_zeros(())
            size = P[has_size, 5].mean() if any(has_size) else P.new_zeros(())
            terms = torch.stack([cls, bce, dice, score, ctr, size])
            inst_loss = inst_loss + (terms * terms.new_tensor(w)).sum()
            self.last_parts.append(terms)
            self.last_matches.append(matches)
        losses = torch.stack([seg_loss.reshape(()), inst_loss])
        if torch.is_grad_enabled() and any(t.requires_grad for t in leaves):
            _AttachGradients._pending = grads
            losses = _AttachGradients.apply(losses, n_seg, *leaves)
        return {"seg_loss": losses[0], "inst_loss": losses[1]}
