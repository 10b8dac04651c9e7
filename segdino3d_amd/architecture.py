"""Baseline3D: forward of SegDINO3D on the MI355X kernels (host side); eval mode = the benchmarked inference path, training
mode = the same forward with autograd nodes over HIP kernels + the criterion.

Mirrors the reference operator interface `segdino3d/models/architecture/baseline3d.py:144-556`:
same constructor kwargs (:146-160), same attributes (`backbone`, `decoder`, `criterion`, `test_cfg`,
`query_num`, ...), `forward(samples, targets)` returning the SAME `targets` list with
`targets[0].pred_pts_seg` attached in eval mode (:333-338), `pred_pts_seg` holding the fields of the
reference `PointData` (:397-404).  The arithmetic (post-processing included) runs through
segdino3d_amd.ops; torch is used for tensor bookkeeping (row gathers by index, concatenation, D2H).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
import torch.nn as nn

from . import _trace, ops
from .builder import ARCHITECTURES, build_backbone, build_decoder, build_loss, build_text_encoder

try:  # pragma: no cover - mmdet3d is absent from the build image
    from mmdet3d.structures import PointData  # type: ignore
except Exception:  # noqa: BLE001
    class PointData:
        """Minimal stand-in for mmdet3d.structures.PointData: attribute container with `.items()`."""

        def __init__(self, **fields):
            self.__dict__.update(fields)

        def items(self):
            return self.__dict__.items()

        def keys(self):
            return self.__dict__.keys()

        def __getitem__(self, k):
            return self.__dict__[k]

        def __contains__(self, k):
            return k in self.__dict__


from . import criterion as _criterion  # noqa: E402,F401 - registers ScanNetUnifiedCriterion

import os as _os  # noqa: E402
# SD3D_FAN_OUT=1: the scenes of a batched evaluation forward run their decoders / post-processing side by side on one side
# stream each.  Bit-identical, but MEASURED SLOWER than one after the other on the forward's stream (3 streams x batch 4:
# 91.8 vs 109.1 scenes/s; more hardware queues make it worse): many concurrent chains of tiny launches take workgroup slots away
# from the persistent convolution kernels of the other batches, whose static tile partition then waits for its slowest workgroup.
FAN_OUT = _os.environ.get("SD3D_FAN_OUT", "0") == "1"


def _cfg_get(cfg, key, default=None):
    if cfg is None:
        return default
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default) if not hasattr(cfg, "get") else cfg.get(key, default)


# SD3D_TOPK_SELECT=0: the top-k (query, class) pairs of the post-processing by a full radix sort of all Q x C scores (rounds 1 - 4)
TOPK_SELECT = _os.environ.get("SD3D_TOPK_SELECT", "1") != "0"


def _sorted_desc(score: torch.Tensor):
    """indices (int32) that sort `score` descending (stable) via the radix sort kernels."""
    keys = ops.keys_from_f32(score, descending=True)
    _, idx = ops.sort_pairs(keys, None, 0, 32)
    return idx


@ARCHITECTURES.register_module()
class Baseline3D(nn.Module):
    def __init__(self, num_classes: int, pointcloud_backbone_cfg: Dict, decoder_cfg: Dict = None, criterion_cfg: Dict = None,
                 text_encoder_cfg: Dict = None, use_sim_classifier: bool = False, query_thr: float = 0.5, test_cfg=None,
                 add_positional_embedding=False, mode_3d_center: str = "mean", query_num=-1,
                 filter_outofbox_points_eval: bool = False):
        super().__init__()
        self.backbone = build_backbone(pointcloud_backbone_cfg)
        self.decoder = build_decoder(decoder_cfg)
        self.criterion = build_loss(criterion_cfg)
        if text_encoder_cfg is not None:
            self.text_encoder = build_text_encoder(text_encoder_cfg)
        self.use_sim_classifier = use_sim_classifier
        if self.use_sim_classifier:
            assert text_encoder_cfg is not None, "Text encoder must be provided when using sim classifier."
        self.query_thr = query_thr
        self.num_classes = num_classes
        self.test_cfg = test_cfg
        self.add_positional_embedding = add_positional_embedding
        self.mode_3d_center = mode_3d_center
        self.query_num = query_num
        self.filter_outofbox_points_eval = filter_outofbox_points_eval
        if self.filter_outofbox_points_eval:
            assert self.decoder.add_box_size_pred, \
                "When filter_outofbox_points_eval is True, decoder must have add_box_size_pred set to True."
        # True: numpy outputs like the reference's `.cpu().numpy()`; "packed": the same with the [n, N] instance masks bit-packed
        # (PackedMasks); False: keep post-processed outputs on the device (bench.py forward timing)
        self.to_host = True
        self._stuff_cols = {}        # stuff-class column list per device (predict_by_feat panoptic branch)

    # ---- get_extra_instance_data (:266-306) ------------------------------------------------------
    def get_extra_instance_data(self, samples, targets, add_instance_centers=False, add_instance_axis_aligned_box=False):
        if not (add_instance_centers or add_instance_axis_aligned_box):
            return None
        scene_range = []
        for i in range(len(targets)):
            pts = samples[i]
            if "elastic_coords" in targets[i]:                 # boxes and scene range of the distorted scene (:280-281)
                pts = (targets[i]["elastic_coords"].to(pts.device).float() * self.backbone.voxel_size).contiguous()
            stats = ops.scene_stats(pts)
            scene_range.append((stats[0:3], stats[3:6]))
            masks = targets[i].get("masks") if hasattr(targets[i], "get") else targets[i]["masks"]
            if masks is not None:
                m = masks[..., 0] if masks.dim() == 3 else masks
                centers, sizes = ops.instance_boxes(pts, m, self.mode_3d_center)
                if add_instance_centers:
                    targets[i].instance_centers = centers
                if add_instance_axis_aligned_box:
                    targets[i].instance_sizes = sizes
        return scene_range

    def forward_backbone(self, samples, targets):
        return self.backbone.forward_wrapper(samples, targets, return_sp_mean_pos=True)

    def forward_decoder(self, sp_features_3d, sp_pos, sp_pos_wo_elastic, queries, queries_pos, targets, scene_range):
        if self.decoder.add_dinox_query_ca:
            query2d_feat = [t["extra_features"]["query2d_feats"] for t in targets]
            query2d_pos = [t["extra_features"]["query2d_pos"] for t in targets]
        else:
            query2d_feat = query2d_pos = None
        return self.decoder(sp_features_3d, sp_pos, sp_pos_wo_elastic, queries, queries_pos, query2d_feat, query2d_pos,
                            scene_range)

    # ---- _select_queries (:207-264) ------------------------------------------------------------------
    def _select_queries(self, x, x_pos=None, targets=None):
        if not self.training and self.query_num == -1:           # evaluation: every superpoint is a query (:227-228)
            return x, x_pos, targets
        queries, queries_pos = [], ([] if self.add_positional_embedding else None)
        if self.query_num > 0:                                   # top-`query_num` superpoints, training AND evaluation (:231-249)
            for i in range(len(x)):
                if x[i].shape[0] > self.query_num:
                    with torch.no_grad():                        # the indices carry no gradient; x[i][ids] below does
                        score = self.decoder.select_scores(x[i].detach())
                    ids = _sorted_desc(score)[: self.query_num].long()
                else:
                    ids = torch.arange(x[i].shape[0], device=x[i].device)
                queries.append(x[i][ids])
                # the reference reads `sp_inst_sem_masks` unconditionally here (:246); a benchmark scene without
                # ground truth is tolerated in evaluation, training needs it for the matcher
                if targets is not None and (self.training or "sp_inst_sem_masks" in targets[i]):
                    targets[i].query_inst_sem_masks = targets[i].sp_inst_sem_masks[:, ids]
                if x_pos is not None and queries_pos is not None:
                    queries_pos.append(x_pos[i][ids])
            return queries, queries_pos, targets
        for i in range(len(x)):                                  # random subset of the superpoints as queries (:250-264)
            if self.query_thr < 1:
                n = (1 - self.query_thr) * torch.rand(1) + self.query_thr           # host RNG, like the reference
                n = (n * len(x[i])).int()
                # (pinned staging + an asynchronous copy: a pageable host -> device copy blocks the host until the stream has drained - the
                #  whole backbone the host had run ahead of - and the decoder's ~340 autograd nodes then start from an empty queue)
                ids = torch.randperm(len(x[i]))[:n].pin_memory().to(x[i].device, non_blocking=True)
                queries.append(x[i][ids])
                targets[i].query_inst_sem_masks = targets[i].sp_inst_sem_masks[:, ids]
                if x_pos is not None and queries_pos is not None:
                    queries_pos.append(x_pos[i][ids])
            else:
                queries.append(x[i])
                targets[i].query_inst_sem_masks = targets[i].sp_inst_sem_masks
                if x_pos is not None and queries_pos is not None:
                    queries_pos.append(x_pos[i])
        return queries, queries_pos, targets

    # ---- forward (:308-346) --------------------------------------------------------------------------------
    @ops.bound_stream
    def forward(self, samples, targets: List = None):
        samples = [s.float().contiguous() for s in samples]
        sp_features_3d, sp_pos, sp_pos_wo_elastic = self.forward_backbone(samples, targets)
        # (baseline3d.py:317 computes these before the backbone; nothing in the backbone reads them, and behind it their launches queue up under the
        #  U-Net's kernels instead of standing - host-bound - in front of the voxelisation every later kernel of the scene waits for)
        scene_range = self.get_extra_instance_data(samples, targets, self.add_positional_embedding,
                                                   self.decoder.add_box_size_pred)
        queries, queries_pos, targets = self._select_queries(sp_features_3d, sp_pos, targets)
        self.decoder.return_hidden_states = not self.training
        self.decoder.return_aux_outputs = True
        if not self.training and len(samples) > 1 and FAN_OUT:
            return self._forward_eval_fanned(samples, targets, sp_features_3d, sp_pos, sp_pos_wo_elastic, queries, queries_pos, scene_range)
        outputs = self.forward_decoder(sp_features_3d, sp_pos, sp_pos_wo_elastic, queries, queries_pos, targets, scene_range)
        cap = _trace.active()
        if cap is not None:                                      # per-call, per-thread (segdino3d_amd/_trace.py)
            cap.outputs, cap.sp_feats, cap.sp_pos = outputs, sp_features_3d, sp_pos
        if self.training:                                        # {"seg_loss", "inst_loss"}, gradients attached (:346)
            return self.criterion(outputs, targets)
        # the reference evaluates one scene per forward (:335-338, bs = 1); here a list of B scenes is B independent evaluations
        # whose backbone ran as one block-diagonal tensor: every target receives its own `pred_pts_seg`
        for b in range(len(targets)):
            pred = self.predict_by_feat(samples, outputs, targets[b]["extra_features"]["super_point_masks"], b)
            targets[b].pred_pts_seg = pred[0]
        return targets

    def _forward_eval_fanned(self, samples, targets, sp_feats, sp_pos, sp_pos_wo, queries, queries_pos, scene_range):
        """Decoder + post-processing of the B scenes of a batched evaluation forward, each scene on its own side stream: per
        scene these are ~300 dependent launches of a few microseconds that fill a handful of CUs, so B of them side by side
        take about as long as one.  Issue order: all decoders, then all threshold-independent post-processing (ending in each
        scene's host read), then the data-dependent selections as their reads arrive.  Same kernels, same per-scene launch
        sequence as the single-scene forward: results are bit-identical to it."""
        B = len(samples)
        main = torch.cuda.current_stream()
        sides = ops.side_streams(B, samples[0].device)
        outs, coms, reads, sems = [None] * B, [None] * B, [None] * B, [None] * B
        pick = lambda lst, b: None if lst is None else [lst[b]]      # noqa: E731
        for b, st in enumerate(sides):
            st.wait_stream(main)
            with ops.use_stream(st):
                outs[b] = self.forward_decoder([sp_feats[b]], pick(sp_pos, b), pick(sp_pos_wo, b), [queries[b]], pick(queries_pos, b),
                                               [targets[b]], pick(scene_range, b))
        ops.baton_yield()
        for b, st in enumerate(sides):
            with ops.use_stream(st):
                coms[b] = self._instances_common([samples[b]], outs[b], targets[b]["extra_features"]["super_point_masks"])
                reads[b] = self._select_begin(coms[b])
                sems[b] = self._semantic(outs[b], targets[b]["extra_features"]["super_point_masks"])
        for b, st in enumerate(sides):
            with ops.use_stream(st):
                pred = self._predict_finish([samples[b]], outs[b], targets[b]["extra_features"]["super_point_masks"], coms[b], reads[b],
                                            sem_pre=sems[b])
            targets[b].pred_pts_seg = pred[0]
            main.wait_stream(st)
        cap = _trace.active()
        if cap is not None:                                      # the per-scene dicts merged back into the decoder's list-of-scenes form
            merged = {k: [o[k][0] for o in outs] for k in outs[0] if k != "aux_outputs"}
            if "aux_outputs" in outs[0]:
                merged["aux_outputs"] = [{k: (None if a[0][k] is None else [x[k][0] for x in a]) for k in a[0]}
                                         for a in zip(*[o["aux_outputs"] for o in outs])]
            cap.outputs, cap.sp_feats, cap.sp_pos = merged, sp_feats, sp_pos
        return targets

    # ---- post-processing -------------------------------------------------------------------------------------
    def _instances_common(self, samples, out, superpoints, b=0):
        """Everything of predict_by_feat_instance (:406-486) that does not depend on the score threshold;
        the reference runs it twice (inst_score_thr and pan_score_thr), here it runs once.  `b`: scene of the batch."""
        cfg = self.test_cfg
        cls = out["cls_preds"][b]
        logits = out["masks"][b]
        pts = samples[b]
        C = self.num_classes
        Q, S = cls.shape[0], logits.shape[1]
        k = int(_cfg_get(cfg, "topk_insts"))
        if Q * C < k:
            raise ValueError(f"topk_insts={k} needs at least {k} (query, class) pairs, got {Q * C}")
        flat, _ = ops.class_scores(cls, C)
        if out.get("scores") is not None and out["scores"][b] is not None:        # objectness head: scores *= out['scores'][0] (:428-429)
            flat = (flat.view(Q, C) * out["scores"][b].reshape(Q, 1)).reshape(-1).contiguous()
        if TOPK_SELECT and flat.numel() <= ops.TOPK_SELECT_MAX_N and k <= 1024:
            order0 = ops.topk_desc(flat, k)                              # top-k (query, class) pairs (:434): radix select, one launch
        else:
            order0 = _sorted_desc(flat)[:k].contiguous()
        top_scores = ops.take_f32(flat, order0)
        labels, qidx, scores = ops.mask_scores(logits, S, order0, top_scores, C, bool(_cfg_get(cfg, "obj_normalization", None)))
        S_pad = (S + 31) // 32 * 32
        centers = out["centers"][b] if "centers" in out else None
        sizes = out["sizes"][b] if "sizes" in out else None
        have_boxes = centers is not None and sizes is not None
        boxes = None
        if _cfg_get(cfg, "nms", None):
            order1 = _sorted_desc(scores)                                # mask_matrix_nms first sort (:71)
            sig, area = ops.gather_sigmoid(logits, S, qidx, order1, S_pad)
            labels1, scores1 = ops.take_pair(order1, labels, scores)
            inter = ops.gather_gemm(sig, sig)                            # torch.mm(masks, masks^T) (:87)
            scores2 = ops.nms_decay(inter, area, labels1, scores1, kernel=_cfg_get(cfg, "matrix_nms_kernel"))
            order2 = _sorted_desc(scores2)                               # final sort (:133)
            # scores2[order2], labels1[order2], sort_inds_record = order1[order2] (:139) and the kept queries' boxes: one launch
            final_scores, final_labels, record, boxes = ops.nms_finish(
                order2, scores2, labels1, order1, qidx, centers.contiguous() if have_boxes else None, sizes.contiguous() if have_boxes else None)
            src_row = order2.contiguous()
        else:
            # the reference leaves sort_inds_record undefined here; identity is the sane reading
            ident = torch.arange(k, dtype=torch.int32, device=cls.device)
            sig, area = ops.gather_sigmoid(logits, S, qidx, ident, S_pad)
            final_scores, final_labels, record, src_row = scores, labels, ident.long(), ident
            if have_boxes:
                q_rec = qidx.long()[record]
                boxes = torch.cat([centers[q_rec], sizes[q_rec]], dim=-1).contiguous()
        # the thresholded rows as a bit table + the point count of every candidate; the [n, N] masks are expanded in `_predict_finish`,
        # for the rows that survive the thresholds only (`ops.MaskBits`: the whole [600, N] byte table was 90 MB written per scene)
        bits = ops.MaskBits(sig, src_row, superpoints.contiguous(), pts, float(_cfg_get(cfg, "sp_score_thr")),
                            boxes if self.filter_outofbox_points_eval else None)
        return dict(scores=final_scores, labels=final_labels, bits=bits, count=bits.count, boxes=boxes, topk_idx=qidx.long(), n_points=pts.shape[0])

    def _select(self, common, thresholds=None):
        """The data-dependent selections of predict_by_feat_instance (:470-476) for the instance and the panoptic score threshold: made on
        the device (`sd3d_select_instances`); the host reads four counts to size the outputs.  (Before: the k scores and point counts went
        to the host, numpy built the row lists and one copy brought them back - with the GPU idle in between; boolean indexing on the
        device would cost a synchronising nonzero per selection.)
        Returns ([(keep int32 rows, score_mask bool[k], npoint_mask bool[n_scored]), (pkeep, None, None)], (union rows int32,
        [positions of keep / pkeep in the union], [keep == union, pkeep == union])), tensors on the device."""
        return self._select_finish(common, thresholds, self._select_begin(common))

    def _select_begin(self, common):
        """Launch the selection and start the host read of its four counts (asynchronous copy + event on the current stream)."""
        cfg = self.test_cfg
        bufs, counts = ops.select_instances(common["scores"], common["count"], float(_cfg_get(cfg, "inst_score_thr")),
                                            float(_cfg_get(cfg, "pan_score_thr")), int(_cfg_get(cfg, "npoint_thr")))
        return ops.HostRead(counts), bufs

    def _select_finish(self, common, thresholds, pending):
        read, (ints, bytes_) = pending
        n_keep, n_pkeep, n_union, n_scored = (int(v) for v in read.wait().tolist())
        k = common["scores"].shape[0]
        keep, pkeep, union = ints[0, :n_keep], ints[1, :n_pkeep], ints[2, :n_union]
        keep_u, pkeep_u = ints[3, :n_keep], ints[4, :n_pkeep]
        score_mask, npoint_mask = bytes_[0, :k].view(torch.bool), bytes_[1, :n_scored].view(torch.bool)
        return [(keep, score_mask, npoint_mask), (pkeep, None, None)], (union, [keep_u, pkeep_u], [n_keep == n_union, n_pkeep == n_union])

    @ops.bound_stream
    def predict_by_feat(self, samples, out, superpoints, b=0):
        cfg = self.test_cfg
        ops.baton_yield()
        com = self._instances_common(samples, out, superpoints, b)
        read = self._select_begin(com)
        return self._predict_finish(samples, out, superpoints, com, read, b, sem_pre=self._semantic(out, superpoints, b))

    def _semantic(self, out, superpoints, b=0):
        """Semantic labels per point (:488-507) and the stuff-class labels the panoptic map starts from (:509-520): four launches that
        depend on no selection - issued BEHIND the read of the selection counts, they run while the counts travel to the host."""
        cfg = self.test_cfg
        sem = out["sem_preds"][b]
        n_sem = sem.shape[1] - 1
        use_index = self.query_num == -1
        sem_res = ops.gather_i64(ops.row_argmax(sem, ncols=n_sem), superpoints, use_index)
        stuff = list(_cfg_get(cfg, "stuff_classes"))
        cols = self._stuff_cols.get(str(sem.device)) if tuple(stuff) == self._stuff_cols.get("classes") else None
        if cols is None:                                         # one H2D copy per device, not one per scene
            if tuple(stuff) != self._stuff_cols.get("classes"):
                self._stuff_cols = {"classes": tuple(stuff)}
            cols = self._stuff_cols[str(sem.device)] = torch.tensor(stuff, dtype=torch.int32, device=sem.device)
        sem_stuff = ops.gather_i64(ops.row_argmax(sem, cols=cols), superpoints, use_index)
        return sem_res, sem_stuff, stuff

    def _predict_finish(self, samples, out, superpoints, com, read, b=0, sem_pre=None):
        """The part of predict_by_feat behind the host read of the scores (data-dependent selections, panoptic; the semantic maps too
        unless the caller launched them while the read travelled: `sem_pre` = `_semantic(...)`)."""
        cfg = self.test_cfg
        # the data-dependent selections need the scores on the host: one polled read (no host thread sits inside a blocking HIP
        # call while other scenes are being issued)
        ((keep, score_mask, npoint_mask), (pkeep, _, _)), (rows, (keep_u, pkeep_u), (keep_all, _)) = self._select_finish(
            com, (float(_cfg_get(cfg, "inst_score_thr")), float(_cfg_get(cfg, "pan_score_thr"))), read)
        masks_u = com["bits"].rows(rows)                           # [rows kept by either threshold, N] bytes: the only point masks made
        inst_masks = None
        if not self.to_host:                                       # (the host path packs the kept rows directly)
            inst_masks = (masks_u if keep_all else masks_u[keep_u.long()]).view(torch.bool)
        inst_labels, inst_scores, inst_boxes = ops.take_instances(keep.contiguous(), com["labels"], com["scores"], com["boxes"])
        sem_res, sem_stuff, stuff = self._semantic(out, superpoints, b) if sem_pre is None else sem_pre
        if pkeep.numel() == 0:
            pan_sem, pan_inst = sem_stuff, sem_stuff
        else:
            pan_sem, pan_inst = ops.panoptic(masks_u, pkeep_u.contiguous(), com["labels"][pkeep].int().contiguous(),
                                             len(stuff), int(_cfg_get(cfg, "npoint_thr")), sem_stuff)
        sort_and_mask = (com["topk_idx"], score_mask, npoint_mask)
        if not self.to_host:
            return [PointData(pts_semantic_mask=[sem_res, pan_sem], pts_instance_mask=[inst_masks, pan_inst],
                              instance_labels=inst_labels, instance_scores=inst_scores, sort_and_mask=sort_and_mask,
                              instance_boxes=inst_boxes)]
        n = inst_scores.shape[0]
        # D2H of the post-processed outputs.  The [n, N] instance masks leave the device BIT-PACKED (sd3d_pack_mask_rows straight from
        # the selected rows of the byte table: 11 MB instead of 90), everything goes through one reused pinned staging buffer with ONE
        # polled wait, and the caller receives pageable arrays (nothing page-locked outlives the forward).  `to_host=True`: the masks are
        # expanded to the [n, N] bool array the reference's evaluator reads (evaluator_3d.py:178) by the C library with the GIL
        # released; `to_host="packed"`: they stay packed (`PackedMasks`: 8 x fewer host bytes, `np.asarray()` / `.unpack()` on demand).
        packed = ops.pack_mask_rows(masks_u, keep_u.contiguous())
        dev = [sem_res, pan_sem, packed, pan_inst, inst_labels, inst_scores] + ([inst_boxes] if inst_boxes is not None else [])
        arr = ops.to_host_arrays(dev)
        N = com["n_points"]
        masks_host = PackedMasks(arr[2], N) if self.to_host == "packed" else ops.unpack_bits_host(arr[2], N)
        return [PointData(
            pts_semantic_mask=[arr[0], arr[1]], pts_instance_mask=[masks_host, arr[3]], instance_labels=arr[4], instance_scores=arr[5],
            sort_and_mask=sort_and_mask, instance_boxes=arr[6] if inst_boxes is not None else np.zeros((n, 6)))]


class PackedMasks:
    """[n, N] boolean instance masks held as bits on the host (`Baseline3D.to_host = "packed"`): `bits` uint8 [n, ceil(N / 8)],
    bit j of byte b = point 8 b + j (numpy bitorder "little").  Behaves like the bool array where consumers only convert it:
    `np.asarray(m)`, `m.unpack()`, `m[i]` (one row or a row selection, unpacked), `len(m)`, `m.shape`."""
    __slots__ = ("bits", "shape")
    dtype = np.dtype(np.bool_)

    def __init__(self, bits: np.ndarray, n_points: int):
        self.bits = bits
        self.shape = (bits.shape[0], int(n_points))

    @property
    def nbytes(self):
        return self.bits.nbytes

    def __len__(self):
        return self.shape[0]

    def unpack(self) -> np.ndarray:
        return ops.unpack_bits_host(self.bits, self.shape[1])

    def __array__(self, dtype=None, copy=None):
        a = self.unpack()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __getitem__(self, idx):
        rows = self.bits[idx]
        if rows.ndim == 1:
            return ops.unpack_bits_host(rows[None], self.shape[1])[0]
        return ops.unpack_bits_host(rows, self.shape[1])
