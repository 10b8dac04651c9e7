#!/usr/bin/env python3
"""Generate golden vectors by importing the *reference* (read-only, /root/reference).

Run ONLY in the build container (the reference does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference needs mmengine / mmdet3d / MinkowskiEngine / spconv / torch_scatter / torchvision /
plyfile / trimesh, none of which are installed here.  The pure-torch parts of the hot path
(decoder, attention, positional encoding, heads, query selection, post-processing, matrix-NMS) do
not actually *use* them, so throw-away stand-in modules (written here, below) are injected into
`sys.modules` before the import.  The sparse backbone cannot be imported (its arithmetic lives in
third-party CUDA libraries), so the architecture-level fixture uses a stand-in backbone that
returns stored superpoint features.

Only inputs and outputs are written (as .npz); weights are regenerated from key names by
`_det.det_param`, so nothing of the reference travels.
"""
import enum
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
REFERENCE = os.environ.get("SEGDINO3D_REFERENCE", "/root/reference")

from _det import assign_det_weights, det_randn  # noqa: E402


# ----------------------------------------------------------------------------------------------
# stand-ins for the third-party modules the reference imports at module import time
# ----------------------------------------------------------------------------------------------
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Registry:
    def __init__(self, name):
        self.name, self.table = name, {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.table[name or cls.__name__] = cls
            return cls
        return deco(module) if module is not None else deco

    def get(self, key):
        return self.table.get(key)


def _build_from_cfg(cfg, registry, default_args=None):
    if cfg is None:
        return None
    args = dict(cfg)
    cls = registry.get(args.pop("type"))
    return cls(**args)


class _PointData:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def items(self):
        return self.__dict__.items()


def _scatter_mean(src, index, dim=0):
    n = int(index.max()) + 1
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    out.index_add_(0, index, src)
    cnt = torch.bincount(index, minlength=n).clamp(min=1).to(src.dtype)
    return out / cnt.view(-1, *([1] * (src.dim() - 1)))


def install_stand_ins():
    _mod("mmengine", Registry=_Registry, build_from_cfg=_build_from_cfg)
    _mod("mmengine.model", BaseModule=nn.Module)
    _mod("mmdet3d")
    _mod("mmdet3d.structures", PointData=_PointData)
    _mod("mmdet3d.structures.bbox_3d")
    _mod("mmdet3d.structures.bbox_3d.utils", rotation_3d_in_axis=None)
    _mod("mmdet3d.datasets")
    _mod("mmdet3d.datasets.transforms", GlobalRotScaleTrans=object)
    _mod("mmdet")
    _mod("mmdet.datasets")
    _mod("mmdet.datasets.transforms", RandomFlip=object)
    _mod("plyfile", PlyData=object, PlyElement=object)
    _mod("trimesh")
    tv = _mod("torchvision")
    tv.transforms = _mod("torchvision.transforms")
    tv.transforms.functional = _mod("torchvision.transforms.functional")
    _mod("torch_scatter", scatter_mean=_scatter_mean)

    class RegionType(enum.Enum):
        HYPER_CUBE = 0
        HYPER_CROSS = 1
        CUSTOM = 2

    me = _mod("MinkowskiEngine", RegionType=RegionType, MinkowskiReLU=nn.ReLU)
    me.MinkowskiOps = _mod("MinkowskiEngine.MinkowskiOps")
    sp = _mod("spconv")
    sp.pytorch = _mod("spconv.pytorch", SparseSequential=nn.Sequential)
    sp.pytorch.modules = _mod("spconv.pytorch.modules", SparseModule=nn.Module)


# ----------------------------------------------------------------------------------------------
def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


DECODER_KW_SCANNET200 = dict(
    add_dinox_query_ca=True, add_dinox_query_ca_mask=True, dinox_query_ca_mask_threshold=0.2,
    num_layers=6, num_instance_queries=0, num_semantic_queries=0, num_instance_classes=198,
    num_semantic_classes=200, num_semantic_linears=1, in_channels=96, d_model=256, num_heads=8,
    hidden_dim=1024, dropout=0.0, activation_fn="gelu", iter_pred=True, attn_mask=True,
    fix_attention=True, objectness_flag=False, add_box_size_pred=True,
    add_positional_embedding=True, pos_type="sine", temperature=20, box_modulate_ca=True,
    normalize_box_prediction=True)

# ScanNetv2 prototype: 18/20 classes, in_channels 32, additive box-size refinement
DECODER_KW_SCANNETV2 = dict(DECODER_KW_SCANNET200, num_instance_classes=18, num_semantic_classes=20,
                            in_channels=32, normalize_box_prediction=False)

# Baseline_ScanNet200 prototype: no positional embedding, no 2D-query attention (configs/prototypes/Baseline_ScanNet200.py)
DECODER_KW_PLAIN = dict(
    add_dinox_query_ca=False, add_dinox_query_ca_mask=True, dinox_query_ca_mask_threshold=0.2,
    num_layers=6, num_instance_queries=0, num_semantic_queries=0, num_instance_classes=198,
    num_semantic_classes=200, num_semantic_linears=1, in_channels=96, d_model=256, num_heads=8,
    hidden_dim=1024, dropout=0.0, activation_fn="gelu", iter_pred=True, attn_mask=True,
    fix_attention=True, objectness_flag=False)

TEST_CFG = dict(topk_insts=600, inst_score_thr=0.0, pan_score_thr=0.5, npoint_thr=100,
                obj_normalization=True, sp_score_thr=0.4, nms=True, matrix_nms_kernel="linear",
                stuff_classes=[0, 1])


class _AttrDict(dict):
    __getattr__ = dict.__getitem__


def decoder_inputs(tag, S, M, C):
    """Superpoint features / positions in an 8 x 6 x 3 m room, deterministic by tag."""
    room = torch.tensor([8.0, 6.0, 3.0])
    pos = det_randn(tag + ".pos", (S, 3)).sigmoid() * room
    pos = torch.floor(pos / 0.02) * 0.02
    pos_wo = pos.clone()
    x = det_randn(tag + ".x", (S, C))
    q2d_pos = pos[torch.arange(M) % S] + det_randn(tag + ".q2dpos", (M, 3), 0.1)
    q2d_feat = det_randn(tag + ".q2dfeat", (M, 256))
    lo = pos.min(0)[0] - 0.05
    hi = pos.max(0)[0] + 0.07
    return x, pos, pos_wo, q2d_feat, q2d_pos, lo, hi


def golden_pe(utils_mod):
    pe = utils_mod.PositionEmbeddingCoordsSine(temperature=20, normalize=True, pos_type="sine", d_pos=256)
    xyz = det_randn("pe.xyz", (1, 17, 3)).sigmoid() * torch.tensor([8.0, 6.0, 3.0])
    lo = xyz.min(1)[0] - 0.1
    hi = xyz.max(1)[0] + 0.2
    mod = 0.5 + det_randn("pe.mod", (1, 17, 3)).sigmoid()
    plain = pe(xyz, input_range=(lo, hi))
    modded = pe(xyz, input_range=(lo, hi), modulated=mod)
    save("pe_sine", xyz=xyz, lo=lo, hi=hi, modulated=mod, out_plain=plain, out_modulated=modded)


def golden_decoder(dec_mod, name, kw, S, M, query_subset=None):
    torch.manual_seed(0)
    dec = dec_mod.ScanNetQueryDecoder(**kw).eval()
    assign_det_weights(dec, "decoder.")
    if not kw["normalize_box_prediction"]:
        # additive size refinement (:751) starts at 0.5; keep the synthetic size head small so that the
        # sizes stay away from 0, otherwise the modulation sigmoid(.)/size (:661) makes the fixture
        # ill-conditioned (a 1e-6 input perturbation moved mask logits by 0.2 with unscaled weights)
        with torch.no_grad():
            for emb in dec.bbox_size_embed:
                emb.layers[-1].weight.mul_(0.05)
                emb.layers[-1].bias.mul_(0.05)
    x, pos, pos_wo, q2d_feat, q2d_pos, lo, hi = decoder_inputs(name, S, M, kw["in_channels"])
    if query_subset is None:
        q, qpos = x, pos
        ids = torch.arange(S)
    else:
        ids = torch.arange(0, S, S // query_subset)[:query_subset]
        q, qpos = x[ids], pos[ids]
    dec.return_hidden_states = True
    dec.return_aux_outputs = True
    with torch.no_grad():
        out = dec([x], [pos], [pos_wo], [q], [qpos], [q2d_feat], [q2d_pos.clone()], [(lo, hi)])
    arrays = dict(x=x, pos=pos, pos_wo=pos_wo, q2d_feat=q2d_feat, q2d_pos=q2d_pos, lo=lo, hi=hi,
                  query_ids=ids,
                  cls_preds=out["cls_preds"][0], sem_preds=out["sem_preds"][0], masks=out["masks"][0],
                  centers=out["centers"][0], sizes=out["sizes"][0], hidden_states=out["hidden_states"][0])
    for li, aux in enumerate(out["aux_outputs"]):
        arrays[f"aux{li}_cls"] = aux["cls_preds"][0]
        arrays[f"aux{li}_masks"] = aux["masks"][0]
        if aux["centers"][0] is not None:
            arrays[f"aux{li}_centers"] = aux["centers"][0]
            arrays[f"aux{li}_sizes"] = aux["sizes"][0]
    save(name, **arrays)
    return dec


def golden_decoder_plain(dec_mod, name, S):
    torch.manual_seed(0)
    dec = dec_mod.ScanNetQueryDecoder(**DECODER_KW_PLAIN).eval()
    assign_det_weights(dec, "decoder.")
    x, pos, pos_wo, q2d_feat, q2d_pos, lo, hi = decoder_inputs(name, S, 3, 96)
    dec.return_hidden_states = True
    dec.return_aux_outputs = True
    with torch.no_grad():
        out = dec([x], None, None, [x], None, None, None, None)
    arrays = dict(x=x, cls_preds=out["cls_preds"][0], sem_preds=out["sem_preds"][0], masks=out["masks"][0],
                  hidden_states=out["hidden_states"][0])
    for li, aux in enumerate(out["aux_outputs"]):
        arrays[f"aux{li}_cls"] = aux["cls_preds"][0]
        arrays[f"aux{li}_masks"] = aux["masks"][0]
    save(name, **arrays)


def golden_nms(arch_mod):
    n, S = 40, 50
    masks = det_randn("nms.masks", (n, S), 2.0).sigmoid()
    labels = (det_randn("nms.labels", (n,)).abs() * 2).long() % 4
    scores = det_randn("nms.scores", (n,)).sigmoid()
    s, l, m, keep, rec = arch_mod.mask_matrix_nms(masks, labels, scores, kernel="linear")
    save("matrix_nms", masks=masks, labels=labels, scores=scores, out_scores=s, out_labels=l,
         out_masks=m, out_keep=keep, out_record=rec)


def golden_architecture(seg, name, query_num, n_points=4000, S=80, M=12, box_filter=True, size_bias=0.0):
    """Full eval-mode Baseline3D.forward with a stand-in backbone (stored superpoint features).
    size_bias: added to the output bias of every layer's box-size head.  With the plain deterministic weights the predicted
    boxes stay near their 0.5 m initial size and filter_outofbox_points (baseline3d.py:348-371) empties almost every mask;
    a positive bias grows the boxes layer by layer so that the filter keeps a large part of each mask and cuts the rest."""
    from segdino3d_amd.synth import make_scene
    points, target = make_scene(scene_idx=7, n_points=n_points, n_superpoints=S, n_query2d=M)
    sp = target.extra_features["super_point_masks"]
    vox = torch.floor(points[:, :3] / 0.02) * 0.02
    sp_pos = _scatter_mean(vox, sp)
    sp_feat = det_randn(name + ".spfeat", (S, 96))

    class _StoredBackbone(nn.Module):
        voxel_size = 0.02

        def __init__(self, **kw):
            super().__init__()

        def forward_wrapper(self, samples, targets, return_sp_mean_pos=True):
            return [sp_feat.clone()], [sp_pos.clone()], [sp_pos.clone()]

    seg.BACKBONES.table["_StoredBackbone"] = _StoredBackbone

    class _NoLoss:
        def __init__(self, **kw):
            pass

    seg.LOSSES.table["_NoLoss"] = _NoLoss
    torch.manual_seed(0)
    model = seg.build_architecture(dict(
        type="Baseline3D", num_classes=198, pointcloud_backbone_cfg=dict(type="_StoredBackbone"),
        decoder_cfg=dict(type="ScanNetQueryDecoder", **DECODER_KW_SCANNET200),
        criterion_cfg=dict(type="_NoLoss"), query_thr=0.5, test_cfg=_AttrDict(TEST_CFG),
        add_positional_embedding=True, mode_3d_center="median", query_num=query_num,
        filter_outofbox_points_eval=box_filter)).eval()
    assign_det_weights(model.decoder, "decoder.")
    if size_bias:
        with torch.no_grad():
            for head in model.decoder.bbox_size_embed:
                head.layers[-1].bias += size_bias
    # make the synthetic scene produce non-trivial instances: bias mask logits through x_mask so
    # that a fair number of superpoints are "on" for each query (random weights alone give ~50 %).
    if query_num > 0:
        target.sp_inst_sem_masks = torch.zeros(1 + 201, S, dtype=torch.bool)
    with torch.no_grad():
        res = model([points], [target])
    pd = res[0].pred_pts_seg
    inst_masks = pd.pts_instance_mask[0]
    topk_idx, score_mask, npoint_mask = pd.sort_and_mask
    save(name, points=points, superpoints=sp, sp_feat=sp_feat, sp_pos=sp_pos,
         q2d_feat=target.extra_features["query2d_feats"], q2d_pos=target.extra_features["query2d_pos"],
         gt_masks=target.masks,
         inst_masks_packed=np.packbits(inst_masks, axis=1), n_points=np.int64(points.shape[0]),
         inst_labels=pd.instance_labels, inst_scores=pd.instance_scores, inst_boxes=pd.instance_boxes,
         sem_mask=pd.pts_semantic_mask[0], pan_sem=pd.pts_semantic_mask[1], pan_inst=pd.pts_instance_mask[1],
         topk_idx=topk_idx, score_mask=score_mask, npoint_mask=npoint_mask,
         instance_centers=res[0].instance_centers, instance_sizes=res[0].instance_sizes, size_bias=np.float32(size_bias))
    print(f"  {name}: {inst_masks.shape[0]} instances kept, mask points {inst_masks.sum()}")


def main():
    install_stand_ins()
    sys.path.insert(0, REFERENCE)
    import segdino3d as seg  # the reference package
    assert os.path.abspath(seg.__file__).startswith(os.path.abspath(REFERENCE))
    from segdino3d.models.module import utils as utils_mod
    from segdino3d.models.decoder import instance_seg_3d_decoder as dec_mod
    from segdino3d.models.architecture import baseline3d as arch_mod

    golden_pe(utils_mod)
    golden_decoder(dec_mod, "decoder_s64_q64", DECODER_KW_SCANNET200, S=64, M=10)
    golden_decoder(dec_mod, "decoder_s96_q16", DECODER_KW_SCANNET200, S=96, M=7, query_subset=16)
    golden_decoder(dec_mod, "decoder_v2_s48", DECODER_KW_SCANNETV2, S=48, M=5)
    # several 32-key tiles per wave and per workgroup, ragged last tile (500 = 15 * 32 + 20), 41 keys in the 2D-query attention
    golden_decoder(dec_mod, "decoder_s500_q32", DECODER_KW_SCANNET200, S=500, M=40, query_subset=32)
    golden_decoder_plain(dec_mod, "decoder_plain_s40", S=40)
    golden_nms(arch_mod)
    golden_architecture(seg, "arch_qall", query_num=-1)
    golden_architecture(seg, "arch_q40", query_num=40)
    golden_architecture(seg, "arch_qall_nobox", query_num=-1, box_filter=False)
    golden_architecture(seg, "arch_qall_widebox", query_num=-1, size_bias=0.3)
    golden_architecture(seg, "arch_q40_widebox", query_num=40, size_bias=0.3)


if __name__ == "__main__":
    main()
