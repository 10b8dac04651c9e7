"""Diagnostic: the architecture goldens through the fused (row-chain) and the op-by-op decoder - label / score agreement with the
imported reference, and how close the reference's own neighbouring scores are where rows swap."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np
import torch
import test_gpu_decoder as T
import segdino3d_amd as seg
from segdino3d_amd import decoder as D
from segdino3d_amd.gtypes import GD3DTarget

d = torch.device("cuda:0")
if seg.BACKBONES.get("_StoredBackbone") is None:
    seg.BACKBONES.register_module(module=T._StoredBackbone)
for name, query_num, box in [("arch_qall", -1, True), ("arch_q40", 40, True), ("arch_qall_nobox", -1, False), ("arch_qall_widebox", -1, True)]:
    g = T.load(name)
    for fused in (True, False):
        D.FUSED_DECODER = fused
        model = seg.build_architecture(dict(
            type="Baseline3D", num_classes=198, pointcloud_backbone_cfg=dict(type="_StoredBackbone"),
            decoder_cfg=dict(type="ScanNetQueryDecoder", **T.DEC_KW), criterion_cfg=None, query_thr=0.5, test_cfg=T.TEST_CFG,
            add_positional_embedding=True, mode_3d_center="median", query_num=query_num, filter_outofbox_points_eval=box)).eval()
        sd = T.decoder_state_dict()
        for i in range(6):
            sd[f"decoder.bbox_size_embed.{i}.layers.2.bias"] = sd[f"decoder.bbox_size_embed.{i}.layers.2.bias"] + float(g["size_bias"])
        model.decoder.load_state_dict({k[len("decoder."):]: v for k, v in sd.items()})
        model.to(d)
        model.backbone.f, model.backbone.p = g["sp_feat"].to(d), g["sp_pos"].to(d)
        tgt = GD3DTarget(masks=g["gt_masks"], extra_features=dict(super_point_masks=g["superpoints"].long(),
                         query2d_feats=g["q2d_feat"], query2d_pos=g["q2d_pos"])).to(d)
        with torch.no_grad(), seg.capture() as cap:
            res = model([g["points"].to(d)], [tgt])
        pd = res[0].pred_pts_seg
        rs, gs = g["inst_scores"].numpy(), pd.instance_scores
        rl, gl = g["inst_labels"].numpy(), pd.instance_labels
        mism = np.flatnonzero(rl != gl)
        gaps = np.abs(np.diff(rs))
        near = [min(gaps[max(i - 1, 0)], gaps[min(i, len(gaps) - 1)]) for i in mism]
        print(f"{name} fused={fused}: n={len(rs)} score max rel err {np.max(np.abs(gs - rs) / np.maximum(np.abs(rs), 1e-9)):.2e}, "
              f"label mismatches {len(mism)}, reference score gap to a neighbour at those rows: "
              f"median {np.median(near) if len(near) else 0:.2e} max {np.max(near) if len(near) else 0:.2e}; "
              f"median gap overall {np.median(gaps):.2e}; cls logits max {cap.outputs['cls_preds'][0].abs().max().item():.3f}")
        if fused:
            fo = cap.outputs
        else:
            for k in ("cls_preds", "masks", "sem_preds", "centers", "sizes"):
                a, b = fo[k][0], cap.outputs[k][0]
                print(f"    {k}: fused vs plain max abs diff {(a - b).abs().max().item():.3e} (max |x| {b.abs().max().item():.3f})")
