"""The reference's model configs as plain dicts (no mmengine needed).

Values restate `configs/models/base_3d.py:1-79` merged with the prototype overrides
`configs/prototypes/SegDINO3D_ScanNet200.py:10-31` and `SegDINO3D_ScanNetv2.py:13-48`; a reference
config file loaded with mmengine yields the same `cfg.model` dict, which `build_architecture` accepts
unchanged (SURVEY.md 8(b) "Config surface")."""
from __future__ import annotations

import copy


def _base(num_instance_classes, num_semantic_classes):
    return dict(
        type="Baseline3D",
        num_classes=num_instance_classes,
        pointcloud_backbone_cfg=dict(
            type="Res16UNet34C", in_channels=256 + 3, out_channels=96,
            config=dict(dilations=[1, 1, 1, 1], conv1_kernel_size=5, bn_momentum=0.02)),
        decoder_cfg=dict(
            type="ScanNetQueryDecoder", add_dinox_query_ca=True, add_dinox_query_ca_mask=True,
            dinox_query_ca_mask_threshold=0.2, num_layers=6, num_instance_queries=0, num_semantic_queries=0,
            num_instance_classes=num_instance_classes, num_semantic_classes=num_semantic_classes, num_semantic_linears=1,
            in_channels=96, d_model=256, num_heads=8, hidden_dim=1024, dropout=0.0, activation_fn="gelu", iter_pred=True,
            attn_mask=True, fix_attention=True, objectness_flag=False),
        text_encoder_cfg=None,
        criterion_cfg=dict(
            type="ScanNetUnifiedCriterion", num_semantic_classes=num_semantic_classes,
            sem_criterion=dict(type="ScanNetSemanticCriterion", ignore_index=num_semantic_classes, loss_weight=0.5),
            inst_criterion=dict(
                type="InstanceCriterion",
                matcher=dict(type="SparseMatcher", costs=[dict(type="QueryClassificationCost", weight=0.5),
                                                          dict(type="MaskBCECost", weight=1.0),
                                                          dict(type="MaskDiceCost", weight=1.0),
                                                          dict(type="CenterL1Cost", weight=0.5),
                                                          dict(type="SizeL1Cost", weight=0.5)], topk=1),
                loss_weight=[0.5, 1.0, 1.0, 0.5, 0.5, 0.5], num_classes=num_instance_classes, non_object_weight=0.1,
                fix_dice_loss_weight=True, iter_matcher=True, fix_mean_loss=True)),
        query_thr=0.5,
        test_cfg=dict(topk_insts=600, inst_score_thr=0.0, pan_score_thr=0.5, npoint_thr=100, obj_normalization=True,
                      sp_score_thr=0.4, nms=True, matrix_nms_kernel="linear", stuff_classes=[0, 1]))


def scannet200_model_cfg(query_num=-1, voxel_size=0.02):
    """configs/prototypes/SegDINO3D_ScanNet200.py"""
    m = _base(198, 200)
    m["pointcloud_backbone_cfg"].update(voxel_size=voxel_size, mode_fuse_2d_feat="early_fusion", add_positional_embedding=True)
    m["decoder_cfg"].update(add_box_size_pred=True, add_positional_embedding=True, pos_type="sine", temperature=20,
                            box_modulate_ca=True, normalize_box_prediction=True)
    m.update(add_positional_embedding=True, mode_3d_center="median", filter_outofbox_points_eval=True, query_num=query_num)
    return copy.deepcopy(m)


def scannetv2_model_cfg(query_num=-1, voxel_size=0.02):
    """configs/prototypes/SegDINO3D_ScanNetv2.py"""
    m = _base(18, 20)
    m["pointcloud_backbone_cfg"] = dict(type="SpConvUNet", num_planes=[32 * (i + 1) for i in range(5)], return_blocks=True,
                                        voxel_size=voxel_size, mode_fuse_2d_feat="early_fusion", add_positional_embedding=True)
    m["decoder_cfg"].update(in_channels=32, add_box_size_pred=True, add_positional_embedding=True, pos_type="sine",
                            temperature=20, box_modulate_ca=True)
    m.update(add_positional_embedding=True, mode_3d_center="median", filter_outofbox_points_eval=True, query_num=query_num)
    return copy.deepcopy(m)


def baseline_scannet200_model_cfg(query_num=-1, voxel_size=0.02):
    """configs/prototypes/Baseline_ScanNet200.py: rgb-only backbone, no 2D-query attention, no positional embedding."""
    m = _base(198, 200)
    m["pointcloud_backbone_cfg"].update(voxel_size=voxel_size, mode_fuse_2d_feat="only_rgb")
    m["decoder_cfg"].update(add_dinox_query_ca=False)
    m["criterion_cfg"]["inst_criterion"]["matcher"]["costs"] = m["criterion_cfg"]["inst_criterion"]["matcher"]["costs"][:3]
    m["criterion_cfg"]["inst_criterion"]["loss_weight"] = [0.5, 1.0, 1.0, 0.5]
    m.update(query_num=query_num)
    return copy.deepcopy(m)
