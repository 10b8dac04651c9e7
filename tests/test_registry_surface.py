"""Host-side surface of the drop-in (SURVEY.md 8(b) "Registry API"), no GPU needed: the reference's registry names build the
AMD classes from the shipped config dicts, the state_dict keys / shapes are the reference's, training and evaluation
entry points exist, and a CPU tensor is refused instead of silently computed on the host."""
import pytest
import torch


def test_registries_build_the_reference_configs():
    import segdino3d_amd as seg
    from segdino3d_amd import augment, criterion  # noqa: F401 - register TRANSFORMS / LOSSES entries
    from segdino3d_amd.builder import BACKBONES, DECODERS, LOSSES, TRANSFORMS
    from segdino3d_amd.configs import scannet200_model_cfg
    for reg, name in ((BACKBONES, "Res16UNet34C"), (BACKBONES, "SpConvUNet"), (DECODERS, "ScanNetQueryDecoder"),
                      (LOSSES, "ScanNetUnifiedCriterion"), (TRANSFORMS, "Scannet200Transforms")):
        assert reg.get(name) is not None, name
    model = seg.build_architecture(scannet200_model_cfg(query_num=200))
    assert type(model).__name__ == "Baseline3D" and type(model.backbone).__name__ == "Res16UNet34C"
    assert type(model.criterion).__name__ == "ScanNetUnifiedCriterion"
    assert model.criterion.inst_criterion.matcher_kind == "SparseMatcher" and model.criterion.inst_criterion.topk == 1
    assert model.criterion.inst_criterion.cost_weights == [0.5, 1.0, 1.0, 0.5, 0.5]
    tf = TRANSFORMS.get("Scannet200Transforms")("train", voxel_size=0.02)
    assert [type(t).__name__ for t in tf.transforms] == ["CustomRandomFlip3D", "CustomGlobalRotScaleTrans", "NormalizePointsColor",
                                                         "ElasticTransfrom", "ToTensor"]


def test_state_dict_keys_are_the_reference_checkpoint_keys():
    import segdino3d_amd as seg
    from oracle import sparse_ref as R
    from segdino3d_amd.configs import scannet200_model_cfg
    model = seg.build_architecture(scannet200_model_cfg(query_num=200))
    sd = model.state_dict()
    want = R.mink_state_dict_shapes(in_channels=259, conv1_kernel_size=5)
    got = {k[len("backbone."):]: tuple(v.shape) for k, v in sd.items() if k.startswith("backbone.")}
    assert got == {k: tuple(v) for k, v in want.items()}
    assert any(k.startswith("decoder.cross_attn_layers.0.") for k in sd) and "decoder.out_cls.0.weight" in sd


def test_cpu_inputs_are_refused():
    import segdino3d_amd as seg
    from segdino3d_amd.configs import scannet200_model_cfg
    from segdino3d_amd.synth import make_scene
    model = seg.build_architecture(scannet200_model_cfg(query_num=200)).eval()
    pts, tgt = make_scene(0, 2000, 40, 5)
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            model([pts], [tgt])


def test_training_plan_records_the_real_network_on_the_cpu():
    """`train_plan.TrainRecorder` against Res16UNet34C's own network definition (no kernel runs): 62 {convolution -> BatchNorm -> ReLU} layers
    (55 sparse + seven 1x1), skip concatenations resolved into column slices of shared buffers, every input gradient but the stem's, and a
    store / add decision for each gradient write that a reverse walk can check."""
    import torch.nn as nn
    from segdino3d_amd import train_ops, train_plan
    from segdino3d_amd.backbone_mink import MinkBN, MinkConv, Res16UNet34C
    m = Res16UNet34C(in_channels=259, out_channels=96, config=dict(dilations=[1, 1, 1, 1], conv1_kernel_size=5, bn_momentum=0.02),
                     voxel_size=0.02, mode_fuse_2d_feat="early_fusion", add_positional_embedding=True)
    pk = {}
    for n, mod in m.named_modules():
        if isinstance(mod, MinkConv):
            pk[n] = train_ops.TrainWeight(mod.kernel, None)
        elif isinstance(mod, MinkBN):
            pk[n] = mod.bn
    rec = train_plan.TrainRecorder(288)
    plan = rec.finish(m._network(rec, pk, rec.input))
    L = plan.layers
    assert plan.n == 62 and len(plan.params) == 62 and all(isinstance(b, nn.BatchNorm1d) for b in plan.bns)
    assert sum(1 for k in plan.rec_keys if k[0] == "id") == 7                      # BasicBlock.downsample: the seven 1x1 convolutions
    assert int(L["need_dx"].sum()) == 61 and int(L["need_dx"][0]) == 0             # only the stem reads the network input
    assert int(L["Cin"][0]) == 288 and int(L["Cout"][-1]) == 96 and plan.out_ch == 96
    # the four skip concatenations: [transposed-convolution output | encoder tensor] side by side in ONE buffer
    cols = {}
    for i in range(plan.n):
        cols.setdefault(int(L["dst"][i]), set()).add(int(L["dst_col"][i]))
    merged = {b for b, c in cols.items() if len(c) > 1}                          # buffers two producers write into
    cat_layers = [i for i in range(plan.n) if int(L["src"][i]) in merged and int(L["src_col"][i]) == 0 and int(L["Cin"][i]) == int(plan.buf_ch[int(L["src"][i])])]
    assert len(merged) == 4
    assert len(cat_layers) == 8                                                    # conv1 + downsample of the first block of block5 .. block8
    assert sorted({int(plan.buf_ch[int(L["src"][i])]) for i in cat_layers}) == [128, 192, 384]
    # every encoder tensor that feeds a concatenation is written into its slice by its producer
    assert {(int(L["dst_col"][i]), int(L["Cout"][i])) for i in range(plan.n) if int(L["dst_col"][i]) > 0} == {(96, 32), (128, 64), (256, 128)}
    # reverse walk: a slice's first gradient write stores, the later ones add; a residual's gradient is always a first write
    seen = {plan.out_id: [(0, plan.out_ch)]}
    for i in range(plan.n - 1, -1, -1):
        if int(L["res"][i]) >= 0:
            b, c, w = int(L["res"][i]), int(L["res_col"][i]), int(L["Cout"][i])
            assert not any(a < c + w and e > c for a, e in seen.get(b, []))
            seen.setdefault(b, []).append((c, c + w))
        if int(L["need_dx"][i]):
            b, c, w = int(L["src"][i]), int(L["src_col"][i]), int(L["Cin"][i])
            covered = any(a <= c and e >= c + w for a, e in seen.get(b, [])) or \
                sum(min(e, c + w) - max(a, c) for a, e in seen.get(b, []) if a < c + w and e > c) >= w
            assert bool(L["dx_accum"][i]) == covered, i
            if not covered:
                seen.setdefault(b, []).append((c, c + w))
    assert int(L["dx_accum"].sum()) == 27
