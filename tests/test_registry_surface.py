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
