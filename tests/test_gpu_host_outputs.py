"""`Baseline3D.to_host` modes on the device (VERDICT r3 item 7 / ADVICE r3): False = device tensors, True = pageable
numpy arrays (the reference's `.cpu().numpy()` contract), "packed" = the same with bit-packed instance masks.  All three must hold the
same bits; nothing page-locked may leave the forward."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,N", [(1, 1), (5, 13), (33, 1000), (600, 20003), (7, 4096)])
def test_pack_mask_rows_matches_numpy(n, N):
    from segdino3d_amd import ops
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(N)
    a = (torch.rand(n, N, generator=g) < 0.3).to(torch.uint8)
    got = ops.pack_mask_rows(a.to(d)).cpu().numpy()
    assert np.array_equal(got, np.packbits(a.numpy(), axis=1, bitorder="little"))
    rows = torch.randperm(n, generator=g)[: max(1, n // 2)].to(torch.int32)
    got = ops.pack_mask_rows(a.to(d), rows.to(d)).cpu().numpy()
    assert np.array_equal(got, np.packbits(a.numpy()[rows.numpy()], axis=1, bitorder="little"))
    assert np.array_equal(ops.unpack_bits_host(got, N).view(np.uint8), a.numpy()[rows.numpy()])


def test_the_three_host_modes_hold_the_same_bits():
    import segdino3d_amd as seg
    from segdino3d_amd.architecture import PackedMasks
    from segdino3d_amd.configs import scannet200_model_cfg
    from segdino3d_amd.synth import make_scene, sharpen_random_model, structure_scene
    d = torch.device("cuda:0")
    pts, tgt = make_scene(1, n_points=8003, n_superpoints=64, n_query2d=8)       # (the smoke scene with a point count that is no multiple of 8)
    structure_scene(pts, tgt)
    cfg = scannet200_model_cfg(query_num=-1)
    cfg["test_cfg"]["npoint_thr"] = 20
    torch.manual_seed(0)
    model = sharpen_random_model(seg.build_architecture(cfg).eval()).to(d)
    outs = {}
    for mode in (False, True, "packed"):
        model.to_host = mode
        with torch.no_grad():
            outs[mode] = model([pts.to(d)], [copy.copy(tgt).to(d)])[0].pred_pts_seg
    dev, host, packed = outs[False], outs[True], outs["packed"]
    ref = dev.pts_instance_mask[0].cpu().numpy()
    assert ref.shape[0] >= 10 and ref.any()
    assert isinstance(host.pts_instance_mask[0], np.ndarray) and host.pts_instance_mask[0].dtype == np.bool_
    assert np.array_equal(host.pts_instance_mask[0], ref)
    assert isinstance(packed.pts_instance_mask[0], PackedMasks) and np.array_equal(np.asarray(packed.pts_instance_mask[0]), ref)
    assert packed.pts_instance_mask[0].nbytes * 7 < ref.nbytes
    for h in (host, packed):
        assert np.array_equal(h.pts_instance_mask[1], dev.pts_instance_mask[1].cpu().numpy())
        assert np.array_equal(h.pts_semantic_mask[0], dev.pts_semantic_mask[0].cpu().numpy())
        assert np.array_equal(h.pts_semantic_mask[1], dev.pts_semantic_mask[1].cpu().numpy())
        assert np.array_equal(h.instance_labels, dev.instance_labels.cpu().numpy())
        assert np.array_equal(h.instance_scores, dev.instance_scores.cpu().numpy())
        assert np.array_equal(h.instance_boxes, dev.instance_boxes.cpu().numpy())
        # pageable copies: the arrays own their memory (no view of a pinned torch tensor keeps it locked)
        for a in (h.pts_semantic_mask[0], h.pts_semantic_mask[1], h.pts_instance_mask[1], h.instance_labels, h.instance_scores):
            assert a.flags["OWNDATA"]
