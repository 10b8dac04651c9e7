"""Property tests for the sparse-backbone oracle (oracle/sparse_ref.py) against an independent
dense formulation: torch's dense conv3d / conv_transpose3d evaluated on the densified volume must
agree with the sparse kernel-map convolution at every occupied voxel.  CPU only.
(The oracle is "parity unpinned" against MinkowskiEngine/spconv - see its header.)"""
import numpy as np
import torch
import torch.nn.functional as F

from _det import det_param, det_randn
from oracle import sparse_ref as R


def _scene(n=400, extent=12, seed=0):
    g = torch.Generator().manual_seed(seed)
    c = torch.randint(0, extent, (n, 3), generator=g).numpy().astype(np.int32)
    uc, _ = R.unique_voxels(c)
    return uc


def _densify(coords, feats, extent, stride=1):
    C = feats.shape[1]
    vol = torch.zeros(C, extent, extent, extent)      # [C, Z, Y, X]
    c = torch.from_numpy(coords.astype(np.int64)) // stride
    vol[:, c[:, 2], c[:, 1], c[:, 0]] = feats.t()
    return vol


def _gather(vol, coords, stride=1):
    c = torch.from_numpy(coords.astype(np.int64)) // stride
    return vol[:, c[:, 2], c[:, 1], c[:, 0]].t()


def _dense_weight(w, k):
    """ME-ordered [K, Cin, Cout] (x fastest) -> conv3d weight [Cout, Cin, kz, ky, kx]."""
    return w.reshape(k, k, k, w.shape[1], w.shape[2]).permute(4, 3, 0, 1, 2).contiguous()


def test_same_map_conv_equals_dense_conv3d():
    uc = _scene()
    feats = det_randn("t.feats", (len(uc), 5))
    for k in (3, 5):
        w = det_randn(f"t.w{k}", (k ** 3, 5, 7))
        sparse = R.sparse_conv(feats, R.kernel_map(uc, uc, R.kernel_offsets(k), 1), w, len(uc))
        dense = F.conv3d(_densify(uc, feats, 12)[None], _dense_weight(w, k), padding=k // 2)[0]
        torch.testing.assert_close(sparse, _gather(dense, uc), rtol=1e-4, atol=1e-4)


def test_strided_and_transposed_conv_equal_dense():
    uc = _scene(extent=12)
    feats = det_randn("t.feats2", (len(uc), 4))
    lv = R.MinkLevels(uc)
    w = det_randn("t.wd", (8, 4, 6))
    down = R.sparse_conv(feats, lv.down(1), w, lv.n(2))
    dense = F.conv3d(_densify(uc, feats, 12)[None], _dense_weight(w, 2), stride=2)[0]
    torch.testing.assert_close(down, _gather(dense, lv.coords[2], 2), rtol=1e-4, atol=1e-4)
    # every coarse voxel has at least one child and coarse coords are multiples of 2
    assert (lv.coords[2] % 2 == 0).all()
    wt = det_randn("t.wt", (8, 6, 3))
    up = R.sparse_conv(down, lv.up(1), wt, lv.n(1))
    dvol = _densify(lv.coords[2], down, 6, 2)
    # conv_transpose3d weight [Cin, Cout, kz, ky, kx]; out[2p + d] += in[p] W[d]
    wd = wt.reshape(2, 2, 2, 6, 3).permute(3, 4, 0, 1, 2).contiguous()
    dense_up = F.conv_transpose3d(dvol[None], wd, stride=2)[0]
    torch.testing.assert_close(up, _gather(dense_up, uc), rtol=1e-4, atol=1e-4)


def test_negative_coordinates_floor():
    xyz = torch.tensor([[-0.001, 0.0, 0.039], [-0.021, 0.02, -0.04]])
    c = R.floor_voxel(xyz, 0.02)
    assert c.tolist() == [[-1, 0, 1], [-2, 1, -2]]
    coarse = R.downsample_coords(np.array([[-1, 0, 1], [-2, 1, -3]], dtype=np.int32), 2)
    assert sorted(map(tuple, coarse.tolist())) == sorted([(-2, 0, 0), (-2, 0, -4)])


def test_voxel_mean_and_inverse():
    xyz = torch.tensor([[0.01, 0.01, 0.01], [0.015, 0.0, 0.019], [0.05, 0.0, 0.0]])
    f = torch.tensor([[1.0, 2.0], [3.0, 6.0], [5.0, 5.0]])
    uc, inv = R.unique_voxels(R.floor_voxel(xyz, 0.02))
    vf = R.segment_mean(f, inv, len(uc))
    assert len(uc) == 2 and inv[0] == inv[1] != inv[2]
    torch.testing.assert_close(vf[inv[0]], torch.tensor([2.0, 4.0]))


def _sd(shapes, prefix):
    return {prefix + k: det_param(prefix + k, s) for k, s in shapes.items()}


def test_res16unet34c_runs_and_matches_key_layout():
    from segdino3d_amd.synth import make_scene
    pts, tgt = make_scene(3, n_points=3000, n_superpoints=40, n_query2d=5)
    sd = _sd(R.mink_state_dict_shapes(), "backbone.")
    n_conv = sum(1 for k in sd if k.endswith(".kernel"))
    assert n_conv == 62                      # SURVEY.md 2 #3a: 62 convs
    n_par = sum(v.numel() for k, v in sd.items() if k.endswith(".kernel"))
    assert abs(n_par / 1e6 - 38.9) < 0.2     # 38.9 M conv params
    f, pos, pos_wo = R.mink_forward_wrapper(sd, pts, tgt.extra_features["points_2dfeats"],
                                            tgt.extra_features["super_point_masks"])
    assert f.shape == (40, 96) and pos.shape == (40, 3) and torch.isfinite(f).all()
    # positions are means of floor-quantised coordinates: within one voxel of the raw mean
    raw = R.segment_mean(pts[:, :3], tgt.extra_features["super_point_masks"].numpy(), 40)
    assert (raw - pos).abs().max() < 0.02 + 1e-6 and ((raw - pos) >= -1e-6).all()


def test_spconvunet_runs():
    from segdino3d_amd.synth import make_scene
    pts, tgt = make_scene(4, n_points=3000, n_superpoints=40, n_query2d=5)
    shapes = R.spconv_state_dict_shapes()
    sd = _sd(shapes, "backbone.")
    n_conv = sum(1 for k, s in shapes.items() if len(s) == 5)
    assert n_conv == 49                      # 49 convs
    f, pos, _ = R.spconv_forward_wrapper(sd, pts, tgt.extra_features["points_2dfeats"],
                                         tgt.extra_features["super_point_masks"])
    assert f.shape == (40, 32) and torch.isfinite(f).all() and pos.shape == (40, 3)


def test_spconvunet_post_activation_variant():
    """normalize_before=False (spconvunet.py:66-81, 166-174, 194-201): convolution first, BatchNorm + ReLU after it, the identity
    branch added after the block's last ReLU.  Same 49 convolutions; with every BatchNorm an identity and the second convolution
    of every block zeroed, a block reduces to `relu(0) + x = x` - the skip path alone - which pins the order of ReLU and add."""
    from segdino3d_amd.synth import make_scene
    pts, tgt = make_scene(4, n_points=3000, n_superpoints=40, n_query2d=5)
    shapes = R.spconv_state_dict_shapes(normalize_before=False)
    assert sum(1 for s in shapes.values() if len(s) == 5) == 49
    assert "blocks.block0.conv_branch.0.weight" in shapes and "blocks.block0.conv_branch.1.running_var" in shapes
    assert "conv.0.weight" in shapes and "conv.1.weight" in shapes and "deconv.1.bias" in shapes and "conv.2.weight" not in shapes
    assert shapes["u.deconv.1.weight"] == (64,) and shapes["conv.1.weight"] == (64,)       # BN after the (de)convolution: its OUTPUT width
    sd = _sd(shapes, "backbone.")
    ef = tgt.extra_features
    f, pos, _ = R.spconv_forward_wrapper(sd, pts, ef["points_2dfeats"], ef["super_point_masks"], normalize_before=False)
    g, _, _ = R.spconv_forward_wrapper(_sd(R.spconv_state_dict_shapes(), "backbone."), pts, ef["points_2dfeats"], ef["super_point_masks"])
    assert f.shape == (40, 32) and torch.isfinite(f).all() and not torch.allclose(f, g)
    assert (f >= 0).all()                    # output_layer ends in a ReLU either way
