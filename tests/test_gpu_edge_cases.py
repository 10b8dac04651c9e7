"""Edge cases of the hot path on the GPU: ragged / empty segments, tiny scenes, hash collisions, empty
prediction sets, key-range overflow.  Each is checked against the oracle (or an exact expectation)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from _det import det_randn  # noqa: E402


def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def test_pool_with_unused_superpoint_ids_and_heavy_voxels():
    """scatter_mean semantics: ids that never occur give zero rows; one voxel may hold hundreds of points."""
    from oracle import sparse_ref as R
    from segdino3d_amd.sparse import SceneMaps
    d = dev()
    g = torch.Generator().manual_seed(3)
    n = 5000
    xyz = torch.rand(n, 3, generator=g) * 0.5
    xyz[:700] = torch.tensor([0.111, 0.222, 0.333]) + 0.001 * torch.rand(700, 3, generator=g)   # ~700 points in one voxel
    pts = torch.cat([xyz, torch.randn(n, 3, generator=g)], 1).contiguous()
    sp = torch.randint(0, 40, (n,), generator=g) * 3            # only ids 0,3,6,...,117 are used
    sp[0] = 119                                                 # S = 120
    maps = SceneMaps(pts.to(d), 0.02, 5, superpoints=sp.to(d))
    c = R.floor_voxel(xyz, 0.02)
    uc, inv = R.unique_voxels(c)
    assert maps.n_vox[0] == len(uc) and maps.n_superpoints == 120
    from helpers import device_level_coords, match_rows
    perm = match_rows(device_level_coords(maps, 0), uc)
    x = det_randn("edge.x", (len(uc), 32))
    f, p = maps.pool(x[perm].to(d).contiguous(), 32)
    rf = R.segment_mean(x[torch.from_numpy(inv)], sp.numpy(), 120)
    rp = R.segment_mean(torch.from_numpy(c).float() * 0.02, sp.numpy(), 120)
    torch.testing.assert_close(f.cpu(), rf, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(p.cpu(), rp, rtol=1e-5, atol=1e-5)
    assert (f.cpu()[1] == 0).all() and (p.cpu()[2] == 0).all()
    vf = maps.voxel_features(pts.to(d), None, 1, 32).cpu()
    ref = R.segment_mean(pts[:, 3:], inv, len(uc))[perm]
    torch.testing.assert_close(vf[:, :3], ref, rtol=1e-5, atol=1e-5)


def test_tiny_scene_single_voxel_and_conv():
    from segdino3d_amd import ops
    from segdino3d_amd.sparse import SceneMaps
    d = dev()
    pts = torch.tensor([[0.001, 0.002, 0.003, 1, 2, 3], [0.004, 0.001, 0.002, 3, 2, 1]], dtype=torch.float32)
    maps = SceneMaps(pts.to(d), 0.02, 5, superpoints=torch.zeros(2, dtype=torch.long, device=d))
    assert maps.n_vox == [1, 1, 1, 1, 1] and maps.n_superpoints == 1
    nbr = maps.same(0, 3).cpu()
    assert nbr.shape == (27, 1) and nbr[13, 0] == 0 and (nbr >= 0).sum() == 1       # only the centre offset hits
    x = torch.ones(1, 32, device=d)
    w = torch.zeros(27, 32, 32); w[13] = torch.eye(32)
    y = ops.gather_gemm(x, w.to(d), nbr=maps.same(0, 3))
    torch.testing.assert_close(y.cpu(), torch.ones(1, 32))
    f, p = maps.pool(y, 32)
    torch.testing.assert_close(f.cpu(), torch.ones(1, 32))
    torch.testing.assert_close(p.cpu(), torch.zeros(1, 3))


def test_hash_table_under_collisions():
    """Dense cube: every voxel has all 27 neighbours inside -> every probe sequence must resolve."""
    from segdino3d_amd.sparse import SceneMaps
    d = dev()
    r = torch.arange(24)
    grid = torch.stack(torch.meshgrid(r, r, r, indexing="ij"), -1).reshape(-1, 3).float() * 0.02 + 0.01
    pts = torch.cat([grid, torch.zeros(len(grid), 3)], 1).contiguous()
    maps = SceneMaps(pts.to(d), 0.02, 5, superpoints=torch.zeros(len(grid), dtype=torch.long, device=d))
    assert maps.n_vox == [24 ** 3, 12 ** 3, 6 ** 3, 3 ** 3, 8]
    nbr = maps.same(0, 3)
    hits = int((nbr >= 0).sum())
    assert hits == sum((24 - abs(dx)) * (24 - abs(dy)) * (24 - abs(dz)) for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1))
    assert int((maps.down(0) >= 0).sum()) == 24 ** 3 and int((maps.up(0) >= 0).sum()) == 24 ** 3


def test_key_range_overflow_is_reported():
    from segdino3d_amd.sparse import SceneMaps
    d = dev()
    pts = torch.zeros(4, 6)
    pts[1, 0] = 2000.0            # 100 000 voxels wide: exceeds the 16-bit-per-axis key range
    with pytest.raises(RuntimeError, match="key range"):
        SceneMaps(pts.to(d), 0.02, 5, superpoints=torch.zeros(4, dtype=torch.long, device=d))


@pytest.mark.parametrize("bad_id", [-1, 2 ** 31 - 1, 2 ** 40])
def test_superpoint_ids_out_of_range_are_reported(bad_id):
    """a negative or > INT32_MAX - 1 superpoint id used to be clamped into the count (max + 1 = 2^31 rows allocated); now the key
    kernel raises flag bit 8 - also on the full-sort retry - and SceneMaps / BatchSceneMaps fail with a clear message."""
    from segdino3d_amd.sparse import BatchSceneMaps, SceneMaps
    d = dev()
    g = torch.Generator().manual_seed(4)
    pts = torch.cat([torch.rand(3000, 3, generator=g) * 2.0, torch.randn(3000, 3, generator=g)], 1).to(d)
    sp = torch.randint(0, 20, (3000,), generator=g)
    sp[17] = bad_id
    with pytest.raises(RuntimeError, match="superpoint ids must lie"):
        SceneMaps(pts, 0.02, 5, superpoints=sp.to(d))
    with pytest.raises(RuntimeError, match="superpoint ids must lie"):
        BatchSceneMaps([pts, pts], 0.02, 5, superpoints=[sp.to(d), torch.randint(0, 20, (3000,), generator=g).to(d)])


def test_no_2d_queries_and_no_surviving_instances():
    """M = 0 cached 2D queries (only the dummy key remains, decoder :724-727) and a test_cfg whose
    thresholds reject every instance (empty outputs, panoptic falls back to the stuff map, :529-530)."""
    import segdino3d_amd as seg
    from oracle import decoder_ref as D
    from segdino3d_amd.configs import scannet200_model_cfg
    from segdino3d_amd.gtypes import GD3DTarget
    from test_gpu_decoder import _StoredBackbone
    from test_oracle_golden import decoder_state_dict
    d = dev()
    if seg.BACKBONES.get("_StoredBackbone") is None:
        seg.BACKBONES.register_module(module=_StoredBackbone)
    cfg = scannet200_model_cfg(query_num=-1)
    cfg["pointcloud_backbone_cfg"] = dict(type="_StoredBackbone")
    cfg["criterion_cfg"] = None
    cfg["test_cfg"]["npoint_thr"] = 10 ** 9
    model = seg.build_architecture(cfg).eval()
    sd = decoder_state_dict()
    model.decoder.load_state_dict({k[len("decoder."):]: v for k, v in sd.items()})
    model.to(d)
    S, N = 40, 900
    g = torch.Generator().manual_seed(5)
    sp_feat = torch.randn(S, 96, generator=g)
    sp_pos = torch.floor(torch.rand(S, 3, generator=g) * 4 / 0.02) * 0.02
    pts = torch.cat([torch.rand(N, 3, generator=g) * 4, torch.randn(N, 3, generator=g)], 1)
    sp = torch.arange(N) % S
    model.backbone.f, model.backbone.p = sp_feat.to(d), sp_pos.to(d)
    tgt = GD3DTarget(masks=None, extra_features=dict(super_point_masks=sp, query2d_feats=torch.zeros(0, 256),
                     query2d_pos=torch.zeros(0, 3))).to(d)
    with seg.capture() as cap:
        res = model([pts.to(d)], [tgt])
    pd = res[0].pred_pts_seg
    assert pd.pts_instance_mask[0].shape == (0, N) and pd.instance_scores.shape == (0,) and pd.instance_boxes.shape == (0, 6)
    assert pd.pts_semantic_mask[0].shape == (N,) and np.array_equal(pd.pts_semantic_mask[1], pd.pts_instance_mask[1])
    out = cap.outputs
    lo, hi = pts[:, :3].min(0)[0], pts[:, :3].max(0)[0]
    ref = D.decoder_forward(sd, D.DecoderCfg(), sp_feat, sp_pos, sp_pos, sp_feat, sp_pos, torch.zeros(0, 256), torch.zeros(0, 3), lo, hi)
    err = (out["masks"][0].cpu() - ref["masks"]).abs()
    bad = (err > 3e-4 + 3e-4 * ref["masks"].abs()).any(dim=1).float().mean().item()
    print(f"no 2D queries: mask-logit rows outside 3e-4: {bad:.1%}, max err {err.max().item():.2e}")
    assert bad <= 0.02


def test_pipelined_runner_matches_sequential():
    """Two scenes in flight on two HIP streams (threads) give the same BITS as back-to-back runs (small ragged scenes; the
    benchmarked mode - four streams, 150 k points - is tests/test_gpu_benchmark_parity.py)."""
    import copy
    import segdino3d_amd as seg
    from segdino3d_amd.configs import scannet200_model_cfg
    from segdino3d_amd.dist_eval import PipelinedRunner
    from segdino3d_amd.synth import make_scene
    d = dev()
    torch.manual_seed(0)
    model = seg.build_architecture(scannet200_model_cfg(query_num=-1)).eval().to(d)
    model.to_host = False
    scenes = []
    for i in range(4):
        pts, tgt = make_scene(20 + i, n_points=6000 + 500 * i, n_superpoints=50, n_query2d=10)
        scenes.append((pts.to(d), tgt.to(d)))
    with torch.no_grad():
        seq = [model([p], [copy.copy(t)])[0].pred_pts_seg for p, t in scenes]
    par = PipelinedRunner(model, 2, d).run([(p, copy.copy(t)) for p, t in scenes])
    torch.cuda.synchronize()
    for i in range(len(scenes)):                       # every kernel sums in a fixed order: bit-identical, not "close"
        a, b = seq[i], par[i][0].pred_pts_seg
        assert torch.equal(a.instance_scores, b.instance_scores) and torch.equal(a.instance_labels, b.instance_labels)
        assert torch.equal(a.pts_instance_mask[0], b.pts_instance_mask[0]) and torch.equal(a.pts_instance_mask[1], b.pts_instance_mask[1])
        assert torch.equal(a.pts_semantic_mask[0], b.pts_semantic_mask[0]) and torch.equal(a.pts_semantic_mask[1], b.pts_semantic_mask[1])


def test_scene_prefetcher_feeds_identical_scenes(tmp_path):
    """Packed files -> pinned staging -> copy stream -> consumer stream: same tensors as a direct upload, in order,
    and the forward on a prefetched scene equals the forward on the directly uploaded one."""
    import copy
    import segdino3d_amd as seg
    from segdino3d_amd import io_scene
    from segdino3d_amd.configs import scannet200_model_cfg
    from segdino3d_amd.synth import make_scene
    d = dev()
    paths, ref = [], []
    for i in range(5):
        pts, tgt = make_scene(40 + i, 6000 + 500 * i, 80, 9)
        ef = tgt.extra_features
        sc = dict(points=pts, super_points=ef["super_point_masks"], points_2dfeats=ef["points_2dfeats"],
                  query2d_feats=ef["query2d_feats"], query2d_pos=ef["query2d_pos"])
        p = os.path.join(tmp_path, f"s{i}.sd3d")
        io_scene.pack_scene(p, sc)
        paths.append(p)
        ref.append(sc)
    model = seg.build_architecture(scannet200_model_cfg(query_num=30)).eval().to(d)
    model.to_host = False
    n = 0
    with torch.no_grad():
        for i, (pts, tgt) in enumerate(io_scene.ScenePrefetcher(paths, d, depth=2)):
            assert torch.equal(pts.cpu(), ref[i]["points"])
            assert torch.equal(tgt.extra_features["points_2dfeats"].cpu(), ref[i]["points_2dfeats"])
            assert torch.equal(tgt.extra_features["super_point_masks"].cpu(), ref[i]["super_points"])
            out = model([pts], [tgt])[0].pred_pts_seg
            direct_tgt = io_scene.to_device_scene({k: v for k, v in ref[i].items()}, d, non_blocking=False)[1]
            out2 = model([ref[i]["points"].to(d)], [direct_tgt])[0].pred_pts_seg
            assert torch.equal(out.instance_scores, out2.instance_scores)
            assert torch.equal(out.pts_semantic_mask[0], out2.pts_semantic_mask[0])
            n += 1
    assert n == 5
