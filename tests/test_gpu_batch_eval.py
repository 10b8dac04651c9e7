"""Batched evaluation forward: B scenes as ONE block-diagonal sparse tensor (sparse.BatchSceneMaps - the collation of the
reference's `utils/dataset_utils.py:215-230` collate_fn_3D + `minkunet.py:624-627`), the decoder's row-wise work for all scenes
in one pass (`ScanNetQueryDecoder._forward_batch`), post-processing per scene.  Evaluation BatchNorm is an affine map and every kernel on the path computes an output row from that row's
own pairs in a fixed order, so EVERY output of every scene must be bit-identical (`torch.equal`) to its single-scene forward
(`baseline3d.py:308-346` runs one scene per forward, `:335-338`).  Integer work (maps, pair lists) is compared exactly."""
import copy
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def _build(cfg, d):
    import segdino3d_amd as seg
    from segdino3d_amd.synth import sharpen_random_model
    torch.manual_seed(0)
    model = sharpen_random_model(seg.build_architecture(cfg).eval(), mask_gain=40.0)
    model.to(d)
    model.to_host = False
    return model


def _fields(pd):
    return dict(masks=pd.pts_instance_mask[0], pan=pd.pts_instance_mask[1], sem=pd.pts_semantic_mask[0], pan_sem=pd.pts_semantic_mask[1],
                labels=pd.instance_labels, scores=pd.instance_scores, boxes=pd.instance_boxes,
                topk=pd.sort_and_mask[0], score_mask=pd.sort_and_mask[1], npoint_mask=pd.sort_and_mask[2])


def _scenes(d, sizes, seed=21, structured=True):
    from segdino3d_amd.synth import make_scene, structure_scene
    out = []
    for i, (n, s, m) in enumerate(sizes):
        pts, tgt = make_scene(seed + i, n, s, m)
        if structured:
            structure_scene(pts, tgt)
        out.append((pts.to(d), tgt.to(d)))
    return out


def _compare(model, scenes, label):
    import segdino3d_amd as seg
    singles, caps = [], []
    with torch.no_grad():
        for p, t in scenes:
            with seg.capture() as cap:
                singles.append(_fields(model([p], [copy.copy(t)])[0].pred_pts_seg))
            caps.append(cap)
        with seg.capture() as capb:
            out = model([p for p, _ in scenes], [copy.copy(t) for _, t in scenes])
    torch.cuda.synchronize()
    assert len(out) == len(scenes)
    for i in range(len(scenes)):
        assert torch.equal(caps[i].sp_feats[0], capb.sp_feats[i]), f"{label}: superpoint features of scene {i} differ"
        assert torch.equal(caps[i].sp_pos[0], capb.sp_pos[i]), f"{label}: superpoint positions of scene {i} differ"
        for k in ("masks", "cls_preds", "sem_preds", "centers", "sizes", "hidden_states"):
            assert torch.equal(caps[i].outputs[k][0], capb.outputs[k][i]), f"{label}: decoder output `{k}` of scene {i} differs"
        for li, (a, b) in enumerate(zip(caps[i].outputs["aux_outputs"], capb.outputs["aux_outputs"])):
            assert torch.equal(a["masks"][0], b["masks"][i]), f"{label}: aux mask logits of layer {li}, scene {i} differ"
        got = _fields(out[i].pred_pts_seg)
        for k, v in singles[i].items():
            if v is None:
                assert got[k] is None
                continue
            assert got[k].shape == v.shape and torch.equal(got[k], v), f"{label}: `{k}` of scene {i} differs from the single-scene forward"
    return singles


def test_batched_forward_is_bit_identical_at_benchmark_size():
    """`model([p1..pB], [t1..tB])` at 150 k points (three scenes of different sizes), query_num = 200."""
    from segdino3d_amd.configs import scannet200_model_cfg
    d = dev()
    model = _build(scannet200_model_cfg(query_num=200), d)
    scenes = _scenes(d, [(150_000, 3000, 300), (142_999, 2900, 300), (135_998, 2800, 280)])
    singles = _compare(model, scenes, "benchmark size")
    stats = [(int(s["scores"].numel()), int(s["masks"].sum())) for s in singles]
    print("batched == sequential, bit for bit; (instances, mask points) per scene:", stats)
    assert all(n >= 100 for n, _ in stats) and sum(m for _, m in stats) > 10000, "scenes must yield instances with content"


def test_batched_forward_one_query_per_superpoint_and_ragged_batch():
    """query_num = -1 (the reference's evaluation setting, `baseline3d.py:227-228`), four scenes from 6 k to 40 k points; the
    decoders of the scenes have different query / key counts."""
    from segdino3d_amd.configs import scannet200_model_cfg
    d = dev()
    cfg = scannet200_model_cfg(query_num=-1)
    cfg["test_cfg"]["npoint_thr"] = 20
    model = _build(cfg, d)
    scenes = _scenes(d, [(40_000, 700, 60), (6_000, 90, 12), (23_456, 333, 40), (12_000, 150, 8)], seed=5)
    _compare(model, scenes, "ragged batch")


def test_batch_maps_equal_the_scenes_own_maps():
    """Integer work: levels, inverse map, neighbour tables and pair lists of the block-diagonal tensor are the scenes' own,
    row-shifted by the scene offsets - exactly."""
    from segdino3d_amd.sparse import BatchSceneMaps, SceneMaps
    d = dev()
    scenes = _scenes(d, [(30_000, 400, 20), (9_000, 120, 10), (17_000, 250, 10)], seed=9, structured=False)
    pts = [p for p, _ in scenes]
    sps = [t.extra_features["super_point_masks"] for _, t in scenes]
    batch = BatchSceneMaps(pts, 0.02, 5, superpoints=sps)
    same, strides = [(0, 5)] + [(l, 3) for l in range(5)], [0, 1, 2, 3]
    batch.prepare(same=same, strides=strides)
    offs = [[0] for _ in range(5)]
    singles = []
    for p, sp in zip(pts, sps):
        m = SceneMaps(p, 0.02, 5, superpoints=sp)
        m.prepare(same=same, strides=strides)
        singles.append(m)
        for l in range(5):
            offs[l].append(offs[l][-1] + m.n_vox[l])
    assert batch.n_vox == [o[-1] for o in offs]
    assert batch.sp_off == [0] + list(torch.tensor([m.n_superpoints for m in singles]).cumsum(0).tolist())
    mask48 = (1 << 48) - 1
    for i, m in enumerate(singles):
        for l in range(5):
            a, b = offs[l][i], offs[l][i + 1]
            assert torch.equal(batch.keys[l][a:b] & mask48, m.keys[l]) and bool(((batch.keys[l][a:b] >> 48) == i).all())
        pa, pb = batch.point_off[i], batch.point_off[i + 1]
        assert torch.equal(batch.inverse[pa:pb], m.inverse + offs[0][i])
        assert torch.equal(batch.icoords[pa:pb], m.icoords)
        for (lvl, k) in same:
            nb, ns = batch.same(lvl, k)[:, offs[lvl][i]:offs[lvl][i + 1]], m.same(lvl, k)
            assert torch.equal(nb, torch.where(ns >= 0, ns + offs[lvl][i], ns)), f"scene {i}: neighbour table (level {lvl}, k {k})"
        for lvl in strides:
            dn, up = batch.down(lvl)[:, offs[lvl + 1][i]:offs[lvl + 1][i + 1]], batch.up(lvl)[:, offs[lvl][i]:offs[lvl][i + 1]]
            assert torch.equal(dn, torch.where(m.down(lvl) >= 0, m.down(lvl) + offs[lvl][i], m.down(lvl)))
            assert torch.equal(up, torch.where(m.up(lvl) >= 0, m.up(lvl) + offs[lvl + 1][i], m.up(lvl)))
    # no neighbour crosses a scene boundary
    for (lvl, k) in same:
        nbr = batch.same(lvl, k)
        row_scene = torch.bucketize(torch.arange(nbr.shape[1], device=d), torch.tensor(offs[lvl][1:], device=d), right=True)
        nb_scene = torch.bucketize(nbr.clamp(min=0), torch.tensor(offs[lvl][1:], device=d), right=True)
        assert bool(((nbr < 0) | (nb_scene == row_scene[None, :])).all())
    # pair lists: every (offset, row) position points at the pair whose input row is the table's entry
    key = ("same", 1, 3)
    pl, nbr = batch.pairs[key], batch.same(1, 3)
    hit = pl.pos >= 0
    assert torch.equal(hit, nbr >= 0)
    assert torch.equal(pl.in_idx[pl.pos[hit].long()], nbr[hit])


def test_voxel_mean_and_pooling_of_the_batch_equal_the_single_scene_kernels():
    """C-ABI entries sd3d_voxel_mean_batch / sd3d_segment_starts_batch / sd3d_keys_from_i64_offset against the single-scene
    entries, including a scene whose superpoint ids have gaps (unused ids give zero rows, `torch_scatter` semantics)."""
    from segdino3d_amd.sparse import BatchSceneMaps, SceneMaps
    d = dev()
    scenes = _scenes(d, [(20_000, 300, 10), (8_000, 100, 10)], seed=13, structured=False)
    sps = [t.extra_features["super_point_masks"].clone() for _, t in scenes]
    sps[1][sps[1] == 7] = 8                                      # id 7 unused in scene 1
    sps[1][sps[1] == 99] = 98                                    # the last id unused: the scene has 99 superpoints now
    pts = [p for p, _ in scenes]
    f2d = [t.extra_features["points_2dfeats"] for _, t in scenes]
    batch = BatchSceneMaps(pts, 0.02, 1, superpoints=sps)
    vf = batch.voxel_features(pts, f2d, 0, 288)
    g = torch.Generator().manual_seed(3)
    feat = torch.randn(batch.n_vox[0], 96, generator=g).to(d)
    fb, pb = batch.pool(feat, 96)
    v0 = 0
    for i in range(2):
        m = SceneMaps(pts[i], 0.02, 1, superpoints=sps[i])
        assert torch.equal(vf[v0:v0 + m.n_vox[0]], m.voxel_features(pts[i], f2d[i], 0, 288)), f"voxel features of scene {i}"
        f1, p1 = m.pool(feat[v0:v0 + m.n_vox[0]].contiguous(), 96)
        a, b = batch.sp_off[i], batch.sp_off[i + 1]
        assert b - a == m.n_superpoints == (300 if i == 0 else 99)
        assert torch.equal(fb[a:b], f1) and torch.equal(pb[a:b], p1), f"pooled rows of scene {i}"
        v0 += m.n_vox[0]
    assert bool((fb[batch.sp_off[1] + 7] == 0).all())


def test_pipelined_runner_with_batches_is_bit_identical_to_sequential():
    """`PipelinedRunner(model, streams=2, batch=2)`: two batches in flight, each fanning its scenes over side streams, against
    back-to-back single-scene forwards (shared model, per-(thread, stream) scratch)."""
    from segdino3d_amd.configs import scannet200_model_cfg
    from segdino3d_amd.dist_eval import PipelinedRunner
    d = dev()
    model = _build(scannet200_model_cfg(query_num=200), d)
    scenes = _scenes(d, [(60_000, 1200, 100), (52_999, 1100, 100), (45_998, 1000, 80)])
    order = [0, 1, 2, 0, 2, 1, 1, 0, 2, 2, 1]                        # 11 scenes: the last batch is a single scene
    with torch.no_grad():
        seq = [_fields(model([p], [copy.copy(t)])[0].pred_pts_seg) for p, t in scenes]
    torch.cuda.synchronize()
    for rep in range(2):
        par = PipelinedRunner(model, 2, d, batch=2).run([(scenes[i][0], copy.copy(scenes[i][1])) for i in order])
        torch.cuda.synchronize()
        assert len(par) == len(order)
        for slot, i in enumerate(order):
            got = _fields(par[slot][0].pred_pts_seg)
            for k, v in seq[i].items():
                assert got[k].shape == v.shape and torch.equal(got[k], v), f"run {rep}, slot {slot} (scene {i}): `{k}` differs"


@pytest.mark.parametrize("rows,cin,cout", [(90, 256, 3072), (200, 96, 256), (200, 256, 256), (200, 512, 768), (200, 256, 1024), (200, 1024, 256),
                                            (200, 256, 199), (301, 256, 3072), (340, 256, 3072), (512, 256, 1536), (700, 256, 3072), (1500, 96, 256),
                                            (3000, 256, 3072), (3000, 96, 256)])
def test_dense_plan_code_reproduces_the_single_scene_kernel(rows, cin, cout):
    """sd3d_dense_plan_code(rows, Cin, Cout) must make sd3d_gather_gemm compute each of MANY rows exactly as its own heuristic computes
    them when it is given `rows` rows: the batched decoder's bit-identity rests on it (csrc/gather_gemm.hip keeps the two together)."""
    from segdino3d_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(rows + cin)
    big = 4 * rows + 37
    x = torch.randn(big, cin, generator=g).to(d)
    w = (torch.randn(cout, cin, generator=g) * cin ** -0.5).to(d)
    b = torch.randn(cout, generator=g).to(d)
    ref = torch.cat([ops.gather_gemm(x[i:i + rows].contiguous(), w, shift=b, act="relu") for i in range(0, big - rows + 1, rows)])
    code = ops.dense_code(rows, cin, cout)
    got = ops.gather_gemm(x, w, shift=b, act="relu", nt=code, exact=True)
    assert torch.equal(got[:ref.shape[0]], ref), f"rows={rows} {cin}->{cout}: code {code} does not reproduce the heuristic's kernel"


def test_batched_forward_in_the_bf16_decoder_mode_and_for_configs0():
    """The batched path in the two other configurations: configs[2] (`compute_dtype = "bf16"`: superpoint-side projections and the
    attention contractions on bf16 operands) and configs[0] (SpConvUNet backbone scene by scene + additive box refinement,
    `SegDINO3D_ScanNetv2`) - bit-identical to single-scene forwards in both."""
    from segdino3d_amd.configs import scannet200_model_cfg, scannetv2_model_cfg
    d = dev()
    cfg = scannet200_model_cfg(query_num=200)
    cfg["decoder_cfg"]["compute_dtype"] = "bf16"
    model = _build(cfg, d)
    _compare(model, _scenes(d, [(90_000, 2500, 200), (84_000, 2300, 180)]), "bf16 decoder")
    cfg0 = scannetv2_model_cfg(query_num=-1)
    cfg0["test_cfg"]["npoint_thr"] = 20
    model0 = _build(cfg0, d)
    _compare(model0, _scenes(d, [(10_000, 300, 50), (8_000, 250, 40), (9_000, 280, 50)], seed=31), "configs[0]")
