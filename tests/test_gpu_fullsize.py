"""Size-independent properties of the hot path at BASELINE.json's full size (configs[1]: 150 k points,
3000 superpoints, 300 2D queries, 200 queries) - sizes at which the CPU oracle takes seconds per call.
Sortedness, permutation, round trips, symmetry of the neighbour tables, exact gathers through the
convolution kernels (identity / one-hot weights), linearity, determinism, conservation of point counts."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_POINTS, N_SP, N_Q2D, N_QUERY = 150_000, 3000, 300, 200


def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def scene():
    from segdino3d_amd.sparse import SceneMaps
    from segdino3d_amd.synth import make_scene
    d = dev()
    pts, tgt = make_scene(7, N_POINTS, N_SP, N_Q2D)
    pts, tgt = pts.to(d), tgt.to(d)
    maps = SceneMaps(pts, 0.02, 5, superpoints=tgt.extra_features["super_point_masks"])
    maps.prepare(same=[(0, 5)] + [(l, 3) for l in range(5)], strides=[0, 1, 2, 3])
    return pts, tgt, maps


def test_sort_is_a_sorted_permutation_at_full_size():
    from segdino3d_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(0)
    keys = torch.randint(0, 1 << 55, (N_POINTS,), generator=g, dtype=torch.int64).to(d)
    sk, sv = ops.sort_pairs(keys.clone(), None, 0, 56)
    assert bool((sk[1:] >= sk[:-1]).all())
    assert torch.equal(keys[sv.long()], sk)                                   # values carry the permutation
    assert torch.equal(torch.sort(sv.long())[0], torch.arange(N_POINTS, device=d))


def test_voxel_levels_are_consistent(scene):
    pts, tgt, maps = scene
    assert sum(1 for _ in maps.n_vox) == 5 and maps.n_vox[0] <= N_POINTS
    for l in range(5):
        k = maps.keys[l]
        assert k.shape[0] == maps.n_vox[l]
        assert bool((k[1:] > k[:-1]).all()), f"level {l} keys not strictly increasing"
    for l in range(4):                                                          # parent of a voxel = its key >> 3
        par = maps.parents[l].long()
        assert bool((par >= 0).all()) and int(par.max()) == maps.n_vox[l + 1] - 1
        assert torch.equal(maps.keys[l + 1][par] & ((1 << 48) - 1), (maps.keys[l] & ((1 << 48) - 1)) >> 3)
    # every point maps to one voxel; segment lengths add up to N
    inv = maps.inverse.long()
    assert int(inv.min()) == 0 and int(inv.max()) == maps.n_vox[0] - 1
    seg = maps.seg_start.long()
    assert int(seg[0]) == 0 and int(seg[-1]) == N_POINTS and bool((seg[1:] > seg[:-1]).all())
    assert torch.equal(torch.bincount(inv, minlength=maps.n_vox[0]), seg[1:] - seg[:-1])


def test_neighbour_tables_are_symmetric(scene):
    pts, tgt, maps = scene
    for lvl, ks in [(0, 5), (0, 3), (2, 3), (4, 3)]:
        nbr = maps.same(lvl, ks).long()
        K, M = nbr.shape
        rows = torch.arange(M, device=nbr.device)
        assert torch.equal(nbr[K // 2], rows)                                   # centre offset = the voxel itself
        for k in (0, 1, K // 2 - 1, K - 2):
            j = nbr[k]
            ok = j >= 0
            assert torch.equal(nbr[K - 1 - k][j[ok]], rows[ok]), f"level {lvl} k={ks}: offset {k} not mirrored"
    for lvl in range(4):                                                        # stride-2 maps: one parent per child
        dn, up = maps.down(lvl).long(), maps.up(lvl).long()
        assert int((up >= 0).sum()) == maps.n_vox[lvl] and bool(((up >= 0).sum(dim=0) == 1).all())
        assert int((dn >= 0).sum()) == maps.n_vox[lvl]
        child = torch.arange(maps.n_vox[lvl], device=dn.device)
        k_of = (up >= 0).float().argmax(dim=0)
        parent = up[k_of, child]
        assert torch.equal(dn[k_of, parent], child)                             # up is the transpose of down


def test_pair_lists_cover_the_rulebook_exactly(scene):
    pts, tgt, maps = scene
    for key in [("same", 0, 5), ("same", 1, 3), ("same", 3, 3), ("down", 1), ("up", 2)]:
        tab = maps.conv_table(*key)
        nbr, pl = tab["nbr"], tab["pairs"]
        K, M = nbr.shape
        valid = nbr >= 0
        pos = pl.pos.long()
        assert torch.equal(pos >= 0, valid)
        p = pos[valid]
        assert torch.equal(torch.sort(p)[0], torch.nonzero(pl.in_idx >= 0).squeeze(1))   # a bijection onto the real entries
        assert torch.equal(pl.in_idx.long()[p], nbr[valid].long())
        tk = pl.tile_k.long()
        n_real, c0, cn = (int(v) for v in tk[-3:])              # number of real tiles, then the centre offset's run of tiles
        assert bool((tk[:n_real] >= 0).all()) and bool((tk[n_real:-3] == -1).all())
        if pl.center >= 0:
            assert cn == (M + 127) // 128 and bool((tk[c0:c0 + cn] == pl.center).all()) and (c0 == 0 or int(tk[c0 - 1]) == pl.center - 1)
        else:
            assert (c0, cn) == (0, 0)
        kk = torch.arange(K, device=nbr.device).unsqueeze(1).expand(K, M)[valid]
        assert torch.equal(tk[p // 128], kk)                                    # every pair sits in a tile of its offset


def test_convolution_kernels_gather_exactly(scene):
    """One-hot weights turn the convolution into a pure gather: out[r] = x[nbr[k][r]] (0 where there is no
    neighbour) must hold bit-exactly for both GEMM passes, every kernel variant and the concat input."""
    from segdino3d_amd import ops
    pts, tgt, maps = scene
    d = pts.device
    g = torch.Generator().manual_seed(1)
    for key, C, k_sel in [(("same", 0, 3), 96, 5), (("same", 1, 3), 32, 13), (("same", 2, 3), 128, 26), (("same", 3, 3), 256, 0),
                          (("same", 0, 5), 32, 77), (("down", 0), 64, 3), (("up", 3), 256, 6)]:
        tab = maps.conv_table(*key)
        nbr, pairs = tab["nbr"], tab["pairs"]
        K, M = nbr.shape
        n_in = int(nbr.max().item()) + 1
        x = torch.randn(n_in, C, generator=g).to(d)
        w = torch.zeros(K, C, C, device=d)
        w[k_sel] = torch.eye(C, device=d)
        idx = nbr[k_sel].long()
        exp = torch.where((idx >= 0).unsqueeze(1), x[idx.clamp(min=0)], torch.zeros((), device=d))
        assert torch.equal(ops.pair_conv(x, w, pairs), exp), f"{key}: pair_conv is not an exact gather"
        if C >= 64:
            h = 32 if C == 96 else C // 2                                        # the concat split must be a multiple of 32
            assert torch.equal(ops.pair_conv(x[:, :h], w, pairs, x2=x[:, h:]), exp), f"{key}: concat input"
        if K <= 27:
            assert torch.equal(ops.gather_gemm(x, w, nbr=nbr), exp), f"{key}: gather_gemm is not an exact gather"


def test_convolution_is_linear(scene):
    from segdino3d_amd import ops
    pts, tgt, maps = scene
    d = pts.device
    g = torch.Generator().manual_seed(2)
    tab = maps.conv_table("same", 1, 3)
    nbr, pairs = tab["nbr"], tab["pairs"]
    K, M = nbr.shape
    x1, x2 = torch.randn(M, 96, generator=g).to(d), torch.randn(M, 96, generator=g).to(d)
    w = (torch.randn(K, 96, 96, generator=g) * (K * 96) ** -0.5).to(d)
    lhs = ops.pair_conv(x1 + 2.0 * x2, w, pairs)
    rhs = ops.pair_conv(x1, w, pairs) + 2.0 * ops.pair_conv(x2, w, pairs)
    assert (lhs - rhs).abs().max().item() < 2e-5                               # fp32 rounding of ~250-term sums of O(1) values
    assert torch.equal(ops.pair_conv(x1, w, pairs), ops.pair_conv(x1, w, pairs))   # deterministic


def test_pooling_conserves_points_and_constants(scene):
    pts, tgt, maps = scene
    d = pts.device
    ones = torch.full((maps.n_vox[0], 32), 3.5, device=d)
    f, pos = maps.pool(ones, 32)
    sp = tgt.extra_features["super_point_masks"]
    used = torch.bincount(sp, minlength=maps.n_superpoints) > 0
    assert f.shape == (maps.n_superpoints, 32)
    assert bool((f[used] == 3.5).all()) and bool((f[~used] == 0).all())        # the mean of a constant is the constant
    # quantised mean positions stay inside the scene's bounding box (one voxel of slack)
    lo, hi = pts[:, :3].min(dim=0)[0] - 0.02, pts[:, :3].max(dim=0)[0] + 0.02
    assert bool(((pos[used] >= lo) & (pos[used] <= hi)).all())


def test_full_forward_is_deterministic_and_counts_add_up():
    import segdino3d_amd as seg
    from segdino3d_amd import ops
    from segdino3d_amd.configs import scannet200_model_cfg
    from segdino3d_amd.synth import make_scene
    d = dev()
    torch.manual_seed(0)
    model = seg.build_architecture(scannet200_model_cfg(query_num=N_QUERY)).eval().to(d)
    model.to_host = False
    pts, tgt = make_scene(8, N_POINTS, N_SP, N_Q2D)
    pts, tgt = pts.to(d), tgt.to(d)
    import copy
    with torch.no_grad():
        with seg.capture() as c1:
            r1 = model([pts], [copy.copy(tgt)])[0].pred_pts_seg
        out1 = {k: c1.outputs[k][0] for k in ("masks", "cls_preds")}
        with seg.capture() as c2:
            r2 = model([pts], [copy.copy(tgt)])[0].pred_pts_seg
        out2 = c2.outputs
    assert torch.equal(out1["masks"], out2["masks"][0]) and torch.equal(out1["cls_preds"], out2["cls_preds"][0])
    m1, m2 = r1.pts_instance_mask[0], r2.pts_instance_mask[0]
    assert m1.shape == m2.shape and m1.shape[1] == N_POINTS and torch.equal(m1, m2)
    assert torch.equal(r1.instance_scores, r2.instance_scores)
    # semantic / panoptic maps label every point
    assert r1.pts_semantic_mask[0].shape == (N_POINTS,) and r1.pts_semantic_mask[1].shape == (N_POINTS,)
    # expand_masks: point counts = row sums when the box filter is off
    sig = torch.rand(600, 3008, device=d)
    src = torch.arange(600, dtype=torch.int32, device=d)
    sp = tgt.extra_features["super_point_masks"].contiguous()
    masks, count = ops.expand_masks(sig, src, sp, pts, 0.4, None)
    assert torch.equal(masks.sum(dim=1, dtype=torch.int64), count.long())
    assert torch.equal(masks.view(torch.bool), (sig[:, :3000] > 0.4)[:, sp])
