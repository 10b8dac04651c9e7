"""Post-processing pieces added in round 4, each against a plain restatement of the reference expression it replaces
(`baseline3d.py:434-476`, `mask_matrix_nms` :71-139): device-side selection, index glue, point masks for a list of rows."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("k", [1, 5, 600, 1024, 1500, 3001])
def test_select_instances_matches_the_host_selection(k):
    """keep / pkeep / union / positions / masks / counts of `sd3d_select_instances` == the numpy selection of rounds 1-3
    (score > thr and count > npoint_thr, ascending), including candidate counts beyond one 1024-row chunk and ties on the thresholds."""
    from segdino3d_amd import ops
    d = _dev()
    g = torch.Generator().manual_seed(k)
    s = torch.rand(k, generator=g)
    s[::7] = 0.3                                                   # exactly on a threshold: `>` must drop them
    c = torch.randint(0, 200, (k,), generator=g, dtype=torch.int32)
    thr0, thr1, npt = 0.3, 0.55, 100
    (ints, bytes_), counts = ops.select_instances(s.to(d), c.to(d), thr0, thr1, npt)
    n_keep, n_pkeep, n_union, n_scored = counts.cpu().tolist()
    sn, cn = s.numpy(), c.numpy()
    np_ok = cn > npt
    m0, m1 = sn > np.float32(thr0), sn > np.float32(thr1)
    keep, pkeep = np.flatnonzero(m0 & np_ok), np.flatnonzero(m1 & np_ok)
    union = np.union1d(keep, pkeep)
    assert (n_keep, n_pkeep, n_union, n_scored) == (keep.size, pkeep.size, union.size, int(m0.sum()))
    ih, bh = ints.cpu().numpy(), bytes_.cpu().numpy()
    assert np.array_equal(ih[0, :n_keep], keep) and np.array_equal(ih[1, :n_pkeep], pkeep) and np.array_equal(ih[2, :n_union], union)
    assert np.array_equal(union[ih[3, :n_keep]], keep) and np.array_equal(union[ih[4, :n_pkeep]], pkeep)
    assert np.array_equal(bh[0, :k].astype(bool), m0)
    assert np.array_equal(bh[1, :n_scored].astype(bool), np_ok[m0])


@pytest.mark.parametrize("N,with_boxes", [(10_000, False), (10_003, True), (37, True)])
def test_expand_rows_equals_the_rows_of_expand_masks(N, with_boxes):
    """`MaskBits(...).rows(list)` == the listed rows of the full [n, N] table of `expand_masks` (and the same point counts)."""
    from segdino3d_amd import ops
    d = _dev()
    g = torch.Generator().manual_seed(N)
    n, S = 77, 96                                                  # S = padded row width of sig (a multiple of 32)
    sig = torch.rand(n, S, generator=g).to(d)
    src = torch.randperm(n, generator=g).to(torch.int32).to(d)
    sp = torch.randint(0, 90, (N,), generator=g).to(d)
    pts = (torch.rand(N, 6, generator=g) * 4).to(d)
    boxes = torch.cat([torch.rand(n, 3, generator=g) * 4, torch.rand(n, 3, generator=g) * 3], 1).to(d) if with_boxes else None
    full, count = ops.expand_masks(sig, src, sp, pts, 0.5, boxes)
    mb = ops.MaskBits(sig, src, sp, pts, 0.5, boxes)
    assert torch.equal(mb.count, count)
    for rows in ([], [0], [5, 6, 7, 40, 76], list(range(n)), [76, 3, 3, 31, 32, 33]):
        r = torch.tensor(rows, dtype=torch.int32, device=d)
        got = mb.rows(r)
        assert got.shape == (len(rows), N)
        assert torch.equal(got, full[r.long()])


def test_index_glue_kernels():
    from segdino3d_amd import ops
    d = _dev()
    g = torch.Generator().manual_seed(3)
    k, Q = 600, 200
    flat = torch.rand(Q * 198, generator=g).to(d)
    idx = torch.randint(0, Q * 198, (k,), generator=g, dtype=torch.int32).to(d)
    assert torch.equal(ops.take_f32(flat, idx), flat[idx.long()])
    labels = torch.randint(0, 198, (k,), generator=g, dtype=torch.int32).to(d)
    scores = torch.rand(k, generator=g).to(d)
    order1 = torch.randperm(k, generator=g).to(torch.int32).to(d)
    l1, s1 = ops.take_pair(order1, labels, scores)
    assert torch.equal(l1, labels[order1.long()]) and torch.equal(s1, scores[order1.long()])
    scores2 = torch.rand(k, generator=g).to(d)
    order2 = torch.randperm(k, generator=g).to(torch.int32).to(d)
    qidx = torch.randint(0, Q, (k,), generator=g, dtype=torch.int32).to(d)
    centers, sizes = torch.rand(Q, 3, generator=g).to(d), torch.rand(Q, 3, generator=g).to(d)
    fs, fl, rec, boxes = ops.nms_finish(order2, scores2, l1, order1, qidx, centers, sizes)
    o2 = order2.long()
    assert torch.equal(fs, scores2[o2]) and torch.equal(fl, l1[o2]) and torch.equal(rec, order1.long()[o2])
    q_rec = qidx.long()[rec]
    assert torch.equal(boxes, torch.cat([centers[q_rec], sizes[q_rec]], -1))
    fs2, fl2, rec2, none = ops.nms_finish(order2, scores2, l1, order1, qidx)
    assert none is None and torch.equal(fs2, fs)
    keep = torch.tensor([0, 3, 4, 599], dtype=torch.int32, device=d)
    lo, so, bo = ops.take_instances(keep, fl, fs, boxes)
    assert lo.dtype == torch.int64 and torch.equal(lo, fl[keep.long()].long()) and torch.equal(so, fs[keep.long()]) and torch.equal(bo, boxes[keep.long()])
    lo, so, bo = ops.take_instances(keep[:0], fl, fs, None)
    assert lo.numel() == 0 and bo is None


@pytest.mark.parametrize("n,k,kind", [(39_600, 600, "sigmoid"), (39_600, 600, "ties"), (40_960, 1024, "normal"), (600, 600, "sigmoid"),
                                      (1000, 1, "normal"), (5000, 37, "constant"), (20_000, 600, "signed")])
def test_topk_select_equals_the_head_of_the_stable_sort(n, k, kind):
    """`sd3d_topk_desc_f32` (round 5: radix select + rank in one workgroup) == `sort_pairs(keys_from_f32(x, descending))[:k]`, the
    index vector `predict_by_feat_instance` took from the full sort before (`baseline3d.py:434`), bit for bit - sigmoid products (top
    byte nearly constant), heavy ties across the k-th place (a stable sort keeps the lower indices), k = n, negative values, zeros."""
    from segdino3d_amd import ops
    d = _dev()
    g = torch.Generator().manual_seed(n + k)
    if kind == "sigmoid":
        x = torch.sigmoid(torch.randn(n, generator=g) * 3) * torch.sigmoid(torch.randn(n, generator=g))
    elif kind == "ties":
        x = (torch.randint(0, 40, (n,), generator=g).float() / 40.0)          # ~1000 copies of every value: the k-th place falls inside a tie
    elif kind == "constant":
        x = torch.full((n,), 0.25)
    elif kind == "signed":
        x = torch.randn(n, generator=g)
        x[::5] = 0.0
        x[1::7] = -0.0
    else:
        x = torch.randn(n, generator=g).abs()
    x = x.to(d)
    _, idx = ops.sort_pairs(ops.keys_from_f32(x, descending=True), None, 0, 32)
    got = ops.topk_desc(x, k)
    assert got.dtype == torch.int32 and got.shape == (k,)
    assert torch.equal(got, idx[:k].to(torch.int32))


def test_post_processing_is_the_same_with_the_select_and_with_the_sort(monkeypatch):
    """`Baseline3D._instances_common` on random decoder outputs: the radix select leaves every product of the threshold-independent
    part (scores, labels, records, mask bits) exactly as the full sort left it."""
    from segdino3d_amd import architecture as A
    import bench
    d = _dev()
    model = bench.build_model(200, d)
    from segdino3d_amd.synth import make_scene
    pts, tgt = make_scene(3, 20000, 400, 50)
    pts, tgt = pts.to(d), tgt.to(d)
    outs = {}
    with torch.no_grad():
        for mode in (True, False):
            monkeypatch.setattr(A, "TOPK_SELECT", mode)
            outs[mode] = model([pts], [tgt])[0].pred_pts_seg
    a, b = outs[True], outs[False]
    assert torch.equal(a.instance_scores, b.instance_scores) and torch.equal(a.instance_labels, b.instance_labels)
    assert torch.equal(a.pts_instance_mask[0], b.pts_instance_mask[0]) and torch.equal(a.pts_instance_mask[1], b.pts_instance_mask[1])
    assert torch.equal(a.pts_semantic_mask[0], b.pts_semantic_mask[0]) and torch.equal(a.pts_semantic_mask[1], b.pts_semantic_mask[1])
