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
