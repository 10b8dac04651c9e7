"""The path at real-scan sizes.  The reference evaluates the whole validation split scene by scene, whatever
its size (`evaluation/evaluate_3d.py:44-68`, `segdino3d/datasets/dataset/scannet200.py:198-289`): ScanNet scans run from ~30 k to
~500 k points and a few hundred to ~10 k superpoints, one query per superpoint.  Every full-size test of rounds 1-3 used the one
benchmark shape (150 k / 3000); here:
  (i)   500 k points / 10 k superpoints / 300 2D queries against the oracle, `query_num` 200 and -1 (pair lists of millions of
        entries x 256 columns, [600, 500 k] point masks, [10 k, 313]-word attention masks);
  (ii)  30 k points / 400 superpoints against the oracle;
  (iii) a batch of [500 k, 30 k, 150 k, 300 k]-point scenes through `model([...])` against single-scene forwards, bit for bit;
  (iv)  the pipelined runner over that mix, bit for bit.
"""
import copy
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import test_gpu_benchmark_parity as P  # noqa: E402  (the parity harness: HIP forward vs oracle.model_ref, every stage printed)

# measured on MI355X (printed by the tests; the one-query-per-superpoint run flips more thresholded mask bits than at 3000 queries)
BIG = dict(P.FULL, bad_rows=0.01, sign=0.9995, twins=0.985, semantic=0.9995)


@pytest.mark.parametrize("query_num", [200, -1])
def test_half_million_points_ten_thousand_superpoints_match_oracle(query_num):
    from segdino3d_amd.configs import scannet200_model_cfg
    P._compare_forward(scannet200_model_cfg(query_num=query_num), "mink", (500_000, 10_000, 300), query_num, 198, BIG, {})


def test_small_scan_matches_oracle():
    from segdino3d_amd.configs import scannet200_model_cfg
    P._compare_forward(scannet200_model_cfg(query_num=-1), "mink", (30_000, 400, 40), -1, 198, BIG, {})


SIZES = [(500_000, 10_000, 300), (30_000, 400, 40), (150_000, 3000, 300), (300_000, 6000, 200)]


def _scenes(d):
    from segdino3d_amd.synth import make_scene, structure_scene
    out = []
    for j, (n, s, m) in enumerate(SIZES):
        pts, tgt = make_scene(60 + j, n, s, m)
        structure_scene(pts, tgt)
        out.append((pts.to(d), tgt.to(d)))
    return out


def _fields(pd):
    return dict(masks=pd.pts_instance_mask[0], pan=pd.pts_instance_mask[1], sem=pd.pts_semantic_mask[0], pan_sem=pd.pts_semantic_mask[1],
                labels=pd.instance_labels, scores=pd.instance_scores, boxes=pd.instance_boxes)


@pytest.mark.parametrize("query_num", [200, -1])
def test_mixed_size_batch_and_pipelined_runner_are_bit_identical_to_single_forwards(query_num):
    """query_num = -1 also routes the 6000- and 10 000-query scenes through the row-chain decoder and the two small ones op by op
    inside ONE call (decoder.FUSED_DECODER = "auto")."""
    from segdino3d_amd.configs import scannet200_model_cfg
    from segdino3d_amd.dist_eval import PipelinedRunner
    d = P.dev()
    cfg = scannet200_model_cfg(query_num=query_num)
    cfg["test_cfg"]["npoint_thr"] = 20                              # (the 30 k-point scene has ~75 points per superpoint)
    model, _ = P._build(cfg, d)
    scenes = _scenes(d)
    with torch.no_grad():
        seq = [_fields(model([p], [copy.copy(t)])[0].pred_pts_seg) for p, t in scenes]
        torch.cuda.synchronize()
        print("single-scene reference:", [(int(s["scores"].numel()), int(s["masks"].sum())) for s in seq], "(instances, mask points) per scene")
        assert sum(s["scores"].numel() >= 50 and int(s["masks"].sum()) > 1000 for s in seq) >= 3, "scenes must yield instances with content"
        batch = model([p for p, _ in scenes], [copy.copy(t) for _, t in scenes])
        torch.cuda.synchronize()
        for i, r in enumerate(batch):
            got = _fields(r.pred_pts_seg)
            for k, v in seq[i].items():
                assert got[k].shape == v.shape and torch.equal(got[k], v), f"batched forward, scene {i}: `{k}` differs from its single-scene forward"
        order = [0, 1, 2, 3, 3, 2, 1, 0]
        for batch_size in (1, 2):
            par = PipelinedRunner(model, 2, d, batch=batch_size).run([(scenes[i][0], copy.copy(scenes[i][1])) for i in order])
            torch.cuda.synchronize()
            for slot, i in enumerate(order):
                got = _fields(par[slot][0].pred_pts_seg)
                for k, v in seq[i].items():
                    assert torch.equal(got[k], v), f"runner (batch {batch_size}), slot {slot} (scene {i}): `{k}` differs"
    print(f"query_num={query_num}: batch of {[s[0] for s in SIZES]} points and the pipelined runner: every output bit-identical to single-scene forwards")


def test_worst_case_pair_capacity_gives_the_same_bits(monkeypatch):
    """`sparse.EXACT_PAIR_CAPACITY`: with ONE scene in flight the rulebook sizes are not read back (pair lists sized K x V: no second
    host synchronisation, 0.24 ms of the single-scene latency); the rulebooks - and every output - are those of the exact-size run."""
    from segdino3d_amd import sparse
    from segdino3d_amd.configs import scannet200_model_cfg
    from segdino3d_amd.synth import make_scene, structure_scene
    d = P.dev()
    model, _ = P._build(scannet200_model_cfg(query_num=200), d)
    pts, tgt = make_scene(21, 150_000, 3000, 300)
    structure_scene(pts, tgt)
    pts, tgt = pts.to(d), tgt.to(d)
    outs = {}
    for mode in (True, False, "auto"):
        monkeypatch.setattr(sparse, "EXACT_PAIR_CAPACITY", mode)
        with torch.no_grad():
            outs[mode] = _fields(model([pts], [copy.copy(tgt)])[0].pred_pts_seg)
    assert outs[True]["scores"].numel() >= 50
    for mode in (False, "auto"):
        for k, v in outs[True].items():
            assert torch.equal(outs[mode][k], v), (mode, k)


def test_fork_join_of_the_table_building_gives_the_same_bits(monkeypatch):
    """`sparse.FORK_JOIN`: with one scene in flight only the stem's neighbour table is built on the scene's stream, the other levels'
    hash tables / kernel maps / pair lists on a side stream between the segments of the U-Net plan (`sd3d_run_layers_ev` waits per
    table).  Same kernels on the same data in a different interleaving: every output of the exact run, twice in a row (a stale side
    stream would show on the second forward)."""
    from segdino3d_amd import sparse
    from segdino3d_amd.configs import scannet200_model_cfg
    from segdino3d_amd.synth import make_scene, structure_scene
    d = P.dev()
    model, _ = P._build(scannet200_model_cfg(query_num=200), d)
    scenes = []
    for seed, n, s in ((23, 150_000, 3000), (24, 60_000, 900)):
        pts, tgt = make_scene(seed, n, s, 300)
        structure_scene(pts, tgt)
        scenes.append((pts.to(d), tgt.to(d)))
    outs = {}
    for mode in (False, True):
        monkeypatch.setattr(sparse, "FORK_JOIN", mode)
        with torch.no_grad():
            outs[mode] = [_fields(model([p], [copy.copy(t)])[0].pred_pts_seg) for p, t in scenes + scenes]
    assert outs[False][0]["scores"].numel() >= 50
    for a, b in zip(outs[False], outs[True]):
        for k, v in a.items():
            assert torch.equal(b[k], v), k
