"""Host logic of the ScanNet AP path (segdino3d_amd.eval_ap) on the CPU: AP from compact scene records against
the reference's own output (tests/golden/ap_protocol.npz), the reference-shaped dictionaries rebuilt from a record,
record packing, and the two-rank exchange of records (gloo)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_oracle_golden import _ap_fixture


def _record_from_oracle(pred_info, gt, opts, valid, class_labels, id_to_label):
    """SceneRecord built from the ORACLE's association (stands in for the GPU kernel in CPU tests)."""
    from oracle import eval_ref as E
    from segdino3d_amd.eval_ap import SceneRecord
    g2p, p2g = E.assign_instances(pred_info, gt, opts, valid, class_labels, id_to_label)
    gts = sorted((g for label in class_labels for g in g2p[label]), key=lambda g: g["instance_id"])
    gid = {g["instance_id"]: i for i, g in enumerate(gts)}
    preds = sorted((p for label in class_labels for p in p2g[label]), key=lambda p: int(p["filename"].split("_")[1]))
    pp, gg, ii = [], [], []
    for pi, p in enumerate(preds):
        for m in p["matched_gt"]:
            pp.append(pi); gg.append(gid[m["instance_id"]]); ii.append(m["intersection"])
    a = lambda x, t=np.int64: np.asarray(x, dtype=t)  # noqa: E731
    return SceneRecord(a([p["label_id"] for p in preds]), a([int(p["filename"].split("_")[1]) for p in preds]),
                       a([p["vert_count"] for p in preds]), a([p["void_intersection"] for p in preds]),
                       a([p["confidence"] for p in preds], np.float64), a([g["label_id"] for g in gts]),
                       a([g["instance_id"] for g in gts]), a([g["vert_count"] for g in gts]), a(pp), a(gg), a(ii))


def _records(options):
    from oracle import eval_ref as E
    z, class_labels, valid, scenes, groups = _ap_fixture()
    id_to_label = {valid[i]: class_labels[i] for i in range(len(valid))}
    opts = E.get_options(options)
    preds = E.aggregate_predictions([s[2] for s in scenes], [s[3] for s in scenes], [s[4] for s in scenes], valid)
    gts = E.rename_gt([s[0] for s in scenes], [s[1] for s in scenes], valid)
    recs = [_record_from_oracle(p, g, opts, valid, class_labels, id_to_label) for p, g in zip(preds, gts)]
    return z, class_labels, valid, groups, opts, id_to_label, preds, gts, recs


@pytest.mark.parametrize("opt_name,options", [("default", None), ("min30", dict(min_region_sizes=np.array([30])))])
def test_ap_from_records_matches_reference_golden(opt_name, options):
    from segdino3d_amd import eval_ap
    z, class_labels, valid, groups, opts, id_to_label, preds, gts, recs = _records(options)
    ap, pr_rc = eval_ap.evaluate_records(recs, class_labels, valid, opts)
    metrics = eval_ap.compute_averages(ap, pr_rc, opts, class_labels)        # default groups = ScanNet200 lists
    for k, v in zip(z[f"{opt_name}_keys"], z[f"{opt_name}_vals"]):
        got = metrics[str(k)]
        assert (np.isnan(got) and np.isnan(v)) or abs(got - v) < 1e-12, (k, got, v)
    cls = np.array([[metrics["classes"][c][f] for f in ("ap", "ap50%", "ap25%", "prec50%", "rec50%")] for c in class_labels])
    assert np.allclose(cls, z[f"{opt_name}_class_ap"], rtol=0, atol=1e-12, equal_nan=True)


def test_reference_dicts_round_trip():
    """to_reference_dicts(record) == what the (oracle's) assign_instances_for_scan returns."""
    from oracle import eval_ref as E
    from segdino3d_amd import eval_ap
    z, class_labels, valid, groups, opts, id_to_label, preds, gts, recs = _records(None)
    for si, (p, g, rec) in enumerate(zip(preds, gts, recs)):
        g2p_ref, p2g_ref = E.assign_instances(p, g, opts, valid, class_labels, id_to_label)
        g2p, p2g = eval_ap.to_reference_dicts(rec, si, class_labels, id_to_label)
        for label in class_labels:
            assert len(g2p[label]) == len(g2p_ref[label]) and len(p2g[label]) == len(p2g_ref[label])
            for a, b in zip(p2g[label], p2g_ref[label]):
                assert a["filename"] == b["filename"] and a["vert_count"] == b["vert_count"] and a["void_intersection"] == b["void_intersection"]
                assert [(m["instance_id"], m["intersection"]) for m in a["matched_gt"]] == [(m["instance_id"], m["intersection"]) for m in b["matched_gt"]]
            for a, b in zip(g2p[label], g2p_ref[label]):
                assert a["instance_id"] == b["instance_id"] and a["vert_count"] == b["vert_count"]
                assert [(m["filename"], m["intersection"]) for m in a["matched_pred"]] == [(m["filename"], m["intersection"]) for m in b["matched_pred"]]
    # and the reference-shaped evaluate_matches (oracle) on the rebuilt dictionaries gives the golden AP again
    matches = {si: dict(zip(("gt", "pred"), eval_ap.to_reference_dicts(rec, si, class_labels, id_to_label))) for si, rec in enumerate(recs)}
    ap, _ = E.evaluate_matches(matches, class_labels, opts)
    ap2, _ = eval_ap.evaluate_records(recs, class_labels, valid, opts)
    assert np.allclose(ap, ap2, atol=1e-12, equal_nan=True)


def test_record_pack_unpack_and_edge_cases():
    from segdino3d_amd import eval_ap
    from segdino3d_amd.eval_ap import SceneRecord
    z, class_labels, valid, groups, opts, id_to_label, preds, gts, recs = _records(None)
    for rec in recs:
        r2 = SceneRecord.unpack(rec.pack())
        for f in rec.__dataclass_fields__:
            assert np.array_equal(getattr(rec, f), getattr(r2, f)), f
    e = np.zeros(0, dtype=np.int64)
    empty = SceneRecord(e, e, e, e, np.zeros(0), e, e, e, e, e, e)
    assert len(SceneRecord.unpack(empty.pack()).pred_label) == 0
    ap, pr_rc = eval_ap.evaluate_records([empty], class_labels, valid, opts)          # no gt, no pred: nan everywhere
    assert np.isnan(ap).all()
    only_gt = SceneRecord(e, e, e, e, np.zeros(0), np.array([valid[0]]), np.array([valid[0] * 1000 + 1]), np.array([500]), e, e, e)
    ap, _ = eval_ap.evaluate_records([only_gt], class_labels, valid, opts)
    assert (ap[0, 0] == 0).all() and np.isnan(ap[0, 1:]).all()                        # gt without predictions: AP 0


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from segdino3d_amd import dist_eval, eval_ap
        z, class_labels, valid, groups, opts, id_to_label, preds, gts, recs = _records(None)
        mine = [(i, recs[i]) for i in dist_eval.shard_scenes(len(recs), rank, world)]
        allrecs = dist_eval.all_gather_ap_records(mine)
        assert [sid for sid, _ in allrecs] == list(range(len(recs)))
        ap, pr_rc = eval_ap.evaluate_records([r for _, r in allrecs], class_labels, valid, opts)
        q.put((rank, eval_ap.compute_averages(ap, pr_rc, opts, class_labels)["all_ap"]))
    finally:
        dist.destroy_process_group()


def test_two_rank_record_exchange_gives_the_single_rank_ap():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = dict(q.get(timeout=10) for _ in range(2))
    z = _ap_fixture()[0]
    ref = dict(zip((str(k) for k in z["default_keys"]), z["default_vals"]))["all_ap"]
    assert abs(got[0] - ref) < 1e-12 and abs(got[1] - ref) < 1e-12


def _worker8(rank, world, port, q, n_scenes):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from segdino3d_amd import dist_eval, eval_ap
        from segdino3d_amd.eval_ap import SceneRecord
        z, class_labels, valid, groups, opts, id_to_label, preds, gts, recs = _records(None)
        e = np.zeros(0, dtype=np.int64)
        empty = SceneRecord(e, e, e, e, np.zeros(0), e, e, e, e, e, e)
        scene = lambda i: empty if i % 5 == 4 else recs[i % len(recs)]      # every fifth scene has neither ground truth nor predictions
        mine = [(i, scene(i)) for i in dist_eval.shard_scenes(n_scenes, rank, world)]
        allrecs = dist_eval.all_gather_ap_records(mine)
        assert [sid for sid, _ in allrecs] == list(range(n_scenes))
        for sid, r in allrecs:                                              # payloads survive the padded exchange unchanged
            want = scene(sid)
            assert all(np.array_equal(getattr(r, f), getattr(want, f)) for f in want.__dataclass_fields__)
        ap, pr_rc = eval_ap.evaluate_records([r for _, r in allrecs], class_labels, valid, opts)
        q.put((rank, len(mine), eval_ap.compute_averages(ap, pr_rc, opts, class_labels)["all_ap"]))
    finally:
        dist.destroy_process_group()


def test_eight_rank_exchange_with_ragged_and_empty_ranks():
    """The 8 x MI355X layout (BASELINE configs[3]) on gloo: 8 ranks, 3 scenes -> five ranks contribute NOTHING; 11 scenes ->
    ragged (2, 2, 2, 1, 1, 1, 1, 1) with empty scene records inside.  Every rank ends up with the same ordered records and the
    same AP as one process evaluating all scenes."""
    from segdino3d_amd import eval_ap
    from segdino3d_amd.eval_ap import SceneRecord
    ctx = mp.get_context("spawn")
    z, class_labels, valid, groups, opts, id_to_label, preds, gts, recs = _records(None)
    e = np.zeros(0, dtype=np.int64)
    empty = SceneRecord(e, e, e, e, np.zeros(0), e, e, e, e, e, e)
    for n_scenes in (3, 11):
        q = ctx.Queue()
        port = 31500 + (os.getpid() % 2000) + n_scenes
        procs = [ctx.Process(target=_worker8, args=(r, 8, port, q, n_scenes)) for r in range(8)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
            assert p.exitcode == 0
        got = sorted(q.get(timeout=10) for _ in range(8))
        assert [g[1] for g in got] == [len(range(r, n_scenes, 8)) for r in range(8)]
        single = [empty if i % 5 == 4 else recs[i % len(recs)] for i in range(n_scenes)]
        ap, pr_rc = eval_ap.evaluate_records(single, class_labels, valid, opts)
        ref = eval_ap.compute_averages(ap, pr_rc, opts, class_labels)["all_ap"]
        assert all(abs(g[2] - ref) < 1e-12 or (np.isnan(g[2]) and np.isnan(ref)) for g in got), (got, ref)


def test_map_inst_markup_matches_reference_golden():
    """segdino3d_amd.eval_ap.map_inst_markup (tensors, not in place) against the reference's evaluator method on the golden scenes
    (evaluator_3d.py:323-349), plus the numpy negative-index wrap it inherits: a thing-class point whose shifted semantic id is
    negative while its instance survives indexes the mapping from the back."""
    from segdino3d_amd import eval_ap
    from test_oracle_golden import _evaluator_fixture
    z, classes, valid, n_stuff, results = _evaluator_fixture()
    for si, (ann, _) in enumerate(results):
        sem0, inst0 = torch.from_numpy(ann["pts_semantic_mask"]), torch.from_numpy(ann["pts_instance_mask"])
        keep = (sem0.clone(), inst0.clone())
        sem, inst = eval_ap.map_inst_markup(sem0, inst0, valid[n_stuff:], n_stuff)
        assert torch.equal(sem, torch.from_numpy(z[f"s{si}_mapped_sem"])) and torch.equal(inst, torch.from_numpy(z[f"s{si}_mapped_inst"]))
        assert torch.equal(sem0, keep[0]) and torch.equal(inst0, keep[1])                      # inputs untouched
    from oracle import eval_ref as E
    sem = np.array([0, 1, 5, 9, 10, 3]); inst = np.array([0, 1, 7, 2, 4, 9])                    # class 0 with a thing instance id: wraps
    rs, ri = E.map_inst_markup(sem, inst, valid[n_stuff:], n_stuff)
    gs, gi = eval_ap.map_inst_markup(torch.from_numpy(sem), torch.from_numpy(inst), valid[n_stuff:], n_stuff)
    assert np.array_equal(gs.numpy(), rs) and np.array_equal(gi.numpy(), ri)


def test_eval_ann_info_matches_the_reference_evaluation_loop():
    """`eval_ap.eval_ann_info` and the oracle's restatement against the records the reference's own `evaluate_3d` loop produced from
    the same `GD3DTarget`s (tests/golden/make_golden_eval_loop.py executes the loop of evaluation/evaluate_3d.py:44-71): the loop
    reads `res_["masks"]`, `res_["labels"]`, `res_["extra_features"]`, `res_["scene_id"]`, `res_.pred_pts_seg` off this package's
    target type, sums the ids of overlapping instance masks, and marks uncovered points -1 / bg_class_id."""
    from oracle import eval_ref as E
    from segdino3d_amd import eval_ap
    from segdino3d_amd.gtypes import GD3DTarget
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eval_loop.npz"))
    bg = int(z["bg_class_id"])
    for i in range(3):
        n = int(z[f"s{i}_n"])
        masks = np.unpackbits(z[f"s{i}_masks"], axis=1)[:, :n].astype(bool)
        inst, sem = E.eval_ann_from_target(masks, z[f"s{i}_labels"], bg)
        assert np.array_equal(inst, z[f"s{i}_inst"]) and np.array_equal(sem, z[f"s{i}_sem"])
        t = GD3DTarget(masks=torch.from_numpy(masks)[:, :, None], labels=torch.from_numpy(z[f"s{i}_labels"]), scene_id=f"s{i}",
                       extra_features=dict(super_point_masks=torch.from_numpy(z[f"s{i}_sp"])))
        ann = eval_ap.eval_ann_info(t, bg)
        assert np.array_equal(ann["pts_instance_mask"].numpy(), z[f"s{i}_inst"]) and np.array_equal(ann["pts_semantic_mask"].numpy(), z[f"s{i}_sem"])
        assert ann["lidar_idx"] == f"s{i}" and torch.equal(ann["sp_pts_mask"], t["extra_features"]["super_point_masks"])
    assert (z["s0_inst"] > 8).sum() == 0 and (z["s0_inst"] == 3).sum() > 0        # the overlap of instances 1 and 2 reads as id 3
