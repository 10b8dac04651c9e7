"""GPU parity of the AP association (sd3d_mask_overlaps through segdino3d_amd.eval_ap.assign_scene): integer work,
bit-exact against the reference's golden association rows and against the oracle on random scenes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_oracle_golden import _ap_fixture  # noqa: E402


def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def _assoc_rows(si, rec):
    rows = []
    starts = np.searchsorted(rec.pair_pred, np.arange(len(rec.pred_label) + 1))
    for p in range(len(rec.pred_label)):
        qs = range(starts[p], starts[p + 1])
        for q in qs:
            g = rec.pair_gt[q]
            rows.append([si, int(rec.pred_index[p]), int(rec.pred_label[p]), int(rec.pred_vert[p]), int(rec.pred_void[p]),
                         int(rec.gt_id[g]), int(rec.gt_vert[g]), int(rec.pair_inter[q])])
        if len(qs) == 0:
            rows.append([si, int(rec.pred_index[p]), int(rec.pred_label[p]), int(rec.pred_vert[p]), int(rec.pred_void[p]), -1, 0, 0])
    return rows


@pytest.mark.parametrize("opt_name,options", [("default", None), ("min30", dict(min_region_sizes=np.array([30])))])
def test_device_association_and_ap_match_reference_golden(opt_name, options):
    from segdino3d_amd import eval_ap
    d = dev()
    z, class_labels, valid, scenes, groups = _ap_fixture()
    opts = eval_ap.get_options(options)
    sem = [torch.from_numpy(s[0]).to(d) for s in scenes]
    inst = [torch.from_numpy(s[1]).to(d) for s in scenes]
    gts = eval_ap.rename_gt(sem, inst, valid)
    rows = []
    for si, s in enumerate(scenes):
        rec = eval_ap.assign_scene(torch.from_numpy(s[2]).to(d), torch.from_numpy(s[3]).to(d), torch.from_numpy(s[4]).to(d), gts[si], opts, valid)
        rows += _assoc_rows(si, rec)
    assert np.array_equal(np.array(sorted(rows), dtype=np.int64), z[f"{opt_name}_assoc"])
    metrics = eval_ap.instance_seg_eval(sem, inst, [torch.from_numpy(s[2]).to(d) for s in scenes], [torch.from_numpy(s[3]).to(d) for s in scenes],
                                        [torch.from_numpy(s[4]).to(d) for s in scenes], valid, class_labels, options=options)
    for k, v in zip(z[f"{opt_name}_keys"], z[f"{opt_name}_vals"]):
        got = metrics[str(k)]
        assert (np.isnan(got) and np.isnan(v)) or abs(got - v) < 1e-12, (k, got, v)


def test_device_association_random_scenes_and_edges():
    """Ragged sizes (N not a multiple of 16, strided mask rows), no predictions, no valid ground truth, all-void scene,
    more ground-truth columns than one histogram pass of 256 threads."""
    from oracle import eval_ref as E
    from segdino3d_amd import eval_ap
    d = dev()
    g = np.random.default_rng(5)
    valid = tuple(range(1, 41))
    class_labels = tuple(f"c{i}" for i in valid)
    id_to_label = dict(zip(valid, class_labels))
    opts = eval_ap.get_options(dict(min_region_sizes=np.array([10])))
    for case, (N, n_inst, n_pred) in enumerate([(20011, 400, 150), (5003, 3, 0), (7001, 0, 20), (16384 * 2 + 5, 60, 64)]):
        gt = np.zeros(N, dtype=np.int64)
        if n_inst:
            owner = g.integers(0, n_inst, N)
            sem = g.integers(0, 45, n_inst)                                   # 0 and 41..44 are not valid -> void
            gt = np.where(np.isin(sem[owner], valid), sem[owner] * 1000 + owner + 1, owner + 1)
        masks = g.random((n_pred, N)) > 0.9 if n_pred else np.zeros((0, N), dtype=bool)
        labels = g.integers(0, len(valid), n_pred)
        scores = g.random(n_pred).astype(np.float32)
        big = torch.zeros(max(n_pred, 1), N + 37, dtype=torch.bool, device=d)   # strided rows
        mt = big[:n_pred, :N]
        mt.copy_(torch.from_numpy(masks))
        rec = eval_ap.assign_scene(mt, torch.from_numpy(labels).to(d), torch.from_numpy(scores).to(d), torch.from_numpy(gt).to(d), opts, valid)
        pred_info = {f"0_{i}": dict(mask=masks[i], label_id=valid[labels[i]], conf=scores[i]) for i in range(n_pred)}
        g2p, p2g = E.assign_instances(pred_info, gt, opts, valid, class_labels, id_to_label)
        exp = []
        for label in class_labels:
            for pr in p2g[label]:
                for m in pr["matched_gt"]:
                    exp.append([0, int(pr["filename"].split("_")[1]), pr["label_id"], pr["vert_count"], pr["void_intersection"],
                                m["instance_id"], m["vert_count"], m["intersection"]])
                if not pr["matched_gt"]:
                    exp.append([0, int(pr["filename"].split("_")[1]), pr["label_id"], pr["vert_count"], pr["void_intersection"], -1, 0, 0])
        assert sorted(_assoc_rows(0, rec)) == sorted(exp), f"case {case}"
        n_gt_ref = sum(len(v) for v in g2p.values())
        assert len(rec.gt_id) == n_gt_ref


def test_full_size_association_conserves_points():
    """600 predictions x 150 k points: every mask point lands in exactly one column (instance or void)."""
    from segdino3d_amd import eval_ap
    d = dev()
    gen = torch.Generator(device="cpu").manual_seed(0)
    N, n = 150_000, 600
    masks = (torch.rand(n, N, generator=gen) > 0.98).to(d)
    owner = torch.randint(0, 80, (N,), generator=gen)
    sem = torch.randint(0, 200, (80,), generator=gen)
    valid = tuple(range(2, 200, 2))
    gt = torch.where(torch.isin(sem[owner], torch.tensor(valid)), sem[owner] * 1000 + owner + 1, owner + 1).to(d)
    opts = eval_ap.get_options(None)
    labels = torch.randint(0, len(valid), (n,), generator=gen).to(d)
    rec = eval_ap.assign_scene(masks, labels, torch.rand(n, generator=gen).to(d), gt, opts, valid)
    assert np.array_equal(rec.pred_vert, masks.sum(dim=1).cpu().numpy()[rec.pred_index])
    # intersections with same-label gts + void <= vert_count; recompute a few rows exactly
    for p in (0, 17, len(rec.pred_label) - 1):
        row = masks[int(rec.pred_index[p])]
        sel = rec.pair_pred == p
        for g_i, inter in zip(rec.pair_gt[sel], rec.pair_inter[sel]):
            assert int((row & (gt == int(rec.gt_id[g_i]))).sum()) == int(inter)


def test_end_to_end_map_of_hip_path_equals_oracle_path():
    """"Identical mAP": the whole HIP forward + device AP association against the whole oracle forward + numpy AP on
    the same scene, weights and ground truth.  The ground truth is labelled from the ORACLE's own predictions, so the
    classes that carry ground truth also carry predictions and any mask / score / label drift of the HIP path moves AP."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from oracle import eval_ref as E
    from oracle import model_ref
    from oracle import postprocess_ref as P
    import segdino3d_amd as seg
    from segdino3d_amd import eval_ap
    from segdino3d_amd.configs import scannet200_model_cfg
    from segdino3d_amd.synth import make_scene
    d = dev()
    pts, tgt = make_scene(21, n_points=8000, n_superpoints=64, n_query2d=8)
    # iid-noise inputs and random weights give every superpoint the same features (and every mask the same points);
    # give the scene structure instead: 2D features and colours are a per-superpoint embedding plus noise, and the mask
    # branch is sharpened so that a prediction switches on a handful of superpoints (a few hundred points)
    ef = tgt.extra_features
    g = torch.Generator().manual_seed(5)
    sp = ef["super_point_masks"]
    ef["points_2dfeats"] = (torch.randn(64, 256, generator=g) * 2.0)[sp] + 0.1 * torch.randn(8000, 256, generator=g)
    pts[:, 3:] = torch.randn(64, 3, generator=g)[sp]
    cfg = scannet200_model_cfg(query_num=-1)
    cfg["test_cfg"]["npoint_thr"] = 20
    cfg["filter_outofbox_points_eval"] = False                            # random-weight boxes would empty every mask
    torch.manual_seed(0)
    model = seg.build_architecture(cfg).eval()
    g2 = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(0.1 * torch.randn(m.num_features, generator=g2))
                m.running_var.copy_(0.5 + torch.rand(m.num_features, generator=g2))
        model.decoder.x_mask[2].weight.mul_(40.0)
        model.decoder.x_mask[2].bias.zero_()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(d)
    model.to_host = False
    with torch.no_grad():
        pd = model([pts.to(d)], [tgt.to(d)])[0].pred_pts_seg
    tgt = tgt.to("cpu")                                                  # .to() moves the target in place
    ef = tgt.extra_features
    ref = model_ref.forward_eval(sd, pts, ef["points_2dfeats"], ef["super_point_masks"], ef["query2d_feats"], ef["query2d_pos"],
                                 tgt.masks, query_num=-1, test_cfg=P.TestCfg(npoint_thr=20), box_filter=False)
    r_masks, r_labels, r_scores = ref["pts_instance_mask"][0].numpy().astype(bool), ref["instance_labels"].numpy(), ref["instance_scores"].numpy()
    assert r_masks.shape[0] >= 10, "the scene must yield predictions for this test to mean anything"
    # ground truth from the oracle's predictions
    n_cls = 198
    valid = tuple(range(2, 2 + n_cls))
    class_labels = tuple(f"c{i}" for i in valid)
    N = r_masks.shape[1]
    # ground truth: the oracle's best predictions, greedily, each reduced to the points no earlier one took
    inst = np.zeros(N, dtype=np.int64)
    sem = np.zeros(N, dtype=np.int64)                                     # 0 = not a valid class -> void
    taken = np.zeros(N, dtype=bool)
    k = 0
    for i in np.argsort(-r_scores, kind="stable"):
        m = r_masks[i] & ~taken
        if m.sum() >= 100 and m.sum() >= 0.5 * r_masks[i].sum():
            k += 1
            inst[m], sem[m] = k, valid[int(r_labels[i])]
            taken |= m
        if k == 10:
            break
    assert k >= 1, k
    opts = dict(min_region_sizes=np.array([50]))
    # oracle path
    id_to_label = dict(zip(valid, class_labels))
    preds = E.aggregate_predictions([r_masks], [r_labels], [r_scores], valid)
    gts = E.rename_gt([sem], [inst], valid)
    m_ref, _, _ = E.scannet_eval(preds, gts, opts, valid, class_labels, id_to_label)
    # HIP path
    m_hip = eval_ap.instance_seg_eval([torch.from_numpy(sem).to(d)], [torch.from_numpy(inst).to(d)], [pd.pts_instance_mask[0]],
                                      [pd.instance_labels], [pd.instance_scores], valid, class_labels, options=opts, groups={})
    assert m_ref["all_ap_25%"] > 0.05                                      # not a trivially-zero operating point
    # prediction sets: every oracle prediction has a twin (same label, identical point mask) in the HIP output
    h_masks = pd.pts_instance_mask[0].cpu().numpy().astype(bool)
    h_labels = pd.instance_labels.cpu().numpy()
    assert h_masks.shape == r_masks.shape
    key = lambda lab, m: (int(lab), np.packbits(m).tobytes())  # noqa: E731
    h_set = {key(l, m) for l, m in zip(h_labels, h_masks)}
    twins = sum(key(l, m) in h_set for l, m in zip(r_labels, r_masks))
    assert twins >= 0.99 * len(r_labels), f"{twins} of {len(r_labels)} oracle predictions found in the HIP output"
    print("mAP oracle / HIP:", {k: (round(float(m_ref[k]), 5), round(float(m_hip[k]), 5)) for k in ("all_ap", "all_ap_50%", "all_ap_25%")})
    for key in ("all_ap", "all_ap_50%", "all_ap_25%"):
        assert abs(m_hip[key] - m_ref[key]) < 1e-3, (key, m_hip[key], m_ref[key])      # north star: mAP within +-0.1 points


def test_evaluator_level_metrics_match_reference_golden():
    """eval_ap.evaluator_instance_metrics with device tensors against the metrics the reference's
    InstanceSeg3DEvaluator.compute_metrics built from the same per-scene results (evaluator_3d.py:124-219)."""
    from segdino3d_amd import eval_ap
    from test_oracle_golden import _evaluator_fixture
    d = dev()
    z, classes, valid, n_stuff, results = _evaluator_fixture()
    on_dev = [(dict(pts_semantic_mask=torch.from_numpy(a["pts_semantic_mask"]).to(d), pts_instance_mask=torch.from_numpy(a["pts_instance_mask"]).to(d)),
               dict(pts_instance_mask=[torch.from_numpy(p["pts_instance_mask"][0]).to(d)], instance_labels=torch.from_numpy(p["instance_labels"]).to(d),
                    instance_scores=torch.from_numpy(p["instance_scores"]).to(d))) for a, p in results]
    metrics = eval_ap.evaluator_instance_metrics(on_dev, classes, valid, n_stuff)
    for k, v in zip(z["keys"], z["vals"]):
        got = metrics[str(k)]
        assert (np.isnan(got) and np.isnan(v)) or abs(got - v) < 1e-12, (k, got, v)
    cls = np.array([[metrics["classes"][c][f] for f in ("ap", "ap50%", "ap25%")] for c in classes[n_stuff:-1]])
    assert np.allclose(cls, z["class_ap"], rtol=0, atol=1e-12, equal_nan=True)
    # numpy inputs (what the reference's evaluator holds) give the same numbers
    host = eval_ap.evaluator_instance_metrics([(a, dict(p, pts_instance_mask=[torch.from_numpy(p["pts_instance_mask"][0]).to(d)])) for a, p in results],
                                              classes, valid, n_stuff)
    assert all((np.isnan(host[str(k)]) and np.isnan(v)) or abs(host[str(k)] - v) < 1e-12 for k, v in zip(z["keys"], z["vals"]))


def test_evaluation_loop_chain_on_the_device_equals_the_oracle_chain():
    """The caller chain either side of the forward, end to end: model output + ground-truth target -> the evaluation loop's record
    (`eval_ann_info`, evaluate_3d.py:49-63) -> `map_inst_markup` -> instance AP (`evaluator_instance_metrics`, evaluator_3d.py:124-219),
    everything on the device, against the same chain through the oracle (CPU forward, numpy records, numpy AP).  Ground truth: two
    stuff "instances" (classes 0 / 1: they must drop out) plus up to ten objects cut from the oracle's own best predictions and
    labelled like them, so that ground truth and predictions meet in the same classes."""
    from oracle import eval_ref as E
    from oracle import model_ref
    from oracle import postprocess_ref as P
    import segdino3d_amd as seg
    from segdino3d_amd import eval_ap
    from segdino3d_amd.configs import scannet200_model_cfg
    from segdino3d_amd.synth import make_scene, sharpen_random_model, structure_scene
    d = dev()
    pts, tgt = make_scene(21, n_points=8000, n_superpoints=64, n_query2d=8)
    structure_scene(pts, tgt)
    cfg = scannet200_model_cfg(query_num=-1)
    cfg["test_cfg"]["npoint_thr"] = 20
    cfg["filter_outofbox_points_eval"] = False
    torch.manual_seed(0)
    model = sharpen_random_model(seg.build_architecture(cfg).eval())
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ef = tgt.extra_features
    ref = model_ref.forward_eval(sd, pts, ef["points_2dfeats"], ef["super_point_masks"], ef["query2d_feats"], ef["query2d_pos"],
                                 tgt.masks, query_num=-1, test_cfg=P.TestCfg(npoint_thr=20), box_filter=False)
    r_masks, r_labels, r_scores = ref["pts_instance_mask"][0].numpy().astype(bool), ref["instance_labels"].numpy(), ref["instance_scores"].numpy()
    N = r_masks.shape[1]
    taken = np.zeros(N, dtype=bool)
    gt_m, gt_l = [], []
    for i in np.argsort(-r_scores, kind="stable"):
        m = r_masks[i] & ~taken
        if m.sum() >= 100 and m.sum() >= 0.5 * r_masks[i].sum():
            gt_m.append(m); gt_l.append(int(r_labels[i]) + 2)            # dataset class = thing index + the two stuff classes
            taken |= m
        if len(gt_m) == 10:
            break
    assert len(gt_m) >= 1
    rest = np.flatnonzero(~taken)
    stuff0, stuff1 = np.zeros(N, dtype=bool), np.zeros(N, dtype=bool)
    stuff0[rest[: len(rest) // 3]] = True                                # "wall" and "floor" take two thirds of the uncovered points,
    stuff1[rest[len(rest) // 3: 2 * len(rest) // 3]] = True              # the last third stays unannotated (-1 / bg_class_id)
    gt_masks = torch.from_numpy(np.stack([stuff0, stuff1] + gt_m))[:, :, None]
    gt_labels = torch.tensor([0, 1] + gt_l)
    tgt.masks, tgt.labels = gt_masks.clone(), gt_labels.clone()
    model.to(d)
    model.to_host = False
    with torch.no_grad():
        out = model([pts.to(d)], [tgt.to(d)])[0]
    classes = tuple(f"c{i}" for i in range(200)) + ("unlabeled",)
    valid = tuple(range(1, 201))
    opts = dict(min_region_sizes=np.array([50]))
    m_hip = eval_ap.evaluator_instance_metrics([(eval_ap.eval_ann_info(out, 200), out.pred_pts_seg)], classes, valid, 2, options=opts, groups={})
    inst, sem = E.eval_ann_from_target(gt_masks[:, :, 0].numpy(), gt_labels.numpy(), 200)
    pred = dict(pts_instance_mask=[r_masks], instance_labels=r_labels, instance_scores=r_scores)
    m_ref, sems, insts = E.evaluator_instance_metrics([(dict(pts_semantic_mask=sem, pts_instance_mask=inst), pred)], classes, valid, 2,
                                                      options=opts, groups={})
    assert (insts[0] >= 0).any() and (insts[0] == -1).any() and set(np.unique(sems[0][insts[0] >= 0])) <= set(valid[2:])
    assert m_ref["all_ap_25%"] > 0.05                                      # not a trivially-zero operating point
    print("evaluator-level mAP oracle / HIP:", {k: (round(float(m_ref[k]), 5), round(float(m_hip[k]), 5)) for k in ("all_ap", "all_ap_50%", "all_ap_25%")})
    for key in ("all_ap", "all_ap_50%", "all_ap_25%"):
        assert abs(m_hip[key] - m_ref[key]) < 1e-3, (key, m_hip[key], m_ref[key])
