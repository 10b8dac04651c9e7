"""Pins the oracle (oracle/*.py) to golden vectors captured from the imported reference
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from _det import det_param
from oracle import decoder_ref as D
from oracle import postprocess_ref as P

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: torch.from_numpy(z[k]) if z[k].dtype != object else z[k] for k in z.files}


def decoder_state_dict(in_channels=96, n_inst=198, n_sem=200, L=6, d=256, hidden=1024, size_embed_scale=1.0, fourier=False):
    """Key names/shapes of the reference decoder state_dict; values by name."""
    shapes = {}

    def lin(name, o, i):
        shapes[name + ".weight"] = (o, i)
        shapes[name + ".bias"] = (o,)

    def ln(name):
        shapes[name + ".weight"] = (d,)
        shapes[name + ".bias"] = (d,)

    lin("input_proj.0", d, in_channels); ln("input_proj.1")
    lin("query_proj.0", d, in_channels); lin("query_proj.2", d, d)
    lin("x_mask.0", d, in_channels); lin("x_mask.2", d, d)
    ln("out_norm"); lin("out_cls.0", d, d); lin("out_cls.2", n_inst + 1, d); lin("out_sem", n_sem + 1, d)
    lin("ca_qpos_proj", d, d)
    for n in ("ref_point_head",):
        lin(n + ".layers.0", d, d); lin(n + ".layers.1", d, d)
    if fourier:                                # pos_type="fourier": Gaussian projection buffer, no box-modulation head (:528)
        shapes["position_embedding.gauss_B"] = (3, d // 2)
    else:
        lin("ref_anchor_head.layers.0", d, d); lin("ref_anchor_head.layers.1", 3, d)
    for i in range(L):
        lin(f"cross_attn_layers.{i}.out_proj", d, d)
        lin(f"self_attn_layers.{i}.out_proj", d, d)
        shapes[f"dinox_query_cross_attn_layers.{i}.attn.in_proj_weight"] = (3 * d, d)
        shapes[f"dinox_query_cross_attn_layers.{i}.attn.in_proj_bias"] = (3 * d,)
        lin(f"dinox_query_cross_attn_layers.{i}.attn.out_proj", d, d)
        ln(f"dinox_query_cross_attn_layers.{i}.norm")
        lin(f"ffn_layers.{i}.net.0", hidden, d); lin(f"ffn_layers.{i}.net.3", d, hidden); ln(f"ffn_layers.{i}.norm")
        for n in ("ca_qcontent_proj", "ca_kcontent_proj", "ca_kpos_proj", "ca_v_proj", "ca_qpos_sine_proj",
                  "sa_qcontent_proj", "sa_qpos_proj", "sa_kcontent_proj", "sa_kpos_proj", "sa_v_proj"):
            lin(f"{n}.{i}", d, d)
        ln(f"norm1.{i}"); ln(f"norm2.{i}")
        for n in ("bbox_embed", "bbox_size_embed"):
            lin(f"{n}.{i}.layers.0", d, d); lin(f"{n}.{i}.layers.1", d, d); lin(f"{n}.{i}.layers.2", 3, d)
    sd = {"decoder." + k: det_param("decoder." + k, s) for k, s in shapes.items()}
    if size_embed_scale != 1.0:          # mirrors tests/golden/make_golden.py (ScanNetv2 variant)
        for i in range(L):
            for leaf in ("weight", "bias"):
                sd[f"decoder.bbox_size_embed.{i}.layers.2.{leaf}"] *= size_embed_scale
    return sd


def test_sine_pe():
    g = load("pe_sine")
    out = D.sine_pe(g["xyz"][0], g["lo"][0], g["hi"][0], 20.0, 256)
    torch.testing.assert_close(out, g["out_plain"][0], rtol=1e-5, atol=1e-5)
    out = D.sine_pe(g["xyz"][0], g["lo"][0], g["hi"][0], 20.0, 256, modulated=g["modulated"][0])
    torch.testing.assert_close(out, g["out_modulated"][0], rtol=1e-5, atol=1e-5)
    assert D.pe_channel_plan(256, 3) == [86, 86, 84]


@pytest.mark.parametrize("name,kw,cfgkw", [
    ("decoder_s64_q64", {}, {}),
    ("decoder_s96_q16", {}, {}),
    ("decoder_s500_q32", {}, {}),
    ("decoder_v2_s48", dict(in_channels=32, n_inst=18, n_sem=20, size_embed_scale=0.05), dict(normalize_box_prediction=False)),
    ("decoder_fourier_s48", dict(fourier=True), dict(pos_type="fourier", box_modulate_ca=False)),
])
def test_decoder_matches_reference(name, kw, cfgkw):
    g = load(name)
    sd = decoder_state_dict(**kw)
    cfg = D.DecoderCfg(**cfgkw)
    ids = g["query_ids"].long()
    out = D.decoder_forward(sd, cfg, g["x"], g["pos"], g["pos_wo"], g["x"][ids], g["pos"][ids],
                            g["q2d_feat"], g["q2d_pos"], g["lo"], g["hi"])
    tol = dict(rtol=2e-4, atol=2e-4)
    for li in range(6):
        torch.testing.assert_close(out["aux"][li]["cls_preds"], g[f"aux{li}_cls"], **tol)
        torch.testing.assert_close(out["aux"][li]["masks"], g[f"aux{li}_masks"], **tol)
        if li > 0:
            torch.testing.assert_close(out["aux"][li]["centers"], g[f"aux{li}_centers"], **tol)
            torch.testing.assert_close(out["aux"][li]["sizes"], g[f"aux{li}_sizes"], **tol)
    for k in ("cls_preds", "sem_preds", "masks", "centers", "sizes", "hidden_states"):
        torch.testing.assert_close(out[k], g[k], **tol)


def test_plain_decoder_with_learned_queries_and_objectness_matches_reference():
    """`num_semantic_queries = 7` learned queries prepended to the projected ones + the `out_score` head (objectness_flag)."""
    g = load("decoder_plain_obj_s40")
    sd = plain_decoder_state_dict(n_learned=7, objectness=True)
    cfg = D.DecoderCfg(add_positional_embedding=False, add_dinox_query_ca=False, add_box_size_pred=False,
                       box_modulate_ca=False, normalize_box_prediction=False)
    out = D.decoder_forward(sd, cfg, g["x"], None, None, g["x"], None, None, None, None, None)
    tol = dict(rtol=2e-4, atol=2e-4)
    assert out["masks"].shape == (47, 40) and out["scores"].shape == (47, 1)
    for li in (0, 2, 4):
        for k, gk in (("cls_preds", "cls"), ("masks", "masks"), ("scores", "scores")):
            torch.testing.assert_close(out["aux"][li][k], g[f"aux{li}_{gk}"], **tol)
    for k in ("cls_preds", "sem_preds", "masks", "scores", "hidden_states"):
        torch.testing.assert_close(out[k], g[k], **tol)


def test_matrix_nms():
    g = load("matrix_nms")
    s, l, m, keep, rec = P.matrix_nms(g["masks"], g["labels"], g["scores"], kernel="linear")
    torch.testing.assert_close(s, g["out_scores"], rtol=1e-5, atol=1e-6)
    assert torch.equal(l, g["out_labels"]) and torch.equal(keep, g["out_keep"]) and torch.equal(rec, g["out_record"])
    torch.testing.assert_close(m, g["out_masks"])


@pytest.mark.parametrize("name,query_num,box", [("arch_qall", -1, True), ("arch_q40", 40, True),
                                                ("arch_qall_nobox", -1, False), ("arch_qall_widebox", -1, True),
                                                ("arch_q40_widebox", 40, True)])
def test_architecture_eval_path(name, query_num, box):
    g = load(name)
    sd = decoder_state_dict()
    for i in range(6):                                # *_widebox: grown boxes, the out-of-box filter keeps ~28 % of the mask points
        sd[f"decoder.bbox_size_embed.{i}.layers.2.bias"] = sd[f"decoder.bbox_size_embed.{i}.layers.2.bias"] + float(g["size_bias"])
    cfg = D.DecoderCfg()
    pts = g["points"]
    lo, hi, centers, sizes = P.scene_range_and_gt_boxes(pts[:, :3], g["gt_masks"], "median")
    torch.testing.assert_close(centers, g["instance_centers"])
    torch.testing.assert_close(sizes, g["instance_sizes"])
    q, qpos, ids = D.select_queries(sd, g["sp_feat"], g["sp_pos"], query_num)
    out = D.decoder_forward(sd, cfg, g["sp_feat"], g["sp_pos"], g["sp_pos"], q, qpos, g["q2d_feat"],
                            g["q2d_pos"], lo, hi)
    res = P.predict_by_feat(out, g["superpoints"].long(), pts[:, :3], 198, P.TestCfg(), box, query_num)
    n = int(g["n_points"])
    ref_masks = torch.from_numpy(np.unpackbits(g["inst_masks_packed"].numpy(), axis=1)[:, :n].astype(bool))
    assert res["pts_instance_mask"][0].shape == ref_masks.shape
    torch.testing.assert_close(res["instance_scores"], g["inst_scores"], rtol=1e-4, atol=1e-6)
    assert torch.equal(res["instance_labels"], g["inst_labels"])
    diff = (res["pts_instance_mask"][0] != ref_masks).float().mean().item()
    assert diff < 1e-4, diff
    torch.testing.assert_close(res["instance_boxes"], g["inst_boxes"].float(), rtol=1e-4, atol=1e-4)
    assert (res["pts_semantic_mask"][0] != g["sem_mask"]).float().mean() < 1e-3
    assert (res["pts_semantic_mask"][1] != g["pan_sem"]).float().mean() < 1e-3
    assert (res["pts_instance_mask"][1] != g["pan_inst"]).float().mean() < 1e-3
    assert torch.equal(torch.sort(res["sort_and_mask"][0])[0], torch.sort(g["topk_idx"])[0])


def plain_decoder_state_dict(in_channels=96, n_inst=198, n_sem=200, L=6, d=256, hidden=1024, n_learned=0, objectness=False):
    """state_dict of the non-positional decoder variant (Baseline_ScanNet200 prototype)."""
    shapes = {}

    def lin(name, o, i):
        shapes[name + ".weight"] = (o, i)
        shapes[name + ".bias"] = (o,)

    def ln(name):
        shapes[name + ".weight"] = (d,)
        shapes[name + ".bias"] = (d,)

    lin("input_proj.0", d, in_channels); ln("input_proj.1")
    lin("query_proj.0", d, in_channels); lin("query_proj.2", d, d)
    lin("x_mask.0", d, in_channels); lin("x_mask.2", d, d)
    ln("out_norm"); lin("out_cls.0", d, d); lin("out_cls.2", n_inst + 1, d); lin("out_sem", n_sem + 1, d)
    if n_learned:
        shapes["query.weight"] = (n_learned, d)          # nn.Embedding of the learned queries (:229-231)
    if objectness:
        lin("out_score.0", d, d); lin("out_score.2", 1, d)
    for i in range(L):
        for n in ("cross_attn_layers", "self_attn_layers"):
            shapes[f"{n}.{i}.attn.in_proj_weight"] = (3 * d, d)
            shapes[f"{n}.{i}.attn.in_proj_bias"] = (3 * d,)
            lin(f"{n}.{i}.attn.out_proj", d, d)
            ln(f"{n}.{i}.norm")
        lin(f"ffn_layers.{i}.net.0", hidden, d); lin(f"ffn_layers.{i}.net.3", d, hidden); ln(f"ffn_layers.{i}.norm")
    return {"decoder." + k: det_param("decoder." + k, s) for k, s in shapes.items()}


def test_plain_decoder_matches_reference():
    g = load("decoder_plain_s40")
    sd = plain_decoder_state_dict()
    cfg = D.DecoderCfg(add_positional_embedding=False, add_dinox_query_ca=False, add_box_size_pred=False,
                       box_modulate_ca=False, normalize_box_prediction=False)
    out = D.decoder_forward(sd, cfg, g["x"], None, None, g["x"], None, None, None, None, None)
    tol = dict(rtol=2e-4, atol=2e-4)
    # reference quirk: without positional embedding `pred_centers` is one entry shorter than `cls_preds`
    # (:653-655 vs :758), and the zip at :781-783 therefore yields 5 aux outputs instead of 6
    assert "aux5_cls" not in g
    for li in range(5):
        torch.testing.assert_close(out["aux"][li]["cls_preds"], g[f"aux{li}_cls"], **tol)
        torch.testing.assert_close(out["aux"][li]["masks"], g[f"aux{li}_masks"], **tol)
    for k in ("cls_preds", "sem_preds", "masks", "hidden_states"):
        torch.testing.assert_close(out[k], g[k], **tol)


# ------------------------------------------------------------------------------------------------
# ScanNet instance-AP protocol: the oracle restatement against the reference's own output
# ------------------------------------------------------------------------------------------------
def _ap_fixture():
    import json
    z = np.load(os.path.join(GOLDEN, "ap_protocol.npz"))
    class_labels = tuple(str(c) for c in z["class_labels"])
    valid = tuple(int(v) for v in z["valid_class_ids"])
    scenes = []
    for si in range(int(z["n_scenes"])):
        n = int(z[f"s{si}_n"])
        masks = np.unpackbits(z[f"s{si}_masks"], axis=1)[:, :n].astype(bool)
        scenes.append((z[f"s{si}_sem"], z[f"s{si}_inst"], masks, z[f"s{si}_labels"], z[f"s{si}_scores"]))
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(here, "segdino3d_amd", "data", "scannet200_groups.json")) as f:
        groups = json.load(f)
    return z, class_labels, valid, scenes, groups


@pytest.mark.parametrize("opt_name,options", [("default", None), ("min30", dict(min_region_sizes=np.array([30])))])
def test_ap_protocol_oracle_matches_reference_golden(opt_name, options):
    from oracle import eval_ref as E
    z, class_labels, valid, scenes, groups = _ap_fixture()
    id_to_label = {valid[i]: class_labels[i] for i in range(len(valid))}
    preds = E.aggregate_predictions([s[2] for s in scenes], [s[3] for s in scenes], [s[4] for s in scenes], valid)
    gts = E.rename_gt([s[0] for s in scenes], [s[1] for s in scenes], valid)
    metrics, ap, pr_rc = E.scannet_eval(preds, gts, options, valid, class_labels, id_to_label, groups)
    # association records: integer work, exact
    opts = E.get_options(options)
    rows = []
    for si, (p, gt) in enumerate(zip(preds, gts)):
        g2p, p2g = E.assign_instances(p, gt, opts, valid, class_labels, id_to_label)
        for label in class_labels:
            for pr in p2g[label]:
                for m in pr["matched_gt"]:
                    rows.append([si, int(pr["filename"].split("_")[1]), pr["label_id"], pr["vert_count"], pr["void_intersection"],
                                 m["instance_id"], m["vert_count"], m["intersection"]])
                if not pr["matched_gt"]:
                    rows.append([si, int(pr["filename"].split("_")[1]), pr["label_id"], pr["vert_count"], pr["void_intersection"], -1, 0, 0])
    assert np.array_equal(np.array(sorted(rows), dtype=np.int64), z[f"{opt_name}_assoc"])
    # averages: same floating-point operations in the same order -> 1e-12
    for k, v in zip(z[f"{opt_name}_keys"], z[f"{opt_name}_vals"]):
        got = metrics[str(k)]
        assert (np.isnan(got) and np.isnan(v)) or abs(got - v) < 1e-12, (k, got, v)
    cls = np.array([[metrics["classes"][c][f] for f in ("ap", "ap50%", "ap25%", "prec50%", "rec50%")] for c in class_labels])
    assert np.allclose(cls, z[f"{opt_name}_class_ap"], rtol=0, atol=1e-12, equal_nan=True)


# ------------------------------------------------------------------------------------------------
# evaluator level (evaluation/evaluator_3d.py:124-219, 323-349): panoptic-style ground truth -> map_inst_markup -> instance AP
# ------------------------------------------------------------------------------------------------
def _evaluator_fixture():
    z = np.load(os.path.join(GOLDEN, "evaluator_scannet.npz"))
    classes = tuple(str(c) for c in z["classes"])
    valid = tuple(int(v) for v in z["valid_class_ids"])
    results = []
    for si in range(4):
        n = int(z[f"s{si}_n"])
        masks = np.unpackbits(z[f"s{si}_masks"], axis=1)[:, :n].astype(bool)
        results.append((dict(pts_semantic_mask=z[f"s{si}_sem"], pts_instance_mask=z[f"s{si}_inst"]),
                        dict(pts_instance_mask=[masks], instance_labels=z[f"s{si}_labels"], instance_scores=z[f"s{si}_scores"])))
    return z, classes, valid, int(z["num_stuff_cls"]), results


def test_evaluator_metrics_oracle_matches_reference_golden():
    """oracle.eval_ref.evaluator_instance_metrics against what the reference's InstanceSeg3DEvaluator.compute_metrics computed on
    the same per-scene results (tests/golden/make_golden_evaluator.py): the mapped ground truth exactly, the metrics to 1e-12."""
    import json
    from oracle import eval_ref as E
    z, classes, valid, n_stuff, results = _evaluator_fixture()
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(here, "segdino3d_amd", "data", "scannet200_groups.json")) as f:
        groups = json.load(f)
    metrics, sems, insts = E.evaluator_instance_metrics(results, classes, valid, n_stuff, groups=groups)
    for si in range(4):
        assert np.array_equal(sems[si], z[f"s{si}_mapped_sem"]) and np.array_equal(insts[si], z[f"s{si}_mapped_inst"])
    for k, v in zip(z["keys"], z["vals"]):
        got = metrics[str(k)]
        assert (np.isnan(got) and np.isnan(v)) or abs(got - v) < 1e-12, (k, got, v)
    cls = np.array([[metrics["classes"][c][f] for f in ("ap", "ap50%", "ap25%")] for c in classes[n_stuff:-1]])
    assert np.allclose(cls, z["class_ap"], rtol=0, atol=1e-12, equal_nan=True)
