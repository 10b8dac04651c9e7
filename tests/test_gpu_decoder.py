"""GPU parity tests of the decoder / post-processing kernels: against the oracle on seeded inputs and
DIRECTLY against the golden vectors captured from the imported reference (tests/golden/*.npz).
fp32 tolerance is written in each test; thresholded outputs are compared as a fraction of bits."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from _det import det_param, det_randn  # noqa: E402
from test_oracle_golden import decoder_state_dict, load  # noqa: E402

DEC_KW = dict(add_dinox_query_ca=True, add_dinox_query_ca_mask=True, dinox_query_ca_mask_threshold=0.2, num_layers=6,
              num_instance_queries=0, num_semantic_queries=0, num_instance_classes=198, num_semantic_classes=200,
              num_semantic_linears=1, in_channels=96, d_model=256, num_heads=8, hidden_dim=1024, dropout=0.0,
              activation_fn="gelu", iter_pred=True, attn_mask=True, fix_attention=True, objectness_flag=False,
              add_box_size_pred=True, add_positional_embedding=True, pos_type="sine", temperature=20,
              box_modulate_ca=True, normalize_box_prediction=True)
TEST_CFG = dict(topk_insts=600, inst_score_thr=0.0, pan_score_thr=0.5, npoint_thr=100, obj_normalization=True,
                sp_score_thr=0.4, nms=True, matrix_nms_kernel="linear", stuff_classes=[0, 1])


def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def unpack(bits, n):
    """int32 [R, W] bit words -> bool [R, n]"""
    b = bits.cpu().numpy().view(np.uint32)
    out = ((b[:, :, None] >> np.arange(32, dtype=np.uint32)[None, None, :]) & 1).astype(bool)
    return torch.from_numpy(out.reshape(b.shape[0], -1)[:, :n])


def test_sine_pe_matches_reference_golden():
    from segdino3d_amd.decoder import ScanNetQueryDecoder
    d = dev()
    g = load("pe_sine")
    dec = ScanNetQueryDecoder(**DEC_KW)
    dim_t, axis = dec.pe_tables(d)
    from segdino3d_amd import ops
    rng = torch.cat([g["lo"][0], g["hi"][0]]).to(d)
    out = ops.sine_pe(g["xyz"][0].to(d), rng, dim_t, axis).cpu()
    torch.testing.assert_close(out, g["out_plain"][0], rtol=1e-4, atol=2e-5)
    ones = torch.ones(17, 3, device=d)
    out = ops.sine_pe(g["xyz"][0].to(d), rng, dim_t, axis, mod_num=g["modulated"][0].to(d), mod_den=ones).cpu()
    torch.testing.assert_close(out, g["out_modulated"][0], rtol=1e-4, atol=2e-5)


def test_layernorm_matches_torch():
    from segdino3d_amd import ops
    d = dev()
    for M, D in ((200, 256), (37, 1024), (5, 96)):
        x, r = det_randn(f"ln.x{M}", (M, D)), det_randn(f"ln.r{M}", (M, D))
        w, b = 1 + det_randn("ln.w", (D,), 0.1), det_randn("ln.b", (D,), 0.1)
        y = ops.layernorm(x.to(d), w.to(d), b.to(d), res=r.to(d), act="relu").cpu()
        ref = torch.relu(torch.nn.functional.layer_norm(x + r, (D,), w, b, 1e-5))
        torch.testing.assert_close(y, ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("Lq,Lk,nsrc,masked", [(200, 3000, 2, True), (64, 64, 1, False), (33, 311, 1, True),
                                               (16, 8, 1, True), (300, 1000, 2, False)])
def test_attention_matches_oracle(Lq, Lk, nsrc, masked):
    from oracle.decoder_ref import attention_core
    from segdino3d_amd import ops
    d = dev()
    H = 8
    q = det_randn(f"at.q{Lq}", (Lq, 256)); k = det_randn(f"at.k{Lk}", (Lk, 256)); v = det_randn(f"at.v{Lk}", (Lk, 256))
    q2 = det_randn(f"at.q2{Lq}", (Lq, 256)); k2 = det_randn(f"at.k2{Lk}", (Lk, 256))
    blocked = None
    bits = None
    if masked:
        blocked = det_randn(f"at.m{Lq}{Lk}", (Lq, Lk)) > 0.3
        blocked[:, : min(40, Lk - 1)] = True            # fully blocked leading key tile(s) for every query
        blocked[0] = True; blocked[0, Lk - 1] = False     # a query with a single open key (the last one)
        blocked[torch.arange(Lq), torch.arange(Lq) % Lk] = False
        nw = (Lk + 31) // 32
        pad = torch.ones(Lq, nw * 32, dtype=torch.bool); pad[:, :Lk] = blocked
        words = (pad.view(Lq, nw, 32).long() << torch.arange(32)).sum(-1)
        bits = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32).to(d)
    if nsrc == 2:
        qc = torch.cat([q.view(Lq, H, 32), q2.view(Lq, H, 32)], 2).reshape(Lq, 512)
        kc = torch.cat([k.view(Lk, H, 32), k2.view(Lk, H, 32)], 2).reshape(Lk, 512)
        ref = attention_core(qc, kc, v, H, blocked)
        out = ops.attention(q.to(d), k.to(d), v.to(d), H, 64 ** -0.5, mask_bits=bits, q2=q2.to(d), k2=k2.to(d)).cpu()
    else:
        ref = attention_core(q, k, v, H, blocked)
        out = ops.attention(q.to(d), k.to(d), v.to(d), H, 32 ** -0.5, mask_bits=bits).cpu()
    torch.testing.assert_close(out, ref, rtol=2e-4, atol=2e-4)


def test_mask_and_distance_bits_match_oracle():
    from oracle import decoder_ref as D
    from segdino3d_amd import ops
    d = dev()
    Q, S, M = 70, 333, 45
    logits = det_randn("mb.logits", (Q, S), 2.0)
    logits[3] = -5.0                                    # dead row -> reset to all-open
    bits = ops.mask_bits(logits.to(d), S, 0.5)
    blocked = torch.sigmoid(logits) < 0.5
    blocked[blocked.all(1)] = False
    assert torch.equal(unpack(bits, S), blocked)
    assert unpack(bits, ((S + 31) // 32) * 32)[:, S:].all()
    pos = det_randn("mb.pos", (S, 3)).sigmoid() * 2
    ctr = pos[:M] + det_randn("mb.ctr", (M, 3), 0.1)
    near = ops.near_bits(pos.to(d), ctr.to(d), 0.2)
    assert torch.equal(unpack(near, S), (torch.cdist(pos, ctr, p=1) < 0.2).t())
    b2 = ops.dinox_mask_bits(bits, near)
    ref = D.dinox_blocked_mask(~blocked, pos, ctr, 0.2)
    assert torch.equal(unpack(b2, M + 1), ref)
    assert unpack(b2, ((M + 1 + 31) // 32) * 32)[:, M + 1:].all()


@pytest.mark.parametrize("Q,S,M", [(2501, 3000, 300), (2048, 333, 45), (3, 70, 1), (201, 4100, 95), (9, 64, 64)])
def test_dinox_mask_bits_sizes(Q, S, M):
    """sd3d_dinox_mask_bits at one query per superpoint (eight queries per workgroup) and at odd sizes (two per workgroup, ragged last
    workgroup, keys filling whole output words, more than 4096 superpoints), single call and the batched entry, against the boolean
    product of instance_seg_3d_decoder.py:721-726."""
    from segdino3d_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(Q + S)
    open_ = torch.rand(Q, S, generator=g) < 0.03
    open_[0] = False                                           # a query with nothing open: every real key blocked, the dummy key open
    near_b = torch.rand(M, S, generator=g) < 0.02
    W = (S + 31) // 32

    def pack(m, fill):                                          # bool [R, S] -> int32 words [R, W], padding bits = fill
        full = torch.full((m.shape[0], W * 32), fill, dtype=torch.bool)
        full[:, :S] = m
        w = (full.view(m.shape[0], W, 32).long() << torch.arange(32)).sum(-1)
        return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)
    blocked_bits, near_bits = pack(~open_, True).to(d), pack(near_b, False).to(d)
    ref = torch.cat([(open_.float() @ near_b.float().t()) == 0, torch.zeros(Q, 1, dtype=torch.bool)], dim=1)
    got = ops.dinox_mask_bits(blocked_bits, near_bits)
    assert torch.equal(unpack(got, M + 1), ref)
    assert unpack(got, got.shape[1] * 32)[:, M + 1:].all()
    half = Q // 2 + 1
    two = ops.dinox_mask_bits_batch([blocked_bits[:half].contiguous(), blocked_bits[half:].contiguous()], [near_bits, near_bits])
    assert torch.equal(torch.cat(two), got)


def _build_decoder(kw_over=None, sd_kw=None):
    from segdino3d_amd.decoder import ScanNetQueryDecoder
    kw = dict(DEC_KW); kw.update(kw_over or {})
    dec = ScanNetQueryDecoder(**kw).eval()
    sd = decoder_state_dict(**(sd_kw or {}))
    dec.load_state_dict({k[len("decoder."):]: v for k, v in sd.items()})
    return dec, sd


def _mask_agree(a, b, thr=0.0):
    return ((a > thr) == (b > thr)).float().mean().item()


# The three evaluation paths of the decoder (segdino3d_amd/decoder.py): op by op (`_forward_scene`), the row-chain path on 16-row
# tiles (csrc/rowchain.hip) and on 4-row tiles (csrc/rowchain_narrow.hip).  Every reference golden runs on all three.
DECODER_PATHS = ["op_by_op", "rowchain16", "rowchain4"]


def _set_decoder_path(monkeypatch, path, dec, rows):
    """Force one decoder path; returns False when the decoder variant has no row-chain path (it runs op by op whatever the switch)."""
    from segdino3d_amd import decoder as D
    monkeypatch.setattr(D, "FUSED_DECODER", path != "op_by_op")
    monkeypatch.setattr(D, "FUSED_NARROW", path == "rowchain4")
    tile = dec._fusable(rows)
    if path == "op_by_op":
        assert tile == 0
        return True
    if tile == 0:
        return False
    assert tile == (16 if path == "rowchain16" else 4), (path, tile)
    return True


# largest accepted fraction of query rows outside the tolerance at a layer after the first mask feedback, per fixture: the
# measured value is 0 rows on every fixture (profiles/r02_parity_numbers.md); a fixture whose near-zero logit flips a mask bit
# on some future kernel change would list its measured fraction here
BAD_ROWS_MAX = {"decoder_s64_q64": 0.0, "decoder_s96_q16": 0.0, "decoder_s500_q32": 0.0, "decoder_v2_s48": 0.0, "decoder_fourier_s48": 0.0}


@pytest.mark.parametrize("name,kw,sdkw", [
    ("decoder_s64_q64", {}, {}), ("decoder_s96_q16", {}, {}), ("decoder_s500_q32", {}, {}),
    ("decoder_v2_s48", dict(num_instance_classes=18, num_semantic_classes=20, in_channels=32, normalize_box_prediction=False),
     dict(in_channels=32, n_inst=18, n_sem=20, size_embed_scale=0.05)),
    ("decoder_fourier_s48", dict(pos_type="fourier", box_modulate_ca=False), dict(fourier=True))])
@pytest.mark.parametrize("path", DECODER_PATHS)
def test_decoder_matches_reference_golden(name, kw, sdkw, path, monkeypatch):
    d = dev()
    g = load(name)
    dec, _ = _build_decoder(kw, sdkw)
    dec.to(d)
    ids = g["query_ids"].long()
    if not _set_decoder_path(monkeypatch, path, dec, len(ids)):
        pytest.skip(f"{name}: this decoder variant has no row-chain path (it runs op by op, covered by the op_by_op case)")
    t = lambda x: x.to(d)
    out = dec([t(g["x"])], [t(g["pos"])], [t(g["pos_wo"])], [t(g["x"][ids])], [t(g["pos"][ids])], [t(g["q2d_feat"])],
              [t(g["q2d_pos"])], [(t(g["lo"]), t(g["hi"]))])
    # The decoder is discontinuous (sigmoid < 0.5 attention masks, :569): a single near-zero logit that
    # rounds differently flips a mask bit and that query row then legitimately diverges in later layers.
    # Measured on MI355X (round 2, printed below): NO row outside 2e-3 at any layer of any fixture, max abs error 1e-4 on
    # mask logits of magnitude ~20, 8e-6 elsewhere, all mask bits equal.  Tolerance: 3e-4 (abs + rel) on EVERY row.
    tol = 3e-4
    report = []

    def rows_ok(got, ref, what, strict):
        err = (got.cpu() - ref).abs()
        lim = tol + tol * ref.abs()
        bad = (err > lim).any(dim=1).float().mean().item()
        report.append(f"{what}: {bad:.1%} rows / max err {err.max().item():.1e}")
        assert bad <= (0.0 if strict else BAD_ROWS_MAX[name]), f"{what}: {bad:.1%} of query rows outside tolerance (max err {err.max().item():.3e})"
    for li in range(6):
        aux = out["aux_outputs"][li]
        rows_ok(aux["cls_preds"][0], g[f"aux{li}_cls"], f"aux{li} cls", li <= 1)
        rows_ok(aux["masks"][0], g[f"aux{li}_masks"], f"aux{li} masks", li <= 1)
        if li > 0:
            rows_ok(aux["centers"][0], g[f"aux{li}_centers"], f"aux{li} centers", li <= 1)
            rows_ok(aux["sizes"][0], g[f"aux{li}_sizes"], f"aux{li} sizes", li <= 1)
    for k in ("cls_preds", "sem_preds", "masks", "centers", "sizes", "hidden_states"):
        rows_ok(out[k][0], g[k], k, False)
    agree = _mask_agree(out["masks"][0].cpu(), g["masks"])
    print(f"{name}: rows outside {tol:g} (abs + rel) / max abs error per tensor:\n  " + "\n  ".join(report) + f"\n  final mask bits equal: {agree:.5f}")
    assert agree >= 0.9999


def test_decoder_matches_oracle_at_benchmark_shape():
    """S = 1000 superpoints, Q = 200 queries, M = 150: the tiled / multi-wave paths of the kernels."""
    from oracle import decoder_ref as D
    d = dev()
    dec, sd = _build_decoder()
    dec.to(d)
    S, Q, M = 1000, 200, 150
    room = torch.tensor([8.0, 6.0, 3.0])
    pos = torch.floor(det_randn("big.pos", (S, 3)).sigmoid() * room / 0.02) * 0.02
    x = det_randn("big.x", (S, 96))
    q2d_pos = pos[:M] + det_randn("big.q2dpos", (M, 3), 0.1)
    q2d_feat = det_randn("big.q2dfeat", (M, 256))
    lo, hi = pos.min(0)[0] - 0.03, pos.max(0)[0] + 0.05
    ids = torch.arange(0, S, S // Q)[:Q]
    ref = D.decoder_forward(sd, D.DecoderCfg(), x, pos, pos, x[ids], pos[ids], q2d_feat, q2d_pos, lo, hi)
    t = lambda a: a.to(d)
    out = dec([t(x)], [t(pos)], [t(pos)], [t(x[ids])], [t(pos[ids])], [t(q2d_feat)], [t(q2d_pos)], [(t(lo), t(hi))])
    # thresholded attention masks make the map discontinuous: allow a handful of rows to diverge
    # measured (round 2): max abs error 5e-5 on mask logits, 5e-6 elsewhere, no row outside, all mask bits equal
    for k, tol in (("cls_preds", 5e-5), ("masks", 5e-4), ("centers", 5e-5), ("sizes", 5e-5)):
        err = (out[k][0].cpu() - ref[k]).abs()
        bad_rows = (err.amax(dim=1) > tol * max(1.0, ref[k].abs().max().item())).float().mean().item()
        print(f"S=1000 Q=200 {k}: {bad_rows:.3%} of query rows outside {tol:g}, max err {err.max().item():.3e}")
        assert bad_rows == 0.0, f"{k}: {bad_rows:.3%} of query rows differ (max err {err.max().item():.3e})"
    print(f"S=1000 Q=200 mask bits equal: {_mask_agree(out['masks'][0].cpu(), ref['masks']):.6f}")
    assert _mask_agree(out["masks"][0].cpu(), ref["masks"]) >= 0.9999


class _StoredBackbone(torch.nn.Module):
    """Test double: returns stored superpoint features (the golden architecture fixtures were captured
    with the same stand-in, tests/golden/make_golden.py)."""
    voxel_size = 0.02

    def __init__(self, **kw):
        super().__init__()
        self.f = self.p = None

    def forward_wrapper(self, samples, targets, return_sp_mean_pos=True):
        return [self.f.clone()], [self.p.clone()], [self.p.clone()]


@pytest.mark.parametrize("name,query_num,box", [("arch_qall", -1, True), ("arch_q40", 40, True), ("arch_qall_nobox", -1, False),
                                                ("arch_qall_widebox", -1, True), ("arch_q40_widebox", 40, True)])
@pytest.mark.parametrize("path", DECODER_PATHS)
def test_architecture_matches_reference_golden(name, query_num, box, path, monkeypatch):
    """The *_widebox fixtures grow the predicted boxes (a bias on the size heads, make_golden.golden_architecture) so that
    filter_outofbox_points keeps ~28 % of the mask points instead of < 1 %: the filter is exercised on real content.
    Runs on every decoder path (op by op, row chain on 16-row and on 4-row tiles)."""
    import segdino3d_amd as seg
    from segdino3d_amd.gtypes import GD3DTarget
    d = dev()
    if seg.BACKBONES.get("_StoredBackbone") is None:
        seg.BACKBONES.register_module(module=_StoredBackbone)
    g = load(name)
    model = seg.build_architecture(dict(
        type="Baseline3D", num_classes=198, pointcloud_backbone_cfg=dict(type="_StoredBackbone"),
        decoder_cfg=dict(type="ScanNetQueryDecoder", **DEC_KW), criterion_cfg=None, query_thr=0.5, test_cfg=TEST_CFG,
        add_positional_embedding=True, mode_3d_center="median", query_num=query_num, filter_outofbox_points_eval=box)).eval()
    sd = decoder_state_dict()
    for i in range(6):
        sd[f"decoder.bbox_size_embed.{i}.layers.2.bias"] = sd[f"decoder.bbox_size_embed.{i}.layers.2.bias"] + float(g["size_bias"])
    model.decoder.load_state_dict({k[len("decoder."):]: v for k, v in sd.items()})
    model.to(d)
    model.backbone.f, model.backbone.p = g["sp_feat"].to(d), g["sp_pos"].to(d)
    n_rows = g["sp_feat"].shape[0] if query_num < 0 else query_num
    assert _set_decoder_path(monkeypatch, path, model.decoder, n_rows), "the SegDINO3D prototype decoder has a row-chain path"
    tgt = GD3DTarget(masks=g["gt_masks"], extra_features=dict(super_point_masks=g["superpoints"].long(),
                     query2d_feats=g["q2d_feat"], query2d_pos=g["q2d_pos"])).to(d)
    res = model([g["points"].to(d)], [tgt])
    pd = res[0].pred_pts_seg
    torch.testing.assert_close(res[0].instance_centers.cpu(), g["instance_centers"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(res[0].instance_sizes.cpu(), g["instance_sizes"], rtol=1e-5, atol=1e-5)
    n = int(g["n_points"])
    ref_masks = np.unpackbits(g["inst_masks_packed"].numpy(), axis=1)[:, :n].astype(bool)
    got_masks = pd.pts_instance_mask[0]
    assert got_masks.shape == ref_masks.shape, (got_masks.shape, ref_masks.shape)
    ref_scores, got_scores = g["inst_scores"].numpy(), pd.instance_scores
    # Rows are ordered by score and the fixture's neighbouring scores are ~1e-6 apart (relative): ONE thresholded quantity that flips
    # (a superpoint whose mask sigmoid sits on 0.5 changes a row's mask score by ~1e-2) moves that row past dozens of others and
    # shifts every row in between by one.  So the rows are ALIGNED first (longest common subsequence over (label, point mask)), a
    # displaced row costs one row, and everything is compared on the aligned pairs.
    import difflib
    from collections import Counter
    ref_keys = [(int(l), m.tobytes()) for l, m in zip(g["inst_labels"].numpy(), np.packbits(ref_masks, axis=1))]
    got_keys = [(int(l), m.tobytes()) for l, m in zip(pd.instance_labels, np.packbits(got_masks, axis=1))]
    blocks = difflib.SequenceMatcher(a=ref_keys, b=got_keys, autojunk=False).get_matching_blocks()
    ia = np.array([i for blk in blocks for i in range(blk.a, blk.a + blk.size)], dtype=np.int64)
    ib = np.array([i for blk in blocks for i in range(blk.b, blk.b + blk.size)], dtype=np.int64)
    aligned = len(ia) / max(1, len(ref_keys))
    twins = sum((Counter(ref_keys) & Counter(got_keys)).values()) / max(1, ref_masks.shape[0])
    in_place = (pd.instance_labels == g["inst_labels"].numpy()).mean()
    bits_in_place = float((got_masks == ref_masks).mean())
    print(f"{name} [{path}]: {int(ref_masks.sum())} mask points in the reference, {int(got_masks.sum())} here; rows aligned in order {aligned:.4f}, "
          f"with an identical (label, mask) twin anywhere {twins:.4f}, equal labels in place {in_place:.4f}, mask bits equal in place {bits_in_place:.6f}")
    assert aligned >= 0.99 and twins >= 0.99, (aligned, twins)
    np.testing.assert_allclose(got_scores[ib], ref_scores[ia], rtol=5e-3, atol=1e-5)
    # What the alignment lets through is bounded: a reference row outside the aligned set must be a DISPLACED row - its
    # (label, point mask) twin exists here and its score is the reference's (a tie that sorted the other way) - or, failing that, one
    # of at most two rows; and the in-place agreement keeps a floor: a displaced row shifts the rows between its two places by one, it
    # does not scramble the table.
    got_by_key = {}
    for i, k in enumerate(got_keys):
        got_by_key.setdefault(k, []).append(i)
    lost, worst_move = 0, 0.0
    for i in sorted(set(range(len(ref_keys))) - set(ia.tolist())):
        cand = got_by_key.get(ref_keys[i], [])
        if not cand:
            lost += 1
            continue
        worst_move = max(worst_move, min(abs(float(got_scores[c]) - float(ref_scores[i])) for c in cand))
    print(f"{name} [{path}]: {len(ref_keys) - len(ia)} reference rows outside the alignment, {lost} without a twin, largest score move of a displaced row {worst_move:.2e}")
    # measured on MI355X, all five fixtures x three decoder paths (profiles/r05_parity_numbers.md): at most 2 of 600 rows outside the
    # alignment, every one of them with a twin whose score moved <= 6e-9 (an exact tie swapped), bits equal in place >= 0.99857
    assert worst_move <= 1e-6, worst_move
    assert lost <= 2, lost
    assert bits_in_place >= 0.998, bits_in_place
    assert in_place >= 0.99, in_place
    box_ok = np.isclose(pd.instance_boxes[ib], g["inst_boxes"].numpy()[ia], rtol=5e-3, atol=5e-3).all(axis=1).mean()
    assert box_ok > 0.99, box_ok
    assert (pd.pts_semantic_mask[0] != g["sem_mask"].numpy()).mean() < 5e-3
    assert (pd.pts_semantic_mask[1] != g["pan_sem"].numpy()).mean() < 5e-3
    assert (pd.pts_instance_mask[1] != g["pan_inst"].numpy()).mean() < 5e-3
    # the top-k (query, class) pairs: equal up to the pair at the k-th place, whose score ties with its neighbour to ~1e-6
    from collections import Counter as _C
    diff = _C(pd.sort_and_mask[0].cpu().tolist()) - _C(g["topk_idx"].tolist())
    assert sum(diff.values()) <= 2, diff
    # STRICT mode (the asserts of rounds 1-3, before the summation order of the attention merge / the row chain changed): where a
    # path reproduces the reference's row ORDER exactly, everything is compared in place with no alignment at all.
    if aligned == 1.0 and len(ref_keys) == len(got_keys):
        np.testing.assert_allclose(got_scores, ref_scores, rtol=5e-3, atol=1e-5)
        assert (pd.instance_labels == g["inst_labels"].numpy()).all()
        assert bits_in_place == 1.0
        STRICT_SEEN.add((name, path))


STRICT_SEEN = set()


def test_architecture_goldens_hold_in_place_on_the_op_by_op_path():
    """Runs after the parametrised golden test (same module, definition order): on the op-by-op decoder path at least three of the
    five architecture fixtures must have matched the reference IN PLACE, row for row (no alignment) - so the aligned comparison
    above is a tolerance for near-tied scores, not a way of passing whatever comes out."""
    if not any(p == "op_by_op" for _, p in STRICT_SEEN) and not STRICT_SEEN:
        pytest.skip("the parametrised architecture golden test did not run in this session")
    n = sum(1 for _, p in STRICT_SEEN if p == "op_by_op")
    print(f"architecture fixtures reproduced in place: {sorted(STRICT_SEEN)}")
    assert n >= 3, sorted(STRICT_SEEN)


@pytest.mark.parametrize("path", DECODER_PATHS)
def test_plain_decoder_matches_reference_golden(path, monkeypatch):
    """Non-positional decoder variant of the Baseline_ScanNet200 prototype.  It has no row-chain path (`_fusable` = 0 whatever the
    switches say): the three cases pin that the switches do not change what this variant computes."""
    from segdino3d_amd import decoder as D
    from segdino3d_amd.decoder import ScanNetQueryDecoder
    monkeypatch.setattr(D, "FUSED_DECODER", path != "op_by_op")
    monkeypatch.setattr(D, "FUSED_NARROW", path == "rowchain4")
    from test_oracle_golden import plain_decoder_state_dict
    d = dev()
    g = load("decoder_plain_s40")
    kw = {k: v for k, v in DEC_KW.items() if k not in ("add_box_size_pred", "add_positional_embedding", "pos_type",
                                                         "temperature", "box_modulate_ca", "normalize_box_prediction")}
    kw["add_dinox_query_ca"] = False
    dec = ScanNetQueryDecoder(**kw).eval()
    sd = plain_decoder_state_dict()
    dec.load_state_dict({k[len("decoder."):]: v for k, v in sd.items()})
    dec.to(d)
    assert dec._fusable(g["x"].shape[0]) == 0
    out = dec([g["x"].to(d)], None, None, [g["x"].to(d)], None, None, None, None)
    assert len(out["aux_outputs"]) == 5 and out["centers"][0] is None

    def rows_ok(got, ref, what, strict):
        err = (got.cpu() - ref).abs()
        bad = (err > 3e-4 + 3e-4 * ref.abs()).any(dim=1).float().mean().item()
        print(f"plain decoder {what}: {bad:.1%} rows outside 3e-4, max err {err.max().item():.1e}")
        assert bad == 0.0, f"{what}: {bad:.1%} rows outside tolerance (max err {err.max().item():.3e})"
    for li in range(5):
        rows_ok(out["aux_outputs"][li]["cls_preds"][0], g[f"aux{li}_cls"], f"aux{li} cls", li <= 1)
        rows_ok(out["aux_outputs"][li]["masks"][0], g[f"aux{li}_masks"], f"aux{li} masks", li <= 1)
    for k in ("cls_preds", "sem_preds", "masks", "hidden_states"):
        rows_ok(out[k][0], g[k], k, False)


def test_plain_decoder_with_learned_queries_and_objectness():
    """Decoder options `num_semantic_queries = 7` (learned query embeddings) and `objectness_flag=True` (`out_score` head) against
    the REFERENCE decoder's outputs (tests/golden/make_golden_plain_obj.py)."""
    from segdino3d_amd.decoder import ScanNetQueryDecoder
    from test_oracle_golden import plain_decoder_state_dict
    d = dev()
    g = load("decoder_plain_obj_s40")
    kw = {k: v for k, v in DEC_KW.items() if k not in ("add_box_size_pred", "add_positional_embedding", "pos_type",
                                                         "temperature", "box_modulate_ca", "normalize_box_prediction")}
    kw.update(add_dinox_query_ca=False, num_semantic_queries=7, objectness_flag=True)
    dec = ScanNetQueryDecoder(**kw).eval()
    sd = plain_decoder_state_dict(n_learned=7, objectness=True)
    dec.load_state_dict({k[len("decoder."):]: v for k, v in sd.items()})
    dec.to(d)
    out = dec([g["x"].to(d)], None, None, [g["x"].to(d)], None, None, None, None)
    assert out["masks"][0].shape == (47, 40) and out["scores"][0].shape == (47, 1)

    def rows_ok(got, ref, what):
        err = (got.cpu() - ref).abs()
        bad = (err > 3e-4 + 3e-4 * ref.abs()).any(dim=1).float().mean().item()
        print(f"learned queries + objectness, {what}: {bad:.1%} rows outside 3e-4, max err {err.max().item():.1e}")
        assert bad == 0.0, f"{what}: {bad:.1%} rows outside tolerance (max err {err.max().item():.3e})"
    for li in (0, 2, 4):
        for k, gk in (("cls_preds", "cls"), ("masks", "masks"), ("scores", "scores")):
            rows_ok(out["aux_outputs"][li][k][0], g[f"aux{li}_{gk}"], f"aux{li} {gk}")
    for k in ("cls_preds", "sem_preds", "masks", "scores", "hidden_states"):
        rows_ok(out[k][0], g[k], k)


def test_baseline_prototype_end_to_end_matches_oracle():
    """Baseline_ScanNet200 prototype (rgb-only Res16UNet34C + non-positional decoder) through build_architecture."""
    import segdino3d_amd as seg
    from oracle import decoder_ref as D, postprocess_ref as P, sparse_ref as R
    from segdino3d_amd.configs import baseline_scannet200_model_cfg
    from segdino3d_amd.synth import make_scene
    d = dev()
    pts, tgt = make_scene(31, n_points=9000, n_superpoints=60, n_query2d=5)
    model = seg.build_architecture(baseline_scannet200_model_cfg()).eval()
    sd = {k: det_param(k, v.shape).to(v.dtype) for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    model.to(d)
    model.to_host = False
    sp = tgt.extra_features["super_point_masks"].clone()
    with torch.no_grad(), seg.capture() as cap:
        res = model([pts.to(d)], [tgt.to(d)])
    out = cap.outputs
    f, _, _ = R.mink_forward_wrapper(sd, pts, None, sp, mode="only_rgb")
    cfg = D.DecoderCfg(add_positional_embedding=False, add_dinox_query_ca=False, add_box_size_pred=False,
                       box_modulate_ca=False, normalize_box_prediction=False)
    ref = D.decoder_forward(sd, cfg, f, None, None, f, None, None, None, None, None)
    err = (out["masks"][0].cpu() - ref["masks"]).abs()
    bad = (err > 5e-4 + 5e-4 * ref["masks"].abs()).any(dim=1).float().mean().item()
    print(f"Baseline_ScanNet200 end to end, mask logits: {bad:.1%} of query rows outside 5e-4, max err {err.max().item():.3e}")
    assert bad <= 0.02, f"{bad:.1%} of query rows differ (max err {err.max().item():.3e})"
    pd = res[0].pred_pts_seg
    assert pd.pts_instance_mask[0].shape[1] == pts.shape[0] and pd.instance_boxes is None


@pytest.mark.parametrize("M,Cin,with_res,act", [(200, 256, True, None), (200, 1024, True, None), (37, 256, False, None), (512, 256, True, "relu"),
                                                (1, 256, True, None), (16, 1024, False, None)])
def test_fused_projection_residual_layernorm(M, Cin, with_res, act):
    """sd3d_linear_layernorm (one launch for <= 512 rows) against float64 LayerNorm(x W^T + b + res), and against the two-launch path it
    replaces (gather_gemm + layernorm kernel) - same formula, different fp32 summation order: 2e-6 of the row scale."""
    from segdino3d_amd import _lib, ops
    d = dev()
    x = det_randn(f"fl.x{M}{Cin}", (M, Cin)); w = det_randn(f"fl.w{Cin}", (256, Cin), Cin ** -0.5); b = det_randn("fl.b", (256,), 0.1)
    res = det_randn(f"fl.r{M}", (M, 256)) if with_res else None
    g = 1 + det_randn("fl.g", (256,), 0.1); beta = det_randn("fl.beta", (256,), 0.1)
    # the C entry itself (ops.linear_layernorm sends contractions longer than 512 to the two-launch path: faster there)
    xd, wd, bd, gd, betad = x.to(d), w.to(d), b.to(d), g.to(d), beta.to(d)
    rd = None if res is None else res.to(d)
    got = torch.empty(M, 256, device=d)
    _lib.check(_lib.load().sd3d_linear_layernorm(xd.data_ptr(), Cin, M, Cin, wd.data_ptr(), 256, bd.data_ptr(), None if rd is None else rd.data_ptr(), 256,
                                                 gd.data_ptr(), betad.data_ptr(), 1e-5, ops.ACT[act], got.data_ptr(), 256, ops._stream()), "linear_layernorm")
    if Cin <= 512:
        assert torch.equal(got, ops.linear_layernorm(xd, wd, bd, gd, betad, res=rd, act=act))
    pre = x.double() @ w.double().T + b.double() + (0 if res is None else res.double())
    ref = torch.nn.functional.layer_norm(pre, (256,), g.double(), beta.double(), 1e-5)
    if act == "relu":
        ref = torch.relu(ref)
    err = (got.cpu().double() - ref).abs().max().item()
    assert err <= 2e-6 * max(1.0, ref.abs().max().item()) * 4, err
    keep = ops.LINEAR_LN_MAX_ROWS
    ops.LINEAR_LN_MAX_ROWS = 0                                    # the two-launch path
    try:
        two = ops.linear_layernorm(x.to(d), w.to(d), b.to(d), g.to(d), beta.to(d), res=None if res is None else res.to(d), act=act)
    finally:
        ops.LINEAR_LN_MAX_ROWS = keep
    assert (two.cpu().double() - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item()) * 4
    assert (got - two).abs().max().item() <= 1e-5
    # rows beyond the limit take the two-launch path by themselves
    big = ops.linear_layernorm(torch.cat([x] * (600 // M + 1)).to(d)[:600] if M < 600 else x.to(d), w.to(d), b.to(d), g.to(d), beta.to(d))
    assert big.shape[1] == 256


@pytest.mark.parametrize("rows", [200, 2441, 3000])
def test_linear_group_equals_single_launches(rows):
    """ops.linear_group (independent Linears sharing launches) against one ops.gather_gemm per job, bit for bit: the few-hundred-row
    group kernel and, from 2017 rows on, the grouped lock-step kernel (jobs of different widths, two inputs, residual, activations,
    a 3072-column job that takes two column tiles per workgroup)."""
    from segdino3d_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(rows)
    rnd = lambda *s: torch.randn(*s, generator=g).to(d)  # noqa: E731
    x, x2, res = rnd(rows, 256), rnd(rows, 256), rnd(rows, 256)
    jobs = [(x, rnd(256, 256) * 0.06, rnd(256), "relu", None, None),
            (x, rnd(199, 256) * 0.06, rnd(199), None, None, None),
            (x, rnd(256, 512) * 0.04, None, None, res, x2),
            (x2, rnd(1024, 256) * 0.06, rnd(1024), "gelu", None, None),
            (x, rnd(32, 256) * 0.06, rnd(32), "sigmoid", None, None),
            (x2, rnd(3072, 256) * 0.06, rnd(3072), None, None, None)]
    got = ops.linear_group(jobs)
    for (xx, w, b, act, r, xx2), y in zip(jobs, got):
        if rows < 2048:                                          # the group kernel = the split-contraction kernel of a single small launch
            ref = ops.gather_gemm(xx, w, x2=xx2, shift=b, act=act, res=r, nt=-1, exact=True)
        else:
            ref = ops.gather_gemm(xx, w, x2=xx2, shift=b, act=act, res=r)
        assert torch.equal(y, ref), (rows, tuple(w.shape), act)
        ref64 = torch.cat([xx, xx2], 1).double() @ w.double().t() if xx2 is not None else xx.double() @ w.double().t()
        if b is not None:
            ref64 = ref64 + b.double()
        if r is not None:
            ref64 = ref64 + r.double()
        ref64 = {None: lambda t: t, "relu": torch.relu, "gelu": torch.nn.functional.gelu, "sigmoid": torch.sigmoid}[act](ref64)
        assert (y.double() - ref64).abs().max().item() < 2e-5 * max(1.0, ref64.abs().max().item())
