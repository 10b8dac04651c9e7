"""Row-chain executor (csrc/rowchain.hip, `sd3d_row_chain`): every op against a plain torch fp32 / float64 restatement of the reference
expression it replaces (instance_seg_3d_decoder.py:606-799, attention.py:361-385, utils.py:53-105), then the fused decoder forward
against the op-by-op forward of rounds 1-3 and B scenes per call against one scene per call (bit for bit)."""
import copy
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 2e-5          # fp32 contraction over <= 1024 channels, different summation order than torch


def _dev():
    return torch.device("cuda:0")


@pytest.fixture(params=[16, 4], ids=["tile16", "tile4"])
def tile(request):
    """rows of a workgroup's tile: 16 (csrc/rowchain.hip) or 4 (csrc/rowchain_narrow.hip)"""
    return request.param


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()


@pytest.mark.parametrize("rows", [1, 16, 37, 200])
@pytest.mark.parametrize("k0,k1,cout,act,use_res", [(96, 0, 256, "relu", False), (256, 0, 256, None, True), (256, 256, 768, None, False),
                                                    (256, 0, 1024, "gelu", False), (1024, 0, 256, None, True), (256, 0, 3, "sigmoid", False),
                                                    (256, 0, 199, None, False), (256, 0, 1, None, False), (256, 256, 256, None, False),
                                                    (48, 0, 32, None, False)])
def test_linear_op(rows, k0, k1, cout, act, use_res, tile):
    from segdino3d_amd.rowchain import Program
    d = _dev()
    g = torch.Generator().manual_seed(rows * 7 + cout)
    x0 = torch.randn(rows, k0, generator=g)
    x1 = torch.randn(rows, k1, generator=g) if k1 else None
    w = torch.randn(cout, k0 + k1, generator=g) / math.sqrt(k0 + k1)
    b = torch.randn(cout, generator=g)
    res = torch.randn(rows, cout, generator=g) if use_res else None
    out = torch.full((rows, cout), float("nan"), device=d)
    out2 = torch.full((rows, cout if cout % 4 == 0 else cout), float("nan"), device=d)
    P = Program(9, rows=tile)
    src = 5 if k0 > 256 else 0                                     # a 1024-wide buffer spans four slots
    dst = 0 if cout > 256 else 3
    P.load(src, x0.to(d))
    if k1:
        P.load(1, x1.to(d))
    if use_res:
        P.load(2, res.to(d))
    P.linear(dst, src, w.to(d), b.to(d), act=act, res=2 if use_res else None, src1=1 if k1 else None, gout=out)
    P.store(dst, out2)                                             # the LDS copy must hold the same values
    P.launch([dict(q0=0, nq=rows)])
    xin = torch.cat([x0, x1], 1) if k1 else x0
    ref = xin.double() @ w.double().T + b.double()
    if use_res:
        ref = ref + res.double()
    ref = {None: lambda t: t, "relu": torch.relu, "gelu": lambda t: torch.nn.functional.gelu(t), "sigmoid": torch.sigmoid}[act](ref)
    assert torch.isfinite(out).all()
    assert _rel(out.cpu(), ref) < TOL
    assert torch.equal(out, out2)


def test_linear_rows_are_independent_of_their_tile(tile):
    """A row's bits do not depend on the rows it shares a tile / launch with (what makes batched == single-scene)."""
    from segdino3d_amd.rowchain import Program
    d = _dev()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(53, 256, generator=g).to(d)
    w, b = (torch.randn(256, 256, generator=g) / 16).to(d), torch.randn(256, generator=g).to(d)

    def run(xs, scenes):
        out = torch.empty(xs.shape[0], 256, device=d)
        P = Program(2, rows=tile)
        P.load(0, xs)
        P.linear(1, 0, w, b, act="gelu", gout=out)
        P.launch(scenes)
        return out
    whole = run(x, [dict(q0=0, nq=53)])
    two = run(x, [dict(q0=0, nq=20), dict(q0=20, nq=33)])              # other tile boundaries, two "scenes"
    part = run(x[7:30].contiguous(), [dict(q0=0, nq=23)])
    assert torch.equal(whole, two) and torch.equal(whole[7:30], part)


def test_layernorm_pe_box_ops(tile):
    from segdino3d_amd import ops
    from segdino3d_amd.rowchain import Program
    d = _dev()
    g = torch.Generator().manual_seed(1)
    rows = 45
    x, res = torch.randn(rows, 256, generator=g).to(d), torch.randn(rows, 256, generator=g).to(d)
    lw, lb = torch.randn(256, generator=g).to(d), torch.randn(256, generator=g).to(d)
    xyz = torch.rand(rows, 3, generator=g).to(d) * 4
    rng = torch.tensor([[0., 0., 0., 4., 5., 3.], [-1., -1., -1., 5., 5., 5.]], device=d)
    temperature, d_pos = 20.0, 256
    # the decoder's tables (utils.py:64-86)
    import segdino3d_amd as seg
    from segdino3d_amd.configs import scannet200_model_cfg
    dec = seg.build_architecture(scannet200_model_cfg()).decoder
    dim_t, axis = dec.pe_tables(d)
    num, den = torch.rand(rows, 3, generator=g).to(d), (torch.rand(rows, 3, generator=g) + 0.5).to(d)
    dc, ds = torch.randn(rows, 3, generator=g).to(d) * 0.1, torch.randn(rows, 3, generator=g).to(d) * 0.1
    sprev = torch.rand(rows, 3, generator=g).to(d)
    pad = lambda t: torch.cat([t, torch.zeros(rows, 1, device=d)], 1).contiguous()        # LOAD wants a multiple of 4 columns  # noqa: E731
    o_ln, o_pe, o_pem = (torch.empty(rows, 256, device=d) for _ in range(3))
    center, size, metric = (torch.empty(rows, 3, device=d) for _ in range(3))
    scenes = [dict(q0=0, nq=20), dict(q0=20, nq=25)]
    P = Program(6, rng, rows=tile)
    P.load(0, x); P.load(1, res)
    P.ln(2, 0, lw, lb, res=1, act="relu", gout=o_ln)
    P.pe(3, xyz, dim_t, axis); P.store(3, o_pe)
    P.load(4, pad(num), width=4)
    P.pe(3, xyz, dim_t, axis, num_slot=4, den=den); P.store(3, o_pem)
    P.load(4, pad(dc), width=4); P.load(5, pad(ds), width=4)
    P.box(xyz, 4, center, size_prev=sprev, ds_slot=5, size=size, size_metric=metric, normalize=True)
    P.launch(scenes)
    rs = torch.cat([torch.zeros(20, dtype=torch.int32), torch.ones(25, dtype=torch.int32)]).to(d)
    # against the stand-alone kernels (pinned to the reference by the decoder goldens); the same expressions, but the compiler is
    # free to contract a * b + c differently in the two kernels: one ulp, not bit equality
    close = lambda a, b: torch.allclose(a, b, rtol=2e-6, atol=2e-6)  # noqa: E731
    assert close(o_ln, ops.layernorm(x, lw, lb, res=res, act="relu"))
    assert close(o_pe, ops.sine_pe(xyz, rng, dim_t, axis, row_scene=rs))
    assert close(o_pem, ops.sine_pe(xyz, rng, dim_t, axis, mod_num=num, mod_den=den, row_scene=rs))
    c2, s2, m2 = ops.box_refine(xyz, dc, sprev, ds, rng, True, row_scene=rs)
    assert close(center, c2) and close(size, s2) and close(metric, m2)


@pytest.mark.parametrize("nq,nk,masked", [(16, 16, False), (37, 37, False), (200, 200, False), (50, 301, True), (200, 13, True), (3, 1, True)])
def test_attention_op(nq, nk, masked, tile):
    """8 heads x 32 channels over the scene's own query rows (self-attention) or its 2D keys with LDS mask bits."""
    from segdino3d_amd import ops
    from segdino3d_amd.rowchain import Program
    d = _dev()
    g = torch.Generator().manual_seed(nq * 31 + nk)
    # two scenes back to back so that the row / key offsets are exercised
    nq2, nk2 = nq + 5, (nk + 3 if masked else nq + 5)
    q = torch.randn(nq + nq2, 256, generator=g).to(d)
    if masked:
        kv = torch.randn(nk + nk2, 512, generator=g).to(d)
    else:
        kv = torch.randn(nq + nq2, 512, generator=g).to(d)
    S = 200
    nw = (S + 31) // 32
    blocked = (torch.rand(nq + nq2, S, generator=g) < 0.7)
    scale = 32 ** -0.5
    out = torch.empty(nq + nq2, 256, device=d)
    scenes = [dict(q0=0, nq=nq), dict(q0=nq, nq=nq2)]
    P = Program(8, rows=tile)
    P.load(0, q)
    bits_ref = None
    if masked:
        near = [(torch.rand(nk - 1, S, generator=g) < 0.02), (torch.rand(nk2 - 1, S, generator=g) < 0.02)]
        pack = lambda m: torch.from_numpy(np.packbits(np.pad(m.numpy(), ((0, 0), (0, nw * 32 - S))), axis=1, bitorder="little").view(np.int32).copy())  # noqa: E731
        blk = pack(blocked).to(d).contiguous()
        nears = [pack(n).to(d).contiguous() for n in near]
        near_all = torch.cat([n.reshape(-1) for n in nears])
        scenes[0].update(m0=0, nm=nk, bits_off=0, nw=nw, near_off=0)
        scenes[1].update(m0=nk, nm=nk2, bits_off=nq * nw, nw=nw, near_off=nears[0].numel())
        P.bits2d(blk, near_all)
        P.attn(1, 0, kv[:, :256], kv[:, 256:], scale, aux=6, keys_2d=True, masked=True)
        bits_ref = [ops.dinox_mask_bits(blk[:nq].contiguous(), nears[0]), ops.dinox_mask_bits(blk[nq:].contiguous(), nears[1])]
    else:
        P.attn(1, 0, kv[:, :256], kv[:, 256:], scale, aux=6)
    P.store(1, out)
    P.launch(scenes, nw_max=nw, nw2_max=(max(nk, nk2) + 31) // 32 if masked else 0)
    # reference: the stand-alone attention kernel (itself pinned to the reference by the decoder goldens) AND torch in float64
    for si, (q0, n, k0, m) in enumerate([(0, nq, 0, nk), (nq, nq2, nk if masked else nq, nk2)]):
        kk = kv[k0:k0 + m] if masked else kv[q0:q0 + n]
        ref_k = ops.attention(q[q0:q0 + n].contiguous(), kk[:, :256], kk[:, 256:], 8, scale, mask_bits=bits_ref[si] if masked else None)
        qq, K, V = q[q0:q0 + n].double().cpu().view(n, 8, 32), kk[:, :256].double().cpu().view(-1, 8, 32), kk[:, 256:].double().cpu().view(-1, 8, 32)
        sc = torch.einsum("qhc,khc->hqk", qq, K) * scale
        if masked:
            bw = bits_ref[si].cpu().numpy().view(np.uint32)
            mb = np.unpackbits(bw.view(np.uint8), axis=1, bitorder="little")[:, :K.shape[0]].astype(bool)
            sc = sc.masked_fill(torch.from_numpy(mb)[None], float("-inf"))
        ref = torch.einsum("hqk,khc->qhc", torch.softmax(sc, -1), V).reshape(n, 256)
        got = out[q0:q0 + n].cpu()
        assert torch.isfinite(got).all()
        assert _rel(got, ref) < TOL, (si, _rel(got, ref))
        assert _rel(got, ref_k.cpu()) < TOL


def test_merge_op_equals_the_attention_kernels_own_merge(tile):
    from segdino3d_amd import ops
    from segdino3d_amd.rowchain import Program
    d = _dev()
    g = torch.Generator().manual_seed(5)
    outs = []
    for Lq, Lk in [(200, 3000), (45, 700), (300, 64)]:              # key split 8 / a few / none
        q, q2 = torch.randn(Lq, 256, generator=g).to(d), torch.randn(Lq, 256, generator=g).to(d)
        k, k2, v = (torch.randn(Lk, 256, generator=g).to(d) for _ in range(3))
        bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (Lq, (Lk + 31) // 32), generator=g, dtype=torch.int64).to(torch.int32).to(d)
        ref = ops.attention(q, k, v, 8, 0.125, mask_bits=bits, q2=q2, k2=k2)
        a = torch.full((Lq, 256), float("nan"), device=d)
        ws, ca = ops.attention_parts([(q, k, v, bits, q2, k2, a)], 8, 0.125)
        out = torch.empty(Lq, 256, device=d)
        P = Program(2, rows=tile)
        P.merge(0, ws, a)
        P.store(0, out)
        P.launch([dict(q0=0, nq=Lq, ksplit=ca[0][0], part_off=ca[0][1])])
        outs.append(ca[0][0])
        assert torch.equal(out, ref), (Lq, Lk, ca)
    assert outs[0] > 1 and outs[-1] == 1                              # both branches were exercised


def _model_and_scenes(sizes, query_num, seed=0):
    import segdino3d_amd as seg
    from segdino3d_amd.configs import scannet200_model_cfg
    from segdino3d_amd.synth import make_scene, sharpen_random_model, structure_scene
    d = _dev()
    cfg = scannet200_model_cfg(query_num=query_num)
    cfg["test_cfg"]["npoint_thr"] = 20
    torch.manual_seed(seed)
    model = sharpen_random_model(seg.build_architecture(cfg).eval()).to(d)
    model.to_host = False
    scenes = []
    for j, (n, s, m) in enumerate(sizes):
        pts, tgt = make_scene(40 + j, n_points=n, n_superpoints=s, n_query2d=m)
        structure_scene(pts, tgt)
        scenes.append((pts.to(d), tgt.to(d)))
    return model, scenes


def _decoder_outputs(model, scenes):
    import segdino3d_amd as seg
    with torch.no_grad(), seg.capture() as cap:
        res = model([p for p, _ in scenes], [copy.copy(t) for _, t in scenes])
    return cap.outputs, res


@pytest.mark.parametrize("query_num,narrow", [(-1, False), (64, False), (-1, True), (64, True)])
def test_fused_decoder_matches_the_op_by_op_decoder(query_num, narrow, monkeypatch):
    """Same weights, same scene: every decoder output of the row-chain path within fp32 summation-order noise of the op-by-op path
    (the path the decoder goldens pinned in rounds 1-3), mask bits equal except where a logit sits on the threshold.  narrow: the
    scene's <= 512 query rows on 4-row tiles (csrc/rowchain_narrow.hip)."""
    from segdino3d_amd import decoder as D
    model, scenes = _model_and_scenes([(30000, 300, 24)], query_num)
    monkeypatch.setattr(D, "FUSED_DECODER", True)
    monkeypatch.setattr(D, "FUSED_NARROW", narrow)
    assert model.decoder._fusable(300 if query_num < 0 else 64) == (4 if narrow else 16)
    fused, _ = _decoder_outputs(model, scenes)
    monkeypatch.setattr(D, "FUSED_DECODER", False)
    plain, _ = _decoder_outputs(model, scenes)
    worst = 0.0
    for key in ("cls_preds", "sem_preds", "masks", "centers", "sizes", "hidden_states"):
        e = _rel(fused[key][0].cpu(), plain[key][0].cpu())
        worst = max(worst, e)
        assert e < 5e-4, (key, e)
    for li, (a, b) in enumerate(zip(fused["aux_outputs"], plain["aux_outputs"])):
        for key in ("cls_preds", "masks", "centers", "sizes"):
            if a[key] is None or a[key][0] is None:
                assert b[key] is None or b[key][0] is None
                continue
            e = _rel(a[key][0].cpu(), b[key][0].cpu())
            assert e < 5e-4, (li, key, e)
    print(f"fused vs op-by-op decoder, query_num={query_num}: worst relative deviation {worst:.2e}")


@pytest.mark.parametrize("policy,narrow,sizes", [(True, False, [(30000, 300, 24), (20000, 180, 10), (25000, 333, 31)]),
                                                 ("auto", False, [(30000, 300, 24), (40000, 700, 10), (25000, 333, 31), (40000, 650, 3)]),
                                                 ("auto", True, [(30000, 300, 24), (40000, 700, 10), (25000, 333, 31), (40000, 650, 3)])])
def test_fused_decoder_batch_is_bit_identical_to_single_scene_calls(policy, narrow, sizes, monkeypatch):
    """policy True: every scene on the row-chain path; "auto": the scenes with more than 512 query rows take it (16-row tiles), the
    others go op by op - or, narrow, on 4-row tiles - in the same call: a scene's bits never depend on what else is in the batch."""
    from segdino3d_amd import decoder as D
    monkeypatch.setattr(D, "FUSED_DECODER", policy)
    monkeypatch.setattr(D, "FUSED_NARROW", narrow)
    model, scenes = _model_and_scenes(sizes, -1)
    _, batch = _decoder_outputs(model, scenes)
    for b, sc in enumerate(scenes):
        _, single = _decoder_outputs(model, [sc])
        pb, ps = batch[b].pred_pts_seg, single[0].pred_pts_seg
        assert torch.equal(pb.pts_instance_mask[0], ps.pts_instance_mask[0]) and torch.equal(pb.instance_scores, ps.instance_scores)
        assert torch.equal(pb.pts_semantic_mask[0], ps.pts_semantic_mask[0]) and torch.equal(pb.instance_labels, ps.instance_labels)
