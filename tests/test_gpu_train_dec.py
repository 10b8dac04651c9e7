"""Autograd nodes of the decoder (segdino3d_amd/train_dec.py, csrc/train_dec.hip) against torch float64
autograd of the same formulas.  Tolerance 2e-5 of the largest entry (fp32 sums over <= 3000 rows / 1024 columns)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from _det import det_randn  # noqa: E402


def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def close(got, ref, what, tol=2e-5):
    err = (got.detach().cpu().double() - ref.detach().cpu()).abs().max().item()
    assert err <= tol * max(ref.abs().max().item(), 1e-3), (what, err, ref.abs().max().item())


ACT64 = {None: lambda t: t, "relu": torch.relu, "gelu": torch.nn.functional.gelu, "sigmoid": torch.sigmoid}


@pytest.mark.parametrize("M,cin,cout,act,res,concat", [(200, 256, 256, None, False, False), (200, 256, 1024, "gelu", False, False),
                                                       (200, 1024, 256, None, True, False), (3000, 96, 256, "relu", False, False),
                                                       (200, 512, 256, None, False, True), (200, 256, 199, None, False, False),
                                                       (200, 256, 3, "sigmoid", False, False), (37, 256, 201, "relu", True, False)])
def test_linear_gradients(M, cin, cout, act, res, concat):
    from segdino3d_amd import train_dec as T
    d = dev()
    x = det_randn(f"tl.x{M}{cin}", (M, cin)); w = det_randn(f"tl.w{cin}{cout}", (cout, cin), cin ** -0.5)
    b = det_randn(f"tl.b{cout}", (cout,), 0.1); r = det_randn(f"tl.r{M}{cout}", (M, cout)) if res else None
    dy = det_randn(f"tl.dy{M}{cout}", (M, cout))
    leaves = [t.to(d).requires_grad_(True) for t in (x, w, b)] + ([r.to(d).requires_grad_(True)] if res else [])
    xd, wd, bd = leaves[:3]
    if concat:
        h = cin // 2
        xa, xb = xd[:, :h].detach().contiguous().requires_grad_(True), xd[:, h:].detach().contiguous().requires_grad_(True)
        y = T.linear(xa, wd, bd, act=act, x2=xb)
    else:
        y = T.linear(xd, wd, bd, act=act, res=leaves[3] if res else None)
    y.backward(dy.to(d))
    l64 = [t.double().requires_grad_(True) for t in (x, w, b)] + ([r.double().requires_grad_(True)] if res else [])
    y64 = ACT64[act](l64[0] @ l64[1].T + l64[2] + (l64[3] if res else 0))
    y64.backward(dy.double())
    close(y, y64, "y")
    if concat:
        close(torch.cat([xa.grad, xb.grad], 1), l64[0].grad, "dx")
    else:
        close(xd.grad, l64[0].grad, "dx")
    close(wd.grad, l64[1].grad, "dw")
    close(bd.grad, l64[2].grad, "db")
    if res:
        close(leaves[3].grad, l64[3].grad, "dres")


@pytest.mark.parametrize("M,D,act,res", [(200, 256, None, True), (3000, 256, "relu", False), (37, 1024, None, True), (5, 96, None, False)])
def test_layernorm_gradients(M, D, act, res):
    from segdino3d_amd import train_dec as T
    d = dev()
    x = det_randn(f"tn.x{M}{D}", (M, D)) * 2 + 0.5; r = det_randn(f"tn.r{M}{D}", (M, D)) if res else None
    w = 1 + det_randn(f"tn.w{D}", (D,), 0.1); b = det_randn(f"tn.b{D}", (D,), 0.1)
    dy = det_randn(f"tn.dy{M}{D}", (M, D))
    xd, wd, bd = (t.to(d).requires_grad_(True) for t in (x, w, b))
    rd = r.to(d).requires_grad_(True) if res else None
    y = T.layernorm(xd, wd, bd, res=rd, act=act)
    y.backward(dy.to(d))
    x64, w64, b64 = (t.double().requires_grad_(True) for t in (x, w, b))
    r64 = r.double().requires_grad_(True) if res else None
    y64 = torch.nn.functional.layer_norm(x64 + (r64 if res else 0), (D,), w64, b64, 1e-5)
    if act == "relu":
        y64 = y64 * (y.detach().cpu() > 0)                      # the fp32 result's own mask (outputs within rounding of 0)
    y64.backward(dy.double())
    close(y, y64, "y")
    close(xd.grad, x64.grad, "dx")
    close(wd.grad, w64.grad, "dw", 5e-5)
    close(bd.grad, b64.grad, "db", 5e-5)
    if res:
        close(rd.grad, r64.grad, "dres")


def test_sine_pe_modulation_gradient():
    from segdino3d_amd import ops, train_dec as T
    from segdino3d_amd.decoder import ScanNetQueryDecoder
    from test_gpu_decoder import DEC_KW
    d = dev()
    dec = ScanNetQueryDecoder(**DEC_KW)
    dim_t, axis = dec.pe_tables(d)
    n = 200
    xyz = (det_randn("tp.x", (n, 3)).sigmoid() * torch.tensor([8.0, 6.0, 3.0])).to(d)
    rng = torch.tensor([0.0, 0.0, 0.0, 8.0, 6.0, 3.0], device=d)
    num = det_randn("tp.n", (n, 3)).sigmoid().to(d).requires_grad_(True)
    den = (0.2 + det_randn("tp.d", (n, 3)).sigmoid()).to(d)
    dy = det_randn("tp.dy", (n, 256)).to(d)
    out = T.sine_pe_modulated(xyz, rng, dim_t, axis, num, den)
    out.backward(dy)
    plain = ops.sine_pe(xyz, rng, dim_t, axis).double()
    n64 = num.detach().double().requires_grad_(True)
    coef = (n64 / den.double())[:, axis.long()]                 # [n, 256]: every channel scaled by its axis' coefficient
    (plain * coef).backward(dy.double())
    close(out, plain * coef, "out", 1e-5)
    close(num.grad, n64.grad, "dnum")
    # one modulation row for all queries (the first layer's broadcast size, ld = 0)
    den1 = den[:1].contiguous().reshape(3)
    num2 = num.detach().clone().requires_grad_(True)
    T.sine_pe_modulated(xyz, rng, dim_t, axis, num2, den1).backward(dy)
    n64 = num.detach().double().requires_grad_(True)
    (plain * (n64 / den1.double())[:, axis.long()]).backward(dy.double())
    close(num2.grad, n64.grad, "dnum broadcast")


@pytest.mark.parametrize("Lq,Lk,nsrc,masked", [(200, 3000, 2, True), (200, 200, 1, False), (200, 301, 1, True), (33, 311, 1, True),
                                               (500, 700, 2, False), (16, 8, 1, True)])
def test_attention_gradients(Lq, Lk, nsrc, masked):
    """dq, dk, dv (and the second score source) against float64 autograd of masked softmax attention; strided views as inputs
    (the decoder slices q / k / v out of packed projections)."""
    from segdino3d_amd import train_dec as T
    d = dev()
    H = 8
    pack_q = det_randn(f"ta.q{Lq}", (Lq, 512)); pack_k = det_randn(f"ta.k{Lk}", (Lk, 768))
    dy = det_randn(f"ta.dy{Lq}", (Lq, 256))
    blocked, bits = None, None
    if masked:
        blocked = det_randn(f"ta.m{Lq}{Lk}", (Lq, Lk)) > 0.3
        blocked[:, : min(40, Lk - 1)] = True
        blocked[0] = True; blocked[0, Lk - 1] = False
        blocked[torch.arange(Lq), torch.arange(Lq) % Lk] = False
        nw = (Lk + 31) // 32
        pad = torch.ones(Lq, nw * 32, dtype=torch.bool); pad[:, :Lk] = blocked
        words = (pad.view(Lq, nw, 32).long() << torch.arange(32)).sum(-1)
        bits = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32).to(d)
    scale = (32 * nsrc) ** -0.5
    pq, pk = pack_q.to(d).requires_grad_(True), pack_k.to(d).requires_grad_(True)
    q, q2 = pq[:, :256], pq[:, 256:]
    k, k2, v = pk[:, :256], pk[:, 256:512], pk[:, 512:]
    out = T.attention(q, k, v, H, scale, mask_bits=bits, q2=q2 if nsrc == 2 else None, k2=k2 if nsrc == 2 else None)
    out.backward(dy.to(d))
    q64, k64 = pack_q.double().requires_grad_(True), pack_k.double().requires_grad_(True)
    s = torch.einsum("qhc,khc->hqk", q64[:, :256].view(Lq, H, 32), k64[:, :256].view(Lk, H, 32))
    if nsrc == 2:
        s = s + torch.einsum("qhc,khc->hqk", q64[:, 256:].view(Lq, H, 32), k64[:, 256:512].view(Lk, H, 32))
    s = s * scale
    if blocked is not None:
        s = s.masked_fill(blocked.unsqueeze(0), float("-inf"))
    ref = torch.einsum("hqk,khc->qhc", torch.softmax(s, -1), k64[:, 512:].view(Lk, H, 32)).reshape(Lq, 256)
    ref.backward(dy.double())
    close(out, ref, "out", 2e-5)
    gq, gk = q64.grad.clone(), k64.grad.clone()
    if nsrc == 1:
        assert float(pq.grad[:, 256:].abs().max()) == 0.0 and float(pk.grad[:, 256:512].abs().max()) == 0.0
    close(pq.grad, gq, "dq", 3e-5)
    close(pk.grad, gk, "dk / dv", 3e-5)
    # bit-reproducible
    pq2 = pack_q.to(d).requires_grad_(True); pk2 = pack_k.to(d).requires_grad_(True)
    T.attention(pq2[:, :256], pk2[:, :256], pk2[:, 512:], H, scale, mask_bits=bits, q2=pq2[:, 256:] if nsrc == 2 else None,
                k2=pk2[:, 256:512] if nsrc == 2 else None).backward(dy.to(d))
    assert torch.equal(pq2.grad, pq.grad) and torch.equal(pk2.grad, pk.grad)


@pytest.mark.parametrize("Lq,Lk,nsrc,masked", [(200, 3000, 2, True), (200, 301, 1, True), (64, 77, 1, False)])
def test_attention_dropout_path_equals_the_fused_attention_at_rate_zero(Lq, Lk, nsrc, masked):
    """`train_dec.attention_dropout` (explicit probabilities + torch dropout, the p > 0 training path) at p = 0 against the fused
    attention node: same mask-bit convention, scaling, two score sources, strided inputs - outputs and gradients within 2e-5."""
    from segdino3d_amd import train_dec as T
    d = dev()
    H = 8
    pack_q, pack_k, dy = det_randn(f"ta.q{Lq}", (Lq, 512)), det_randn(f"ta.k{Lk}", (Lk, 768)), det_randn(f"ta.dy{Lq}", (Lq, 256)).to(d)
    bits = None
    if masked:
        blocked = det_randn(f"ta.m{Lq}{Lk}", (Lq, Lk)) > 0.3
        blocked[torch.arange(Lq), torch.arange(Lq) % Lk] = False
        nw = (Lk + 31) // 32
        pad = torch.ones(Lq, nw * 32, dtype=torch.bool); pad[:, :Lk] = blocked
        words = (pad.view(Lq, nw, 32).long() << torch.arange(32)).sum(-1)
        bits = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32).to(d)
    scale = (32 * nsrc) ** -0.5
    res = []
    for fn in (T.attention, lambda *a, **k: T.attention_dropout(*a, p=0.0, **k)):
        pq, pk = pack_q.to(d).requires_grad_(True), pack_k.to(d).requires_grad_(True)
        out = fn(pq[:, :256], pk[:, :256], pk[:, 512:], H, scale, mask_bits=bits, q2=pq[:, 256:] if nsrc == 2 else None,
                 k2=pk[:, 256:512] if nsrc == 2 else None)
        out.backward(dy)
        res.append((out.detach(), pq.grad, pk.grad))
    for a, b, name in zip(res[0], res[1], ("out", "dq", "dk / dv")):
        close(b, a.double(), name, 2e-5)


def test_decoder_trains_with_dropout():
    """dropout > 0 in training (no shipped config: configs/models/base_3d.py:30 is 0.0): nn.Dropout after every attention /
    FFN projection and on the attention probabilities (instance_seg_3d_decoder.py:48-51, 128-131, 166-170, 499, 515, 690, 708).
    The random stream is torch's on the device, so the check is behavioural: same seed -> same bits, other seed -> other
    outputs, finite gradients for every parameter, mean over seeds close to the rate-0 forward, evaluation untouched."""
    from test_gpu_decoder import _build_decoder
    from test_oracle_golden import load
    d = dev()
    g = load("decoder_s96_q16")
    dec0, _ = _build_decoder()
    dec, _ = _build_decoder(dict(dropout=0.1))
    dec.load_state_dict(dec0.state_dict())
    dec0.to(d).train(); dec.to(d).train()
    ids = g["query_ids"].long()
    t = lambda a: a.to(d)
    x, q = t(g["x"].detach()), t(g["x"].detach()[ids])

    def run(m, seed, grad=False):
        torch.manual_seed(seed)
        xx, qq = x.clone().requires_grad_(grad), q.clone().requires_grad_(grad)
        out = m([xx], [t(g["pos"])], [t(g["pos_wo"])], [qq], [t(g["pos"][ids])], [t(g["q2d_feat"])], [t(g["q2d_pos"])], [(t(g["lo"]), t(g["hi"]))])
        return out, xx
    base, _ = run(dec0, 0)
    a, xa = run(dec, 1, grad=True)
    b, _ = run(dec, 1)
    c, _ = run(dec, 2)
    assert torch.equal(a["cls_preds"][0], b["cls_preds"][0]) and torch.equal(a["masks"][0], b["masks"][0])
    assert not torch.equal(a["cls_preds"][0], c["cls_preds"][0])
    assert not torch.allclose(a["cls_preds"][0], base["cls_preds"][0], atol=1e-4), "dropout must change the training forward"
    from decoder_grad_case import objective
    pick = lambda o: {k: (None if o.get(k) is None or o[k][0] is None else o[k][0]) for k in ("cls_preds", "masks", "centers", "sizes", "sem_preds")}

    def grads(m, out):
        objective([pick(z) for z in out["aux_outputs"]] + [pick(out)]).backward()
        return {n: p.grad for n, p in m.named_parameters()}
    ga = grads(dec, a)
    g0 = grads(dec0, run(dec0, 0, grad=True)[0])
    assert xa.grad is not None and torch.isfinite(xa.grad).all()
    assert {n for n, v in ga.items() if v is None} == {n for n, v in g0.items() if v is None}, "the same parameters take part"
    assert all(torch.isfinite(v).all() for v in ga.values() if v is not None)
    assert sum(v is not None for v in ga.values()) > 150
    # under no_grad the module still drops (nn.Dropout looks at .training only)
    with torch.no_grad():
        e, _ = run(dec, 1)
    assert torch.equal(e["cls_preds"][0], a["cls_preds"][0].detach())
    # inverted dropout is unbiased: the first layer's class logits averaged over seeds approach the rate-0 ones
    first = torch.stack([run(dec, 100 + s)[0]["aux_outputs"][1]["cls_preds"][0].detach() for s in range(48)]).mean(0)
    ref = base["aux_outputs"][1]["cls_preds"][0].detach()
    one = run(dec, 100)[0]["aux_outputs"][1]["cls_preds"][0].detach()
    assert (first - ref).abs().mean().item() < 0.5 * (one - ref).abs().mean().item()
    # evaluation: dropout is the identity
    dec.eval(); dec0.eval()
    with torch.no_grad():
        assert torch.equal(run(dec, 5)[0]["masks"][0], run(dec0, 6)[0]["masks"][0])


def test_transposed_weights_are_batched_across_threads():
    """The W^T of all Linears of a step come from ONE `_TransposedWeights._run` (sd3d_transpose_batch) although the forward registers
    from the calling thread and the backward asks from the autograd engine's device thread; a second model whose parameters may land
    on the addresses of the first (freed) one gets its own transposes, not the cached ones."""
    import gc
    from segdino3d_amd import train_dec as T
    d = dev()
    calls = []
    orig = T._TransposedWeights._run

    def counted(self, ws):
        calls.append(len(ws))
        return orig(self, ws)
    T._TransposedWeights._run = counted
    try:
        for seed in (1, 2, 3):
            torch.manual_seed(seed)
            lins = [torch.nn.Linear(256, 256).to(d) for _ in range(5)] + [torch.nn.Linear(256, 199).to(d)]
            x = torch.randn(300, 256, device=d, requires_grad=True)
            h = x
            for m in lins[:5]:
                h = T.linear(h, m.weight, m.bias, act="relu")
            y = T.linear(h, lins[5].weight, lins[5].bias)
            n0 = len(calls)
            y.square().mean().backward()
            assert len(calls) == n0 + 1 and calls[-1] >= 6, calls          # one batched run for the six weights (plus leftovers)
            x2 = x.detach().clone().requires_grad_(True)
            h = x2
            for m in lins[:5]:
                h = torch.relu(torch.nn.functional.linear(h, m.weight, m.bias))
            torch.nn.functional.linear(h, lins[5].weight, lins[5].bias).square().mean().backward()
            close(x.grad, x2.grad.double(), f"dx, model {seed}", 2e-5)
            del lins, x, x2, h, y
            gc.collect(); torch.cuda.empty_cache()
    finally:
        T._TransposedWeights._run = orig


def test_decoder_training_gradients_match_reference():
    """The whole query decoder in training mode on the device (autograd nodes over HIP kernels) against the gradients of the
    REFERENCE decoder itself (tests/golden/decoder_grad_s96_q16.npz: reference autograd, train mode, same weights / inputs):
    every parameter's gradient norm and leading entries, d/d(superpoint features), d/d(query features), within 2e-3 (fp32;
    the float32 oracle sits at the same distance, tests/test_decoder_grad_oracle.py)."""
    from decoder_grad_case import Z, compare, objective
    from test_gpu_decoder import _build_decoder
    from test_oracle_golden import load
    d = dev()
    g = load("decoder_s96_q16")
    dec, _ = _build_decoder()
    dec.to(d).train()
    dec.return_hidden_states = False
    ids = g["query_ids"].long()
    x = g["x"].detach().clone().to(d).requires_grad_(True)
    q = g["x"].detach()[ids].clone().to(d).requires_grad_(True)
    t = lambda a: a.to(d)
    out = dec([x], [t(g["pos"])], [t(g["pos_wo"])], [q], [t(g["pos"][ids])], [t(g["q2d_feat"])], [t(g["q2d_pos"])], [(t(g["lo"]), t(g["hi"]))])
    assert "hidden_states" not in out
    assert torch.equal(out["masks"][0].detach().cpu() > 0, torch.from_numpy(Z["masks"]) > 0), "mask signs differ: gradients are not comparable"
    pick = lambda o: {k: (None if o.get(k) is None or o[k][0] is None else o[k][0]) for k in ("cls_preds", "masks", "centers", "sizes", "sem_preds")}
    sets = [pick(a) for a in out["aux_outputs"]] + [pick(out)]
    obj = objective(sets)
    obj.backward()
    grads = {n: p.grad for n, p in dec.named_parameters()}
    worst = compare(grads, x.grad, q.grad, obj.detach().cpu(), 2e-3)
    print("worst parameter gradient error vs the reference:", worst)
    # eval mode afterwards is the untouched forward path
    dec.eval()
    with torch.no_grad():
        again = dec([x.detach()], [t(g["pos"])], [t(g["pos_wo"])], [q.detach()], [t(g["pos"][ids])], [t(g["q2d_feat"])], [t(g["q2d_pos"])],
                    [(t(g["lo"]), t(g["hi"]))])
    assert (again["masks"][0] - out["masks"][0].detach()).abs().max().item() < 1e-4


def test_full_model_training_step_matches_float64_oracle():
    """`model.train(); losses = model(samples, targets); (seg + inst).backward()` on the device against the oracle's training
    forward (oracle/model_ref.forward_train: backbone with batch statistics -> query subset -> decoder -> criterion) in float64
    with torch autograd: both losses and the gradient of EVERY parameter of backbone and decoder (relative L2).
    The backbone's ReLUs are dropped on both sides (train_ops.TrainBackend.IGNORE_ACT): a smooth backbone keeps the decoder's
    thresholded attention masks and the matcher's choices identical between float32 and float64, so the comparison is exact
    up to rounding; the ReLU path itself is covered by test_gpu_train_ops."""
    import segdino3d_amd as seg
    from oracle import decoder_ref as D, model_ref as MR, sparse_ref as R
    from segdino3d_amd import train_ops
    from segdino3d_amd.configs import scannet200_model_cfg
    from segdino3d_amd.synth import add_training_targets, make_scene
    d = dev()
    torch.manual_seed(0)
    model = seg.build_architecture(scannet200_model_cfg(query_num=-1)).to(d).train()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    pts, tgt = make_scene(21, n_points=20000, n_superpoints=150, n_query2d=20)
    tgt = add_training_targets(pts, tgt, n_instances=8, seed=1)
    cpu = dict(pts=pts.clone(), f2d=tgt.extra_features["points_2dfeats"].clone(), sp=tgt.extra_features["super_point_masks"].clone(),
               q2d_feat=tgt.extra_features["query2d_feats"].clone(), q2d_pos=tgt.extra_features["query2d_pos"].clone(),
               masks=tgt.masks.clone(), labels=tgt.labels.clone(), spm=tgt.sp_inst_sem_masks.clone())
    pts_d, tgt_d = pts.to(d), tgt.to(d)
    train_ops.TrainBackend.IGNORE_ACT = True
    try:
        torch.manual_seed(7)
        with seg.capture() as cap:
            losses = model([pts_d], [tgt_d])
        (losses["seg_loss"] + losses["inst_loss"]).backward()
    finally:
        train_ops.TrainBackend.IGNORE_ACT = False
    # the query subset the model drew (baseline3d.py:252-254)
    S = 150
    torch.manual_seed(7)
    n = ((1 - model.query_thr) * torch.rand(1) + model.query_thr)
    ids = torch.randperm(S)[: int((n * S).int())]
    assert cap.outputs["masks"][0].shape[0] == len(ids)
    sd64 = {k: (v.double().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    ic = model.criterion.inst_criterion
    loss_cfg = dict(matcher="sparse", topk=ic.topk, cost_weights=ic.cost_weights, loss_weight=ic.loss_weight, num_classes=ic.num_classes,
                    num_semantic_classes=model.criterion.num_semantic_classes, sem_ignore_index=model.criterion.sem_criterion.ignore_index,
                    sem_loss_weight=model.criterion.sem_criterion.loss_weight, non_object_weight=ic.class_weight[-1],
                    fix_dice_loss_weight=ic.fix_dice_loss_weight, iter_matcher=ic.iter_matcher, fix_mean_loss=ic.fix_mean_loss)
    relu = torch.relu
    torch.relu = lambda t: t                                       # the oracle backbone's ReLUs (decoder_ref uses F.relu / torch.relu too:
    try:                                                           # patch only while the backbone runs)
        R.BN_TRAIN = True
        sp_feat, sp_pos, sp_pos_wo = R.mink_forward_wrapper(sd64, cpu["pts"].double(), cpu["f2d"].double(), cpu["sp"])
    finally:
        R.BN_TRAIN = False
        torch.relu = relu
    orig = R.mink_forward_wrapper
    R.mink_forward_wrapper = lambda *a, **k: (sp_feat, sp_pos, sp_pos_wo)      # forward_train reuses the smooth backbone's output
    try:
        ref, out64 = MR.forward_train(sd64, cpu["pts"].double(), cpu["f2d"].double(), cpu["sp"], cpu["q2d_feat"].double(), cpu["q2d_pos"].double(),
                                      cpu["masks"], cpu["labels"], cpu["spm"], ids, loss_cfg, dec_cfg=D.DecoderCfg())
    finally:
        R.mink_forward_wrapper = orig
    same_bits = torch.equal(cap.outputs["masks"][0].detach().cpu() > 0, out64["masks"].detach() > 0)
    assert same_bits, "mask signs differ between the float32 device run and the float64 oracle: gradients are not comparable"
    (ref["seg_loss"] + ref["inst_loss"]).backward()
    for k in ("seg_loss", "inst_loss"):
        assert abs(float(losses[k].detach()) - float(ref[k].detach())) <= 2e-4 * abs(float(ref[k].detach())), (k, float(losses[k]), float(ref[k]))
    rows = []
    for name, p in model.named_parameters():
        r = sd64[name].grad
        if r is None or float(r.norm()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) < 1e-6, f"{name} has a gradient the oracle does not produce"
            continue
        assert p.grad is not None, name
        rows.append((_rel(p.grad.cpu(), r), float(r.norm()), name))
    floor = 1e-2 * sorted(r[1] for r in rows)[len(rows) // 2]       # key biases: true gradient zero (softmax shift invariance)
    worst = sorted(((e * n / max(n, floor), name) for e, n, name in rows), reverse=True)
    print(f"{len(rows)} parameters; worst relative L2 gradient error vs the float64 oracle: {worst[:3]}")
    assert worst[0][0] <= 2e-3, worst[:6]


def _rel(a, b):
    return ((a.double() - b).norm() / b.norm().clamp(min=1e-30)).item()


def test_plain_decoder_training_gradients_match_float64_oracle():
    """Non-positional decoder variant (Baseline_ScanNet200 prototype) in training mode against float64 autograd of the oracle."""
    from oracle import decoder_ref as D
    from segdino3d_amd.decoder import ScanNetQueryDecoder
    from test_gpu_decoder import DEC_KW
    from test_oracle_golden import load, plain_decoder_state_dict
    d = dev()
    g = load("decoder_plain_s40")
    kw = {k: v for k, v in DEC_KW.items() if k not in ("add_box_size_pred", "add_positional_embedding", "pos_type",
                                                         "temperature", "box_modulate_ca", "normalize_box_prediction")}
    kw["add_dinox_query_ca"] = False
    dec = ScanNetQueryDecoder(**kw)
    sd = plain_decoder_state_dict()
    dec.load_state_dict({k[len("decoder."):]: v for k, v in sd.items()})
    dec.to(d).train()
    x = g["x"].detach().clone().to(d).requires_grad_(True)
    q = g["x"].detach().clone().to(d).requires_grad_(True)
    out = dec([x], None, None, [q], None, None, None, None)
    sets = [dict(cls_preds=a["cls_preds"][0], masks=a["masks"][0]) for a in out["aux_outputs"]] + \
           [dict(cls_preds=out["cls_preds"][0], masks=out["masks"][0], sem_preds=out["sem_preds"][0])]
    from decoder_grad_case import objective
    objective(sets).backward()
    sd64 = {k: (v.double().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    x64 = g["x"].detach().clone().double().requires_grad_(True)
    q64 = g["x"].detach().clone().double().requires_grad_(True)
    cfg = D.DecoderCfg(add_positional_embedding=False, add_dinox_query_ca=False, add_box_size_pred=False, box_modulate_ca=False)
    ref = D.decoder_forward(sd64, cfg, x64, None, None, q64, None, None, None, None, None)
    assert torch.equal(out["masks"][0].detach().cpu() > 0, ref["masks"].detach() > 0)
    sets64 = [dict(cls_preds=a["cls_preds"], masks=a["masks"]) for a in ref["aux"][:len(sets) - 1]] + \
             [dict(cls_preds=ref["cls_preds"], masks=ref["masks"], sem_preds=ref["sem_preds"])]
    objective(sets64).backward()
    close(x.grad, x64.grad, "dx", 1e-4)
    close(q.grad, q64.grad, "dq", 1e-4)
    worst = sorted(((_rel(p.grad.cpu(), sd64["decoder." + n].grad), n) for n, p in dec.named_parameters()
                    if sd64["decoder." + n].grad is not None and float(sd64["decoder." + n].grad.norm()) > 1e-6), reverse=True)
    assert worst[0][0] <= 1e-3, worst[:5]


def test_training_step_is_bit_reproducible():
    """Same scene, same query subset, twice: every parameter gradient and both losses are bit-identical (all reductions of
    the HIP kernels run in a fixed order; no atomics on floating-point data)."""
    import segdino3d_amd as seg
    from segdino3d_amd.configs import scannet200_model_cfg
    from segdino3d_amd.synth import add_training_targets, make_scene
    d = dev()
    torch.manual_seed(0)
    model = seg.build_architecture(scannet200_model_cfg(query_num=-1)).to(d).train()
    pts, tgt = make_scene(23, n_points=20000, n_superpoints=200, n_query2d=30)
    tgt = add_training_targets(pts, tgt, n_instances=8, seed=3)
    pts, tgt = pts.to(d), tgt.to(d)
    runs = []
    for _ in range(2):
        for p in model.parameters():
            p.grad = None
        for k in ("query_inst_sem_masks", "instance_centers", "instance_sizes"):
            tgt.__dict__.pop(k, None)
        torch.manual_seed(11)
        losses = model([pts], [tgt])
        (losses["seg_loss"] + losses["inst_loss"]).backward()
        runs.append((losses["seg_loss"].detach().clone(), losses["inst_loss"].detach().clone(),
                     {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    assert runs[0][2].keys() == runs[1][2].keys() and len(runs[0][2]) > 400
    diff = [n for n in runs[0][2] if not torch.equal(runs[0][2][n], runs[1][2][n])]
    assert not diff, f"{len(diff)} parameter gradients differ between two identical steps, e.g. {diff[:3]}"
    assert all(bool(torch.isfinite(g).all()) for g in runs[0][2].values())


@pytest.mark.parametrize("M,cin,cout", [(2441, 256, 256), (3000, 96, 256), (777, 1024, 256), (5, 256, 3), (2441, 260, 200), (4096, 256, 1024), (1, 32, 32)])
def test_linear_wgrad_one_launch_matches_float64(M, cin, cout):
    """`sd3d_linear_wgrad` (round 6: weight + bias gradient of a Linear on a few thousand rows in one launch) against a float64 product, the
    pair-list kernel it replaces for such shapes, itself run twice (bit-reproducible), and its accumulate flag."""
    import ctypes
    from segdino3d_amd import _lib, ops, train_ops
    d = dev()
    lib = _lib.load()
    gen = torch.Generator().manual_seed(M + cin + cout)
    c_pad = (cout + 31) // 32 * 32
    g = torch.zeros(M, c_pad)
    g[:, :cout] = torch.randn(M, cout, generator=gen)
    x = torch.randn(M, cin, generator=gen)
    g, x = g.to(d), x.to(d)

    def run(flags=0, dw=None, db=None):
        dw = torch.empty(cout, cin, device=d) if dw is None else dw
        db = torch.empty(cout, device=d) if db is None else db
        rc = lib.sd3d_linear_wgrad(g.data_ptr(), g.stride(0), x.data_ptr(), x.stride(0), M, cin, cout, dw.data_ptr(), db.data_ptr(), flags, ops._stream())
        assert rc == 0, lib.sd3d_last_error()
        return dw, db
    dw, db = run()
    dw2, db2 = run()
    assert torch.equal(dw, dw2) and torch.equal(db, db2), "must be bit-reproducible"
    ref_w = g[:, :cout].double().t() @ x.double()
    ref_b = g[:, :cout].double().sum(0)
    scale = max(ref_w.abs().max().item(), 1.0)
    assert (dw.double() - ref_w).abs().max().item() <= 3e-5 * scale
    assert (db.double() - ref_b).abs().max().item() <= 3e-5 * max(ref_b.abs().max().item(), 1.0)
    acc_w, acc_b = run(flags=1, dw=dw.clone(), db=db.clone())                 # SD3D_WGRAD_ACCUMULATE
    assert torch.allclose(acc_w, 2 * dw, rtol=1e-6, atol=0) and torch.allclose(acc_b, 2 * db, rtol=1e-6, atol=0)
    if cin % 4 == 0 and M >= 2:                                               # the two-launch path on identity pair lists
        nbr = torch.arange(M, dtype=torch.int32, device=d).unsqueeze(0).contiguous()
        old = train_ops.pair_wgrad(g, x, ops.pair_lists(nbr, M))[0, :cout, :cin]
        assert (old.double() - ref_w).abs().max().item() <= 3e-5 * scale
        assert (old - dw).abs().max().item() <= 3e-5 * scale
