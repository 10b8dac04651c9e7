"""The round-3 paths of the pair-major sparse convolution (csrc/pair_gemm.hip, entry sd3d_pair_conv_ex) against each other and
against a float64 model of `ME.MinkowskiConvolution(Transpose) + BN + ReLU + residual` (`minkunet.py:135-192, 234-250`):
  * per-row partial-product lists in pass 2 (stride-1 tables, down convolutions),
  * direct epilogue: one pair per output row (transposed k2s2 convolutions) - pass 1 writes the output rows, no pass 2.
The older combination (pos-based pass 2 over all offsets) stays reachable by stripping the optional products from a
PairLists object, so every path is compared in ONE process on the same rulebook.  fp32 tolerance 2e-6 of the row magnitude
(the paths differ in summation order only); each path must be bit-reproducible run to run."""

import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def scene_maps():
    from segdino3d_amd.sparse import SceneMaps
    from segdino3d_amd.synth import make_scene
    d = dev()
    pts, tgt = make_scene(3, 60_000, 800, 40)
    maps = SceneMaps(pts.to(d), 0.02, 5, superpoints=tgt.extra_features["super_point_masks"].to(d))
    maps.prepare(same=[(0, 5)] + [(l, 3) for l in range(5)], strides=[0, 1, 2, 3])
    return maps


def _variant(pl, **kw):
    """A PairLists over the SAME device arrays with some optional products switched off."""
    from segdino3d_amd.ops import PairLists
    v = PairLists(pl.pos, pl.in_idx, pl.tile_k, pl.p_cap, pl.K, pl.M, rlist=pl.rlist, rl_stride=pl.rl_stride, center=pl.center,
                  out_idx=pl.out_idx, direct=pl.direct)
    for k, val in kw.items():
        setattr(v, k, val)
    return v


def _ref64(x, w, nbr, scale, shift, res, act):
    """float64 model: out[r] = act(scale * sum_k x[nbr[k][r]] . w[k]^T + shift + res[r])"""
    K, M = nbr.shape
    out = torch.zeros(M, w.shape[1], dtype=torch.float64, device=x.device)
    xd, wd = x.double(), w.double()
    for k in range(K):
        rows = (nbr[k] >= 0).nonzero().squeeze(1)
        if rows.numel():
            out[rows] += xd[nbr[k][rows].long()] @ wd[k].t()
    if scale is not None:
        out = out * scale.double()
    if shift is not None:
        out = out + shift.double()
    if res is not None:
        out = out + res.double()
    return torch.relu(out) if act == "relu" else out


def _case(maps, key, cin, cout, split=0, use_res=True, act="relu", seed=0):
    d = maps.device
    tab = maps.conv_table(*key)
    nbr, pl = tab["nbr"], tab["pairs"]
    K, M = nbr.shape
    n_in = int(nbr.max().item()) + 1
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n_in, cin, generator=g).to(d)
    w = (torch.randn(K, cout, cin, generator=g) * (K * cin) ** -0.5 * 3).to(d)
    scale = (0.5 + torch.rand(cout, generator=g)).to(d)
    shift = (0.1 * torch.randn(cout, generator=g)).to(d)
    res = torch.randn(M, cout, generator=g).to(d) if use_res else None
    x1 = x2 = None
    if split:
        x1, x2 = x[:, :split].contiguous(), x[:, split:].contiguous()
    return dict(nbr=nbr, pl=pl, x=x, x1=x1, x2=x2, w=w, scale=scale, shift=shift, res=res, act=act)


def _run(c, pl):
    from segdino3d_amd import ops
    if c["x1"] is not None:
        return ops.pair_conv(c["x1"], c["w"], pl, x2=c["x2"], scale=c["scale"], shift=c["shift"], res=c["res"], act=c["act"])
    return ops.pair_conv(c["x"], c["w"], pl, scale=c["scale"], shift=c["shift"], res=c["res"], act=c["act"])


def _close(a, ref, tol=2e-6):
    mag = ref.abs().amax(dim=1, keepdim=True).clamp(min=1.0)
    err = ((a.double() - ref).abs() / mag).max().item()
    return err


@pytest.mark.parametrize("key,cin,cout,split", [
    (("same", 0, 3), 96, 96, 0), (("same", 0, 3), 128, 96, 96), (("same", 1, 3), 32, 32, 0), (("same", 2, 3), 64, 64, 0),
    (("same", 2, 3), 192, 128, 128), (("same", 3, 3), 256, 256, 0), (("same", 3, 3), 384, 256, 256), (("same", 4, 3), 256, 256, 0),
    (("same", 0, 5), 288, 32, 0), (("same", 1, 3), 160, 96, 96)])
def test_stride1_row_lists_match_pos_based_pass2(scene_maps, key, cin, cout, split):
    """The stride-1 tables (plain lists): per-row lists = the row's neighbours in offset order; pass 2 over them equals the pos-based
    pass 2 of rounds 1-2 bit for bit; both sit at fp32 round-off of the float64 model."""
    c = _case(scene_maps, key, cin, cout, split)
    pl = c["pl"]
    assert pl.center == -1 and pl.rlist is not None
    cnt_ref = (c["nbr"] >= 0).sum(dim=0)
    assert torch.equal(pl.rlist[:, 0].long(), cnt_ref)
    r = int(cnt_ref.argmax())
    ks = [k for k in range(pl.K) if int(c["nbr"][k][r]) >= 0]
    assert pl.rlist[r, 1:1 + len(ks)].tolist() == [int(pl.pos[k][r]) for k in ks]
    rl = _run(c, pl)
    assert torch.equal(rl, _run(c, pl)), "the convolution must be bit-reproducible"
    old = _run(c, _variant(pl, rlist=None))                                 # round-1/2 path: pass 1 over all offsets + pos-based pass 2
    assert torch.equal(rl, old), "same offsets in the same order: the row-list pass 2 must equal the pos-based one bit for bit"
    ref = _ref64(c["x"], c["w"], c["nbr"], c["scale"], c["shift"], c["res"], c["act"])
    e = _close(rl, ref)
    print(f"{key} {cin}->{cout}: err {e:.2e} of the row magnitude")
    assert e < 2e-6


def test_identity_table_as_a_pair_table():
    """K = 1 (a 1x1 convolution as a pair table, train_ops identity lists)."""
    from segdino3d_amd import ops
    d = dev()
    M = 5000
    nbr = torch.arange(M, dtype=torch.int32, device=d).view(1, M)
    pl = ops.pair_lists(nbr, M)
    g = torch.Generator().manual_seed(1)
    x, w = torch.randn(M, 64, generator=g).to(d), (torch.randn(1, 96, 64, generator=g) * 0.1).to(d)
    res = torch.randn(M, 96, generator=g).to(d)
    y = ops.pair_conv(x, w, pl, res=res, act="relu")
    ref = torch.relu(x.double() @ w[0].double().t() + res.double())
    assert _close(y, ref) < 2e-6
    with pytest.raises(Exception):                              # the dense centre kernel of round 3 is gone: a centre offset is refused
        ops.pair_lists(nbr, M, center=0)


@pytest.mark.parametrize("lvl,cin,cout", [(0, 32, 32), (1, 32, 64), (2, 64, 128), (3, 128, 256)])
def test_down_convolution_row_lists_match_pos_based_pass2(scene_maps, lvl, cin, cout):
    c = _case(scene_maps, ("down", lvl), cin, cout, use_res=False)
    pl = c["pl"]
    assert pl.center == -1 and pl.rlist is not None and not pl.direct
    a = _run(c, pl)
    b = _run(c, _variant(pl, rlist=None))
    assert torch.equal(a, b), "same offsets in the same order: the row-list pass 2 must equal the pos-based one bit for bit"
    ref = _ref64(c["x"], c["w"], c["nbr"], c["scale"], c["shift"], None, c["act"])
    assert _close(a, ref) < 2e-6


@pytest.mark.parametrize("lvl,cin,cout,use_res", [(0, 96, 96, False), (1, 128, 96, True), (2, 256, 128, False), (3, 256, 256, True),
                                                   (0, 64, 32, False), (1, 96, 64, False)])
def test_transposed_convolution_direct_epilogue(scene_maps, lvl, cin, cout, use_res):
    c = _case(scene_maps, ("up", lvl), cin, cout, use_res=use_res)
    pl = c["pl"]
    assert pl.direct and pl.out_idx is not None and pl.rlist is None
    assert bool(((c["nbr"] >= 0).sum(dim=0) == 1).all()), "every fine voxel has exactly one parent"
    # out_idx[pos[k][r]] == r, -1 on the padding
    hit = pl.pos >= 0
    rows = torch.arange(pl.M, device=pl.pos.device).expand_as(pl.pos)
    assert torch.equal(pl.out_idx[pl.pos[hit].long()].long(), rows[hit])
    assert int((pl.out_idx >= 0).sum()) == pl.M
    from segdino3d_amd import ops
    direct = _run(c, pl)
    with ops.scenes_in_flight(4):                               # the lock-step kernels (the pipelined runner's choice)
        direct_ls = _run(c, pl)
    two_pass = _run(c, _variant(pl, direct=False))              # pass 1 -> partial products -> pos-based pass 2
    ref = _ref64(c["x"], c["w"], c["nbr"], c["scale"], c["shift"], c["res"], c["act"])
    print(f"up {lvl} {cin}->{cout}: direct == two-pass bitwise: {torch.equal(direct, two_pass)}, lock-step == weight-stationary: {torch.equal(direct, direct_ls)}")
    assert _close(direct, ref) < 2e-6 and _close(two_pass, ref) < 2e-6
    assert torch.equal(direct, direct_ls), "the two pass-1 variants must write the same bits"


def test_plan_and_eager_agree_with_the_new_paths():
    """The whole Res16UNet34C forward through sd3d_run_layers (tables carry rlist / centre / out_idx) equals the layer-by-layer path."""
    import segdino3d_amd as seg
    from segdino3d_amd import plan
    from segdino3d_amd.configs import scannet200_model_cfg
    from segdino3d_amd.synth import make_scene
    d = dev()
    torch.manual_seed(0)
    model = seg.build_architecture(scannet200_model_cfg(query_num=-1)).eval().to(d)
    pts, tgt = make_scene(2, 30_000, 400, 20)
    pts, tgt = pts.to(d), tgt.to(d)
    with torch.no_grad():
        a = model.backbone.forward_wrapper([pts], [tgt], return_sp_mean_pos=True)[0][0]
        old = plan.USE_PLAN
        plan.USE_PLAN = False
        try:
            b = model.backbone.forward_wrapper([pts], [tgt], return_sp_mean_pos=True)[0][0]
        finally:
            plan.USE_PLAN = old
    assert torch.equal(a, b)


# ---- chained lists (round 4): mirror offsets {k, K-1-k} and the centre share one partial product --------------------------------
@pytest.mark.parametrize("level,cin,cout,split", [(0, 96, 96, 0), (0, 128, 96, 96), (1, 32, 32, 0), (1, 96, 96, 0), (2, 64, 64, 0),
                                                  (2, 128, 128, 0), (2, 192, 128, 128), (3, 256, 256, 0), (3, 384, 256, 256), (4, 256, 256, 0)])
def test_chained_lists_give_the_plain_convolution(scene_maps, level, cin, cout, split):
    """`ops.pair_lists(..., center=PAIR_CHAINED)` on the real 3^3 tables of a scene: the convolution equals the float64 model at
    2e-6 of the row magnitude (like every other path) and the plain pair-major path at fp32 summation-order noise; it is
    bit-reproducible; the lists hold exactly the rulebook's entries and fewer partial rows."""
    from segdino3d_amd import ops
    c = _case(scene_maps, ("same", level, 3), cin, cout, split=split, seed=level * 10 + cout)
    nbr = c["nbr"]
    K, M = nbr.shape
    P = int((nbr >= 0).sum())
    chained = ops.pair_lists(nbr, P, center=ops.PAIR_CHAINED)
    assert chained.center == ops.PAIR_CHAINED and chained.rlist is not None
    n_tiles = int(chained.tile_k[chained.p_cap // 128].item())
    tk = chained.tile_k[:n_tiles]
    assert int((chained.in_idx[: n_tiles * 128] >= 0).sum()) == P, "a chained list holds every rulebook entry exactly once"
    n_partial_rows = int(chained.rlist[:, 0].sum())
    plain_rows = int(c["pl"].rlist[:, 0].sum()) if c["pl"].rlist is not None else P
    assert plain_rows == P and n_partial_rows < P
    assert bool(((tk & 0x3FFFFFFF) < K).all()) and not bool((tk[-1:] & 0x40000000).any())
    # every row's partial positions point at the LAST sub-tile of a chain
    pos = chained.rlist[:, 1:][torch.arange(chained.rlist.shape[1] - 1, device=nbr.device)[None] < chained.rlist[:, :1]]
    assert bool((tk[(pos // 128).long()] & 0x40000000).eq(0).all())
    x, x2 = (c["x1"], c["x2"]) if split else (c["x"], None)
    kw = dict(x2=x2, scale=c["scale"], shift=c["shift"], res=c["res"], act=c["act"])
    got = ops.pair_conv(x, c["w"], chained, **kw)
    again = ops.pair_conv(x, c["w"], chained, **kw)
    plain = ops.pair_conv(x, c["w"], c["pl"], **kw)
    ref = _ref64(c["x"], c["w"], nbr, c["scale"], c["shift"], c["res"], c["act"])
    mag = ref.abs().amax(dim=1, keepdim=True).clamp_min(1e-3)
    err = ((got.double() - ref).abs() / mag).max().item()
    err_plain = ((plain.double() - ref).abs() / mag).max().item()
    print(f"level {level} {cin}->{cout}: P = {P}, partial rows {plain_rows} -> {n_partial_rows} ({1 - n_partial_rows / plain_rows:.1%} fewer); "
          f"max error / row magnitude: chained {err:.2e}, plain {err_plain:.2e}")
    assert torch.equal(got, again)
    assert err < 2e-6, err


def test_chained_lists_edge_cases():
    """Isolated voxels (centre only), a table smaller than one tile, and rows whose only neighbours are one-sided."""
    from segdino3d_amd import ops
    from segdino3d_amd.sparse import SceneMaps
    d = dev()
    g = torch.Generator().manual_seed(5)
    # a few far-apart points, one tight cluster, and a straight line (every interior voxel has exactly the +-x mirror pair)
    iso = torch.rand(40, 3, generator=g) * 50.0
    cluster = 0.5 + 0.05 * torch.rand(300, 3, generator=g)
    line = torch.stack([torch.arange(200) * 0.02 + 10.0, torch.full((200,), 3.0), torch.full((200,), 3.0)], 1)
    pts = torch.cat([iso, cluster, line]).float()
    pts = torch.cat([pts, torch.zeros(pts.shape[0], 3)], 1).to(d)
    maps = SceneMaps(pts, 0.02, 2)
    maps.prepare(same=[(0, 3), (1, 3)], strides=[0])
    for lvl in (0, 1):
        nbr = maps.same(lvl, 3)
        K, M = nbr.shape
        P = int((nbr >= 0).sum())
        ch = ops.pair_lists(nbr, P, center=ops.PAIR_CHAINED)
        x = torch.randn(M, 32, generator=g).to(d)
        w = (torch.randn(K, 64, 32, generator=g) * 0.1).to(d)
        got = ops.pair_conv(x, w, ch)
        ref = _ref64(x, w, nbr, None, None, None, None)
        assert ((got.double() - ref).abs() / ref.abs().amax(dim=1, keepdim=True).clamp_min(1e-3)).max().item() < 2e-6
        assert int(ch.rlist[:, 0].min()) >= 1                       # every row has at least its centre


@pytest.mark.parametrize("which", ["same5", "same3", "down", "up", "chained"])
def test_lean_tables_equal_the_full_tables(scene_maps, which):
    """`ops.pair_lists(..., lean=True)` (round 5: the evaluation forward's tables - no [K, M] position table, built by the row-block
    kernels, the unused capacity behind the last tile left unwritten): in_idx / tile_k over the real tiles, the tile count, the
    per-row lists and out_idx are those of the full builder, entry for entry, and a convolution on them gives the same bits."""
    from segdino3d_amd import ops
    maps = scene_maps
    nbr, direct, center = {"same5": (maps.same(0, 5), False, -1), "same3": (maps.same(3, 3), False, -1), "down": (maps.down(1), False, -1),
                           "up": (maps.up(1), True, -1), "chained": (maps.same(1, 3), False, ops.PAIR_CHAINED)}[which]
    K, M = nbr.shape
    P = int((nbr >= 0).sum())
    full, lean = ops.pair_lists_batch([(nbr, K * M, center, direct, False), (nbr, K * M, center, direct, True)])   # worst-case capacity: a long unused tail
    assert lean.pos is None or which == "chained"
    nt = int(full.tile_k[full.p_cap // 128])
    assert nt == int(lean.tile_k[lean.p_cap // 128]) and full.p_cap == lean.p_cap
    assert full.tile_k[-3:].tolist() == lean.tile_k[-3:].tolist()
    assert torch.equal(full.tile_k[:nt], lean.tile_k[:nt]) and torch.equal(full.in_idx[: nt * 128], lean.in_idx[: nt * 128])
    if direct:
        assert torch.equal(full.out_idx[: nt * 128], lean.out_idx[: nt * 128])
    else:
        cnt = full.rlist[:, 0]
        assert torch.equal(cnt, lean.rlist[:, 0])
        live = torch.arange(full.rlist.shape[1] - 1, device=nbr.device)[None] < cnt[:, None]
        assert torch.equal(full.rlist[:, 1:][live], lean.rlist[:, 1:][live])
    g = torch.Generator().manual_seed(5)
    cin, cout = 32, 32
    x = torch.randn(int(nbr.max()) + 1, cin, generator=g).to(nbr.device)
    w = (torch.randn(K, cout, cin, generator=g) * 0.1).to(nbr.device)
    assert torch.equal(ops.pair_conv(x, w, full), ops.pair_conv(x, w, lean))


# ---- the shared tail of the lock-step pass 1 (round 5; epoch-stamped counters since round 6) ---------------------------------------
def _pool(on):
    from segdino3d_amd import _lib
    return _lib.load().sd3d_set_pair_pool(int(on))


def _pool_launches():
    import ctypes
    from segdino3d_amd import _lib
    n = ctypes.c_int64(0)
    assert _lib.load().sd3d_pair_pool_launches(ctypes.byref(n)) == 0
    return n.value


def _pool_check():
    from segdino3d_amd import _lib
    lib = _lib.load()
    n = lib.sd3d_pair_pool_check()
    assert n >= 0, lib.sd3d_last_error()
    return n


@pytest.fixture(scope="module")
def bench_scene_maps():
    """The benchmark scene (150 k points): pooled launches need >= 6 tiles per workgroup, i.e. real sizes."""
    from segdino3d_amd.sparse import SceneMaps
    from segdino3d_amd.synth import make_scene
    d = dev()
    pts, tgt = make_scene(0, 150_000, 3000, 300)
    maps = SceneMaps(pts.to(d), 0.02, 5, superpoints=tgt.extra_features["super_point_masks"].to(d))
    maps.prepare(same=[(0, 5)] + [(l, 3) for l in range(5)], strides=[0, 1, 2, 3])
    return maps


@pytest.mark.parametrize("level,cin,cout,split", [(0, 96, 96, 0), (0, 128, 96, 96), (1, 96, 96, 0), (2, 128, 128, 0), (2, 192, 128, 128),
                                                  (3, 256, 256, 0), (3, 384, 256, 256), (4, 256, 256, 0)])
def test_shared_tail_gives_the_same_bits(bench_scene_maps, level, cin, cout, split):
    """Pool on vs off in ONE process (`sd3d_set_pair_pool`), chained and plain tables on the lock-step kernel: `torch.equal`.  Then the
    ring is overwritten with garbage (what a launch that died half-way, or 16 384 launches of drift, would leave): the next launches
    still produce the same bits - a pooled launch assumes nothing about the slot it is handed - and `sd3d_pair_pool_check` stays 0 on the
    healthy ring."""
    from segdino3d_amd import _lib, ops
    lib = _lib.load()
    c = _case(bench_scene_maps, ("same", level, 3), cin, cout, split=split, seed=level * 7 + cout)
    nbr = c["nbr"]
    P = int((nbr >= 0).sum())
    chained = ops.pair_lists(nbr, P, center=ops.PAIR_CHAINED)
    x, x2 = (c["x1"], c["x2"]) if split else (c["x"], None)
    kw = dict(x2=x2, scale=c["scale"], shift=c["shift"], res=c["res"], act=c["act"])
    prev = _pool(1)
    try:
        with ops.scenes_in_flight(4):                           # several scenes in flight: the lock-step kernel for the plain tables too
            t0 = _pool_launches()
            on_c = ops.pair_conv(x, c["w"], chained, **kw)
            on_p = ops.pair_conv(x, c["w"], c["pl"], **kw)
            assert _pool_launches() == t0 + 2, "both launches must have taken a slot of the ring"
            assert _pool_check() == 0
            _pool(0)
            t1 = _pool_launches()
            off_c = ops.pair_conv(x, c["w"], chained, **kw)
            off_p = ops.pair_conv(x, c["w"], c["pl"], **kw)
            assert _pool_launches() == t1, "switched off: no ticket is taken"
            _pool(1)
            assert torch.equal(on_c, off_c) and torch.equal(on_p, off_p), "the shared tail must not change a bit"
            for word in (-1, 0x0000000100000000 * 7 + 5, 3):      # a future epoch, an old epoch mid-way, epoch 0 with a count
                assert lib.sd3d_pair_pool_poison(word) == 0, lib.sd3d_last_error()
                dirty_c = ops.pair_conv(x, c["w"], chained, **kw)
                dirty_p = ops.pair_conv(x, c["w"], c["pl"], **kw)
                assert torch.equal(dirty_c, on_c) and torch.equal(dirty_p, on_p), f"a dirty ring ({word:#x}) changed the result"
            assert _pool_check() > 0, "the check must see the poisoned ring"
            assert lib.sd3d_pair_pool_poison(0) == 0
            assert _pool_check() == 0
    finally:
        _pool(prev)


@pytest.mark.parametrize("n_scenes", [1, 4])
def test_shared_tail_through_the_whole_unet(n_scenes):
    """Every layer output of Res16UNet34C (the activation arena of the plan: all layers' rows back to back) at 150 k points per scene,
    one and four scenes per forward, with the shared tail on and off: bit-identical; the ring is healthy after the forwards, after the
    batched forward and after the pipelined runner."""
    import copy
    import segdino3d_amd as seg
    from segdino3d_amd.configs import scannet200_model_cfg
    from segdino3d_amd.dist_eval import PipelinedRunner
    from segdino3d_amd.synth import make_scene
    d = dev()
    torch.manual_seed(0)
    model = seg.build_architecture(scannet200_model_cfg(query_num=200)).eval().to(d)
    scenes = [make_scene(10 + i, 150_000, 3000, 300) for i in range(n_scenes)]
    pts = [p.to(d) for p, _ in scenes]
    tgts = [t.to(d) for _, t in scenes]
    arenas = {}
    prev = _pool(1)
    try:
        for on in (1, 0):
            _pool(on)
            with torch.no_grad(), seg.capture(keep_arena=True) as cap:
                model(pts, [copy.copy(t) for t in tgts])
            assert len(cap.arenas) >= 1
            arenas[on] = [a.clone() for a in cap.arenas]
        assert len(arenas[0]) == len(arenas[1])
        for a, b in zip(arenas[1], arenas[0]):
            assert a.shape == b.shape and torch.equal(a, b), "a U-Net layer output differs between shared tail on and off"
        _pool(1)
        assert _pool_check() == 0
        runner = PipelinedRunner(model, 2, d, batch=2)
        with torch.no_grad():
            runner.run([(pts[i % n_scenes], copy.copy(tgts[i % n_scenes])) for i in range(6)])
        torch.cuda.synchronize()
        assert _pool_check() == 0
    finally:
        _pool(prev)


def test_pass2_without_any_list_is_refused(scene_maps):
    """a lean table has no position table; asking for the pos-based pass 2 on it must fail with an argument error, not fault."""
    from segdino3d_amd import ops
    c = _case(scene_maps, ("same", 1, 3), 32, 32)
    nbr = c["nbr"]
    lean = ops.pair_lists_batch([(nbr, int((nbr >= 0).sum()), -1, False, True)])[0]
    assert lean.pos is None and lean.rlist is not None
    ok = ops.pair_conv(c["x"], c["w"], lean)
    assert torch.isfinite(ok).all()
    with pytest.raises(RuntimeError, match="pass 2 needs"):
        ops.pair_conv(c["x"], c["w"], _variant(lean, rlist=None))
