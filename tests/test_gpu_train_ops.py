"""Backward of the sparse convolution (segdino3d_amd/train_ops.py, csrc/pair_wgrad.hip) against a
float64 gather + matmul model of the same convolution differentiated by torch autograd.
Tolerance: 3e-5 of the largest entry (fp32 MFMA accumulation over up to ~1e4 pairs per offset)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def scene():
    from segdino3d_amd.sparse import SceneMaps
    from segdino3d_amd.synth import make_scene
    d = dev()
    pts, tgt = make_scene(3, 30_000, 600, 40)
    pts = pts.to(d)
    maps = SceneMaps(pts, 0.02, 5, superpoints=tgt.extra_features["super_point_masks"].to(d))
    maps.prepare(same=[(0, 5)] + [(l, 3) for l in range(5)], strides=[0, 1, 2, 3])
    return maps


def reference(x, w, nbr):
    """float64 on the device: y[r] = sum_k x[nbr[k][r]] @ w[k]^T."""
    y = 0
    for k in range(nbr.shape[0]):
        idx = nbr[k].long()
        ok = (idx >= 0).unsqueeze(1)
        y = y + torch.where(ok, x[idx.clamp(min=0)], torch.zeros((), dtype=x.dtype, device=x.device)) @ w[k].T
    return y


CASES = [("same", 0, 3, 32, 32), ("same", 1, 3, 64, 96), ("same", 2, 3, 128, 128), ("same", 3, 3, 256, 256), ("same", 0, 5, 32, 32),
         ("down", 0, 2, 32, 64), ("down", 2, 2, 96, 128), ("up", 1, 2, 128, 64), ("up", 3, 2, 256, 160), ("same", 4, 3, 192, 224)]


@pytest.mark.parametrize("kind,level,ks,cin,cout", CASES)
def test_sparse_conv_gradients(scene, kind, level, ks, cin, cout):
    from segdino3d_amd import train_ops
    maps = scene
    d = maps.keys[0].device
    tab = maps.conv_table(kind, level, ks) if kind == "same" else maps.conv_table(kind, level)
    nbr = tab["nbr"]
    K, M = nbr.shape
    n_in = int(nbr.max().item()) + 1
    g = torch.Generator().manual_seed(level * 10 + cin)
    x = torch.randn(n_in, cin, generator=g).to(d).requires_grad_(True)
    w = (torch.randn(K, cout, cin, generator=g) * (K * cin) ** -0.5).to(d).requires_grad_(True)
    dy = torch.randn(M, cout, generator=g).to(d)
    y = train_ops.sparse_conv(x, w, maps, kind, level, ks)
    y.backward(dy)
    x64, w64 = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    y64 = reference(x64, w64, nbr)
    y64.backward(dy.double())
    assert (y.detach().double() - y64.detach()).abs().max().item() < 3e-5 * y64.abs().max().item()
    for name, got, ref in (("dx", x.grad, x64.grad), ("dw", w.grad, w64.grad)):
        err = (got.double() - ref).abs().max().item()
        assert err <= 3e-5 * ref.abs().max().item(), (name, err, ref.abs().max().item())
    # bit-reproducible
    dw2 = train_ops.pair_wgrad(dy, x.detach(), tab["pairs"])
    assert torch.equal(dw2, w.grad)
    # accumulate into an existing gradient
    acc = w.grad.clone()
    train_ops.pair_wgrad(dy, x.detach(), tab["pairs"], dw=acc, accumulate=True)
    assert torch.equal(acc, w.grad + w.grad)


def test_wgrad_counts_pairs_exactly(scene):
    """x = 1, dy = 1 -> dW[k][co][ci] = number of pairs of offset k (integers: exact in fp32)."""
    from segdino3d_amd import train_ops
    maps = scene
    d = maps.keys[0].device
    for key in [("same", 0, 3), ("down", 1), ("up", 0)]:
        tab = maps.conv_table(*key)
        nbr = tab["nbr"]
        K, M = nbr.shape
        n_in = int(nbr.max().item()) + 1
        dw = train_ops.pair_wgrad(torch.ones(M, 32, device=d), torch.ones(n_in, 64, device=d), tab["pairs"])
        counts = (nbr >= 0).sum(dim=1).float()
        assert torch.equal(dw, counts.view(K, 1, 1).expand(K, 32, 64))


def test_scalar_loss_gradient_reaches_conv_weights(scene):
    """A scalar loss built with torch ops flows through SparseConv into the weights (autograd plumbing)."""
    from segdino3d_amd import train_ops
    maps = scene
    d = maps.keys[0].device
    M = maps.n_vox[2]
    g = torch.Generator().manual_seed(9)
    x = torch.randn(M, 32, generator=g).to(d)
    w = (torch.randn(27, 32, 32, generator=g) * 0.03).to(d).requires_grad_(True)
    y = train_ops.sparse_conv(x, w, maps, "same", 2, 3)
    loss = (y * y).mean()
    loss.backward()
    x64, w64 = x.double(), w.detach().double().requires_grad_(True)
    (reference(x64, w64, maps.conv_table("same", 2, 3)["nbr"]) ** 2).mean().backward()
    assert (w.grad.double() - w64.grad).abs().max().item() <= 3e-5 * w64.grad.abs().max().item()


def test_batch_norm_act_matches_float64(scene):
    """Batch-statistics BatchNorm + residual + ReLU (csrc/train.hip) against torch float64: outputs, gradients of the
    input / residual / gamma / beta, and the running statistics nn.BatchNorm1d would hold."""
    from segdino3d_amd import train_ops
    d = dev()
    g = torch.Generator().manual_seed(4)
    for M, C, act, with_res in [(30_011, 96, "relu", True), (5_000, 32, "relu", False), (777, 256, None, True), (2049, 128, None, False)]:
        x = (torch.randn(M, C, generator=g) * 3 + 50).to(d).requires_grad_(True)       # large mean: the variance must not cancel
        res = torch.randn(M, C, generator=g).to(d).requires_grad_(True) if with_res else None
        bn = torch.nn.BatchNorm1d(C, eps=1e-5, momentum=0.02).to(d)
        with torch.no_grad():
            bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(C, generator=g))
        dy = torch.randn(M, C, generator=g).to(d)
        y = train_ops.batch_norm_act(x, bn, res=res, act=act)
        y.backward(dy)
        bn64 = torch.nn.BatchNorm1d(C, eps=1e-5, momentum=0.02).to(d).double()
        with torch.no_grad():
            bn64.weight.copy_(bn.weight); bn64.bias.copy_(bn.bias)
        x64 = x.detach().double().requires_grad_(True)
        r64 = res.detach().double().requires_grad_(True) if with_res else None
        y64 = bn64(x64) + (r64 if with_res else 0)
        if act == "relu":                                   # the fp32 result's own mask: outputs within rounding of 0 may differ in sign
            y64 = y64 * (y.detach() > 0)
        y64.backward(dy.double())
        tol = lambda ref: 2e-5 * max(ref.abs().max().item(), 1e-3)
        assert (y.detach().double() - y64.detach()).abs().max().item() <= tol(y64)
        assert (x.grad.double() - x64.grad).abs().max().item() <= tol(x64.grad)
        assert (bn.weight.grad.double() - bn64.weight.grad).abs().max().item() <= 5e-5 * bn64.weight.grad.abs().max().item()
        assert (bn.bias.grad.double() - bn64.bias.grad).abs().max().item() <= 5e-5 * bn64.bias.grad.abs().max().item()
        if with_res:
            assert (res.grad.double() - r64.grad).abs().max().item() <= tol(r64.grad)
        assert (bn.running_mean.double() - bn64.running_mean).abs().max().item() <= 1e-5
        assert (bn.running_var.double() - bn64.running_var).abs().max().item() <= 1e-5 * bn64.running_var.abs().max().item()
        assert int(bn.num_batches_tracked) == 1


def _backbone_training_case(smooth):
    """-> (features, {param: grad}) of the HIP path, and a function dtype -> the same from the oracle."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from _det import det_param
    from oracle import sparse_ref as R
    from segdino3d_amd import train_ops
    from segdino3d_amd.backbone_mink import Res16UNet34C
    from segdino3d_amd.synth import make_scene
    d = dev()
    pts, tgt = make_scene(14, n_points=20000, n_superpoints=150, n_query2d=20)
    m = Res16UNet34C(in_channels=259, out_channels=96, config=dict(dilations=[1, 1, 1, 1], conv1_kernel_size=5, bn_momentum=0.02),
                     voxel_size=0.02, mode_fuse_2d_feat="early_fusion", add_positional_embedding=True)
    sd = {k: det_param("backbone." + k, v.shape).to(v.dtype) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m.to(d).train()
    g = torch.Generator().manual_seed(1)
    train_ops.TrainBackend.IGNORE_ACT = smooth
    try:
        f, pos, _ = m.forward_wrapper([pts.to(d)], [tgt.to(d)], return_sp_mean_pos=True)
        R_w = torch.randn(f[0].shape, generator=g)
        (f[0] * R_w.to(d)).sum().backward()
    finally:
        train_ops.TrainBackend.IGNORE_ACT = False
    tgt = tgt.to("cpu")

    def oracle_grads(dtype):
        rsd = {"backbone." + k: (v.to(dtype).requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
        R.BN_TRAIN = True
        relu = torch.relu
        if smooth:
            torch.relu = lambda x: x
        try:
            rf, _, _ = R.mink_forward_wrapper(rsd, pts.to(dtype), tgt.extra_features["points_2dfeats"].to(dtype),
                                              tgt.extra_features["super_point_masks"])
            (rf * R_w.to(dtype)).sum().backward()
        finally:
            R.BN_TRAIN = False
            torch.relu = relu
        return rf.detach(), {k[len("backbone."):]: v.grad for k, v in rsd.items() if v.is_floating_point() and v.grad is not None}

    return f[0].detach().cpu(), {n: p.grad.cpu() for n, p in m.named_parameters()}, oracle_grads


def _rel(a, b):
    return ((a.double() - b).norm() / b.norm().clamp(min=1e-30)).item()


def test_res16unet34c_backward_without_relu_matches_float64_oracle():
    """The whole training-mode backbone with the ReLUs dropped on both sides (convolutions, batch-statistics BatchNorm,
    residuals, skip concatenations, superpoint pooling): a smooth function, so every parameter gradient must agree with
    the float64 oracle to fp32 rounding (relative L2 <= 2e-4 after 34 layers)."""
    f, grads, oracle = _backbone_training_case(smooth=True)
    rf, ref = oracle(torch.float64)
    assert (f.double() - rf).abs().max().item() <= 2e-4 * max(rf.abs().max().item(), 1.0)
    worst = sorted(((_rel(grads[n], ref[n]), n) for n in grads), reverse=True)
    assert set(grads) == set(ref)
    assert worst[0][0] <= 2e-4, f"parameter gradients differ: {worst[:5]}"


def test_res16unet34c_training_step_matches_float64_oracle():
    """With the ReLUs: an output within rounding of zero switches its mask between two roundings of the same network,
    which moves the gradients of the coarse levels (a few hundred voxels) by up to ~1 %.  The yardstick is therefore the
    oracle itself run in float32: the HIP path must sit about as close to the float64 gradients as that does."""
    f, grads, oracle = _backbone_training_case(smooth=False)
    rf, ref = oracle(torch.float64)
    _, ref32 = oracle(torch.float32)
    err = (f.double() - rf).abs().max().item()
    assert err <= 1e-3 * max(rf.abs().max().item(), 1.0), f"training-mode features differ: {err}"
    rows = sorted(((_rel(grads[n], ref[n]), _rel(ref32[n], ref[n]), n) for n in grads), reverse=True)
    floor = max(r[1] for r in rows)
    med = lambda i: sorted(r[i] for r in rows)[len(rows) // 2]
    print(f"gradient error vs float64: worst {rows[0][0]:.2e} (float32 oracle {floor:.2e}), median {med(0):.2e} (float32 oracle {med(1):.2e})")
    assert rows[0][0] <= max(4.0 * floor, 2e-3), f"parameter gradients differ (mine, float32 oracle, name): {rows[:5]}"
    assert med(0) <= max(4.0 * med(1), 1e-3), f"median gradient error {med(0)} vs float32 oracle {med(1)}"


@pytest.mark.parametrize("normalize_before", [True, False])
def test_spconvunet_backward_without_relu_matches_float64_oracle(normalize_before):
    """SpConvUNet (ScanNetv2 prototype) in training mode, ReLUs dropped on both sides: pre-activation BatchNorms
    (or, normalize_before=False, BatchNorm after each convolution and the identity added after the block - spconvunet.py:66-81),
    1x1 identity branches, stride-2 / transposed convolutions, skip concatenations - gradients of every parameter
    against the float64 oracle (relative L2 <= 2e-4)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from _det import det_param
    from oracle import sparse_ref as R
    from segdino3d_amd import train_ops
    from segdino3d_amd.backbone_spconv import SpConvUNet
    from segdino3d_amd.synth import make_scene
    d = dev()
    pts, tgt = make_scene(15, n_points=10000, n_superpoints=100, n_query2d=20)
    m = SpConvUNet(num_planes=[32 * (i + 1) for i in range(5)], return_blocks=True, voxel_size=0.02,
                   mode_fuse_2d_feat="early_fusion", add_positional_embedding=True, normalize_before=normalize_before)
    sd = {k: det_param("backbone." + k, v.shape).to(v.dtype) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m.to(d).train()
    g = torch.Generator().manual_seed(2)
    train_ops.TrainBackend.IGNORE_ACT = True
    try:
        f, _, _ = m.forward_wrapper([pts.to(d)], [tgt.to(d)], return_sp_mean_pos=True)
        R_w = torch.randn(f[0].shape, generator=g)
        (f[0] * R_w.to(d)).sum().backward()
    finally:
        train_ops.TrainBackend.IGNORE_ACT = False
    tgt = tgt.to("cpu")
    rsd = {"backbone." + k: (v.double().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    R.BN_TRAIN = True
    relu, floor_voxel = torch.relu, R.floor_voxel
    torch.relu = lambda x: x
    R.floor_voxel = lambda xyz, vs: floor_voxel(xyz.float(), vs)              # voxelise exactly as the fp32 pipeline does
    try:
        rf, _, _ = R.spconv_forward_wrapper(rsd, pts.double(), tgt.extra_features["points_2dfeats"].double(),
                                            tgt.extra_features["super_point_masks"], normalize_before=normalize_before)
        (rf * R_w.double()).sum().backward()
    finally:
        R.BN_TRAIN = False
        torch.relu, R.floor_voxel = relu, floor_voxel
    assert (f[0].detach().cpu().double() - rf.detach()).abs().max().item() <= 2e-4 * max(rf.abs().max().item(), 1.0)
    grads = {n: (p.grad.cpu(), rsd["backbone." + n].grad) for n, p in m.named_parameters()}
    # normalize_before=False with the ReLUs dropped: the last block's BatchNorm bias is a per-channel constant in front of
    # output_layer's batch-statistics BatchNorm - its gradient is zero in exact arithmetic (1e-20 in float64, rounding noise in
    # fp32), so it is held to an absolute bound relative to the other bias gradients instead of a relative one
    scale = max(g64.norm().item() for n, (_, g64) in grads.items() if n.endswith(".bias"))
    null = [n for n, (_, g64) in grads.items() if g64.norm().item() < 1e-9 * scale]
    assert null == ([] if normalize_before else ["blocks_tail.block1.conv_branch.4.bias"]), null
    for n in null:
        assert grads[n][0].norm().item() < 1e-5 * scale, (n, grads[n][0].norm().item(), scale)
    worst = sorted(((_rel(g, g64), n) for n, (g, g64) in grads.items() if n not in null), reverse=True)
    assert worst[0][0] <= 2e-4, f"parameter gradients differ: {worst[:5]}"


@pytest.mark.parametrize("n_scenes", [1, 2])
def test_train_plan_equals_the_autograd_node_path(n_scenes):
    """The U-Net's training step as ONE autograd node over two C calls (`segdino3d_amd/train_plan.py`, csrc/train_plan.hip) against the
    node-per-layer path (`train_ops.TrainBackend`): same kernels, same order per tensor - the features are the same bits, every parameter
    gradient agrees to fp32 association noise (a tensor with three consumers adds its gradients in another order), the BatchNorm running
    statistics advance identically.  One scene and a batch of two (block-diagonal tables, batch statistics)."""
    import copy, os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from _det import det_param
    from segdino3d_amd import train_plan
    from segdino3d_amd.backbone_mink import Res16UNet34C
    from segdino3d_amd.synth import make_scene
    d = dev()
    scenes = [make_scene(50 + i, n_points=15000 + 3000 * i, n_superpoints=120, n_query2d=12) for i in range(n_scenes)]
    m = Res16UNet34C(in_channels=259, out_channels=96, config=dict(dilations=[1, 1, 1, 1], conv1_kernel_size=5, bn_momentum=0.02),
                     voxel_size=0.02, mode_fuse_2d_feat="early_fusion", add_positional_embedding=True)
    sd = {k: det_param("backbone." + k, v.shape).to(v.dtype) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m.to(d).train()
    g = torch.Generator().manual_seed(2)
    res = {}
    old = train_plan.USE_TRAIN_PLAN
    try:
        for use in (True, False):
            train_plan.USE_TRAIN_PLAN = use
            m.load_state_dict(sd)
            m.zero_grad(set_to_none=True)
            feats, _, _ = m.forward_wrapper([p.to(d) for p, _ in scenes], [copy.copy(t).to(d) for _, t in scenes], return_sp_mean_pos=True)
            if use:
                assert m._train_plan is not None and m._train_plan.n == 62
                R_w = [torch.randn(f.shape, generator=g).to(d) for f in feats]
            sum((f * w).sum() for f, w in zip(feats, R_w)).backward()
            res[use] = ([f.detach().clone() for f in feats], {n: p.grad.detach().clone() for n, p in m.named_parameters()},
                        {n: b.detach().clone() for n, b in m.named_buffers()})
    finally:
        train_plan.USE_TRAIN_PLAN = old
    for a, b in zip(res[True][0], res[False][0]):
        assert torch.equal(a, b), "the forward of the plan must give the bits of the node-per-layer path"
    assert set(res[True][1]) == set(res[False][1])
    worst = sorted(((_rel(res[True][1][n], res[False][1][n].double()), n) for n in res[True][1]), reverse=True)
    print("plan vs autograd nodes, worst parameter gradients:", worst[:3])
    assert worst[0][0] <= 2e-5, worst[:5]
    for n in res[True][2]:
        assert torch.equal(res[True][2][n], res[False][2][n]), f"running statistics differ: {n}"


def test_batched_training_uses_batch_statistics_like_minkowski():
    """Two scenes in one training batch: ME collates them into ONE sparse tensor (minkunet.py:624-627), so every BatchNorm takes
    its statistics over the voxels of both scenes while convolutions stay inside their scene.  Oracle: the single-scene network
    on the union of the two voxel sets, the second shifted by 16384 voxels (a multiple of every tensor stride: no neighbour
    relation crosses, the coarse levels stay aligned).  ReLUs dropped on both sides (smooth network, see above)."""
    import os, sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from _det import det_param
    from oracle import sparse_ref as R
    from segdino3d_amd import train_ops
    from segdino3d_amd.backbone_mink import Res16UNet34C
    from segdino3d_amd.synth import make_scene
    d = dev()
    scenes = [make_scene(40, n_points=9000, n_superpoints=90, n_query2d=10), make_scene(41, n_points=12000, n_superpoints=110, n_query2d=10)]
    m = Res16UNet34C(in_channels=259, out_channels=96, config=dict(dilations=[1, 1, 1, 1], conv1_kernel_size=5, bn_momentum=0.02),
                     voxel_size=0.02, mode_fuse_2d_feat="early_fusion", add_positional_embedding=True)
    sd = {k: det_param("backbone." + k, v.shape).to(v.dtype) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m.to(d).train()
    g = torch.Generator().manual_seed(3)
    train_ops.TrainBackend.IGNORE_ACT = True
    try:
        feats, _, _ = m.forward_wrapper([p.to(d) for p, _ in scenes], [t.to(d) for _, t in scenes], return_sp_mean_pos=True)
        R_w = [torch.randn(f.shape, generator=g) for f in feats]
        sum((f * w.to(d)).sum() for f, w in zip(feats, R_w)).backward()
    finally:
        train_ops.TrainBackend.IGNORE_ACT = False
    # ---- oracle on the union
    rsd = {"backbone." + k: (v.double().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    ucs, invs, vfs = [], [], []
    for i, (pts, tgt) in enumerate(scenes):
        tgt = tgt.to("cpu")
        c = R.floor_voxel(pts[:, :3], 0.02)
        uc, inv = R.unique_voxels(c)
        f = torch.cat([pts[:, 3:], tgt.extra_features["points_2dfeats"]], dim=1).double()
        vfs.append(R.segment_mean(f, inv, len(uc)))
        ucs.append(uc + np.array([16384 * i, 0, 0], dtype=uc.dtype))
        invs.append(inv)
    R.BN_TRAIN = True
    relu = torch.relu
    torch.relu = lambda t: t
    try:
        x = R.res16unet34c(rsd, np.concatenate(ucs), torch.cat(vfs), "backbone.", 5, "x_fastest")
    finally:
        R.BN_TRAIN = False
        torch.relu = relu
    n0 = len(ucs[0])
    ref_feats = []
    for i, (pts, tgt) in enumerate(scenes):
        xi = x[:n0] if i == 0 else x[n0:]
        sp = tgt.extra_features["super_point_masks"].cpu()
        ref_feats.append(R.segment_mean(xi[torch.from_numpy(invs[i])], sp.numpy(), int(sp.max()) + 1))
    sum((f * w.double()).sum() for f, w in zip(ref_feats, R_w)).backward()
    for f, r in zip(feats, ref_feats):
        assert (f.detach().cpu().double() - r.detach()).abs().max().item() <= 2e-4 * max(r.abs().max().item(), 1.0)
    worst = sorted(((_rel(p.grad.cpu(), rsd["backbone." + n].grad), n) for n, p in m.named_parameters()), reverse=True)
    assert worst[0][0] <= 2e-4, f"parameter gradients differ: {worst[:5]}"
    # and the batch statistics really differ from per-scene statistics: one scene alone gives other features
    m.zero_grad()
    train_ops.TrainBackend.IGNORE_ACT = True
    try:
        alone, _, _ = m.forward_wrapper([scenes[0][0].to(d)], [scenes[0][1].to(d)], return_sp_mean_pos=True)
    finally:
        train_ops.TrainBackend.IGNORE_ACT = False
    assert (alone[0].detach() - feats[0].detach()).abs().max().item() > 1e-3


def test_spconv_batched_training_runs_on_one_block_diagonal_tensor():
    """SpConvUNet with two scenes in training mode: batch-wide BatchNorm statistics (the features of a scene depend on its
    batch mate), gradients reach every parameter, and one scene alone still matches the single-scene path."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from _det import det_param
    from segdino3d_amd.backbone_spconv import SpConvUNet
    from segdino3d_amd.synth import make_scene
    d = dev()
    scenes = [make_scene(50, n_points=6000, n_superpoints=60, n_query2d=10), make_scene(51, n_points=8000, n_superpoints=70, n_query2d=10)]
    m = SpConvUNet(num_planes=[32 * (i + 1) for i in range(5)], return_blocks=True, voxel_size=0.02,
                   mode_fuse_2d_feat="early_fusion", add_positional_embedding=True)
    sd = {k: det_param("backbone." + k, v.shape).to(v.dtype) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m.to(d).train()
    both, pos, pos_wo = m.forward_wrapper([p.to(d) for p, _ in scenes], [t.to(d) for _, t in scenes], return_sp_mean_pos=True)
    sum((f * f).mean() for f in both).backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())
    assert both[0].shape == (60, 32) and both[1].shape == (70, 32) and pos[1].shape == (70, 3)
    alone, _, _ = m.forward_wrapper([scenes[0][0].to(d)], [scenes[0][1].to(d)], return_sp_mean_pos=True)
    assert (alone[0].detach() - both[0].detach()).abs().max().item() > 1e-4      # other statistics
    m.eval()
    with torch.no_grad():
        e2, _, _ = m.forward_wrapper([p.to(d) for p, _ in scenes], [t.to(d) for _, t in scenes], return_sp_mean_pos=True)
        e1, _, _ = m.forward_wrapper([scenes[1][0].to(d)], [scenes[1][1].to(d)], return_sp_mean_pos=True)
    assert torch.equal(e2[1], e1[0])                                           # eval: scenes are independent
