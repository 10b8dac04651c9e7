"""EXPERIMENTAL library (`make -C segdino3d_amd/csrc experimental`, include/segdino3d_hip_experimental.h) - not on the product path.
Output-stationary sparse convolution (csrc/experimental/slab_conv.hip, C entry sd3d_slab_conv) against a float64 gather + matmul model
and against the pair-major kernel it replaces, on real neighbour tables: every (Cin, Cout) shape of the two U-Nets,
concatenated inputs, residual, folded BatchNorm, ReLU, k=2 stride / transposed tables, the 5^3 stem, levels with few rows
(offset split + slab_reduce_kernel), a ragged last slab, rows without any neighbour, and run-to-run determinism."""
import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.experimental]


def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from segdino3d_amd import experimental
    if not experimental.available():
        pytest.skip("libsegdino3d_hip_experimental.so is not built (make -C segdino3d_amd/csrc experimental)")
    return torch.device("cuda:0")


def _fp64_conv(x, w, nbr, scale, shift, res, relu):
    idx = nbr.cpu().long()
    K, M = idx.shape
    cout = w.shape[1]
    ref = torch.zeros(M, cout, dtype=torch.float64)
    mag = torch.zeros(M, cout, dtype=torch.float64)
    xx, ww = x.double(), w.double()
    for k in range(K):
        ok = idx[k] >= 0
        rows = xx[idx[k].clamp(min=0)] * ok[:, None]
        ref += rows @ ww[k].T
        mag += rows.abs() @ ww[k].abs().T
    if scale is not None:
        ref, mag = ref * scale.double() + shift.double(), mag * scale.double() + shift.abs().double()
    if res is not None:
        ref, mag = ref + res.double(), mag + res.abs().double()
    if relu:
        ref = torch.relu(ref)
    return ref, mag + 1e-30


@pytest.fixture(scope="module")
def maps():
    from segdino3d_amd.sparse import SceneMaps
    from segdino3d_amd.synth import make_scene
    d = dev()
    pts, tgt = make_scene(6, 40000, 400, 40)
    m = SceneMaps(pts.to(d), 0.02, 5, superpoints=tgt.extra_features["super_point_masks"].to(d))
    m.prepare(same=[(0, 5)] + [(l, 3) for l in range(5)], strides=[0, 1, 2, 3])
    return m


CASES = [  # (table, Cin, Cout, concat split or 0)
    (("same", 1, 3), 32, 32, 0), (("same", 2, 3), 32, 64, 0), (("same", 2, 3), 64, 64, 0), (("same", 3, 3), 64, 128, 0),
    (("same", 3, 3), 128, 128, 0), (("same", 4, 3), 128, 256, 0), (("same", 4, 3), 256, 256, 0), (("same", 3, 3), 384, 256, 256),
    (("same", 2, 3), 192, 128, 128), (("same", 1, 3), 128, 96, 96), (("same", 0, 3), 96, 96, 0), (("same", 0, 3), 128, 96, 96),
    (("down", 0), 32, 32, 0), (("down", 2), 64, 64, 0), (("up", 3), 256, 256, 0), (("up", 2), 256, 128, 0), (("up", 0), 96, 96, 0),
    (("same", 0, 5), 32, 32, 0),
]


@pytest.mark.parametrize("key,cin,cout,split", CASES)
def test_slab_conv_matches_fp64_and_pair_major(maps, key, cin, cout, split):
    from segdino3d_amd import experimental, ops
    d = dev()
    tab = maps.conv_table(*key)
    nbr, pairs = tab["nbr"], tab["pairs"]
    K, M = nbr.shape
    P = int((nbr >= 0).sum())
    assert experimental.slab_conv_supported(K, cin, cout, M, P)
    g = torch.Generator().manual_seed(hash((key, cin, cout)) % 1000)
    n_in = int(nbr.max().item()) + 1
    x = torch.randn(n_in, cin, generator=g) * torch.exp(0.5 * torch.randn(n_in, 1, generator=g))
    w = torch.randn(K, cout, cin, generator=g) * (K * cin) ** -0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    res = torch.randn(M, cout, generator=g)
    xd, wd = x.to(d), w.to(d)
    xin, x2 = (xd[:, :split], xd[:, split:]) if split else (xd, None)
    for kw, (sc, sh, rs, relu) in ((dict(scale=scale.to(d), shift=shift.to(d), res=res.to(d), act="relu"), (scale, shift, res, True)),
                                   (dict(), (None, None, None, False))):
        got = experimental.slab_conv(xin, wd, nbr, n_pairs=P, x2=x2, **kw)
        again = experimental.slab_conv(xin, wd, nbr, n_pairs=P, x2=x2, **kw)
        assert torch.equal(got, again), "slab_conv is not deterministic"
        ref, mag = _fp64_conv(x, w, nbr, sc, sh, rs, relu)
        err = ((got.double().cpu() - ref).abs() / mag).max().item()
        old = ops.pair_conv(xin, wd, pairs, x2=x2, **kw)
        err_old = ((old.double().cpu() - ref).abs() / mag).max().item()
        print(f"{key} {cin}->{cout} M={M} P={P}: slab err {err:.2e} of the row magnitude (pair-major {err_old:.2e})")
        assert err < 2e-6, err                     # fp32 sums of <= 27 x 384 products: a few ulp of the magnitude


def test_unsupported_shapes_are_reported_not_guessed():
    """Shapes without a kernel variant (the 288-channel stem, odd widths) answer 0 workspace bytes = "keep the pair-major
    path"; calling anyway raises."""
    from segdino3d_amd import experimental, ops
    d = dev()
    assert not experimental.slab_conv_supported(125, 288, 32, 1000, 5000)
    assert not experimental.slab_conv_supported(27, 96, 100, 1000, 5000)
    with pytest.raises(ValueError):
        experimental.slab_conv(torch.zeros(10, 288, device=d), torch.zeros(125, 32, 288, device=d), torch.zeros(125, 10, dtype=torch.int32, device=d))


def test_rows_without_neighbours_and_ragged_slabs():
    """A table whose rows mostly have NO neighbour at all (output = epilogue of zero), M not a multiple of the slab size, and
    a single-row table."""
    from segdino3d_amd import experimental, ops
    d = dev()
    g = torch.Generator().manual_seed(0)
    for M, n_in, K, p in ((1000 + 37, 500, 27, 0.02), (1, 5, 8, 0.5), (70, 70, 27, 1.0)):
        nbr = torch.where(torch.rand(K, M, generator=g) < p, torch.randint(0, n_in, (K, M), generator=g), torch.full((K, M), -1)).int()
        x, w = torch.randn(n_in, 64, generator=g), torch.randn(K, 64, 64, generator=g) * 0.05
        shift = torch.randn(64, generator=g)
        got = experimental.slab_conv(x.to(d), w.to(d), nbr.to(d), shift=shift.to(d), act="relu").double().cpu()
        ref, mag = _fp64_conv(x, w, nbr, torch.ones(64), shift, None, True)
        assert ((got - ref).abs() / mag).max().item() < 2e-6
        empty = (nbr < 0).all(dim=0)
        if empty.any():
            assert torch.equal(got[empty], torch.relu(shift.double()).expand(int(empty.sum()), 64))
