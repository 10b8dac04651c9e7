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
