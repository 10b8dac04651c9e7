"""Backward of the sparse convolution (first kernels of the training step).

The reference gets these gradients from MinkowskiEngine / spconv autograd (`train_engine_3d.py:88-122` calls
`loss.backward()` on a graph whose convolutions are `ME.MinkowskiConvolution` / `spconv.SubMConv3d`,
`segdino3d/models/backbone/backbone_3d/mink_unet.py`, `spconv_unet.py`).  Here:

* input gradient = the FORWARD pair-major convolution (`ops.pair_conv`) on the transposed rulebook with transposed
  weights.  No transposed table is ever built: a submanifold table is its own transpose with the offsets mirrored
  (`nbr[K-1-k][i] = r  <=>  nbr[k][r] = i`), and the stride-2 "down" and "up" tables are each other's transposes
  offset by offset (tests/test_gpu_fullsize.py checks both properties on real scenes);
* weight gradient = `sd3d_pair_wgrad` (csrc/pair_wgrad.hip): per offset a [Cout x pairs] x [pairs x Cin] product on
  the fp32 matrix cores, fixed summation order.

`SparseConv.apply` ties both to torch autograd so that a loss computed from its output (e.g. by
`criterion.ScanNetUnifiedCriterion`) reaches the weights.  No CPU fallback.
"""
from __future__ import annotations

import torch

from . import _lib, ops

_WS = ops._PerThread()


def pair_out_rows(pairs: "ops.PairLists") -> torch.Tensor:
    """out_idx[p] = output row of pair p (-1 on padding); cached on the lists."""
    if pairs.out_idx is None:
        lib = _lib.load()
        out = torch.empty(pairs.p_cap, dtype=torch.int32, device=pairs.pos.device)
        _lib.check(lib.sd3d_pair_out_rows(pairs.pos.data_ptr(), pairs.K, pairs.M, pairs.p_cap, out.data_ptr(), ops._stream()), "pair_out_rows")
        pairs.out_idx = out
    return pairs.out_idx


def pair_wgrad(dy: torch.Tensor, x: torch.Tensor, pairs: "ops.PairLists", dw: torch.Tensor = None, accumulate: bool = False) -> torch.Tensor:
    """dw[k] (+)= dy[out rows of offset k]^T @ x[in rows of offset k]; dy [M, Cout], x [V_in, Cin] -> dw [K, Cout, Cin]."""
    lib = _lib.load()
    pdy, ldy = ops._rows(dy, "dy")
    px, ldx = ops._rows(x, "x")
    Cout, Cin, K = dy.shape[1], x.shape[1], pairs.K
    if dy.shape[0] != pairs.M:
        raise ValueError(f"dy has {dy.shape[0]} rows, the rulebook {pairs.M} outputs")
    if dw is None:
        dw = torch.empty(K, Cout, Cin, dtype=torch.float32, device=dy.device)
        accumulate = False
    elif dw.shape != (K, Cout, Cin) or not dw.is_contiguous():
        raise ValueError("dw must be a contiguous [K, Cout, Cin] tensor")
    nb = lib.sd3d_pair_wgrad_ws_bytes(K, Cin, Cout)
    ws = _WS.get(nb, dy.device)
    _lib.check(lib.sd3d_pair_wgrad(pdy, ldy, px, ldx, pairs.in_idx.data_ptr(), pair_out_rows(pairs).data_ptr(), pairs.tile_k.data_ptr(),
                                   pairs.p_cap, K, Cin, Cout, dw.data_ptr(), 1 if accumulate else 0, ws.data_ptr(), ws.numel(),
                                   ops._stream()), "pair_wgrad")
    return dw


def transposed_weights(w: torch.Tensor, mirrored: bool) -> torch.Tensor:
    """[K, Cout, Cin] -> [K, Cin, Cout] for the transposed rulebook (offsets mirrored for a submanifold table)."""
    wt = w.transpose(1, 2)
    return (wt.flip(0) if mirrored else wt).contiguous()


class SparseConv(torch.autograd.Function):
    """y = pair_conv(x, w, pairs).  `pairs_t` = lists of the transposed table (`pairs` itself for a submanifold
    convolution, the "up" lists for a "down" convolution and vice versa); `mirrored` = True for the submanifold case."""

    @staticmethod
    def forward(ctx, x, w, pairs, pairs_t, mirrored):
        ctx.save_for_backward(x, w)
        ctx.pairs, ctx.pairs_t, ctx.mirrored = pairs, pairs_t, mirrored
        return ops.pair_conv(x.detach(), w.detach(), pairs)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = ops.pair_conv(dy, transposed_weights(w, ctx.mirrored), ctx.pairs_t)
        if ctx.needs_input_grad[1]:
            dw = pair_wgrad(dy, x, ctx.pairs)
        return dx, dw, None, None, None


def sparse_conv(x, w, maps, kind: str, level: int, ksize: int = 3):
    """Differentiable sparse convolution on a scene's cached maps: kind "same" (submanifold, kernel `ksize`),
    "down" (stride 2, level -> level + 1) or "up" (transposed, level + 1 -> level)."""
    if kind == "same":
        t = maps.conv_table("same", level, ksize)
        return SparseConv.apply(x, w, t["pairs"], t["pairs"], True)
    other = "up" if kind == "down" else "down"
    return SparseConv.apply(x, w, maps.conv_table(kind, level)["pairs"], maps.conv_table(other, level)["pairs"], False)
