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


def pair_wgrad(dy: torch.Tensor, x: torch.Tensor, pairs: "ops.PairLists", dw: torch.Tensor = None, accumulate: bool = False,
               bf16_operands: bool = False, bias: bool = False):
    """dw[k] (+)= dy[out rows of offset k]^T @ x[in rows of offset k]; dy [M, Cout], x [V_in, Cin] -> dw [K, Cout, Cin].
    bf16_operands: both operands rounded to bf16 before the (fp32-accumulated) product.
    bias (K = 1, a Linear): returns (dw, db) with db [Cout] = column sums of dy, out of the same launch (SD3D_WGRAD_BIAS)."""
    lib = _lib.load()
    pdy, ldy = ops._rows(dy, "dy")
    px, ldx = ops._rows(x, "x")
    Cout, Cin, K = dy.shape[1], x.shape[1], pairs.K
    if dy.shape[0] != pairs.M:
        raise ValueError(f"dy has {dy.shape[0]} rows, the rulebook {pairs.M} outputs")
    if bias:
        if K != 1 or dw is not None:
            raise ValueError("pair_wgrad: bias=True is for K = 1 and a fresh output")
        flat = torch.empty(Cout * Cin + Cout, dtype=torch.float32, device=dy.device)
        dw, accumulate = flat, False
    elif dw is None:
        dw = torch.empty(K, Cout, Cin, dtype=torch.float32, device=dy.device)
        accumulate = False
    elif dw.shape != (K, Cout, Cin) or not dw.is_contiguous():
        raise ValueError("dw must be a contiguous [K, Cout, Cin] tensor")
    nb = lib.sd3d_pair_wgrad_ws_bytes(K, Cin, Cout)
    ws = _WS.get(nb, dy.device)
    flags = (1 if accumulate else 0) | (2 if bf16_operands else 0) | (4 if bias else 0)
    _lib.check(lib.sd3d_pair_wgrad(pdy, ldy, px, ldx, pairs.in_idx.data_ptr(), pair_out_rows(pairs).data_ptr(), pairs.tile_k.data_ptr(),
                                   pairs.p_cap, K, Cin, Cout, dw.data_ptr(), flags, ws.data_ptr(), ws.numel(), ops._stream()), "pair_wgrad")
    if bias:
        return dw[:Cout * Cin].view(1, Cout, Cin), dw[Cout * Cin:]
    return dw


def pair_wgrad_native(dy: torch.Tensor, x: torch.Tensor, pairs: "ops.PairLists") -> torch.Tensor:
    """The weight gradient in the PARAMETER's layout [K, Cin, Cout] (`ME.MinkowskiConvolution.kernel`): the same kernel with the two
    operands - and their index lists - exchanged (dW^T[k][ci][co] = sum_p x[in(p)][ci] dy[out(p)][co]); no transposed copy afterwards."""
    lib = _lib.load()
    pdy, ldy = ops._rows(dy, "dy")
    px, ldx = ops._rows(x, "x")
    Cout, Cin, K = dy.shape[1], x.shape[1], pairs.K
    if dy.shape[0] != pairs.M:
        raise ValueError(f"dy has {dy.shape[0]} rows, the rulebook {pairs.M} outputs")
    dw = torch.empty(K, Cin, Cout, dtype=torch.float32, device=dy.device)
    nb = lib.sd3d_pair_wgrad_ws_bytes(K, Cout, Cin)
    ws = _WS.get(nb, dy.device)
    _lib.check(lib.sd3d_pair_wgrad(px, ldx, pdy, ldy, pair_out_rows(pairs).data_ptr(), pairs.in_idx.data_ptr(), pairs.tile_k.data_ptr(),
                                   pairs.p_cap, K, Cout, Cin, dw.data_ptr(), 0, ws.data_ptr(), ws.numel(), ops._stream()), "pair_wgrad")
    return dw


class TrainWeight:
    """A convolution parameter in its native layout (`kernel` [K, Cin, Cout], or [Cin, Cout] for a 1x1) together with the [K, Cout, Cin]
    copy the forward kernels read, made for ALL convolutions of a network by one launch per step (`transpose_all`)."""
    __slots__ = ("param", "fwd", "pad_cin")

    def __init__(self, param, fwd, pad_cin=0):
        self.param, self.fwd, self.pad_cin = param, fwd, pad_cin


def transpose_all(named_params, pad_cin=None):
    """{name: kernel parameter} -> {name: TrainWeight}: every [K, Cin, Cout] -> [K, Cout, Cin (zero-padded to pad_cin[name])] in ONE launch
    (sd3d_transpose_batch, one job of K matrices per convolution).  Replaces a permute + copy per convolution in every forward (61 launches
    for Res16UNet34C) and the flipped / transposed copies of every backward."""
    import numpy as np
    from .train_dec import _TJOB_DT
    pad_cin = pad_cin or {}
    names = list(named_params)
    arr = np.zeros(len(names), dtype=_TJOB_DT)
    out, keep = {}, []
    for i, n in enumerate(names):
        p = named_params[n]
        k3 = p.detach() if p.dim() == 3 else p.detach().unsqueeze(0)
        if not k3.is_contiguous() or k3.dtype != torch.float32:
            k3 = k3.contiguous().float()
        K, cin, cout = k3.shape
        ld = max(int(pad_cin.get(n, 0)), cin)
        fwd = torch.empty(K, cout, ld, dtype=torch.float32, device=p.device)
        arr[i] = (k3.data_ptr(), fwd.data_ptr(), cin, cout, ld, K)
        keep.append(k3)
        out[n] = TrainWeight(p, fwd, ld)
    if names:
        _lib.check(_lib.load().sd3d_transpose_batch(len(names), arr.ctypes.data, ops._stream()), "transpose_batch")
    return out


class SparseConvNative(torch.autograd.Function):
    """y = pair_conv(x, W) with the parameter in its native layout: forward on the per-step [K, Cout, Cin] copy (`TrainWeight.fwd`), input
    gradient on the parameter AS IT LIES (mirrored offsets for a stride-1 table: `ops.pair_conv(mirror_w=True)`), weight gradient written in
    the parameter's layout (`pair_wgrad_native`).  Same kernels and the same bits as `SparseConv` fed with permuted copies."""

    @staticmethod
    def forward(ctx, x, kernel, fwd, pairs, pairs_t, mirrored, res=None):
        ctx.save_for_backward(x, kernel)
        ctx.pairs, ctx.pairs_t, ctx.mirrored = pairs, pairs_t, mirrored
        return ops.pair_conv(x.detach(), fwd, pairs, res=None if res is None else res.detach())

    @staticmethod
    def backward(ctx, dy):
        x, kernel = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dk = None
        k3 = kernel.detach() if kernel.dim() == 3 else kernel.detach().unsqueeze(0)       # [K, Cin, Cout] = the transposed set [K, Cout', Cin']
        if ctx.needs_input_grad[0]:
            if k3.shape[1] != x.shape[1]:                       # (the stem's input channels are zero-padded to a multiple of 32: it needs no input gradient)
                raise NotImplementedError("SparseConvNative: input gradient through a channel-padded convolution")
            dx = ops.pair_conv(dy, k3, ctx.pairs_t, mirror_w=ctx.mirrored)
        if ctx.needs_input_grad[1]:
            dk = pair_wgrad_native(dy, x, ctx.pairs)            # [K, Cin (padded), Cout]
            if dk.shape[1] != k3.shape[1]:
                dk = dk[:, :k3.shape[1]]
            dk = dk.reshape(kernel.shape) if dk.is_contiguous() else dk.contiguous().reshape(kernel.shape)
        return dx, dk, None, None, None, None, (dy if len(ctx.needs_input_grad) > 6 and ctx.needs_input_grad[6] else None)


def transposed_weights(w: torch.Tensor, mirrored: bool) -> torch.Tensor:
    """[K, Cout, Cin] -> [K, Cin, Cout] for the transposed rulebook (offsets mirrored for a submanifold table)."""
    wt = w.transpose(1, 2)
    return (wt.flip(0) if mirrored else wt).contiguous()


class SparseConv(torch.autograd.Function):
    """y = pair_conv(x, w, pairs).  `pairs_t` = lists of the transposed table (`pairs` itself for a submanifold
    convolution, the "up" lists for a "down" convolution and vice versa); `mirrored` = True for the submanifold case."""

    @staticmethod
    def forward(ctx, x, w, pairs, pairs_t, mirrored, res=None):
        ctx.save_for_backward(x, w)
        ctx.pairs, ctx.pairs_t, ctx.mirrored = pairs, pairs_t, mirrored
        return ops.pair_conv(x.detach(), w.detach(), pairs, res=None if res is None else res.detach())   # residual added in pass 2

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = ops.pair_conv(dy, transposed_weights(w, ctx.mirrored), ctx.pairs_t)
        if ctx.needs_input_grad[1]:
            dw = pair_wgrad(dy, x, ctx.pairs)
        n_in = len(ctx.needs_input_grad)
        return (dx, dw, None, None, None, (dy if ctx.needs_input_grad[5] else None))[:n_in] if n_in > 5 else (dx, dw, None, None, None)


def sparse_conv(x, w, maps, kind: str, level: int, ksize: int = 3, res=None):
    """Differentiable sparse convolution on a scene's cached maps: kind "same" (submanifold, kernel `ksize`),
    "down" (stride 2, level -> level + 1) or "up" (transposed, level + 1 -> level); `res` is added to the output."""
    if kind == "same":
        t = maps.conv_table("same", level, ksize)
        return SparseConv.apply(x, w, t["pairs"], t["pairs"], True, res)
    other = "up" if kind == "down" else "down"
    return SparseConv.apply(x, w, maps.conv_table(kind, level)["pairs"], maps.conv_table(other, level)["pairs"], False, res)


# ------------------------------------------------------------------------------------------------------------------
# Training-mode BatchNorm (+ residual + ReLU) and the superpoint pooling, as autograd nodes over csrc/train.hip
# ------------------------------------------------------------------------------------------------------------------
_WS_BN = ops._PerThread()


class _BatchNormAct(torch.autograd.Function):
    """y = act(BN_batch(x) + res) for x [M, C] (`minkunet.py:234-250, 302-304`); returns (y, mean, biased var)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, act, eps, running=None):
        """running = (running_mean, running_var, num_batches_tracked, momentum) or None: the module's buffers, advanced in place by
        the launch that makes the batch statistics."""
        lib = _lib.load()
        px, ldx = ops._rows(x, "x")
        M, C = x.shape
        dev = x.device
        mean, var, rstd = (torch.empty(C, dtype=torch.float32, device=dev) for _ in range(3))
        nb = lib.sd3d_bn_ws_bytes(M, C)
        ws = _WS_BN.get(nb, dev)
        rm, rv, nt, mom = running if running is not None else (None, None, None, 0.0)
        _lib.check(lib.sd3d_bn_stats_running(px, ldx, M, C, float(eps), mean.data_ptr(), var.data_ptr(), rstd.data_ptr(),
                                             ops._ptr(rm, torch.float32, "running_mean"), ops._ptr(rv, torch.float32, "running_var"),
                                             ops._ptr(nt, torch.int64, "num_batches_tracked"), float(mom), ws.data_ptr(), ws.numel(),
                                             ops._stream()), "bn_stats")
        ctx.set_materialize_grads(False)                        # no zero tensors for the two statistics outputs in backward
        y = torch.empty(M, C, dtype=torch.float32, device=dev)
        pr, ldr = ops._rows(res, "res") if res is not None else (None, 0)
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        _lib.check(lib.sd3d_bn_apply(px, ldx, mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), b.data_ptr(), pr, ldr, M, C, ops.ACT[act],
                                     y.data_ptr(), C, ops._stream()), "bn_apply")
        ctx.save_for_backward(x, y, mean, rstd, g)
        ctx.act, ctx.has_res = act, res is not None
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        lib = _lib.load()
        x, y, mean, rstd, g = ctx.saved_tensors
        M, C = x.shape
        dev = x.device
        dy = dy.contiguous()
        dx = torch.empty(M, C, dtype=torch.float32, device=dev)
        dres = torch.empty(M, C, dtype=torch.float32, device=dev) if ctx.has_res else None
        dgamma, dbeta = torch.empty(C, dtype=torch.float32, device=dev), torch.empty(C, dtype=torch.float32, device=dev)
        px, ldx = ops._rows(x, "x")
        nb = lib.sd3d_bn_ws_bytes(M, C)
        ws = _WS_BN.get(nb, dev)
        _lib.check(lib.sd3d_bn_backward(dy.data_ptr(), C, y.data_ptr(), C, px, ldx, mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), M, C,
                                        ops.ACT[ctx.act], dx.data_ptr(), C, None if dres is None else dres.data_ptr(), C, dgamma.data_ptr(),
                                        dbeta.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()), "bn_backward")
        return dx, dgamma, dbeta, dres, None, None, None


def batch_norm_act(x, bn: "torch.nn.BatchNorm1d", res=None, act=None):
    """Training-mode BatchNorm1d over the rows of x with the residual add and activation folded in; updates the
    module's running statistics like nn.BatchNorm1d (momentum, unbiased variance)."""
    tracked = bn.track_running_stats and bn.running_mean is not None
    if tracked and bn.momentum is not None and bn.running_mean.is_contiguous() and bn.running_var.is_contiguous():
        running = (bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum)
        y = _BatchNormAct.apply(x, bn.weight, bn.bias, res, act, bn.eps, running)[0]
        # the kernel advanced the buffers behind torch's back: bump their version counters like the mul_ / add_ it replaces did, so
        # that DerivedWeights._sources_stamp (folded-BN copies of the eval path) sees the write whatever called this
        torch.autograd.graph.increment_version((bn.running_mean, bn.running_var, bn.num_batches_tracked))
        return y
    y, mean, var = _BatchNormAct.apply(x, bn.weight, bn.bias, res, act, bn.eps)
    if tracked:                                                  # momentum=None: cumulative average, needs the step count on the host
        with torch.no_grad():
            m = x.shape[0]
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
            bn.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
            bn.running_var.mul_(1 - mom).add_(var * (m / max(m - 1, 1)), alpha=mom)
            bn.num_batches_tracked += 1
    return y


class _PoolSuperpoints(torch.autograd.Function):
    """SceneMaps.pool with a backward: features [V, C] -> superpoint means [S, C] (positions carry no gradient)."""

    @staticmethod
    def forward(ctx, feat, maps, C):
        f, pos = maps.pool(feat.detach(), C)
        ctx.maps, ctx.C, ctx.V = maps, C, feat.shape[0]
        ctx.mark_non_differentiable(pos)
        ctx.set_materialize_grads(False)
        return f, pos

    @staticmethod
    def backward(ctx, df, _dpos):
        lib = _lib.load()
        maps, C = ctx.maps, ctx.C
        df = df.contiguous()
        out = torch.empty(ctx.V, C, dtype=torch.float32, device=df.device)
        _lib.check(lib.sd3d_pool_superpoints_backward(df.data_ptr(), C, maps.superpoints.data_ptr(), maps.sidx.data_ptr(), maps.seg_start.data_ptr(),
                                                      maps._sp_start.data_ptr(), ctx.V, out.data_ptr(), C, ops._stream()), "pool_superpoints_backward")
        return out, None, None


def pool_superpoints(feat, maps, C):
    return _PoolSuperpoints.apply(feat, maps, C)


class TrainBackend:
    """The plan-backend interface of the U-Net definitions (`plan.EagerBackend`) for training: every call is an
    autograd node over HIP kernels - sparse convolution (pair-major forward, transposed-rulebook input gradient,
    MFMA weight gradient), then batch-statistics BatchNorm with the residual and ReLU folded in.  `affine` is the
    layer's nn.BatchNorm1d (or None), `wt` the [K, Cout, Cin] view of its live parameter."""

    IGNORE_ACT = False      # test hook: drop the ReLUs (a smooth network has no mask flips between roundings, so its
                            # gradients can be compared with the float64 oracle at rounding-level tolerance)

    def __init__(self, maps):
        self.maps = maps
        self._identity = {}

    def _identity_pairs(self, n_rows, device):
        if n_rows not in self._identity:
            nbr = torch.arange(n_rows, dtype=torch.int32, device=device).unsqueeze(0).contiguous()
            self._identity[n_rows] = ops.pair_lists(nbr, n_rows)
        return self._identity[n_rows]

    @staticmethod
    def _cat(x, x2):
        return x if x2 is None else torch.cat([x, x2], dim=1)

    def _conv(self, xin, wt, kind, level, ksize, res=None):
        if isinstance(wt, TrainWeight):
            if kind == "same":
                t = self.maps.conv_table("same", level, ksize)["pairs"]
                return SparseConvNative.apply(xin, wt.param, wt.fwd, t, t, True, res)
            other = "up" if kind == "down" else "down"
            return SparseConvNative.apply(xin, wt.param, wt.fwd, self.maps.conv_table(kind, level)["pairs"],
                                          self.maps.conv_table(other, level)["pairs"], False, res)
        return sparse_conv(xin, wt, self.maps, kind, level, ksize, res=res)

    def conv(self, x, wt, affine, key, x2=None, res=None, act=None):
        kind, level = key[0], key[1]
        act = None if self.IGNORE_ACT else act
        ksize = key[2] if kind == "same" else 2
        if affine is None:
            if act is not None:
                raise NotImplementedError("TrainBackend: activation without BatchNorm")
            return self._conv(self._cat(x, x2), wt, kind, level, ksize, res=res)
        y = self._conv(self._cat(x, x2), wt, kind, level, ksize)
        return batch_norm_act(y, affine, res=res, act=act)

    def dense(self, x, wt, affine, x2=None, res=None, act=None):
        xin = self._cat(x, x2)
        pairs = self._identity_pairs(xin.shape[0], xin.device)
        act = None if self.IGNORE_ACT else act
        if isinstance(wt, TrainWeight):
            apply = lambda r: SparseConvNative.apply(xin, wt.param, wt.fwd, pairs, pairs, True, r)      # noqa: E731
        else:
            if wt.dim() == 2:
                wt = wt.unsqueeze(0)
            apply = lambda r: SparseConv.apply(xin, wt, pairs, pairs, True, r)                           # noqa: E731
        if affine is None:
            if act is not None:
                raise NotImplementedError("TrainBackend: activation without BatchNorm")
            return apply(res)
        y = apply(None)                                                         # K = 1: its own mirror
        return batch_norm_act(y, affine, res=res, act=act)

    def affine(self, x, affine, x2=None, act=None, add=None):
        """Pre-activation BatchNorm (+ ReLU) of the spconv residual blocks (`spconvunet.py:48-64`); `add` = the identity branch
        of a normalize_before=False block, summed in after the activation (`spconvunet.py:66-81, 95-97`)."""
        act = None if self.IGNORE_ACT else act
        y = batch_norm_act(self._cat(x, x2), affine, act=act)
        return y if add is None else y + add
