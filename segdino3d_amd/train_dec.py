"""Autograd nodes of the query decoder over HIP kernels (in progress).

The reference differentiates the decoder (`instance_seg_3d_decoder.py:640-797`) with torch autograd over nn.Linear,
nn.LayerNorm, bmm / softmax attention and elementwise ops.  Here every node is a `torch.autograd.Function` whose forward is
the eval path's kernel and whose backward runs on the device too:

* `linear`     y = act(x W^T + b (+ res)), optionally on the concatenation [x | x2]:  g = dy act'(.) (`sd3d_act_backward`),
               dx = g W (`sd3d_gather_gemm` on W^T), dW = g^T x (`sd3d_pair_wgrad` on identity pair lists: the fp32 MFMA
               kernel of the sparse convolution's weight gradient), db = column sums of g (`sd3d_col_sums`);
* `layernorm`  y = act(LN(x + res) w + b): `sd3d_layernorm_backward`;
* `sine_pe_modulated`: gradient w.r.t. the box modulation only (positions and sizes are detached in the reference, `:740, :753`).

torch pads / transposes / concatenates small weight and activation tensors (layout plumbing); no arithmetic of the
gradients is done by torch.  No CPU fallback.
"""
from __future__ import annotations

import threading

import torch

from . import _lib, ops, train_ops

_WS = ops._PerThread()
_IDENTITY = {}
# mixed-precision training (decoder compute_dtype "bf16"): True (SD3D_BF16_BACKWARD=1) = bf16 operands in the Linear backward
# products as well - what autograd does under autocast; False (default) = fp32 backward products.  Measured at the training
# shapes (tools/linear_dtype_quick.py, 2441 query rows): these products are launch-bound, bf16 operands buy no time (dx 16 vs
# 14 us; the weight gradient runs on the fp32 kernel with operands rounded while they are staged), so the equally fast and more
# precise fp32 backward is the default and the autocast-faithful one is the opt-in.
import os as _os
BF16_BACKWARD = _os.environ.get("SD3D_BF16_BACKWARD", "0") == "1"
BATCH_WT = _os.environ.get("SD3D_BATCH_WT", "1") != "0"
# rows up to which a Linear's weight gradient takes the one-launch kernel (sd3d_linear_wgrad); 0: always the pair-list kernel + reduce (rounds 2 - 5)
LINEAR_WGRAD_ROWS = int(_os.environ.get("SD3D_LINEAR_WGRAD_ROWS", "8192"))


def _identity_pairs(n_rows: int, device):
    """Pair lists of the identity table [1, n_rows] (a Linear is a 1-offset convolution); the query count changes from step
    to step (random subset), so the cache is bounded."""
    key = (n_rows, str(device))
    hit = _IDENTITY.pop(key, None)
    if hit is None:
        nbr = torch.arange(n_rows, dtype=torch.int32, device=device).unsqueeze(0).contiguous()
        hit = ops.pair_lists(nbr, n_rows)
        while len(_IDENTITY) >= 32:
            _IDENTITY.pop(next(iter(_IDENTITY)))
    _IDENTITY[key] = hit                                       # most recently used last
    return hit


def _round(n, m):
    return (n + m - 1) // m * m


class _TransposedWeights:
    """W^T of every Linear weight of the running training step, made by ONE launch.  The input gradient of a Linear is a
    product with W^T ([Cin, Cout padded to 32]); transposing each weight inside its own backward cost one copy kernel and a
    handful of host-side tensor ops per Linear (145 per step).  Forward passes register their weights; the first backward
    that needs a transposed weight transposes all registered ones (sd3d_transpose_batch) and the following ones look theirs
    up.  The next registration after a backward starts a new step.  Keyed by (storage pointer, shape, version): packed weights
    (torch.cat views rebuilt every step) are fresh tensors each time and simply register again.
    ONE instance per process, behind a lock: forward passes register from the calling thread, the backward of device tensors runs
    on the autograd engine's device thread (a thread-local registry never saw a registered weight there, transposed them one by
    one, and kept stale entries whose addresses later parameters reused)."""

    def __init__(self):
        self.pending, self.done, self.closed = [], {}, False
        self.lock = threading.Lock()

    @staticmethod
    def _key(w):
        return (w.data_ptr(), tuple(w.shape), w._version)

    def register(self, w):
        with self.lock:
            if self.closed:
                self.pending, self.done, self.closed = [], {}, False
            if len(self.pending) >= 4096:                        # forwards without a backward: do not hold their weights for ever
                del self.pending[:2048]
            self.pending.append(w)

    def get(self, w):
        key = self._key(w)
        with self.lock:
            hit = self.done.get(key)
            if hit is None:
                todo = [t for t in self.pending if self._key(t) not in self.done]
                if not any(self._key(t) == key for t in todo):
                    todo.append(w)
                self._run(todo)
                self.pending, self.closed = [], True
                hit = self.done[key]
            return hit

    def _run(self, ws):
        import ctypes as C
        import numpy as np
        lib = _lib.load()
        seen, jobs = set(), []
        for w in ws:
            k = self._key(w)
            if k in seen or w.dim() != 2 or not w.is_cuda or w.dtype != torch.float32:
                continue
            seen.add(k)
            src = w.detach()
            if not src.is_contiguous():
                src = src.contiguous()
            cout, cin = src.shape
            dst = torch.empty(cin, _round(cout, 32), dtype=torch.float32, device=w.device)
            self.done[k] = dst
            jobs.append((src, dst, cout, cin))
        if not jobs:
            return
        arr = np.zeros(len(jobs), dtype=_TJOB_DT)
        for i, (src, dst, cout, cin) in enumerate(jobs):
            arr[i] = (src.data_ptr(), dst.data_ptr(), cout, cin, dst.shape[1], 0)
        _lib.check(lib.sd3d_transpose_batch(len(jobs), arr.ctypes.data, ops._stream()), "transpose_batch")
        self._keep = jobs                                        # sources stay alive until the launch has been issued


import numpy as _np
_TJOB_DT = _np.dtype([("src", "<u8"), ("dst", "<u8"), ("rows", "<i4"), ("cols", "<i4"), ("ld_dst", "<i4"), ("pad_", "<i4")], align=True)
assert _TJOB_DT.itemsize == 32
_WT = _TransposedWeights()


def act_backward(dy, ref, act, c_pad):
    lib = _lib.load()
    M, C = dy.shape
    if act is None and c_pad == C and dy.is_contiguous():        # nothing to apply, nothing to pad: g IS dy
        return dy
    g = torch.empty(M, c_pad, dtype=torch.float32, device=dy.device)
    pd, ldd = ops._rows(dy, "dy")
    pr, ldr = (None, 0) if ref is None else ops._rows(ref, "ref")
    _lib.check(lib.sd3d_act_backward(pd, ldd, pr, ldr, ops.ACT[act], M, C, c_pad, g.data_ptr(), c_pad, ops._stream()), "act_backward")
    return g


def col_sums(x):
    lib = _lib.load()
    M, C = x.shape
    px, ld = ops._rows(x, "x")
    out = torch.empty(C, dtype=torch.float32, device=x.device)
    nb = lib.sd3d_col_sums_ws_bytes(M, C)
    ws = _WS.get(nb, x.device)
    _lib.check(lib.sd3d_col_sums(px, ld, M, C, out.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()), "col_sums")
    return out


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, act, res, x2, force_exact=False):
        xd, wd = x.detach(), w.detach()
        bd = None if b is None else b.detach()
        kw = dict(x2=None if x2 is None else x2.detach(), shift=bd, res=None if res is None else res.detach())
        # mixed precision (decoder compute_dtype "bf16"): the forward product may run with bf16 operands (the backward
        # products follow it, see ctx.bf16_bwd); w is a live parameter here, so no rounded copy is cached across steps
        exact = force_exact or not ops.bf16_decoder_active()
        ws = None if exact or xd.shape[0] < ops.BF16_MIN_ROWS or wd.shape[1] % 32 else ops.split_weights(wd.unsqueeze(0), 1)
        y = ops.gather_gemm(xd, wd, act=act, exact=True, wt_split=ws, **kw)
        ref = y
        if act == "gelu":                                      # its derivative needs the pre-activation
            ref = ops.gather_gemm(xd, wd, act=None, exact=True, wt_split=ws, **kw)
        ctx.save_for_backward(x, w, ref if act is not None else None, x2)
        ctx.act, ctx.has_b, ctx.has_res = act, b is not None, res is not None
        if ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[5]):
            _WT.register(w)                                     # its transpose is made with all the others at the first backward
        # BASELINE configs[4] (autocast(bf16) around the decoder, train_engine_3d.py:88-100): the two backward products of a
        # Linear whose forward ran on bf16 operands run on bf16 operands too (fp32 accumulation), as autograd under autocast does
        ctx.bf16_bwd = ws is not None and BF16_BACKWARD
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, ref, x2 = ctx.saved_tensors
        cout, cin = w.shape
        c_pad = _round(cout, 32)
        g = act_backward(dy.contiguous(), ref, ctx.act, c_pad)                  # [M, c_pad], zero beyond cout
        dx = dx2 = dw = db = dres = None
        if ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[5]):
            if BATCH_WT:
                wt = _WT.get(w)                                 # [cin, c_pad], zero beyond cout
            elif c_pad == cout:
                wt = w.detach().t().contiguous()
            else:
                wt = torch.zeros(cin, c_pad, dtype=torch.float32, device=w.device)
                wt[:, :cout] = w.detach().t()
            if ctx.bf16_bwd and c_pad % 32 == 0:
                dxa = ops.gather_gemm(g, wt, wt_split=ops.split_weights(wt.unsqueeze(0), 1))     # bf16 operands, fp32 accumulation
            else:
                dxa = ops.gather_gemm(g, wt, exact=True)                        # [M, cin]
            if x2 is None:
                dx = dxa
            else:
                c0 = x.shape[1]
                dx, dx2 = dxa[:, :c0], dxa[:, c0:]
        if ctx.needs_input_grad[1]:
            xin = x.detach() if x2 is None else torch.cat([x.detach(), x2.detach()], dim=1)
            if xin.shape[1] % 4:
                xin = torch.nn.functional.pad(xin, (0, 4 - xin.shape[1] % 4))
            # bf16 mode: the same kernel with both operands rounded to bf16 as they are staged (products of bf16 values are exact in
            # fp32, so this IS the bf16-operand / fp32-accumulate product, at the fp32 kernel's speed)
            # the bias gradient (column sums of g) comes out of the same launch; in the bf16 mode the kernel stages ROUNDED
            # operands, so there the sums keep their own fp32 pass
            want_b = ctx.has_b and ctx.needs_input_grad[2]
            fuse_b = want_b and not ctx.bf16_bwd
            if 0 < g.shape[0] <= LINEAR_WGRAD_ROWS:
                # a few thousand rows (the decoder's Linears): ONE launch, a 32 x 32 block of dW per workgroup over all rows (round 6; the
                # pair-list kernel splits the rows into ranges and adds their partial blocks in a second launch: 21 + 13 us for 0.3 GFLOP)
                xin = xin.contiguous()
                dw = torch.empty(cout, cin, dtype=torch.float32, device=g.device)
                db = torch.empty(cout, dtype=torch.float32, device=g.device) if fuse_b else None
                _lib.check(_lib.load().sd3d_linear_wgrad(g.data_ptr(), g.stride(0), xin.data_ptr(), xin.stride(0), g.shape[0], cin, cout, dw.data_ptr(),
                                                         None if db is None else db.data_ptr(), 2 if ctx.bf16_bwd else 0, ops._stream()), "linear_wgrad")
            else:
                dwp = train_ops.pair_wgrad(g, xin.contiguous(), _identity_pairs(g.shape[0], g.device), bf16_operands=ctx.bf16_bwd, bias=fuse_b)   # [1, c_pad, cin(+pad)]
                if fuse_b:
                    dwp, dbp = dwp
                    db = dbp[:cout]
                dw = dwp[0, :cout, :cin]
        if db is None and ctx.has_b and ctx.needs_input_grad[2]:
            db = col_sums(g)[:cout]
        if ctx.has_res and ctx.needs_input_grad[4]:
            dres = g[:, :cout]
        return dx, dw, db, None, dres, dx2, None


def linear(x, weight, bias=None, act=None, res=None, x2=None, exact=False):
    """Differentiable `ops.linear` / two-source `ops.gather_gemm`: weight [Cout, Cin(total)]; `exact` keeps the forward product
    fp32 inside a bf16 scope (the mask head: its logits feed thresholds)."""
    return _Linear.apply(x, weight, bias, act, res, x2, exact)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, res, act, eps):
        y = ops.layernorm(x.detach(), w.detach(), b.detach(), res=None if res is None else res.detach(), act=act, eps=eps)
        ctx.save_for_backward(x, w, res, y if act is not None else None)
        ctx.act, ctx.eps = act, eps
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w, res, y = ctx.saved_tensors
        M, D = x.shape
        dev = x.device
        dy = dy.contiguous()
        dxin = torch.empty(M, D, dtype=torch.float32, device=dev)
        dw, db = torch.empty(D, dtype=torch.float32, device=dev), torch.empty(D, dtype=torch.float32, device=dev)
        nb = lib.sd3d_layernorm_backward_ws_bytes(M, D)
        ws = _WS.get(nb, dev)
        px, ldx = ops._rows(x.detach(), "x")
        pr, ldr = (None, 0) if res is None else ops._rows(res.detach(), "res")
        _lib.check(lib.sd3d_layernorm_backward(dy.data_ptr(), D, None if y is None else y.data_ptr(), D, px, ldx, pr, ldr,
                                               w.detach().contiguous().data_ptr(), float(ctx.eps), M, D, ops.ACT[ctx.act], dxin.data_ptr(), D,
                                               dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()), "layernorm_backward")
        return dxin, dw, db, (dxin if res is not None else None), None, None


def layernorm(x, weight, bias, res=None, act=None, eps=1e-5):
    return _LayerNorm.apply(x, weight, bias, res, act, eps)


class _SinePEModulated(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, rng, dim_t, axis, mod_num, mod_den):
        out = ops.sine_pe(xyz, rng, dim_t, axis, mod_num=mod_num.detach(), mod_den=mod_den)
        ctx.save_for_backward(xyz, rng, dim_t, axis, mod_den)
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        xyz, rng, dim_t, axis, mod_den = ctx.saved_tensors
        n = xyz.shape[0]
        d_out = d_out.contiguous()
        d_num = torch.empty(n, 3, dtype=torch.float32, device=xyz.device)
        px, ldx = ops._rows(xyz, "xyz")
        ld_den = mod_den.stride(0) if mod_den.dim() == 2 else 0
        _lib.check(lib.sd3d_sine_pe_mod_backward(d_out.data_ptr(), d_out.shape[1], px, ldx, n, rng.data_ptr(), dim_t.data_ptr(), axis.data_ptr(),
                                                 dim_t.numel(), mod_den.data_ptr(), ld_den, d_num.data_ptr(), ops._stream()), "sine_pe_mod_backward")
        return None, None, None, None, d_num, None


def sine_pe_modulated(xyz, rng, dim_t, axis, mod_num, mod_den):
    """`ops.sine_pe` with box modulation, differentiable w.r.t. `mod_num` ([n, 3])."""
    return _SinePEModulated.apply(xyz, rng, dim_t, axis, mod_num, mod_den)


_WS_ATT = ops._PerThread()
_WS_ATT2 = ops._PerThread()


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, num_heads, scale, mask_bits, q2, k2):
        lib = _lib.load()
        qd, kd, vd = q.detach(), k.detach(), v.detach()
        pq, ldq = ops._rows(qd, "q"); pk, ldk = ops._rows(kd, "k"); pv, ldv = ops._rows(vd, "v")
        pq2, ldq2, pk2, ldk2 = None, 0, None, 0
        if q2 is not None:
            pq2, ldq2 = ops._rows(q2.detach(), "q2"); pk2, ldk2 = ops._rows(k2.detach(), "k2")
        Lq, Lk = q.shape[0], k.shape[0]
        out = torch.empty(Lq, num_heads * 32, dtype=torch.float32, device=q.device)
        lse = torch.empty(num_heads, Lq, dtype=torch.float32, device=q.device)
        ws = _WS_ATT.get(lib.sd3d_attention_ws_bytes(Lq, num_heads), q.device)
        fn = lib.sd3d_attention_lse_bf16 if ops.bf16_decoder_active() else lib.sd3d_attention_lse
        _lib.check(fn(pq, ldq, pq2, ldq2, pk, ldk, pk2, ldk2, pv, ldv, ops._ptr(mask_bits, torch.int32, "mask_bits"),
                      Lq, Lk, num_heads, float(scale), out.data_ptr(), out.shape[1], lse.data_ptr(), ws.data_ptr(),
                      ws.numel(), ops._stream()), "attention_lse")
        ctx.save_for_backward(q, k, v, q2, k2, mask_bits, out, lse)
        ctx.H, ctx.scale = num_heads, float(scale)
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        q, k, v, q2, k2, bits, out, lse = ctx.saved_tensors
        d_out = d_out.contiguous()
        Lq, Lk, H = q.shape[0], k.shape[0], ctx.H
        dev = q.device
        new = lambda n: torch.empty(n, H * 32, dtype=torch.float32, device=dev)
        dq, dk, dv = new(Lq), new(Lk), new(Lk)
        dq2, dk2 = (new(Lq), new(Lk)) if q2 is not None else (None, None)
        pq, ldq = ops._rows(q.detach(), "q"); pk, ldk = ops._rows(k.detach(), "k"); pv, ldv = ops._rows(v.detach(), "v")
        pq2, ldq2, pk2, ldk2 = None, 0, None, 0
        if q2 is not None:
            pq2, ldq2 = ops._rows(q2.detach(), "q2"); pk2, ldk2 = ops._rows(k2.detach(), "k2")
        ws = _WS_ATT2.get(lib.sd3d_attention_backward_ws_bytes(Lq, H), dev)
        W = H * 32
        _lib.check(lib.sd3d_attention_backward(pq, ldq, pq2, ldq2, pk, ldk, pk2, ldk2, pv, ldv, ops._ptr(bits, torch.int32, "mask_bits"), Lq, Lk, H,
                                               ctx.scale, out.data_ptr(), W, lse.data_ptr(), d_out.data_ptr(), W, dq.data_ptr(), W,
                                               ops._ptr(dq2), W, dk.data_ptr(), W, ops._ptr(dk2), W, dv.data_ptr(), W, ws.data_ptr(), ws.numel(),
                                               ops._stream()), "attention_backward")
        return dq, dk, dv, None, None, None, dq2, dk2


def attention(q, k, v, num_heads, scale, mask_bits=None, q2=None, k2=None):
    """Differentiable `ops.attention` (fp32): gradients for q, k, v and the optional second source q2 / k2."""
    return _Attention.apply(q, k, v, num_heads, scale, mask_bits, q2, k2)


class _SplitCols(torch.autograd.Function):
    """Equal column blocks of a packed projection's output as views.  Plain slicing gives every block its own SliceBackward:
    a zero-filled full-width tensor per block and a chain of full-width adds (12 x 30 MB fills + 11 adds for the packed key /
    value projection); here the blocks' gradients are concatenated once."""

    @staticmethod
    def forward(ctx, t, width):
        ctx.width, ctx.shape = width, t.shape
        ctx.set_materialize_grads(False)
        return tuple(t[:, c:c + width] for c in range(0, t.shape[1], width))

    @staticmethod
    def backward(ctx, *grads):
        if all(g is not None for g in grads):
            return torch.cat(grads, dim=1), None
        out = torch.zeros(ctx.shape, dtype=torch.float32, device=next(g for g in grads if g is not None).device)
        for i, g in enumerate(grads):
            if g is not None:
                out[:, i * ctx.width:(i + 1) * ctx.width] = g
        return out, None


def split_cols(t, width):
    if t.shape[1] % width:
        raise ValueError("split_cols: the width must divide the column count")
    return list(_SplitCols.apply(t, width))


def attention_dropout(q, k, v, num_heads, scale, mask_bits=None, q2=None, k2=None, p=0.0):
    """Attention with dropout on the softmax probabilities - what `nn.MultiheadAttention(dropout=p)` does in training
    (`instance_seg_3d_decoder.py:48-49, 128-129`).  The fused kernel has no random stream, so for p > 0 - no shipped config,
    `configs/models/base_3d.py:30` is 0.0 - the probabilities are formed explicitly on the device ([H, Lq, Lk] fp32: 19 MB at
    200 x 3000, 288 MB at 3000 x 3000) and torch's dropout and autograd carry the rest.  Same arguments as `attention`;
    bit = 1 in mask_bits blocks a key."""
    Lq, Lk, H = q.shape[0], k.shape[0], num_heads
    heads = lambda t, L: t.reshape(L, H, -1).transpose(0, 1)  # noqa: E731   [H, L, width / H]
    s = heads(q, Lq) @ heads(k, Lk).transpose(1, 2)
    if q2 is not None:
        s = s + heads(q2, Lq) @ heads(k2, Lk).transpose(1, 2)
    s = s * scale
    if mask_bits is not None:
        cols = torch.arange(Lk, device=q.device)
        blocked = (mask_bits[:, cols >> 5] >> (cols & 31)) & 1             # [Lq, Lk]
        s = s.masked_fill(blocked.bool().unsqueeze(0), float("-inf"))
    prob = torch.nn.functional.dropout(torch.softmax(s, dim=-1), p, training=True)
    return (prob @ heads(v, Lk)).transpose(0, 1).reshape(Lq, -1)


class _BoxRefine(torch.autograd.Function):
    """`ops.box_refine`: (center, size, size_metric); gradients reach d_center and d_size through center / size_metric only
    (the refined points and sizes that feed the next layer are detached in the reference, `:740, :753`)."""

    @staticmethod
    def forward(ctx, ref_points, d_center, size_prev, d_size, rng, normalize):
        center, size, size_metric = ops.box_refine(ref_points, d_center.detach().contiguous(), size_prev,
                                                   None if d_size is None else d_size.detach().contiguous(), rng, normalize)
        ctx.save_for_backward(size, rng)
        ctx.normalize, ctx.has_size, ctx.Q = normalize, d_size is not None, ref_points.shape[0]
        ctx.set_materialize_grads(False)                        # backward takes None for the outputs nothing differentiates
        if size is None:
            return center, None, None
        ctx.mark_non_differentiable(size)
        return center, size, size_metric

    @staticmethod
    def backward(ctx, d_center, _d_size, d_metric):
        lib = _lib.load()
        size, rng = ctx.saved_tensors
        dev = rng.device
        d_dc = torch.empty(ctx.Q, 3, dtype=torch.float32, device=dev)
        d_ds = torch.empty(ctx.Q, 3, dtype=torch.float32, device=dev) if ctx.has_size else None
        c = None if d_center is None else d_center.contiguous()
        m = None if d_metric is None else d_metric.contiguous()
        _lib.check(lib.sd3d_box_refine_backward(ops._ptr(c), ops._ptr(m), ops._ptr(size), rng.data_ptr(), int(ctx.normalize), ctx.Q,
                                                d_dc.data_ptr(), ops._ptr(d_ds), ops._stream()), "box_refine_backward")
        return None, d_dc, None, d_ds, None, None


def box_refine(ref_points, d_center, size_prev, d_size, rng, normalize):
    return _BoxRefine.apply(ref_points, d_center, size_prev, d_size, rng, normalize)
