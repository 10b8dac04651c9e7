"""Training step of a sparse U-Net as ONE autograd node over two C calls (`csrc/train_plan.hip`).

`Res16UNetBase.forward` in training mode (`minkunet.py:531-601`; the reference's `loss.backward()` of `train_engine_3d.py:88-122` differentiates
it through MinkowskiEngine's autograd) is a straight list of {sparse convolution -> batch-statistics BatchNorm (+ residual) -> ReLU} layers.
`train_ops.TrainBackend` runs every one of them as two torch.autograd nodes (convolution, BatchNorm) - ~250 nodes, each with its own ctypes
calls, tensor allocations and, in backward, `add_` kernels wherever a tensor has two consumers and `cat` / `narrow` copies around the skip
connections.  Here the network definition is RECORDED once (`TrainRecorder`, the same backend interface `plan.Recorder` implements for
evaluation) and a step is `sd3d_unet_train_forward` + `sd3d_unet_train_backward`:
  * a skip concatenation is two producers writing their column slices of one buffer - no `cat`, and the consumer's input gradient reaches
    both producers with one launch;
  * gradients of a tensor with several consumers: the first writer (in reverse layer order) stores, the others add in the epilogue of their
    own kernel - which is which is decided here, at record time (`dx_accum`);
  * parameters stay in their native [K, Cin, Cout] layout (`train_ops.transpose_all`: one launch per step for the forward copies).
The autograd-node path stays (`SD3D_TRAIN_PLAN=0`, SpConvUNet, BatchNorm without momentum): both run the same kernels in the same order per
tensor and are compared in tests/test_gpu_train_ops.py.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import _lib, ops, train_ops
from .plan import BUF_DT, table_level

USE_TRAIN_PLAN = os.environ.get("SD3D_TRAIN_PLAN", "1") != "0"

TTABLE_DT = np.dtype([("in_idx", "<u8"), ("tile_k", "<u8"), ("pos", "<u8"), ("rlist", "<u8"), ("out_rows", "<u8"), ("p_cap", "<i8"), ("M", "<i8"),
                      ("K", "<i4"), ("rl_stride", "<i4"), ("center", "<i4"), ("direct", "<i4")], align=True)
TLAYER_DT = np.dtype([("table", "<i4"), ("table_t", "<i4"), ("mirrored", "<i4"), ("src", "<i4"), ("src_col", "<i4"), ("res", "<i4"),
                      ("res_col", "<i4"), ("dst", "<i4"), ("dst_col", "<i4"), ("raw", "<i4"), ("K", "<i4"), ("Cin", "<i4"), ("Cout", "<i4"),
                      ("act", "<i4"), ("need_dx", "<i4"), ("dx_accum", "<i4"), ("stats", "<i4"), ("pad_", "<i4"), ("eps", "<f4"),
                      ("momentum", "<f4"), ("wt_fwd", "<u8"), ("kernel", "<u8"), ("dkernel", "<u8"), ("gamma", "<u8"), ("beta", "<u8"),
                      ("dgamma", "<u8"), ("dbeta", "<u8"), ("running_mean", "<u8"), ("running_var", "<u8"), ("num_batches", "<u8")], align=True)
assert TTABLE_DT.itemsize == 72 and TLAYER_DT.itemsize == 160


class _Storage:
    """One buffer of the activation arena: `ch` columns on the rows of `level`.  A concatenation merges two storages into a new one
    (`parent`, `off`): everything recorded against the old ones resolves into column slices of the new one."""
    __slots__ = ("level", "ch", "parent", "off", "id")

    def __init__(self, level, ch):
        self.level, self.ch, self.parent, self.off, self.id = level, ch, None, 0, -1

    def root(self) -> Tuple["_Storage", int]:
        s, off = self, 0
        while s.parent is not None:
            off += s.off
            s = s.parent
        return s, off


class _Sym:
    __slots__ = ("st", "col", "ch", "level")

    def __init__(self, st, col, ch, level):
        self.st, self.col, self.ch, self.level = st, col, ch, level

    def loc(self):
        r, off = self.st.root()
        return r, off + self.col


class TrainRecorder:
    """The plan-backend interface (`conv`, `dense`; see plan.EagerBackend) that records a training plan.  `wt` = train_ops.TrainWeight
    placeholder or anything with `.param` (only identity and shapes matter at record time), `affine` = the layer's nn.BatchNorm1d."""

    def __init__(self, in_channels: int, in_level: int = 0):
        self.input = _Sym(_Storage(in_level, in_channels), 0, in_channels, in_level)
        self.layers: List[dict] = []

    def _cat(self, x: _Sym, x2: _Sym) -> _Sym:
        if x2 is None:
            return x
        (ra, ca), (rb, cb) = x.loc(), x2.loc()
        if ra is rb:
            if cb != ca + x.ch:
                raise NotImplementedError("train plan: the two halves of a concatenation lie in one buffer but not side by side")
            return _Sym(ra, ca, x.ch + x2.ch, x.level)
        if ca != 0 or cb != 0 or ra.ch != x.ch or rb.ch != x2.ch or x.level != x2.level:
            raise NotImplementedError("train plan: concatenation of tensors that are slices of other concatenations")
        new = _Storage(x.level, x.ch + x2.ch)
        ra.parent, ra.off = new, 0
        rb.parent, rb.off = new, x.ch
        return _Sym(new, 0, new.ch, x.level)

    def _layer(self, key, x, wt, affine, x2, res, act):
        if affine is None:
            raise NotImplementedError("train plan: a convolution without BatchNorm")
        if act not in (None, "relu"):
            raise NotImplementedError("train plan: activations other than ReLU")
        src = self._cat(x, x2)
        param = wt.param
        cout = param.shape[-1]
        level = table_level(key) if key[0] != "id" else x.level
        out = _Sym(_Storage(level, cout), 0, cout, level)
        self.layers.append(dict(key=key, src=src, res=res, out=out, param=param, bn=affine, act=ops.ACT[act], K=1 if param.dim() == 2 else param.shape[0]))
        return out

    def conv(self, x, wt, affine, key, x2=None, res=None, act=None):
        return self._layer(key, x, wt, affine, x2, res, act)

    def dense(self, x, wt, affine, x2=None, res=None, act=None):
        return self._layer(("id", x.level), x, wt, affine, x2, res, act)

    def affine(self, *a, **k):
        raise NotImplementedError("train plan: pre-activation BatchNorm layers (SpConvUNet) run on train_ops.TrainBackend")

    def finish(self, output: _Sym) -> "TrainPlan":
        return TrainPlan(self, output)


def _transposed_key(key):
    """Table of the transposed rulebook and whether it is the same table with mirrored offsets."""
    if key[0] in ("same", "id"):
        return key, True
    return (("up" if key[0] == "down" else "down"), key[1]), False


class TrainPlan:
    def __init__(self, rec: TrainRecorder, output: _Sym):
        # storages -> buffer ids (roots only)
        roots: List[_Storage] = []

        def bid(sym):
            r, col = sym.loc()
            if r.id < 0:
                r.id = len(roots)
                roots.append(r)
            return r.id, col
        in_id, _ = bid(rec.input)
        assert in_id == 0
        self.tables: List[Tuple] = []

        def tid(key):
            if key not in self.tables:
                self.tables.append(key)
            return self.tables.index(key)
        self.n = len(rec.layers)
        L = np.zeros(self.n, dtype=TLAYER_DT)
        self.params, self.bns = [], []
        stats = 0
        for i, r in enumerate(rec.layers):
            kt, mirrored = _transposed_key(r["key"])
            (src, src_col), (dst, dst_col) = bid(r["src"]), bid(r["out"])
            res, res_col = bid(r["res"]) if r["res"] is not None else (-1, 0)
            need_dx = 0 if src == 0 else 1
            L[i]["table"], L[i]["table_t"], L[i]["mirrored"] = tid(r["key"]), (tid(kt) if need_dx else -1), int(mirrored)
            L[i]["src"], L[i]["src_col"], L[i]["res"], L[i]["res_col"], L[i]["dst"], L[i]["dst_col"] = src, src_col, res, res_col, dst, dst_col
            L[i]["raw"], L[i]["K"], L[i]["Cin"], L[i]["Cout"], L[i]["act"] = i, r["K"], r["src"].ch, r["out"].ch, r["act"]
            L[i]["need_dx"], L[i]["stats"] = need_dx, stats
            L[i]["eps"], L[i]["momentum"] = r["bn"].eps, r["bn"].momentum
            stats += 3 * r["out"].ch
            self.params.append(r["param"])
            self.bns.append(r["bn"])
        self.stats_floats = stats
        self.buf_level = np.array([s.level for s in roots], dtype=np.int64)
        self.buf_ch = np.array([s.ch for s in roots], dtype=np.int64)
        self.out_id, self.out_col = bid(output)
        self.out_ch = output.ch
        if self.out_col != 0 or self.buf_ch[self.out_id] != output.ch:
            raise NotImplementedError("train plan: the network output must own its buffer")
        # which gradient writes are the first into their slice (reverse layer order): BatchNorm residual gradients must be, input
        # gradients of convolutions may be either (store / add through the residual input of pass 2)
        written: Dict[int, List[Tuple[int, int]]] = {self.out_id: [(0, int(self.out_ch))]}

        def state(b, c0, c1):
            ov = [(a, e) for a, e in written.get(b, []) if a < c1 and e > c0]
            if not ov:
                return 0
            covered = sum(min(e, c1) - max(a, c0) for a, e in ov)
            return 2 if covered >= c1 - c0 else 1
        for i in range(self.n - 1, -1, -1):
            r = rec.layers[i]
            if L[i]["res"] >= 0:
                b, c = int(L[i]["res"]), int(L[i]["res_col"])
                if state(b, c, c + int(L[i]["Cout"])) != 0:
                    raise NotImplementedError("train plan: a residual whose gradient is not the first its tensor receives")
                written.setdefault(b, []).append((c, c + int(L[i]["Cout"])))
            if L[i]["need_dx"]:
                b, c, w = int(L[i]["src"]), int(L[i]["src_col"]), int(L[i]["Cin"])
                st = state(b, c, c + w)
                if st == 1:
                    raise NotImplementedError("train plan: an input gradient over a partly written slice")
                L[i]["dx_accum"] = 1 if st == 2 else 0
                if st == 0:
                    written.setdefault(b, []).append((c, c + w))
        self.layers = L
        self.rec_keys = [r["key"] for r in rec.layers]

    # ---- per step ---------------------------------------------------------------------------------------------------------------
    def _tables(self, maps, identity):
        tabs = np.zeros(len(self.tables), dtype=TTABLE_DT)
        keep = []
        for i, key in enumerate(self.tables):
            pl = identity(key[1]) if key[0] == "id" else maps.conv_table(*key)["pairs"]
            if pl.pos is None:
                raise RuntimeError("train plan: the training step needs full pair lists (position tables)")
            rows = train_ops.pair_out_rows(pl)
            tabs[i] = (pl.in_idx.data_ptr(), pl.tile_k.data_ptr(), pl.pos.data_ptr(), 0 if pl.rlist is None else pl.rlist.data_ptr(),
                       rows.data_ptr(), pl.p_cap, pl.M, pl.K, pl.rl_stride, pl.center, 1 if pl.direct else 0)
            keep.append(pl)
        return tabs, keep


class _UNetTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, plan: TrainPlan, maps, fwd_weights, *params):
        lib, libq = _lib.load_nogil(), _lib.load()
        dev = x.device
        n = plan.n
        kernels, gammas, betas = params[:n], params[n:2 * n], params[2 * n:3 * n]
        n_vox = np.asarray(maps.n_vox, dtype=np.int64)
        ident = {}

        def identity(level):
            if level not in ident:
                rows = int(n_vox[level])
                nbr = torch.arange(rows, dtype=torch.int32, device=dev).unsqueeze(0).contiguous()
                ident[level] = ops.pair_lists(nbr, rows)
            return ident[level]
        tabs, keep = plan._tables(maps, identity)
        rows = n_vox[plan.buf_level]
        sizes = rows * plan.buf_ch
        sizes[0] = 0
        offs = np.concatenate(([0], np.cumsum(sizes)[:-1]))
        arena = torch.empty(int(sizes.sum()) + 4, dtype=torch.float32, device=dev)
        bufs = np.zeros(len(rows), dtype=BUF_DT)
        bufs["ptr"], bufs["rows"], bufs["ld"] = arena.data_ptr() + offs * 4, rows, plan.buf_ch
        if x.shape[0] != rows[0] or x.shape[1] != plan.buf_ch[0] or x.stride(1) != 1 or x.dtype != torch.float32:
            raise ValueError("train plan: input does not match the recorded network input")
        bufs["ptr"][0], bufs["ld"][0] = x.data_ptr(), x.stride(0)
        L = plan.layers.copy()
        out_rows = np.array([tabs["M"][t] for t in L["table"]], dtype=np.int64)
        raw_sizes = out_rows * L["Cout"]
        raw_offs = np.concatenate(([0], np.cumsum(raw_sizes)[:-1]))
        raw_arena = torch.empty(int(raw_sizes.sum()) + 4, dtype=torch.float32, device=dev)
        raws = np.zeros(n, dtype=BUF_DT)
        raws["ptr"], raws["rows"], raws["ld"] = raw_arena.data_ptr() + raw_offs * 4, out_rows, L["Cout"]
        stats = torch.empty(plan.stats_floats, dtype=torch.float32, device=dev)
        for i in range(n):
            bn = plan.bns[i]
            L[i]["wt_fwd"], L[i]["kernel"] = fwd_weights[i].data_ptr(), kernels[i].data_ptr()
            L[i]["gamma"], L[i]["beta"] = gammas[i].data_ptr(), betas[i].data_ptr()
            L[i]["running_mean"], L[i]["running_var"], L[i]["num_batches"] = bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr()
        # scratch: partial products of the widest convolution (forward: Cout columns; backward input gradient: Cin columns), BatchNorm / weight-gradient workspace
        p_caps = tabs["p_cap"]
        part_f = int(max(int(p_caps[L["table"][i]]) * int(L["Cout"][i]) for i in range(n)))
        part_b = int(max([int(p_caps[L["table_t"][i]]) * int(L["Cin"][i]) for i in range(n) if L["need_dx"][i]] or [0]))
        ws_bytes = 0
        for i in range(n):
            ws_bytes = max(ws_bytes, libq.sd3d_bn_ws_bytes(int(out_rows[i]), int(L["Cout"][i])),
                           libq.sd3d_pair_wgrad_ws_bytes(int(L["K"][i]), int(L["Cout"][i]), int(L["Cin"][i])))
        part = ops._WS3.get(max(part_f, part_b) * 4, dev)
        ws = train_ops._WS.get(ws_bytes, dev)
        rc = lib.sd3d_unet_train_forward(L.ctypes.data, n, tabs.ctypes.data, len(tabs), bufs.ctypes.data, len(bufs), raws.ctypes.data, n,
                                         stats.data_ptr(), part.data_ptr(), part.numel(), ws.data_ptr(), ws.numel(), ops._stream())
        if rc:
            _lib.check(rc, "unet_train_forward")
        for bn in plan.bns:                                      # the kernels advanced the running statistics behind torch's back
            torch.autograd.graph.increment_version((bn.running_mean, bn.running_var, bn.num_batches_tracked))
        ctx.plan, ctx.maps, ctx.keep = plan, maps, (keep, ident, fwd_weights, x)
        ctx.state = (L, tabs, bufs, raws, arena, raw_arena, stats, sizes, offs, int(raw_sizes.max()), ws_bytes, max(part_f, part_b))
        ctx.save_for_backward(*params)
        o = int(offs[plan.out_id])
        m = int(rows[plan.out_id])
        return arena[o:o + m * plan.out_ch].view(m, plan.out_ch)

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load_nogil()
        plan = ctx.plan
        n = plan.n
        L, tabs, bufs, raws, arena, raw_arena, stats, sizes, offs, graw_floats, ws_bytes, part_floats = ctx.state
        params = ctx.saved_tensors
        dev = dout.device
        dout = dout.contiguous()
        garena = torch.empty(int(sizes.sum()) + 4, dtype=torch.float32, device=dev)
        grads = bufs.copy()
        grads["ptr"] = garena.data_ptr() + offs * 4
        grads["ptr"][0] = 0                                      # the network input carries no gradient
        grads["ld"][0] = bufs["ld"][0]
        grads["ptr"][plan.out_id], grads["ld"][plan.out_id] = dout.data_ptr(), dout.stride(0)
        dk, dg, db = [], [], []
        L = L.copy()
        for i in range(n):
            K, cin, cout = int(L["K"][i]), int(L["Cin"][i]), int(L["Cout"][i])
            dk.append(torch.empty(K, cin, cout, dtype=torch.float32, device=dev))
            L[i]["dkernel"] = dk[-1].data_ptr()
        bn_flat = torch.empty(2 * int(L["Cout"].sum()), dtype=torch.float32, device=dev)
        o = 0
        for i in range(n):
            c = int(L["Cout"][i])
            dg.append(bn_flat[o:o + c]); db.append(bn_flat[o + c:o + 2 * c])
            L[i]["dgamma"], L[i]["dbeta"] = dg[-1].data_ptr(), db[-1].data_ptr()
            o += 2 * c
        graw = torch.empty(graw_floats + 4, dtype=torch.float32, device=dev)
        part = ops._WS3.get(part_floats * 4, dev)
        ws = train_ops._WS.get(ws_bytes, dev)
        rc = lib.sd3d_unet_train_backward(L.ctypes.data, n, tabs.ctypes.data, len(tabs), bufs.ctypes.data, grads.ctypes.data, len(bufs),
                                          raws.ctypes.data, n, stats.data_ptr(), graw.data_ptr(), graw_floats, part.data_ptr(), part.numel(),
                                          ws.data_ptr(), ws.numel(), ops._stream())
        if rc:
            _lib.check(rc, "unet_train_backward")
        out = []
        for i in range(n):
            p = params[i]
            g = dk[i]
            if p.dim() == 2:
                g = g[0]
            if g.shape[-2] != p.shape[-2]:                       # the stem: input channels padded to a multiple of 32
                g = g[..., :p.shape[-2], :]
            out.append(g if ctx.needs_input_grad[4 + i] else None)
        out += [dg[i] if ctx.needs_input_grad[4 + n + i] else None for i in range(n)]
        out += [db[i] if ctx.needs_input_grad[4 + 2 * n + i] else None for i in range(n)]
        ctx.state = ctx.keep = None
        return (None, None, None, None, *out)


def supported(bns) -> bool:
    return all(bn.track_running_stats and bn.running_mean is not None and bn.momentum is not None and bn.affine for bn in bns)


def run(plan: TrainPlan, maps, x: torch.Tensor, fwd_weights) -> torch.Tensor:
    """x [V_0, C_in (padded)] -> the network output [V_out, C] as ONE autograd node; `fwd_weights` = the [K, Cout, Cin] copies of this
    step in plan order (train_ops.transpose_all)."""
    params = list(plan.params) + [bn.weight for bn in plan.bns] + [bn.bias for bn in plan.bns]
    return _UNetTrain.apply(x, plan, maps, fwd_weights, *params)
