"""Layer plans: a sparse U-Net forward as ONE call into the C library (`sd3d_run_layers`).

The network is written once against a small backend interface (`conv`, `dense`, `affine`):
  * `EagerBackend` issues every layer through `segdino3d_amd.ops` (one ctypes call + one tensor per layer),
  * `Recorder` records the same calls into a `LayerPlan` - an array of `sd3d_layer` structs with the packed
    weight pointers baked in - which is then replayed per scene with the scene's neighbour tables and one
    activation arena.
Both run the same kernels in the same order, so their results are bit-identical
(`tests/test_gpu_sparse.py::test_layer_plan_matches_eager`).  Why: ~110 Python-level launches per scene held
the GIL for ~2 ms of the ~6.5 ms a forward costs on the host, and the host - not the GPU - limits scenes/s.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import numpy as np
import torch

from . import _lib, _trace, ops

KIND_PAIR_CONV, KIND_DENSE, KIND_AFFINE = 0, 1, 2
# SD3D_PLAN=0: issue every layer from Python (EagerBackend) - tuning / ablation / instrumentation
USE_PLAN = os.environ.get("SD3D_PLAN", "1") != "0"

LAYER_DT = np.dtype([("kind", "<i4"), ("table", "<i4"), ("src0", "<i4"), ("src1", "<i4"), ("res", "<i4"), ("dst", "<i4"),
                     ("K", "<i4"), ("Cin", "<i4"), ("C0", "<i4"), ("Cout", "<i4"), ("act", "<i4"), ("pad_", "<i4"),
                     ("wt", "<u8"), ("scale", "<u8"), ("shift", "<u8")], align=True)
TABLE_DT = np.dtype([("in_idx", "<u8"), ("tile_k", "<u8"), ("pos", "<u8"), ("p_cap", "<i8"), ("M", "<i8"), ("K", "<i4"),
                     ("pad_", "<i4"), ("rlist", "<u8"), ("out_idx", "<u8"), ("rl_stride", "<i4"), ("center", "<i4")], align=True)
BUF_DT = np.dtype([("ptr", "<u8"), ("rows", "<i8"), ("ld", "<i4"), ("pad_", "<i4")], align=True)
assert LAYER_DT.itemsize == 72 and TABLE_DT.itemsize == 72 and BUF_DT.itemsize == 24


def table_level(key: Tuple) -> int:
    """Level whose voxels are the OUTPUT rows of a neighbour table."""
    return key[1] + 1 if key[0] == "down" else key[1]


class EagerBackend:
    """Layer-by-layer execution through ops (tensors in, tensors out)."""

    def __init__(self, maps):
        self.maps = maps

    def conv(self, x, wt, affine, key, x2=None, res=None, act=None):
        s, b = affine if affine is not None else (None, None)
        return ops.gather_gemm(x, wt, x2=x2, scale=s, shift=b, res=res, act=act, **self.maps.conv_table(*key))

    def dense(self, x, wt, affine, x2=None, res=None, act=None):
        s, b = affine if affine is not None else (None, None)
        return ops.gather_gemm(x, wt, x2=x2, scale=s, shift=b, res=res, act=act)

    def affine(self, x, affine, x2=None, act=None, add=None):
        return ops.scale_shift_act(x, affine[0], affine[1], act=act, x2=x2, add=add)


class _Sym:
    __slots__ = ("id", "level", "ch")

    def __init__(self, id_, level, ch):
        self.id, self.level, self.ch = id_, level, ch


class Recorder:
    """Records the backend calls of one network definition into a LayerPlan."""

    def __init__(self, in_channels: int, in_level: int = 0):
        self.layers: List[dict] = []
        self.bufs: List[Tuple[int, int]] = [(in_level, in_channels)]       # buffer 0 = the network input
        self.tables: List[Tuple] = []
        self.keep = []                                                      # weight tensors the plan points into
        self.input = _Sym(0, in_level, in_channels)

    def _new(self, level, ch):
        self.bufs.append((level, ch))
        return _Sym(len(self.bufs) - 1, level, ch)

    def _table(self, key):
        if key not in self.tables:
            self.tables.append(key)
        return self.tables.index(key)

    def _ptr(self, t):
        if t is None:
            return 0
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("plan weights must be contiguous fp32 device tensors")
        self.keep.append(t)
        return t.data_ptr()

    def _layer(self, kind, x, x2, res, out, wt, affine, act, table=-1):
        s, b = affine if affine is not None else (None, None)
        K, Cout, Cin = (wt.shape if wt is not None else (1, out.ch, out.ch))
        self.layers.append(dict(kind=kind, table=table, src0=x.id, src1=-1 if x2 is None else x2.id,
                                res=-1 if res is None else res.id, dst=out.id, K=K, Cin=Cin, C0=x.ch, Cout=Cout,
                                act=ops.ACT[act], pad_=0, wt=self._ptr(wt), scale=self._ptr(s), shift=self._ptr(b)))
        return out

    def conv(self, x, wt, affine, key, x2=None, res=None, act=None):
        if wt.dim() != 3 or wt.shape[2] != x.ch + (x2.ch if x2 is not None else 0):
            raise ValueError("plan: weight / input channel mismatch")
        out = self._new(table_level(key), wt.shape[1])
        return self._layer(KIND_PAIR_CONV, x, x2, res, out, wt, affine, act, self._table(key))

    def dense(self, x, wt, affine, x2=None, res=None, act=None):
        if wt.dim() == 2:
            wt = wt.unsqueeze(0)
        out = self._new(x.level, wt.shape[1])
        return self._layer(KIND_DENSE, x, x2, res, out, wt, affine, act)

    def affine(self, x, affine, x2=None, act=None, add=None):
        """out = act(cat[x, x2] * scale + shift) + add: for THIS layer kind the `res` slot is summed in after the activation."""
        out = self._new(x.level, x.ch + (x2.ch if x2 is not None else 0))
        if add is not None and (add.level != out.level or add.ch != out.ch):
            raise ValueError("plan: affine add operand does not match the output rows / channels")
        return self._layer(KIND_AFFINE, x, x2, add, out, None, affine, act)

    def finish(self, output: _Sym) -> "LayerPlan":
        return LayerPlan(self, output)


class LayerPlan:
    def __init__(self, rec: Recorder, output: _Sym):
        self.layers = np.zeros(len(rec.layers), dtype=LAYER_DT)
        for i, L in enumerate(rec.layers):
            for k, v in L.items():
                self.layers[k][i] = v
        self.buf_level = np.array([b[0] for b in rec.bufs], dtype=np.int64)
        self.buf_ch = np.array([b[1] for b in rec.bufs], dtype=np.int64)
        self.table_keys = list(rec.tables)
        self.keep = rec.keep
        self.out_id, self.out_ch = output.id, output.ch
        # partial-product columns per table = widest convolution that uses it
        self.table_cout = np.zeros(len(self.table_keys), dtype=np.int64)
        for L in rec.layers:
            if L["kind"] == KIND_PAIR_CONV:
                self.table_cout[L["table"]] = max(self.table_cout[L["table"]], L["Cout"])
        self._lib_nogil = None

    def run(self, maps, x: torch.Tensor) -> torch.Tensor:
        """x [V_in_level, C] (row stride may exceed C).  Returns the output activation [V, C_out] (a view into the
        scene's activation arena)."""
        lib = _lib.load_nogil()
        n_vox = np.asarray(maps.n_vox, dtype=np.int64)
        rows = n_vox[self.buf_level]
        sizes = rows * self.buf_ch
        sizes[0] = 0                                             # buffer 0 is the caller's tensor
        offs = np.concatenate(([0], np.cumsum(sizes)[:-1]))
        arena = torch.empty(int(sizes.sum()) + 4, dtype=torch.float32, device=x.device)
        bufs = np.zeros(len(rows), dtype=BUF_DT)
        bufs["ptr"] = arena.data_ptr() + offs * 4
        bufs["rows"] = rows
        bufs["ld"] = self.buf_ch
        bufs["ptr"][0], bufs["ld"][0] = x.data_ptr(), x.stride(0)
        if x.shape[0] != rows[0] or x.shape[1] < self.buf_ch[0] or x.stride(1) != 1 or x.dtype != torch.float32:
            raise ValueError("plan input does not match the recorded network input")
        tabs = np.zeros(len(self.table_keys), dtype=TABLE_DT)
        evs = np.zeros(len(self.table_keys), dtype=np.uint64)
        ws = ops._WS2.get(256, x.device)

        def fill_tables():
            """-> (floats of partial-product scratch the tables present need, do all tables exist?)"""
            part_floats, complete = 0, True
            pending = getattr(maps, "events", None) or {}
            for i, key in enumerate(self.table_keys):
                pl = maps.pairs.get(key)
                if pl is None:
                    complete = False
                    continue
                tabs[i] = (pl.in_idx.data_ptr(), pl.tile_k.data_ptr(), 0 if pl.pos is None else pl.pos.data_ptr(), pl.p_cap, pl.M, pl.K, 0,
                           0 if pl.rlist is None else pl.rlist.data_ptr(), pl.out_idx.data_ptr() if pl.direct else 0, pl.rl_stride, pl.center)
                ev = pending.get(key)
                evs[i] = 0 if ev is None else ev.cuda_event
                part_floats = max(part_floats, pl.p_cap * int(self.table_cout[i]))
            return part_floats, complete

        def run(a, b, part_floats):
            part = ops._WS3.get(part_floats * 4, x.device)
            rc = lib.sd3d_run_layers_ev(self.layers.ctypes.data + a * LAYER_DT.itemsize, b - a, tabs.ctypes.data, len(tabs), bufs.ctypes.data,
                                        len(bufs), part.data_ptr(), part.numel(), ws.data_ptr(), ws.numel(),
                                        evs.ctypes.data if evs.any() else None, ops._stream())
            if rc:
                _lib.check(rc, "run_layers")

        # Tables built on the scene's side stream (SceneMaps.prepare(fork=True)): this stream waits for a table's event before the first
        # layer that reads it.  Only the stem's table exists when this is called: the layers up to the first one that needs another table
        # go first - the stem convolves while the host is still issuing table kernels - then the next group of tables, and so on.
        n, done = len(self.layers), 0
        while True:
            part_floats, complete = fill_tables()
            if complete:
                run(done, n, part_floats)
                break
            have = np.array([maps.pairs.get(k) is not None for k in self.table_keys])
            needs = (self.layers["kind"] == KIND_PAIR_CONV) & ~have[np.clip(self.layers["table"], 0, len(have) - 1)]
            cut = int(np.argmax(needs))                          # first layer whose table does not exist yet
            if cut > done:
                run(done, cut, part_floats)
                done = cut
            if not getattr(maps, "next_fork", lambda: False)():
                missing = [k for k in self.table_keys if maps.pairs.get(k) is None]
                raise RuntimeError(f"neighbour tables {missing} have no pair lists (SceneMaps.prepare not called for them)")
        if getattr(maps, "events", None):
            maps.join()                                          # (tables no layer read: nothing of the side stream outlives this call unordered)
        getattr(maps, "release_side", lambda: None)()            # the side stream's pool may reuse the tables' blocks only behind these layers
        o = int(offs[self.out_id])
        n = int(rows[self.out_id])
        cap = _trace.active()
        if cap is not None and cap.keep_arena:
            cap.arenas.append(arena[:int(sizes.sum())])
        return arena[o:o + n * self.out_ch].view(n, self.out_ch)
