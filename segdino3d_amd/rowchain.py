"""Host side of the row-chain executor (csrc/rowchain.hip, `sd3d_row_chain`): builds the `sd3d_rc_program` a launch interprets.

A program is a list of ops over LDS slots ([rows][260] fp32 each, rows = 16 or 4) that one workgroup runs for that many consecutive
query rows of one scene - the row-local part of a decoder layer (`instance_seg_3d_decoder.py:606-799`) as ONE launch.  The struct layout below mirrors
`include/segdino3d_hip.h` (checked against `sd3d_row_chain_program_bytes()` when the library is loaded).
"""
from __future__ import annotations

import ctypes as C
import struct
import threading

import numpy as np
import torch

from . import _lib, ops

MAX_OPS, MAX_PROGRAMS, MAX_BATCH = 44, 4, 16
LOAD, STORE, LINEAR, LN, PE, BOX, MERGE, BITS2D, ATTN = 1, 2, 3, 4, 5, 6, 7, 8, 9
F_NO_LDS_DST, F_NORMALIZE, F_KEYS_2D, F_MASK_BITS2D, F_INPLACE = 1, 2, 4, 8, 16
NONE = 0xFF

OP_DT = np.dtype([("type", "u1"), ("act", "u1"), ("src0", "u1"), ("src1", "u1"), ("dst", "u1"), ("res", "u1"), ("flag", "u1"), ("aux", "u1"),
                  ("k0", "<u2"), ("k1", "<u2"), ("cout", "<u2"), ("pad_", "<u2"), ("ld", "<i4"), ("f0", "<f4"),
                  ("p0", "<u8"), ("p1", "<u8"), ("p2", "<u8"), ("p3", "<u8"), ("p4", "<u8")], align=True)
SCENE_DT = np.dtype([("q0", "<i4"), ("nq", "<i4"), ("m0", "<i4"), ("nm", "<i4"), ("bits_off", "<i4"), ("nw", "<i4"), ("near_off", "<i4"),
                     ("ksplit", "<i4"), ("part_off", "<i8")], align=True)
PROGRAM_DT = np.dtype([("n_scenes", "<i4"), ("n_programs", "<i4"), ("n_slots", "<i4"), ("nw_max", "<i4"), ("nw2_max", "<i4"), ("tile_rows", "<i4"),
                       ("rng", "<u8"), ("tile0", "<i4", (MAX_BATCH + 1,)), ("prog_begin", "<i4", (MAX_PROGRAMS + 1,)),
                       ("scenes", SCENE_DT, (MAX_BATCH,)), ("ops", OP_DT, (MAX_OPS,))], align=True)
assert OP_DT.itemsize == 64 and SCENE_DT.itemsize == 40
_checked = False


def _f32(t, name):
    if t is None:
        return 0
    if not t.is_cuda:
        raise RuntimeError(f"row_chain {name}: expected a tensor on the HIP device, got {t.device} (no CPU fallback)")
    if t.dtype != torch.float32:
        raise TypeError(f"row_chain {name}: expected float32, got {t.dtype}")
    return t.data_ptr()


def _rows(t, name):
    """2-D fp32 device tensor with contiguous rows -> (ptr, row stride)."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"row_chain {name}: expected a 2-D tensor with contiguous rows")
    return _f32(t, name), t.stride(0)


_OP = struct.Struct("<8B4HifQQQQQ")
_SCENE = struct.Struct("<8iq")
_HEAD = struct.Struct("<6iQ17i5i")
_OFF_SCENES = _HEAD.size
_OFF_OPS = _OFF_SCENES + MAX_BATCH * _SCENE.size
assert _OP.size == 64 and _SCENE.size == 40 and _OFF_SCENES == 120 and _OFF_OPS + MAX_OPS * _OP.size == PROGRAM_DT.itemsize


def _nslots(width):
    ld = 260 if width <= 256 else width + 4
    return (16 * ld + 4159) // 4160


_PACK_CACHE = {}
_PACK_LOCK = threading.Lock()


def pack_weight(w, rows=16):
    """nn.Linear weight [cout, K] -> the MFMA-fragment order the LINEAR op streams (include/segdino3d_hip.h):
    16-row tiles: P[tile][group][lane = 16 * kq + c16][e] = w[min(16 * tile + c16, cout - 1)][16 * group + 4 * kq + e];
    4-row tiles:  P[tile][quad][lane][e] = w[min(64 * tile + lane, cout - 1)][4 * quad + e].
    Cached per source tensor (address, shape, version counter; the entry keeps its source alive); owners that rewrite weights behind
    the version counter call `clear_pack_cache()` (ScanNetQueryDecoder._derived_reset does)."""
    key = (w.data_ptr(), tuple(w.shape), w._version, rows)
    with _PACK_LOCK:
        hit = _PACK_CACHE.get(key)
    if hit is not None:
        return hit[0]
    cout, K = w.shape
    if K % 16:
        raise ValueError("row_chain LINEAR: K must be a multiple of 16")
    tc = 16 if rows == 16 else 64
    tiles = (cout + tc - 1) // tc
    src = w.detach()
    if tiles * tc != cout:
        src = torch.cat([src, src[-1:].expand(tiles * tc - cout, K)])
    if rows == 16:
        packed = src.reshape(tiles, 16, K // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous()   # [tile, group, kq, c16, e]
    else:
        packed = src.reshape(tiles, 64, K // 4, 4).permute(0, 2, 1, 3).contiguous()          # [tile, quad, lane, e]
    with _PACK_LOCK:
        if len(_PACK_CACHE) > 4096:
            _PACK_CACHE.clear()
        _PACK_CACHE[key] = (packed, w)
    return packed


def clear_pack_cache():
    with _PACK_LOCK:
        _PACK_CACHE.clear()


class Program:
    """One launch of `sd3d_row_chain`.  Slots are small integers; `begin()` starts a further program (gridDim.y) over the same rows.
    The struct is packed with `struct.pack_into` (one call per op: the decoder issues ~300 ops per scene)."""

    def __init__(self, n_slots: int, rng=None, rows: int = 16):
        """rows: query rows per workgroup - 16 (csrc/rowchain.hip, 16 waves) or 4 (csrc/rowchain_narrow.hip, 8 waves; a few hundred rows)."""
        if rows not in (4, 16):
            raise ValueError("row_chain: 16 or 4 rows per tile")
        self.rows = rows
        self.buf = C.create_string_buffer(PROGRAM_DT.itemsize)
        self.n_ops = 0
        self.n_slots = n_slots
        self.rng = 0 if rng is None else _f32(rng, "rng")
        self.prog_begin = [0]
        self._keep = [rng]

    def begin(self):
        if len(self.prog_begin) >= MAX_PROGRAMS:
            raise ValueError("row_chain: at most 4 programs per launch")
        self.prog_begin.append(self.n_ops)

    def _op(self, keep=(), type=0, act=0, src0=NONE, src1=NONE, dst=NONE, res=NONE, flag=0, aux=0, k0=0, k1=0, cout=0, ld=0, f0=0.0,
            p0=0, p1=0, p2=0, p3=0, p4=0):
        self._keep.extend(keep)          # every tensor an op points at lives at least until the launch is enqueued (stream-ordered allocator)
        if self.n_ops >= MAX_OPS:
            raise ValueError(f"row_chain: more than {MAX_OPS} ops in one launch")
        _OP.pack_into(self.buf, _OFF_OPS + 64 * self.n_ops, type, act, src0, src1, dst, res, flag, aux, k0, k1, cout, 0, ld, f0, p0, p1, p2, p3, p4)
        self.n_ops += 1

    # ---- ops ---------------------------------------------------------------------------------------------------------------
    def load(self, dst, t, width=None):
        ptr, ld = _rows(t, "LOAD")
        self._op(keep=(t,), type=LOAD, dst=dst, cout=t.shape[1] if width is None else width, ld=ld, p0=ptr)

    def store(self, src, t, width=None):
        ptr, ld = _rows(t, "STORE")
        self._op(keep=(t,), type=STORE, src0=src, cout=t.shape[1] if width is None else width, ld=ld, p0=ptr)

    def linear(self, dst, src0, w, b=None, act=None, res=None, src1=None, gout=None):
        """dst = act([src0 | src1] w^T + b (+ res)); dst None: only the global copy `gout` [rows, cout]."""
        cout, K = w.shape
        wp = pack_weight(w, self.rows)
        k1 = K // 2 if src1 is not None else 0
        gp, gld = (0, 0) if gout is None else _rows(gout, "LINEAR out")
        if gout is not None and gout.shape[1] != cout:
            raise ValueError("row_chain LINEAR: global output width != cout")
        flag = F_NO_LDS_DST if dst is None else 0
        if dst is not None:                                      # dst overlapping a source: the kernel separates reads from stores by a barrier
            d0, d1 = dst, dst + _nslots(cout)
            if (src0 < d1 and d0 < src0 + _nslots(K - k1)) or (src1 is not None and src1 < d1 and d0 < src1 + _nslots(k1)):
                flag |= F_INPLACE
        self._op(keep=(wp, b, gout), type=LINEAR, act=ops.ACT[act], src0=src0, src1=NONE if src1 is None else src1, dst=NONE if dst is None else dst,
                 res=NONE if res is None else res, flag=flag, k0=K - k1, k1=k1, cout=cout, ld=gld,
                 p0=_f32(wp, "weight"), p1=_f32(b, "bias"), p2=gp)

    def ln(self, dst, src, g, b, res=None, act=None, eps=1e-5, gout=None):
        gp, gld = (0, 0) if gout is None else _rows(gout, "LN out")
        self._op(keep=(g, b, gout), type=LN, act=ops.ACT[act], src0=src, dst=dst, res=NONE if res is None else res, f0=eps, ld=gld, p0=_f32(g, "ln weight"),
                 p1=_f32(b, "ln bias"), p2=gp)

    def pe(self, dst, xyz, dim_t, axis, num_slot=None, den=None):
        if xyz.shape[1] != 3 or not xyz.is_contiguous() or (den is not None and (tuple(den.shape) != tuple(xyz.shape) or not den.is_contiguous())):
            raise ValueError("row_chain PE: xyz / denominator must be contiguous [rows, 3]")
        if axis.dtype != torch.int8:
            raise TypeError("row_chain PE: axis table must be int8")
        self._op(keep=(xyz, dim_t, axis, den), type=PE, dst=dst, src0=NONE if num_slot is None else num_slot, p0=_f32(xyz, "xyz"), p1=_f32(dim_t, "dim_t"), p2=axis.data_ptr(),
                 p3=_f32(den, "den"))

    def box(self, ref, dc_slot, center, size_prev=None, ds_slot=None, size=None, size_metric=None, normalize=False):
        for t in (ref, center, size_prev, size, size_metric):
            if t is not None and (t.dim() != 2 or t.shape[1] != 3 or not t.is_contiguous()):
                raise ValueError("row_chain BOX: tensors must be contiguous [rows, 3]")
        self._op(keep=(ref, center, size_prev, size, size_metric), type=BOX, src0=dc_slot, src1=NONE if ds_slot is None else ds_slot, flag=F_NORMALIZE if normalize else 0, p0=_f32(ref, "ref"),
                 p1=_f32(size_prev, "size_prev"), p2=_f32(center, "center"), p3=_f32(size, "size"), p4=_f32(size_metric, "size_metric"))

    def merge(self, dst, parts, out):
        ptr, ld = _rows(out, "MERGE out")
        self._op(keep=(parts, out), type=MERGE, dst=dst, ld=ld, p0=0 if parts is None else parts.data_ptr(), p1=ptr)

    def bits2d(self, blocked_all, near_all):
        self._op(keep=(blocked_all, near_all), type=BITS2D, p0=blocked_all.data_ptr(), p1=near_all.data_ptr())

    def attn(self, dst, q_slot, k, v, scale, aux, keys_2d=False, masked=False):
        kp, ldk = _rows(k, "ATTN keys")
        vp, ldv = _rows(v, "ATTN values")
        if ldk != ldv:
            raise ValueError("row_chain ATTN: keys and values must share their row stride")
        self._op(keep=(k, v), type=ATTN, dst=dst, src0=q_slot, aux=aux, flag=(F_KEYS_2D if keys_2d else 0) | (F_MASK_BITS2D if masked else 0), ld=ldk,
                 f0=scale, p0=kp, p1=vp)

    # ---- launch ------------------------------------------------------------------------------------------------------------
    def launch(self, scenes, nw_max=0, nw2_max=0):
        """scenes: list of dicts(q0, nq[, m0, nm, bits_off, nw, near_off, ksplit, part_off])."""
        global _checked
        lib = _lib.load()
        if not _checked:
            if lib.sd3d_row_chain_program_bytes() != PROGRAM_DT.itemsize:
                raise RuntimeError(f"sd3d_rc_program: library {lib.sd3d_row_chain_program_bytes()} bytes, binding {PROGRAM_DT.itemsize}")
            _checked = True
        n = len(scenes)
        if n > MAX_BATCH:
            raise ValueError("row_chain: at most 16 scenes per launch")
        tile0, tiles = [0] * (MAX_BATCH + 1), 0
        for i, sc in enumerate(scenes):
            tile0[i] = tiles
            tiles += (sc["nq"] + self.rows - 1) // self.rows
            g = sc.get
            _SCENE.pack_into(self.buf, _OFF_SCENES + 40 * i, sc["q0"], sc["nq"], g("m0", 0), g("nm", 0), g("bits_off", 0), g("nw", 0),
                             g("near_off", 0), g("ksplit", 0), g("part_off", 0))
        tile0[n] = tiles
        pb = self.prog_begin + [self.n_ops]
        n_prog = len(self.prog_begin)
        pb += [0] * (MAX_PROGRAMS + 1 - len(pb))
        _HEAD.pack_into(self.buf, 0, n, n_prog, self.n_slots, nw_max, nw2_max, self.rows, self.rng, *tile0, *pb)
        _lib.check(lib.sd3d_row_chain(C.addressof(self.buf), ops._stream()), "row_chain")
