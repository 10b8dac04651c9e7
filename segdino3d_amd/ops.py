"""Python operator layer over the C ABI: torch tensors in, torch tensors out.

torch is used here for device memory (caching allocator) and the current HIP stream only; all
arithmetic happens in libsegdino3d_hip.so.  Every op raises if a tensor is not on a HIP device -
there is deliberately no CPU path (the CPU restatement is oracle/, test infrastructure).
"""
from __future__ import annotations

import ctypes
import threading
from collections import OrderedDict as _collections_od
from typing import Optional

import torch

from . import _lib

ACT = {None: 0, "none": 0, "relu": 1, "gelu": 2, "sigmoid": 3}

# Optional instrumentation for bench.py: an object with before(meta: dict) / after() called around
# every gather_gemm launch on the current stream (meta: K, Cin, Cout, M, nbr tensor or None).
GG_HOOK = None
# Debug / tuning override of the gather_gemm tiling heuristic (see csrc/gather_gemm.hip launch codes).
import os as _os
GG_FORCE_NT = int(_os.environ["SD3D_GG_NT"]) if _os.environ.get("SD3D_GG_NT") else None


_STREAM_TLS = threading.local()


def _stream():
    s = getattr(_STREAM_TLS, "handle", None)
    return s if s is not None else torch.cuda.current_stream().cuda_stream


class stream_scope:
    """Binds this thread's current HIP stream handle for the duration of a forward.

    torch.cuda.current_stream() costs ~8 us per lookup on the host - more than the launch it precedes,
    and a forward makes ~360 of them.  Inside the scope every op enqueues on the stream that was current
    when the outermost scope was entered; do not switch torch streams inside it."""

    def __enter__(self):
        self.prev = getattr(_STREAM_TLS, "handle", None)
        if self.prev is None:
            _STREAM_TLS.handle = torch.cuda.current_stream().cuda_stream
        return self

    def __exit__(self, *exc):
        _STREAM_TLS.handle = self.prev
        return False


class use_stream:
    """Issue on `stream` (a torch.cuda.Stream) inside the scope: torch's current stream AND the bound handle of this thread's
    ops move together.  The caller orders the streams (`stream.wait_stream(...)` before, `....wait_stream(stream)` after)."""

    def __init__(self, stream):
        self.stream = stream
        self.ctx = torch.cuda.stream(stream)

    def __enter__(self):
        self.prev = getattr(_STREAM_TLS, "handle", None)
        self.ctx.__enter__()
        _STREAM_TLS.handle = self.stream.cuda_stream
        return self

    def __exit__(self, *exc):
        _STREAM_TLS.handle = self.prev
        return self.ctx.__exit__(*exc)


_SIDE_TLS = threading.local()


def side_streams(n: int, device=None):
    """`n` side streams owned by the calling host thread (created once, reused by every forward of the thread)."""
    pool = getattr(_SIDE_TLS, "pool", None)
    if pool is None:
        pool = _SIDE_TLS.pool = []
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def bound_stream(fn):
    """Decorator form of stream_scope for the public forward entry points."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **kw):
        with stream_scope():
            return fn(*a, **kw)
    return wrapped


def _ptr(t: Optional[torch.Tensor], dtype=None, name="tensor"):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on the HIP device, got {t.device} (no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t.data_ptr()


def _rows(t: torch.Tensor, name="tensor"):
    """2-D fp32 tensor whose rows may be strided (last dim contiguous) -> (ptr, ld)."""
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on the HIP device, got {t.device} (no CPU fallback)")
    if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: expected a 2-D fp32 tensor with contiguous rows")
    return t.data_ptr(), t.stride(0)


class HostRead:
    """Small device -> host read: asynchronous copy into pinned memory + an event the caller waits on through
    `wait_event` (which hands the issue baton of the pipelined runner to another scene's thread meanwhile)."""

    __slots__ = ("buf", "event")

    def __init__(self, dev_tensor: torch.Tensor):
        if BLOCKING_SYNC:
            self.buf, self.event = dev_tensor.cpu(), None
            return
        self.buf = torch.empty(dev_tensor.shape, dtype=dev_tensor.dtype, pin_memory=True)
        self.buf.copy_(dev_tensor, non_blocking=True)
        self.event = torch.cuda.Event(blocking=BLOCKING_EVENTS)
        self.event.record()

    def wait(self):
        if self.event is not None:
            wait_event(self.event)
        return self.buf


# hipEventBlockingSync: a thread waiting for its scene sleeps instead of spinning on a core (8 ranks x 4 scene threads
# share one host in the multi-GPU runs).  SD3D_SPIN_EVENTS=1 restores spinning waits.
BLOCKING_EVENTS = _os.environ.get("SD3D_SPIN_EVENTS") != "1"
BLOCKING_SYNC = _os.environ.get("SD3D_BLOCKING_SYNC") == "1"      # A/B switch: blocking .cpu() / event.synchronize()


_BATON_TLS = threading.local()


def set_baton(lock):
    """Install (or clear, with None) this thread's issue baton: the pipelined runner lets only the baton
    holder run Python; a thread hands the baton over exactly while it waits for the GPU (wait_event)."""
    _BATON_TLS.lock = lock


import collections as _collections
_BATON_WAITERS = _collections.deque()                          # append / pop are atomic: threads that want the baton back
BATON_YIELD = _os.environ.get("SD3D_BATON_YIELD", "1") != "0"


def baton_yield():
    """Called between the big issue blocks of a forward (decoder layers, post-processing): if another scene's
    thread has finished waiting for the GPU and wants to issue its next (short) phase, let it go first - its
    stream is empty, ours still has milliseconds of queued work."""
    baton = getattr(_BATON_TLS, "lock", None)
    if baton is not None and BATON_YIELD and _BATON_WAITERS:
        import time
        baton.release()
        time.sleep(0)
        _BATON_WAITERS.append(1)
        baton.acquire()
        _BATON_WAITERS.pop()


def wait_event(ev):
    """Wait for `ev` without keeping other scene threads from issuing."""
    import time
    baton = getattr(_BATON_TLS, "lock", None)
    if baton is not None:
        baton.release()
        try:
            ev.synchronize()                                   # blocks inside HIP with the GIL released
        finally:
            _BATON_WAITERS.append(1)
            baton.acquire()
            _BATON_WAITERS.pop()
        return
    if BLOCKING_SYNC:
        ev.synchronize()
        return
    spins = 0
    while not ev.query():
        spins += 1
        time.sleep(0 if spins < 200 else 5e-5)


def stream_event():
    ev = torch.cuda.Event(blocking=BLOCKING_EVENTS)
    ev.record()
    return ev


class Workspace:
    """Grow-only scratch buffer (bytes) per device."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes: int, device):
        nbytes = max(int(nbytes), 256)
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != torch.device(device):
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self.buf


class _PerThread:
    """One scratch buffer per (host thread, HIP stream): each worker thread of the pipelined runner drives its own stream, a
    batched forward additionally fans its scenes' decoders out over side streams (`use_stream`), and scratch must never be
    shared between streams."""

    def __init__(self):
        self._tls = threading.local()

    def get(self, nbytes: int, device):
        by_stream = getattr(self._tls, "ws", None)
        if by_stream is None:
            by_stream = self._tls.ws = {}
        key = _stream()
        ws = by_stream.get(key)
        if ws is None:
            ws = by_stream[key] = Workspace()
        return ws.get(nbytes, device)


_WS = _PerThread()
_WS2 = _PerThread()       # split-K partial sums of gather_gemm
_WS3 = _PerThread()       # partial products of pair_conv
_WS4 = _PerThread()       # pair_lists scratch
_WS5 = _PerThread()       # expand_masks bit table
_WS6 = _PerThread()       # attention key-split partial states


# --------------------------------------------------------------------------------------------
# sort / scan
# --------------------------------------------------------------------------------------------
def sort_pairs(keys: torch.Tensor, vals: Optional[torch.Tensor] = None, begin_bit=0, end_bit=64):
    """Stable ascending sort of uint64-as-int64 keys; returns (sorted_keys, sorted_vals[int32]).
    `keys` (int64 storage of u64 bit patterns) and `vals` are clobbered.  No padding pass: after an even number of 8-bit radix passes
    the result sits in the input / scratch buffers and those are what is returned (`sd3d_sort_pairs_u64_ex`)."""
    lib = _lib.load()
    n = keys.numel()
    dev = keys.device
    keys_out = torch.empty_like(keys)
    vals_out = torch.empty(n, dtype=torch.int32, device=dev)
    scratch = torch.empty(n, dtype=torch.int32, device=dev) if vals is None else None
    ws = _WS.get(lib.sd3d_sort_ws_bytes(n), dev)
    landed = ctypes.c_int(0)
    _lib.check(lib.sd3d_sort_pairs_u64_ex(_ptr(keys, torch.int64, "keys"), _ptr(vals, torch.int32, "vals"),
                                          _ptr(keys_out), _ptr(vals_out), _ptr(scratch), n, begin_bit, end_bit,
                                          ws.data_ptr(), ws.numel(), ctypes.addressof(landed), _stream()), "sort_pairs")
    if landed.value:
        return keys, (vals if vals is not None else scratch)
    return keys_out, vals_out


def scan_exclusive(x: torch.Tensor):
    lib = _lib.load()
    n = x.numel()
    out = torch.empty_like(x)
    total = torch.empty(1, dtype=torch.int32, device=x.device)
    ws = _WS.get(lib.sd3d_scan_ws_bytes(n), x.device)
    _lib.check(lib.sd3d_scan_exclusive_i32(_ptr(x, torch.int32, "x"), _ptr(out), n, _ptr(total), ws.data_ptr(),
                                           ws.numel(), _stream()), "scan_exclusive")
    return out, total


def keys_from_f32(x: torch.Tensor, descending=False):
    lib = _lib.load()
    keys = torch.empty(x.numel(), dtype=torch.int64, device=x.device)
    _lib.check(lib.sd3d_keys_from_f32(_ptr(x, torch.float32, "x"), x.numel(), int(descending), _ptr(keys), _stream()),
               "keys_from_f32")
    return keys


def keys_from_i64(x: torch.Tensor, check=None, max_out=None):
    """check = (bits, flag int32 [1], value): flag |= value when an id needs more than `bits` bits.  max_out int32 [1] (zeroed by the
    caller; needs `check`, bits = 64 for none): receives the largest id."""
    lib = _lib.load()
    keys = torch.empty(x.numel(), dtype=torch.int64, device=x.device)
    if max_out is not None:
        bits, flag, value = check
        _lib.check(lib.sd3d_keys_from_i64_checked_max(_ptr(x, torch.int64, "x"), x.numel(), _ptr(keys), int(bits), _ptr(flag, torch.int32, "flag"),
                                                      int(value), _ptr(max_out, torch.int32, "max_out"), _stream()), "keys_from_i64_checked_max")
        return keys
    if check is not None:
        bits, flag, value = check
        _lib.check(lib.sd3d_keys_from_i64_checked(_ptr(x, torch.int64, "x"), x.numel(), _ptr(keys), int(bits), _ptr(flag, torch.int32, "flag"),
                                                  int(value), _stream()), "keys_from_i64_checked")
        return keys
    _lib.check(lib.sd3d_keys_from_i64(_ptr(x, torch.int64, "x"), x.numel(), _ptr(keys), _stream()), "keys_from_i64")
    return keys


# --------------------------------------------------------------------------------------------
# voxelisation / coordinate maps
# --------------------------------------------------------------------------------------------
def scene_stats(points: torch.Tensor, out: Optional[torch.Tensor] = None):
    """points [N, >=3] fp32 -> stats[9] = (min xyz, max xyz, sum xyz); `out`: a contiguous fp32 [9] row to write into."""
    lib = _lib.load()
    p, ld = _rows(points, "points")
    stats = torch.empty(9, dtype=torch.float32, device=points.device) if out is None else out
    ws = _WS.get(lib.sd3d_scene_stats_ws_bytes(), points.device)
    _lib.check(lib.sd3d_scene_stats(p, ld, points.shape[0], _ptr(stats), ws.data_ptr(), ws.numel(), _stream()),
               "scene_stats")
    return stats


def voxel_keys(points, inv_voxel: float, stats, shift_to_min=False, batch_index=0, want_icoords=True, out=None, err=None):
    """`out` = (keys [n] int64, icoords [n, 3] int32 | None, err [1] int32): slices of batch-wide buffers to write into
    (the error flag is OR-ed, so several scenes may share it)."""
    lib = _lib.load()
    p, ld = _rows(points, "points")
    n = points.shape[0]
    dev = points.device
    if out is not None:
        keys, icoords, err = out
    else:
        keys = torch.empty(n, dtype=torch.int64, device=dev)
        icoords = torch.empty(n, 3, dtype=torch.int32, device=dev) if want_icoords else None
        if err is None:
            err = torch.zeros(1, dtype=torch.int32, device=dev)
    origin = torch.empty(3, dtype=torch.int32, device=dev)
    _lib.check(lib.sd3d_voxel_keys(p, ld, n, float(inv_voxel), _ptr(stats, torch.float32, "stats"), int(shift_to_min),
                                   int(batch_index), _ptr(origin), _ptr(keys), _ptr(icoords), _ptr(err), _stream()),
               "voxel_keys")
    return keys, icoords, origin, err


def unique_sorted(keys, src_idx, n_cap, n_dev=None, shift=0, want_seg_start=False, want_map=True, map_size=None,
                  clip=None, nuniq_out=None):
    """Run-length unique over sorted keys.  Returns (ukeys[n_cap], seg_start[n_cap+1]|None, map|None, n_unique[1])."""
    lib = _lib.load()
    dev = keys.device
    ukeys = torch.empty(n_cap, dtype=torch.int64, device=dev)
    seg = torch.empty(n_cap + 1, dtype=torch.int32, device=dev) if want_seg_start else None
    mp = torch.empty(map_size if map_size is not None else n_cap, dtype=torch.int32, device=dev) if want_map else None
    nuniq = torch.empty(1, dtype=torch.int32, device=dev) if nuniq_out is None else nuniq_out
    ws = _WS.get(lib.sd3d_unique_ws_bytes(n_cap), dev)
    _lib.check(lib.sd3d_unique_sorted(_ptr(keys, torch.int64, "keys"), _ptr(src_idx, torch.int32, "src_idx"), n_cap,
                                      _ptr(n_dev, torch.int32, "n_dev"), shift, _ptr(ukeys), _ptr(seg), _ptr(mp),
                                      _ptr(nuniq), ws.data_ptr(), ws.numel(),
                                      _ptr(clip[0], torch.float32, "clip stats") if clip else None,
                                      float(clip[1]) if clip else 0.0, int(clip[2]) if clip else 0,
                                      int(clip[3]) if clip else 0, _stream()), "unique_sorted")
    return ukeys, seg, mp, nuniq


def unique_levels(keys0: torch.Tensor, n_cap: int, n0_dev: torch.Tensor, n_extra: int, counts_out=None):
    """All coarser levels from the sorted level-0 unique keys in four launches (`sd3d_unique_levels`): ([ukeys_1..], [parent_1..],
    counts int32 [n_extra]) - what `unique_sorted(keys_l, None, n_cap, n_l, 3, want_map=True)` gives level after level (no extent clip)."""
    lib = _lib.load()
    dev = keys0.device
    uk = torch.empty(n_extra, n_cap, dtype=torch.int64, device=dev)
    par = torch.empty(n_extra, n_cap, dtype=torch.int32, device=dev)
    counts = torch.empty(n_extra, dtype=torch.int32, device=dev) if counts_out is None else counts_out
    ws = _WS.get(lib.sd3d_unique_levels_ws_bytes(n_cap, n_extra), dev)
    up = (ctypes.c_void_p * n_extra)(*[uk.data_ptr() + 8 * n_cap * l for l in range(n_extra)])
    pp = (ctypes.c_void_p * n_extra)(*[par.data_ptr() + 4 * n_cap * l for l in range(n_extra)])
    _lib.check(lib.sd3d_unique_levels(_ptr(keys0, torch.int64, "keys0"), n_cap, _ptr(n0_dev, torch.int32, "n0"), n_extra,
                                      ctypes.addressof(up), ctypes.addressof(pp), _ptr(counts), ws.data_ptr(), ws.numel(), _stream()), "unique_levels")
    return [uk[l] for l in range(n_extra)], [par[l] for l in range(n_extra)], counts


def voxel_levels_all(skeys, sidx, n_levels: int, counts_out):
    """Every level of a (batch of) scene(s) from its sorted point keys in four launches (`sd3d_voxel_levels_all`): what
    `unique_sorted(skeys, sidx, N, None, 0, want_seg_start=True, want_map=True)` followed by `unique_levels` returns -
    (ukeys [L x [N]], seg_start [N + 1], inverse [N], parents [(L - 1) x [N]]); counts_out int32 [L] receives the voxels per level."""
    lib = _lib.load()
    N, dev, L = skeys.numel(), skeys.device, int(n_levels)
    uk = torch.empty(L, N, dtype=torch.int64, device=dev)
    par = torch.empty(max(1, L - 1), N, dtype=torch.int32, device=dev)
    seg = torch.empty(N + 1, dtype=torch.int32, device=dev)
    inv = torch.empty(N, dtype=torch.int32, device=dev)
    ws = _WS.get(lib.sd3d_unique_levels_ws_bytes(N, L), dev)
    up = (ctypes.c_void_p * L)(*[uk.data_ptr() + 8 * N * l for l in range(L)])
    pp = (ctypes.c_void_p * max(1, L - 1))(*[par.data_ptr() + 4 * N * l for l in range(L - 1)])
    _lib.check(lib.sd3d_voxel_levels_all(_ptr(skeys, torch.int64, "skeys"), _ptr(sidx, torch.int32, "sidx"), N, L, ctypes.addressof(up), _ptr(seg),
                                         _ptr(inv), ctypes.addressof(pp), _ptr(counts_out, torch.int32, "counts_out"), ws.data_ptr(), ws.numel(),
                                         _stream()), "voxel_levels_all")
    return [uk[l] for l in range(L)], seg, inv, [par[l] for l in range(L - 1)]


_VOX_DESC_DT = None


def voxelise_scene(points, inv_voxel: float, shift_to_min: bool, key_bits: int, n_levels: int, superpoints=None, sp_bits: int = 64):
    """One scene's voxelisation chain from ONE C call (`sd3d_voxelise_scene`): what scene_stats + voxel_keys + sort_pairs + unique_sorted
    (segment starts, point -> voxel map) + unique_levels + keys_from_i64(max_out) give when called one after the other - the same kernels
    in the same order, issued from C instead of from ~10 Python calls in front of everything else the scene needs.
    -> dict(stats, origin, icoords, skeys, sidx, ukeys [L x [N]], seg_start, inverse, parents [L-1 x [N]], readback int32 [L + 2], sp_keys | None)"""
    global _VOX_DESC_DT
    import numpy as np
    if _VOX_DESC_DT is None:
        _VOX_DESC_DT = np.dtype([("points", "<u8"), ("n", "<i8"), ("ld", "<i4"), ("shift_to_min", "<i4"), ("inv_voxel", "<f4"), ("key_bits", "<i4"),
                                 ("n_levels", "<i4"), ("sp_bits", "<i4"), ("stats", "<u8"), ("origin", "<u8"), ("icoords", "<u8"),
                                 ("keys_a", "<u8"), ("keys_b", "<u8"), ("vals_a", "<u8"), ("vals_b", "<u8"), ("ukeys0", "<u8"),
                                 ("seg_start", "<u8"), ("inverse", "<u8"), ("ukeys", "<u8"), ("parents", "<u8"), ("readback", "<u8"),
                                 ("superpoints", "<u8"), ("sp_keys", "<u8"), ("ws", "<u8"), ("ws_bytes", "<u8")], align=True)
        assert _VOX_DESC_DT.itemsize == 176, _VOX_DESC_DT.itemsize
    lib = _lib.load()
    p, ld = _rows(points, "points")
    N, dev, L = points.shape[0], points.device, int(n_levels)
    Np = (N + 63) // 64 * 64                                                                                 # (every array keeps the alignment of an allocation of its own)
    i64 = torch.empty(L + 2 + (1 if superpoints is not None else 0), Np, dtype=torch.int64, device=dev)[:, :N]   # key ping-pong, unique keys per level, id keys
    i32 = torch.empty(L + 2, Np, dtype=torch.int32, device=dev)[:, :N]                                       # value ping-pong, map, parents per level
    icoords = torch.empty(N, 3, dtype=torch.int32, device=dev)
    seg = torch.empty(N + 1, dtype=torch.int32, device=dev)
    stats = torch.empty(9, dtype=torch.float32, device=dev)
    small = torch.empty(3 + L + 2, dtype=torch.int32, device=dev)                                            # origin, read-back
    origin, rb = small[:3], small[3:]
    uk_ptr = (ctypes.c_void_p * max(1, L - 1))(*[i64[3 + l].data_ptr() for l in range(L - 1)])
    par_ptr = (ctypes.c_void_p * max(1, L - 1))(*[i32[3 + l].data_ptr() for l in range(L - 1)])
    ws = _WS.get(lib.sd3d_voxelise_scene_ws_bytes(N, L), dev)
    sp_keys = i64[L + 2] if superpoints is not None else None
    d = np.zeros(1, dtype=_VOX_DESC_DT)
    d[0] = (p, N, ld, int(bool(shift_to_min)), float(inv_voxel), int(key_bits), L, int(sp_bits), stats.data_ptr(), origin.data_ptr(), icoords.data_ptr(),
            i64[0].data_ptr(), i64[1].data_ptr(), i32[0].data_ptr(), i32[1].data_ptr(), i64[2].data_ptr(), seg.data_ptr(), i32[2].data_ptr(),
            ctypes.addressof(uk_ptr), ctypes.addressof(par_ptr), rb.data_ptr(),
            0 if superpoints is None else _ptr(superpoints, torch.int64, "superpoints"), 0 if sp_keys is None else sp_keys.data_ptr(),
            ws.data_ptr(), ws.numel())
    in_a = ctypes.c_int(0)
    _lib.check(lib.sd3d_voxelise_scene(d.ctypes.data, ctypes.addressof(in_a), _stream()), "voxelise_scene")
    a = 0 if in_a.value else 1
    return dict(stats=stats, origin=origin, icoords=icoords, skeys=i64[a], sidx=i32[a], ukeys=[i64[2]] + [i64[3 + l] for l in range(L - 1)],
                seg_start=seg, inverse=i32[2], parents=[i32[3 + l] for l in range(L - 1)], readback=rb, sp_keys=sp_keys)


def hash_build(ukeys: torch.Tensor, n: int):
    lib = _lib.load()
    cap = 1
    while cap < 2 * n + 2:
        cap *= 2
    tk = torch.empty(cap, dtype=torch.int64, device=ukeys.device)
    tv = torch.empty(cap, dtype=torch.int32, device=ukeys.device)
    _lib.check(lib.sd3d_hash_build(_ptr(ukeys, torch.int64, "ukeys"), n, _ptr(tk), _ptr(tv), cap, _stream()), "hash_build")
    return tk, tv


def kernel_map(out_keys, n_out, table, offsets_i8, pair_count=None, mirrored=False):
    """offsets_i8: int8 [K,3] device tensor.  Returns nbr int32 [K, n_out].  pair_count: optional int32 [64]
    device counters (pre-zeroed) whose sum is the rulebook size.  mirrored: the table is the voxel set's own and
    offsets[K-1-k] == -offsets[k] (centred odd kernel) - half the probes."""
    lib = _lib.load()
    tk, tv = table
    K = offsets_i8.shape[0]
    nbr = torch.empty(K, n_out, dtype=torch.int32, device=out_keys.device)
    _lib.check(lib.sd3d_kernel_map(_ptr(out_keys, torch.int64, "out_keys"), n_out, _ptr(tk), _ptr(tv), tk.numel(),
                                   _ptr(offsets_i8, torch.int8, "offsets"), K, 1 if mirrored else 0, _ptr(nbr),
                                   _ptr(pair_count, torch.int32, "pair_count"), _stream()), "kernel_map")
    return nbr


def kernel_maps_hier(keys, parents, n_vox, offs3, offs5, inv27, pair_counts=None, perm8=None):
    """The 3^3 kernel maps of every level (+ the 5^3 map of level 0 when `offs5` is given) from ONE call, through the level hierarchy of
    the sorted keys instead of hash tables (`sd3d_kernel_maps_hier`).  keys: per level int64 [n_l] (finest first); parents: per level
    but the last int32 [n_l]; offs3 / offs5: device int8 [27, 3] / [125, 3]; inv27: numpy int8 [27] ((dx+1) + 3 (dy+1) + 9 (dz+1) -> row
    of offs3).  pair_counts: zeroed int32 [(L + 1), 64] or None.  perm8 (int32 [8], as `stride_maps`): the stride-2 maps of every level
    pair come out of the same launches.  -> ([nbr3 per level], nbr5 | None, [(nbr_down, nbr_up) per level pair] | None)"""
    lib = _lib.load()
    L = len(keys)
    dev = keys[0].device
    n = (ctypes.c_int64 * L)(*[int(v) for v in n_vox])
    nbr3 = [torch.empty(27, int(n_vox[l]), dtype=torch.int32, device=dev) for l in range(L)]
    nbr5 = torch.empty(125, int(n_vox[0]), dtype=torch.int32, device=dev) if offs5 is not None else None
    kp = (ctypes.c_void_p * L)(*[_ptr(keys[l], torch.int64, "keys") for l in range(L)])
    pp = (ctypes.c_void_p * L)(*[(_ptr(parents[l], torch.int32, "parent") if l + 1 < L else None) for l in range(L)])
    np_ = (ctypes.c_void_p * L)(*[t.data_ptr() for t in nbr3])
    inv = (ctypes.c_int8 * 27)(*[int(v) for v in inv27])
    strides, dp, up = None, None, None
    if perm8 is not None and L > 1:
        strides = [(torch.empty(8, int(n_vox[l + 1]), dtype=torch.int32, device=dev), torch.empty(8, int(n_vox[l]), dtype=torch.int32, device=dev))
                   for l in range(L - 1)]
        dp = (ctypes.c_void_p * L)(*([t[0].data_ptr() for t in strides] + [None]))
        up = (ctypes.c_void_p * L)(*([t[1].data_ptr() for t in strides] + [None]))
    ws = _WS.get(lib.sd3d_kernel_maps_hier_ws_bytes(L, ctypes.addressof(n)), dev)
    _lib.check(lib.sd3d_kernel_maps_hier(L, ctypes.addressof(kp), ctypes.addressof(pp), ctypes.addressof(n), ctypes.addressof(np_),
                                         _ptr(nbr5), _ptr(offs3, torch.int8, "offs3"), _ptr(offs5, torch.int8, "offs5"),
                                         ctypes.addressof(inv), _ptr(pair_counts, torch.int32, "pair_counts"),
                                         _ptr(perm8, torch.int32, "perm8") if strides is not None else None,
                                         ctypes.addressof(dp) if strides is not None else None, ctypes.addressof(up) if strides is not None else None,
                                         ws.data_ptr(), ws.numel(), _stream()), "kernel_maps_hier")
    return nbr3, nbr5, strides


def stride_maps(fine_keys, parent, n_fine, n_coarse, perm8, want_down=True, want_up=True):
    lib = _lib.load()
    dev = fine_keys.device
    down = torch.empty(8, n_coarse, dtype=torch.int32, device=dev) if want_down else None
    up = torch.empty(8, n_fine, dtype=torch.int32, device=dev) if want_up else None
    _lib.check(lib.sd3d_stride_maps(_ptr(fine_keys, torch.int64, "fine_keys"), _ptr(parent, torch.int32, "parent"),
                                    n_fine, n_coarse, _ptr(perm8, torch.int32, "perm8"), _ptr(down), _ptr(up),
                                    _stream()), "stride_maps")
    return down, up


def voxel_mean(points, feats2d, mode, stats, sorted_idx, seg_start, n_vox, ld_out):
    lib = _lib.load()
    p, ld = _rows(points, "points")
    F = 0 if feats2d is None else feats2d.shape[1]
    out = torch.empty(n_vox, ld_out, dtype=torch.float32, device=points.device)
    _lib.check(lib.sd3d_voxel_mean(p, ld, _ptr(feats2d, torch.float32, "feats2d"), F, mode,
                                   _ptr(stats, torch.float32, "stats"), points.shape[0],
                                   _ptr(sorted_idx, torch.int32, "sorted_idx"), _ptr(seg_start, torch.int32, "seg_start"),
                                   n_vox, _ptr(out), ld_out, _stream()), "voxel_mean")
    return out


def voxel_mean_batch(scenes, mode, ukeys, sorted_idx, seg_start, n_vox, ld_out):
    """scenes: [(points [N_i, >=6], feats2d [N_i, F] | None, stats [9], point_off)] of one batch (sparse.BatchSceneMaps);
    one launch over the voxels of all scenes."""
    import numpy as np
    lib = _lib.load()
    dt = np.dtype([("points", "<u8"), ("feats2d", "<u8"), ("stats", "<u8"), ("point_off", "<i8"), ("n_points", "<i8"),
                   ("ld_points", "<i4"), ("pad_", "<i4")], align=True)
    assert dt.itemsize == 48
    tab = np.zeros(len(scenes), dtype=dt)
    F = 0
    for i, (pts, f2d, stats, off) in enumerate(scenes):
        p, ld = _rows(pts, "points")
        if f2d is not None:
            if F and f2d.shape[1] != F:
                raise ValueError("voxel_mean_batch: every scene must carry the same number of 2D feature channels")
            F = f2d.shape[1]
        tab[i] = (p, 0 if f2d is None else _ptr(f2d, torch.float32, "feats2d"), _ptr(stats, torch.float32, "stats"), int(off),
                  pts.shape[0], ld, 0)
    dev = scenes[0][0].device
    out = torch.empty(n_vox, ld_out, dtype=torch.float32, device=dev)
    _lib.check(lib.sd3d_voxel_mean_batch(tab.ctypes.data, len(scenes), F, mode, _ptr(ukeys, torch.int64, "ukeys"),
                                         _ptr(sorted_idx, torch.int32, "sorted_idx"), _ptr(seg_start, torch.int32, "seg_start"),
                                         n_vox, _ptr(out), ld_out, _stream()), "voxel_mean_batch")
    return out


def keys_from_i64_offset(x: torch.Tensor, add: int, out: torch.Tensor, check=None, max_out=None):
    """out[i] = x[i] + add (out: a contiguous int64 slice of the batch-wide key buffer).  check = (bits, flag, value) + max_out: as
    `keys_from_i64`, taken on x[i]."""
    lib = _lib.load()
    if out.numel() != x.numel():
        raise ValueError("keys_from_i64_offset: output slice has another length")
    if max_out is not None:
        bits, flag, value = check
        _lib.check(lib.sd3d_keys_from_i64_offset_checked_max(_ptr(x, torch.int64, "x"), x.numel(), int(add), _ptr(out, torch.int64, "out"), int(bits),
                                                             _ptr(flag, torch.int32, "flag"), int(value), _ptr(max_out, torch.int32, "max_out"),
                                                             _stream()), "keys_from_i64_offset_checked_max")
        return out
    _lib.check(lib.sd3d_keys_from_i64_offset(_ptr(x, torch.int64, "x"), x.numel(), int(add), _ptr(out, torch.int64, "out"), _stream()),
               "keys_from_i64_offset")
    return out


def segment_starts_batch(sorted_ids, n, S, id_off):
    """sorted (scene << 32 | id) keys -> start[S + 1] over the dense ids id_off[scene] + id."""
    import ctypes as C
    lib = _lib.load()
    start = torch.empty(S + 1, dtype=torch.int32, device=sorted_ids.device)
    offs = (C.c_int32 * len(id_off))(*[int(v) for v in id_off])
    _lib.check(lib.sd3d_segment_starts_batch(_ptr(sorted_ids, torch.int64, "sorted_ids"), n, S, offs, len(id_off), _ptr(start),
                                             _stream()), "segment_starts_batch")
    return start


def segment_starts(sorted_ids, n, S):
    lib = _lib.load()
    start = torch.empty(S + 1, dtype=torch.int32, device=sorted_ids.device)
    _lib.check(lib.sd3d_segment_starts(_ptr(sorted_ids, torch.int64, "sorted_ids"), n, S, _ptr(start), _stream()),
               "segment_starts")
    return start


def pool_superpoints(feat, C, inverse, icoords, voxel_size, sorted_idx, start, S):
    lib = _lib.load()
    f, ld = _rows(feat, "feat")
    out_feat = torch.empty(S, C, dtype=torch.float32, device=feat.device)
    out_pos = torch.empty(S, 3, dtype=torch.float32, device=feat.device)
    _lib.check(lib.sd3d_pool_superpoints(f, ld, C, _ptr(inverse, torch.int32, "inverse"),
                                         _ptr(icoords, torch.int32, "icoords"), float(voxel_size),
                                         _ptr(sorted_idx, torch.int32, "sorted_idx"), _ptr(start, torch.int32, "start"),
                                         S, _ptr(out_feat), _ptr(out_pos), _stream()), "pool_superpoints")
    return out_feat, out_pos


# --------------------------------------------------------------------------------------------
# gather-GEMM
# --------------------------------------------------------------------------------------------
# Opt-in arithmetic mode of the lock-step gather-GEMM (csrc/gather_gemm_split.hip): None = exact fp32 MFMA
# (default, the mode every parity claim and the bench headline are made in), "bf16x3" / "bf16x6" = fp32
# products evaluated as 3 / 6 bf16 MFMA products with fp32 accumulation.
GEMM_MODE = _os.environ.get("SD3D_GEMM_MODE") or None
if GEMM_MODE not in (None, "bf16x3", "bf16x6"):
    raise ValueError(f"SD3D_GEMM_MODE must be bf16x3 or bf16x6, got {GEMM_MODE!r}")
SPLIT_MIN_ROWS = 2048          # below this the launch is latency-bound and stays on the fp32 kernel
_SPLIT_CACHE = _collections_od()
_BF16_TLS = threading.local()
# projections of fewer rows are launch-latency bound and measured FASTER on the exact fp32 small-M kernel (200-query decoder:
# 2.6 ms fp32, 3.5 ms with every Linear on the bf16 kernel); exact fp32 is never less precise than the bf16 the mode allows
BF16_MIN_ROWS = int(_os.environ.get("SD3D_BF16_MIN_ROWS", "1024"))


class bf16_decoder_scope:
    """BASELINE config #3 ("bf16 decoder"): inside the scope every dense projection (`linear`, `gather_gemm` without a
    neighbour table and without `exact=True`) runs with bf16 operands and fp32 accumulation, and `attention` runs its two
    contractions on the bf16 MFMA.  LayerNorm, softmax, the positional encodings, the mask head's logits and every
    threshold stay fp32 ("bf16 decoder").  Per thread; nests."""

    def __init__(self, enabled=True):
        self.enabled = bool(enabled)

    def __enter__(self):
        self.prev = getattr(_BF16_TLS, "on", False)
        _BF16_TLS.on = self.enabled or self.prev
        return self

    def __exit__(self, *exc):
        _BF16_TLS.on = self.prev
        return False


def bf16_decoder_active() -> bool:
    return getattr(_BF16_TLS, "on", False)


def split_weights(wt, terms):
    """fp32 [K, Cout, Cin] -> bf16 [terms_per_operand, K, Cout, Cin] with wt = sum of the terms (to ~2^-17 / 2^-25)."""
    ns = {1: 1, 3: 2, 6: 3}[terms]
    if ns == 1:                                          # plain rounding: one conversion kernel (training re-rounds live weights per step)
        return wt.detach().to(torch.bfloat16).unsqueeze(0).contiguous()
    r = wt.detach().to(torch.float32).clone()
    parts = []
    for _ in range(ns):
        t = r.to(torch.bfloat16)
        parts.append(t)
        r = r - t.to(torch.float32)
    return torch.stack(parts).contiguous()


_SPLIT_CACHE_MAX = 1024          # a decoder forward touches ~110 distinct Linear weights in the bf16 mode (17 per layer + heads); owners clear on change
_SPLIT_LOCK = threading.Lock()   # the scene threads of dist_eval.PipelinedRunner share the cache


def _cached_split(wt, terms):
    """bf16 rounding(s) of a weight tensor, cached.  An entry keeps its SOURCE tensor alive, so the address in the key
    cannot be handed to another tensor while the entry exists (the caching allocator reuses addresses of freed blocks: a
    rebuilt set of packed weights lands where the old one was); in-place updates move `_version`.  Least-recently-used
    entries are dropped beyond _SPLIT_CACHE_MAX; owners drop everything with clear_split_cache() when their weights change.
    Lookups, insertions and evictions happen under one lock (a hit moves the entry to the end, nothing is popped on the way)."""
    key = (wt.data_ptr(), tuple(wt.shape), tuple(wt.stride()), terms, wt._version)
    with _SPLIT_LOCK:
        hit = _SPLIT_CACHE.get(key)
        if hit is not None:
            _SPLIT_CACHE.move_to_end(key)
            return hit[0]
    made = (split_weights(wt, terms), wt)                 # outside the lock: a conversion launch; a racing thread makes an equal copy
    with _SPLIT_LOCK:
        hit = _SPLIT_CACHE.setdefault(key, made)
        _SPLIT_CACHE.move_to_end(key)
        while len(_SPLIT_CACHE) > _SPLIT_CACHE_MAX:
            _SPLIT_CACHE.popitem(last=False)
    return hit[0]


_IN_FLIGHT = [1]


def scenes_in_flight_now() -> int:
    """How many independent scenes the caller keeps in flight on this GPU (set by `scenes_in_flight`, default 1)."""
    return _IN_FLIGHT[0]


class scenes_in_flight:
    """Context: tell the launchers that `n` independent scenes run concurrently on this GPU (sd3d_set_scenes_in_flight)."""

    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        self.prev = _lib.load().sd3d_set_scenes_in_flight(self.n)
        self.prev_py, _IN_FLIGHT[0] = _IN_FLIGHT[0], self.n
        return self

    def __exit__(self, *exc):
        _lib.load().sd3d_set_scenes_in_flight(self.prev)
        _IN_FLIGHT[0] = self.prev_py
        return False


def clear_split_cache():
    with _SPLIT_LOCK:
        _SPLIT_CACHE.clear()


class PairLists:
    """Offset-major layout of one neighbour table (csrc/pair_gemm.hip): shared by every convolution that uses it.
    Optional products (sd3d_pair_lists_desc): `rlist` [M, rl_stride] per-row partial-product lists, `center` = -1 (plain lists) or
    PAIR_CHAINED (mirror offsets + centre share one partial product), `direct` = every output row has exactly one pair (`out_idx`
    is then built and pass 1 writes the output rows itself)."""
    __slots__ = ("pos", "in_idx", "tile_k", "p_cap", "K", "M", "out_idx", "rlist", "rl_stride", "center", "direct")

    def __init__(self, pos, in_idx, tile_k, p_cap, K, M, rlist=None, rl_stride=0, center=-1, out_idx=None, direct=False):
        self.pos, self.in_idx, self.tile_k, self.p_cap, self.K, self.M = pos, in_idx, tile_k, p_cap, K, M
        self.out_idx = out_idx                            # output row of every pair; built on demand otherwise (train_ops.pair_out_rows)
        self.rlist, self.rl_stride, self.center, self.direct = rlist, rl_stride, center, direct


def pair_lists(nbr, n_pairs, center=-1, direct=False):
    """nbr int32 [K, M]; n_pairs = number of entries >= 0 (host int, e.g. from kernel_map's pair counter).  `center` / `direct`:
    see PairLists (the caller vouches for them: an odd stride-1 table of a voxel set onto itself / a one-pair-per-row table)."""
    return pair_lists_batch([(nbr, n_pairs, center, direct)])[0]


# Sparse convolutions run pair-major (pair_conv) whenever the caller hands over the table's PairLists;
# SD3D_PAIR_CONV=0 keeps the output-stationary gather_gemm kernels (tuning / ablation).
PAIR_CONV = _os.environ.get("SD3D_PAIR_CONV", "1") != "0"
_PAIR_DESC_DT = None


PAIR_CHAINED = -2          # SD3D_PAIR_CHAINED: `center` value of a chained table (include/segdino3d_hip.h)


def pair_lists_batch(tables):
    """tables: list of (nbr int32 [K, M], n_pairs[, center[, direct[, lean]]]) -> list of PairLists, built by ONE launch set
    (csrc/pair_gemm.hip: count, scan, fill, per-row lists).  center = PAIR_CHAINED: chained lists (a stride-1 table of a voxel set
    onto itself, odd symmetric kernel): the entries of a row's mirror groups and its centre share partial products."""
    global _PAIR_DESC_DT
    import numpy as np
    if _PAIR_DESC_DT is None:
        _PAIR_DESC_DT = np.dtype([("nbr", "<u8"), ("pos", "<u8"), ("in_idx", "<u8"), ("tile_k", "<u8"), ("rlist", "<u8"), ("out_idx", "<u8"),
                                  ("M", "<i8"), ("p_cap", "<i8"), ("K", "<i4"), ("center", "<i4"), ("rl_stride", "<i4"), ("meta", "<i4")],
                                 align=True)
        assert _PAIR_DESC_DT.itemsize == 80
    lib = _lib.load()
    out = []
    for start in range(0, len(tables), 16):
        chunk = tables[start:start + 16]
        dev = chunk[0][0].device
        desc = np.zeros(len(chunk), dtype=_PAIR_DESC_DT)
        res, nb = [], 0
        for i, t in enumerate(chunk):
            nbr, n_pairs = t[0], t[1]
            center = int(t[2]) if len(t) > 2 else -1
            direct = bool(t[3]) if len(t) > 3 else False
            lean = bool(t[4]) if len(t) > 4 else False            # evaluation: no [K, M] position table, the unused capacity stays unwritten
            K, M = nbr.shape
            chained = center == PAIR_CHAINED
            if chained:
                if direct or K % 2 == 0:
                    raise ValueError("pair_lists: chained lists need an odd symmetric kernel")
                p_cap = (int(n_pairs) + 127 * (11 * (K // 2) + 1) + 127) // 128 * 128
                pos = torch.empty(0, dtype=torch.int32, device=dev)       # (a chained table has no [K, M] position table: its rows' partial positions are `rlist`)
            else:
                p_cap = (int(n_pairs) + 127 * K + 127) // 128 * 128
                lean = lean and K <= 128
                pos = None if lean else torch.empty(K, M, dtype=torch.int32, device=dev)
            in_idx = torch.empty(p_cap, dtype=torch.int32, device=dev)
            tile_k = torch.empty(p_cap // 128 + 3, dtype=torch.int32, device=dev)     # [p_cap / 128]: number of real tiles, then the centre run
            # per-row list {count, partial positions}: K entries at most - K / 2 + 1 (mirror groups + the lone centre) for a chained table
            rl_stride = ((K // 2 + 2) + 3) // 4 * 4 if chained else (K + 4 + 3) // 4 * 4
            rlist = None if direct else torch.empty(M, rl_stride, dtype=torch.int32, device=dev)
            out_idx = torch.empty(p_cap, dtype=torch.int32, device=dev) if direct else None
            desc[i] = (_ptr(nbr, torch.int32, "nbr"), 0 if pos is None else pos.data_ptr(), in_idx.data_ptr(), tile_k.data_ptr(),
                       0 if rlist is None else rlist.data_ptr(), 0 if out_idx is None else out_idx.data_ptr(), M, p_cap, K, center,
                       rl_stride, 3 if lean else 1)
            nb += (lib.sd3d_pair_lists_ws_bytes(K, M) + 255) // 256 * 256
            res.append(PairLists(pos, in_idx, tile_k, p_cap, K, M, rlist=rlist, rl_stride=rl_stride, center=center, out_idx=out_idx,
                                 direct=direct))
        ws = _WS4.get(nb, dev)
        _lib.check(lib.sd3d_pair_lists_desc(len(chunk), desc.ctypes.data, ws.data_ptr(), ws.numel(), _stream()), "pair_lists_desc")
        out += res
    return out


def pair_conv(x, wt, pairs, x2=None, scale=None, shift=None, res=None, act=None, out=None, mirror_w=False):
    """Same contract as gather_gemm(x, wt, nbr=...) for the table `pairs` was built from.
    mirror_w: offset k multiplies wt[K - 1 - k] (SD3D_PAIR_MIRROR_W: the input gradient of a stride-1 convolution on its own table with
    the parameter in its native [K, Cin, Cout] layout, train_ops.SparseConvNative)."""
    lib = _lib.load()
    K, Cout, Cin = wt.shape
    if K != pairs.K:
        raise ValueError(f"weights have {K} offsets, the pair lists {pairs.K}")
    p0, ld0 = _rows(x, "x")
    C0 = x.shape[1]
    p1, ld1 = (None, 0)
    if x2 is not None:
        p1, ld1 = _rows(x2, "x2")
        if C0 + x2.shape[1] != Cin:
            raise ValueError(f"concat channels {C0}+{x2.shape[1]} != Cin {Cin}")
    elif C0 != Cin:
        raise ValueError(f"input channels {C0} != Cin {Cin}")
    M = pairs.M
    if out is None:
        out = torch.empty(M, Cout, dtype=torch.float32, device=x.device)
    po, ldo = _rows(out, "out")
    pr, ldr = (None, 0)
    if res is not None:
        pr, ldr = _rows(res, "res")
    part = _WS3.get(pairs.p_cap * Cout * 4, x.device)
    hook = GG_HOOK
    if hook is not None:
        hook.before(dict(K=K, Cin=Cin, Cout=Cout, M=M, nbr=None, pairs=pairs))
    _lib.check(lib.sd3d_pair_conv_ex(p0, ld0, C0, p1, ld1, pairs.in_idx.data_ptr(), pairs.tile_k.data_ptr(), pairs.p_cap,
                                     None if pairs.pos is None else pairs.pos.data_ptr(), None if pairs.rlist is None else pairs.rlist.data_ptr(), pairs.rl_stride,
                                     pairs.center - 2 if mirror_w else pairs.center, pairs.out_idx.data_ptr() if pairs.direct else None,
                                     _ptr(wt, torch.float32, "wt"), K, Cin, Cout, M,
                                     _ptr(scale, torch.float32, "scale"), _ptr(shift, torch.float32, "shift"), pr, ldr, po, ldo,
                                     ACT[act], part.data_ptr(), part.numel(), _stream()), "pair_conv")
    if hook is not None:
        hook.after()
    return out


_LINEAR_JOB_DT = None


def dense_code(rows: int, cin: int, cout: int) -> int:
    """Tiling code of sd3d_gather_gemm that reproduces, for ANY number of rows, the kernel its heuristic picks for a plain Linear
    on `rows` rows (sd3d_dense_plan_code - the C library's own answer): the batched decoder passes it so that the rows of several
    scenes run on the SAME kernel - same summation order - as one scene's rows.  0 = the lock-step kernel (>= 64 row tiles: its order
    does not depend on the row count), -1 = one 32-column tile per wave with the contraction split over the four waves of a
    workgroup, n > 0 = n column tiles per wave, no split."""
    return _lib.load().sd3d_dense_plan_code(int(rows), int(cin), int(cout))


def small_rows_code(cin: int) -> int:
    """dense_code for <= 512 rows and <= 1024 output columns (the decoder's query tensors)."""
    return -1 if cin // 32 >= 8 else 1


def linear_group(jobs, force_small=False):
    """jobs: list of (x, weight [Cout, Cin], bias | None, act | None, res | None, x2 | None) - INDEPENDENT plain Linears on a few
    hundred rows each; returns their outputs.  Up to 8 per launch (csrc/gather_gemm.hip gather_gemm_group_kernel).  Jobs on
    >= 2017 rows each go out together as well, every one on the lock-step tiling its own launch would use
    (gather_gemm_lds_group_kernel: same bits).  Mixed sizes, the bf16 decoder scope on >= BF16_MIN_ROWS rows and the
    instrumentation hook run one by one.
    force_small: the rows are several scenes' few-hundred-row tensors back to back - always the group kernel."""
    global _LINEAR_JOB_DT
    import numpy as np
    if GG_HOOK is not None or GEMM_MODE is not None or GG_FORCE_NT is not None:
        if force_small:
            return [gather_gemm(x, w, x2=x2, shift=b, act=act, res=res, nt=-1, exact=True) for (x, w, b, act, res, x2) in jobs]
        return [gather_gemm(x, w, x2=x2, shift=b, act=act, res=res) for (x, w, b, act, res, x2) in jobs]
    if len(jobs) == 1:
        x, w, b, act, res, x2 = jobs[0]
        if force_small:                                         # the kernel the heuristic picks for ONE scene's rows
            return [gather_gemm(x, w, x2=x2, shift=b, act=act, res=res, nt=small_rows_code(w.shape[-1]), exact=True)]
        return [gather_gemm(x, w, x2=x2, shift=b, act=act, res=res)]
    bf16 = getattr(_BF16_TLS, "on", False)
    small = [force_small or (x.shape[0] < 2048 and not (bf16 and x.shape[0] >= BF16_MIN_ROWS) and x.shape[0] > 0) for (x, *_r) in jobs]
    # a few thousand rows each (one query per superpoint): the lock-step kernel of the single launch, several jobs per launch
    large = not bf16 and not force_small and all(w.dim() == 2 and w.shape[1] % 32 == 0 and dense_code(x.shape[0], w.shape[1], w.shape[0]) == 0
                                                   for (x, w, *_r) in jobs)
    if not all(small) and not large:
        return [gather_gemm(x, w, x2=x2, shift=b, act=act, res=res) for (x, w, b, act, res, x2) in jobs]
    if _LINEAR_JOB_DT is None:
        _LINEAR_JOB_DT = np.dtype([("in0", "<u8"), ("in1", "<u8"), ("wt", "<u8"), ("shift", "<u8"), ("res", "<u8"), ("out", "<u8"),
                                   ("M", "<i8"), ("ld0", "<i4"), ("C0", "<i4"), ("ld1", "<i4"), ("Cin", "<i4"), ("Cout", "<i4"),
                                   ("ld_res", "<i4"), ("ld_out", "<i4"), ("act", "<i4")], align=True)
        assert _LINEAR_JOB_DT.itemsize == 88
    lib = _lib.load()
    outs = []
    for start in range(0, len(jobs), 8):
        chunk = jobs[start:start + 8]
        tab = np.zeros(len(chunk), dtype=_LINEAR_JOB_DT)
        for i, (x, w, b, act, res, x2) in enumerate(chunk):
            if not (x.is_cuda and w.is_cuda) or x.dtype is not torch.float32 or w.dtype is not torch.float32 or x.dim() != 2 \
                    or x.stride(1) != 1 or not w.is_contiguous() or w.dim() != 2:
                raise ValueError("linear_group: expected 2-D fp32 device tensors with contiguous rows")
            M, C0 = x.shape
            Cout, Cin = w.shape
            if (x2 is None and C0 != Cin) or (x2 is not None and (C0 + x2.shape[1] != Cin or x2.stride(1) != 1 or x2.shape[0] != M)):
                raise ValueError(f"linear_group: input channels do not match Cin {Cin}")
            out = torch.empty(M, Cout, dtype=torch.float32, device=x.device)
            if res is not None and (res.shape != out.shape or res.stride(1) != 1 or res.dtype is not torch.float32):
                raise ValueError("linear_group: residual must be an fp32 [rows, Cout] tensor")
            if b is not None and (b.numel() != Cout or not b.is_contiguous() or b.dtype is not torch.float32):
                raise ValueError("linear_group: bias must be a contiguous fp32 [Cout] tensor")
            tab[i] = (x.data_ptr(), 0 if x2 is None else x2.data_ptr(), w.data_ptr(), 0 if b is None else b.data_ptr(),
                      0 if res is None else res.data_ptr(), out.data_ptr(), M, x.stride(0), C0, 0 if x2 is None else x2.stride(0), Cin, Cout,
                      0 if res is None else res.stride(0), Cout, ACT[act])
            outs.append(out)
        rc = lib.sd3d_linear_group(len(chunk), tab.ctypes.data, _stream())
        if rc:
            _lib.check(rc, "linear_group")
    return outs


def _dense_linear(x, wt, x2, shift, res, act):
    """Short host path of the plain Linear y = act(x wt^T + shift + res) on [x | x2] (the decoder issues ~150 of them per
    forward; the general entry below spends ~14 us of Python per call, this one ~4): same kernel, same checks that matter -
    device tensors, fp32, contiguous rows, matching widths."""
    if not (x.is_cuda and wt.is_cuda):
        raise RuntimeError("gather_gemm: expected tensors on the HIP device (no CPU fallback)")
    if x.dtype is not torch.float32 or wt.dtype is not torch.float32 or x.dim() != 2 or x.stride(1) != 1 or not wt.is_contiguous():
        raise ValueError("gather_gemm: expected 2-D fp32 tensors with contiguous rows")
    rows, C0 = x.shape
    Cout, Cin = wt.shape
    p1, ld1 = None, 0
    if x2 is not None:
        if x2.dtype is not torch.float32 or x2.stride(1) != 1 or not x2.is_cuda or C0 + x2.shape[1] != Cin:
            raise ValueError(f"gather_gemm: bad second source for Cin {Cin}")
        p1, ld1 = x2.data_ptr(), x2.stride(0)
    elif C0 != Cin:
        raise ValueError(f"input channels {C0} != Cin {Cin}")
    pr, ldr = None, 0
    if res is not None:
        if res.dtype is not torch.float32 or res.stride(1) != 1 or not res.is_cuda or res.shape[0] != rows or res.shape[1] != Cout:
            raise ValueError("gather_gemm: residual must be an fp32 [rows, Cout] device tensor")
        pr, ldr = res.data_ptr(), res.stride(0)
    ps = None
    if shift is not None:
        if not shift.is_cuda or shift.dtype is not torch.float32 or shift.numel() != Cout or not shift.is_contiguous():
            raise ValueError("gather_gemm: shift must be a contiguous fp32 [Cout] device tensor")
        ps = shift.data_ptr()
    out = torch.empty((rows, Cout), dtype=torch.float32, device=x.device)
    rc = _lib.load().sd3d_gather_gemm(x.data_ptr(), x.stride(0), C0, p1, ld1, None, wt.data_ptr(), 1, Cin, Cout, rows, None, ps, pr, ldr,
                                      out.data_ptr(), Cout, ACT[act], 0, None, 0, _stream())
    if rc:
        _lib.check(rc, "gather_gemm")
    return out


def gather_gemm(x, wt, nbr=None, x2=None, scale=None, shift=None, res=None, act=None, out=None, M=None, nt=0,
                density=None, wt_split=None, pairs=None, exact=False):
    """out[r, n] = act(scale[n] * sum_k sum_c X[nbr[k, r], c] * wt[k, n, c] + shift[n] + res[r, n]).

    x [V_in, C0] (rows may be strided), optional x2 [V_in, C1] = concatenated channels,
    wt [K, Cout, Cin] contiguous, nbr int32 [K, M] or None (identity rows, K = 1).  With `pairs` (the table's
    PairLists) a sparse convolution runs pair-major (pair_conv); `density` is informational (kept for callers that
    pass SceneMaps.conv_table(...) as keyword arguments)."""
    if nbr is None and pairs is None and GG_HOOK is None and scale is None and out is None and wt_split is None and nt == 0 \
            and M is None and GEMM_MODE is None and GG_FORCE_NT is None and wt.dim() == 2 and not getattr(_BF16_TLS, "on", False):
        return _dense_linear(x, wt, x2, shift, res, act)
    lib = _lib.load()
    if wt.dim() == 2:
        wt = wt.unsqueeze(0)
    if pairs is not None and PAIR_CONV and nt == 0 and wt_split is None and GEMM_MODE is None and GG_FORCE_NT is None \
            and wt.shape[1] % 4 == 0:
        return pair_conv(x, wt, pairs, x2=x2, scale=scale, shift=shift, res=res, act=act, out=out)
    K, Cout, Cin = wt.shape
    p0, ld0 = _rows(x, "x")
    C0 = x.shape[1]
    p1, ld1 = (None, 0)
    if x2 is not None:
        p1, ld1 = _rows(x2, "x2")
        if C0 + x2.shape[1] != Cin:
            raise ValueError(f"concat channels {C0}+{x2.shape[1]} != Cin {Cin}")
    elif C0 != Cin:
        raise ValueError(f"input channels {C0} != Cin {Cin}")
    if M is None:
        M = nbr.shape[1] if nbr is not None else x.shape[0]
    if nbr is not None and (nbr.shape[0] != K or nbr.shape[1] != M):
        raise ValueError(f"nbr shape {tuple(nbr.shape)} != ({K}, {M})")
    if out is None:
        out = torch.empty(M, Cout, dtype=torch.float32, device=x.device)
    po, ldo = _rows(out, "out")
    pr, ldr = (None, 0)
    if res is not None:
        pr, ldr = _rows(res, "res")
    terms = 0
    if wt_split is not None:
        terms = {1: 1, 2: 3, 3: 6}[wt_split.shape[0]]
    elif nbr is None and not exact and bf16_decoder_active() and M >= BF16_MIN_ROWS and Cin % 32 == 0 and (x2 is None or C0 % 32 == 0):
        terms = 1                                  # bf16 decoder: plain bf16 operands (weights rounded once, cached)
    elif GEMM_MODE is not None and nt == 0 and M >= SPLIT_MIN_ROWS and Cin % 32 == 0:
        terms = 3 if GEMM_MODE == "bf16x3" else 6
    if terms and wt_split is None:
        wt_split = _cached_split(wt, terms)
    if GG_FORCE_NT is not None and nt == 0:
        sub = (Cout + 31) // 32
        eff = GG_FORCE_NT if GG_FORCE_NT > 0 else (-GG_FORCE_NT - 10 if GG_FORCE_NT <= -11 else 1)
        if sub % eff == 0:
            nt = GG_FORCE_NT
    hook = GG_HOOK
    if hook is not None:
        hook.before(dict(K=K, Cin=Cin, Cout=Cout, M=M, nbr=nbr))
    ws_ptr, ws_n = None, 0
    if nbr is not None and K >= 8 and M * Cout <= (1 << 22):      # small launch: allow split-K partials
        ws = _WS2.get(8 * M * Cout * 4, x.device)
        ws_ptr, ws_n = ws.data_ptr(), ws.numel()
    if terms:
        if wt_split.dtype != torch.bfloat16 or not wt_split.is_contiguous() or tuple(wt_split.shape[1:]) != (K, Cout, Cin):
            raise ValueError("wt_split must be contiguous bf16 [1|2|3, K, Cout, Cin]")
        _lib.check(lib.sd3d_gather_gemm_split(p0, ld0, C0, p1, ld1, _ptr(nbr, torch.int32, "nbr"), wt_split.data_ptr(),
                                              terms, K, Cin, Cout, M, _ptr(scale, torch.float32, "scale"),
                                              _ptr(shift, torch.float32, "shift"), pr, ldr, po, ldo, ACT[act],
                                              max(nt, 0), ws_ptr, ws_n, _stream()), "gather_gemm_split")
        if hook is not None:
            hook.after()
        return out
    _lib.check(lib.sd3d_gather_gemm(p0, ld0, C0, p1, ld1, _ptr(nbr, torch.int32, "nbr"), _ptr(wt, torch.float32, "wt"),
                                    K, Cin, Cout, M, _ptr(scale, torch.float32, "scale"),
                                    _ptr(shift, torch.float32, "shift"), pr, ldr, po, ldo, ACT[act], nt, ws_ptr, ws_n,
                                    _stream()), "gather_gemm")
    if hook is not None:
        hook.after()
    return out


def linear(x, weight, bias=None, act=None, res=None, out=None):
    """y = act(x @ weight.T + bias (+ res)); weight [Cout, Cin] as stored by nn.Linear."""
    return gather_gemm(x, weight, shift=bias, act=act, res=res, out=out)


# --------------------------------------------------------------------------------------------
# decoder ops
# --------------------------------------------------------------------------------------------
def layernorm(x, weight, bias, res=None, act=None, eps=1e-5, out=None):
    lib = _lib.load()
    px, ldx = _rows(x, "x")
    pr, ldr = (None, 0) if res is None else _rows(res, "res")
    if out is None:
        out = torch.empty(x.shape[0], x.shape[1], dtype=torch.float32, device=x.device)
    po, ldo = _rows(out, "out")
    _lib.check(lib.sd3d_layernorm(px, ldx, pr, ldr, _ptr(weight, torch.float32, "weight"), _ptr(bias, torch.float32, "bias"),
                                  float(eps), x.shape[0], x.shape[1], po, ldo, ACT[act], _stream()), "layernorm")
    return out


LINEAR_LN_MAX_ROWS = int(_os.environ.get("SD3D_LINEAR_LN_MAX_ROWS", "512"))      # 0 switches the fused launch off


def linear_layernorm(x, weight, bias, ln_weight, ln_bias, res=None, act=None, eps=1e-5, max_rows=None):
    """act(LayerNorm(x @ weight^T + bias + res) * ln_weight + ln_bias).  Few rows and a 256-wide output (the decoder's query tensors):
    one fused launch (sd3d_linear_layernorm); anything else - or an instrumented / split-precision run - is the projection followed
    by the LayerNorm kernel."""
    M, Cin = x.shape
    # (measured at 200 rows: Cin = 256 14.5 us fused vs 21.9 us in two launches; Cin = 1024 39.9 vs 22.2 - a 16-row workgroup walks
    # the whole contraction alone - so long contractions keep the two launches)
    # max_rows: the rows of several scenes of <= LINEAR_LN_MAX_ROWS rows each take the fused launch too (16-row workgroups:
    # per row the same arithmetic whatever M is - the batched decoder must not switch kernels with the batch size)
    if (M > (LINEAR_LN_MAX_ROWS if max_rows is None else max_rows) or Cin > 512 or weight.shape[0] != 256 or Cin % 16 or GG_HOOK is not None or GEMM_MODE is not None or GG_FORCE_NT is not None
            or not weight.is_contiguous() or (res is not None and res.stride(1) != 1)):
        return layernorm(gather_gemm(x, weight, shift=bias, res=res), ln_weight, ln_bias, act=act, eps=eps)
    lib = _lib.load()
    px, ldx = _rows(x, "x")
    pr, ldr = (None, 0) if res is None else _rows(res, "res")
    out = torch.empty(M, 256, dtype=torch.float32, device=x.device)
    _lib.check(lib.sd3d_linear_layernorm(px, ldx, M, Cin, _ptr(weight, torch.float32, "weight"), 256,
                                         None if bias is None else _ptr(bias, torch.float32, "bias"), pr, ldr,
                                         _ptr(ln_weight, torch.float32, "ln_weight"), _ptr(ln_bias, torch.float32, "ln_bias"), float(eps), ACT[act],
                                         out.data_ptr(), 256, _stream()), "linear_layernorm")
    return out


def sine_pe(xyz, rng, dim_t, axis, mod_num=None, mod_den=None, row_scene=None):
    """xyz [n,3]; rng [6] = (lo, hi); dim_t [d] fp32, axis [d] int8 -> [n, d].  row_scene int32 [n]: rows of several scenes,
    rng is then [n_scenes, 6] and row r uses rng[row_scene[r]] (sd3d_sine_pe_rows)."""
    lib = _lib.load()
    px, ldx = _rows(xyz, "xyz")
    n, d = xyz.shape[0], dim_t.numel()
    out = torch.empty(n, d, dtype=torch.float32, device=xyz.device)
    pn, ldn, pd, ldd = None, 0, None, 0
    if mod_num is not None:
        pn, ldn = _rows(mod_num, "mod_num")
        if mod_den.dim() == 1:
            pd, ldd = _ptr(mod_den, torch.float32, "mod_den"), 0
        else:
            pd, ldd = _rows(mod_den, "mod_den")
    if row_scene is not None:
        _lib.check(lib.sd3d_sine_pe_rows(px, ldx, n, _ptr(rng, torch.float32, "rng"), _ptr(row_scene, torch.int32, "row_scene"),
                                         _ptr(dim_t, torch.float32, "dim_t"), _ptr(axis, torch.int8, "axis"), d, pn, ldn, pd, ldd,
                                         _ptr(out), d, _stream()), "sine_pe_rows")
        return out
    _lib.check(lib.sd3d_sine_pe(px, ldx, n, _ptr(rng, torch.float32, "rng"), _ptr(dim_t, torch.float32, "dim_t"),
                                _ptr(axis, torch.int8, "axis"), d, pn, ldn, pd, ldd, _ptr(out), d, _stream()), "sine_pe")
    return out


def fourier_pe(xyz, rng, gauss_b, d_pos, row_scene=None):
    """xyz [n,3]; rng [6] = (lo, hi); gauss_b [3, >= d_pos / 2] fp32 -> [n, d_pos] = [sin | cos] (utils.py:107-142)."""
    lib = _lib.load()
    px, ldx = _rows(xyz, "xyz")
    pb, ldb = _rows(gauss_b, "gauss_b")
    n = xyz.shape[0]
    if gauss_b.shape[0] != 3 or gauss_b.shape[1] < d_pos // 2:
        raise ValueError("fourier_pe: gauss_b must be [3, >= d_pos / 2]")
    out = torch.empty(n, d_pos, dtype=torch.float32, device=xyz.device)
    if row_scene is not None:
        _lib.check(lib.sd3d_fourier_pe_rows(px, ldx, n, _ptr(rng, torch.float32, "rng"), _ptr(row_scene, torch.int32, "row_scene"), pb, ldb,
                                            d_pos, _ptr(out), d_pos, _stream()), "fourier_pe_rows")
        return out
    _lib.check(lib.sd3d_fourier_pe(px, ldx, n, _ptr(rng, torch.float32, "rng"), pb, ldb, d_pos, _ptr(out), d_pos, _stream()), "fourier_pe")
    return out


def attention(q, k, v, num_heads, scale, mask_bits=None, q2=None, k2=None, out=None):
    """q/k [L, H*32] (+ optional second source concatenated per head), v [Lk, H*32] -> [Lq, H*32] (`out`: rows to write into)."""
    lib = _lib.load()
    pq, ldq = _rows(q, "q")
    pk, ldk = _rows(k, "k")
    pv, ldv = _rows(v, "v")
    pq2, ldq2, pk2, ldk2 = None, 0, None, 0
    if q2 is not None:
        pq2, ldq2 = _rows(q2, "q2")
        pk2, ldk2 = _rows(k2, "k2")
    Lq, Lk = q.shape[0], k.shape[0]
    if q.shape[1] != num_heads * 32 or v.shape[1] != num_heads * 32:
        raise ValueError("attention: head slices must be 32 channels wide")
    if mask_bits is not None and tuple(mask_bits.shape) != (Lq, (Lk + 31) // 32):
        raise ValueError(f"attention: mask bits shape {tuple(mask_bits.shape)} != ({Lq}, {(Lk + 31) // 32})")
    if out is None:
        out = torch.empty(Lq, num_heads * 32, dtype=torch.float32, device=q.device)
    elif tuple(out.shape) != (Lq, num_heads * 32) or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError("attention: `out` must be a contiguous fp32 [Lq, H * 32] tensor")
    ws = _WS6.get(lib.sd3d_attention_ws_bytes(Lq, num_heads), q.device)
    fn = lib.sd3d_attention_bf16 if bf16_decoder_active() else lib.sd3d_attention
    _lib.check(fn(pq, ldq, pq2, ldq2, pk, ldk, pk2, ldk2, pv, ldv, _ptr(mask_bits, torch.int32, "mask_bits"),
                  Lq, Lk, num_heads, float(scale), _ptr(out), out.shape[1], ws.data_ptr(), ws.numel(), _stream()), "attention")
    return out


_ATTN_JOB_DT = None


def attention_batch(jobs, num_heads, scale):
    """jobs: list of (q, k, v, mask_bits | None, q2 | None, k2 | None, out) - the same attention for several scenes in ONE launch
    (sd3d_attention_batch): per scene the rows are the bits of `attention` on that scene alone."""
    global _ATTN_JOB_DT
    import numpy as np
    lib = _lib.load()
    if _ATTN_JOB_DT is None:
        _ATTN_JOB_DT = np.dtype([("q0", "<u8"), ("q1", "<u8"), ("k0", "<u8"), ("k1", "<u8"), ("v", "<u8"), ("bits", "<u8"), ("out", "<u8"),
                                 ("ldq0", "<i4"), ("ldq1", "<i4"), ("ldk0", "<i4"), ("ldk1", "<i4"), ("ldv", "<i4"), ("ldo", "<i4"),
                                 ("Lq", "<i4"), ("Lk", "<i4")], align=True)
        assert _ATTN_JOB_DT.itemsize == 88
    tab = np.zeros(len(jobs), dtype=_ATTN_JOB_DT)
    nb = 0
    for i, (q, k, v, bits, q2, k2, out) in enumerate(jobs):
        pq, ldq = _rows(q, "q")
        pk, ldk = _rows(k, "k")
        pv, ldv = _rows(v, "v")
        pq2 = pk2 = ldq2 = ldk2 = 0
        if q2 is not None:
            pq2, ldq2 = _rows(q2, "q2")
            pk2, ldk2 = _rows(k2, "k2")
        Lq, Lk = q.shape[0], k.shape[0]
        if q.shape[1] != num_heads * 32 or v.shape[1] != num_heads * 32:
            raise ValueError("attention_batch: head slices must be 32 channels wide")
        if bits is not None and tuple(bits.shape) != (Lq, (Lk + 31) // 32):
            raise ValueError("attention_batch: mask bits shape mismatch")
        if tuple(out.shape) != (Lq, num_heads * 32) or not out.is_contiguous() or out.dtype != torch.float32:
            raise ValueError("attention_batch: `out` must be a contiguous fp32 [Lq, H * 32] tensor")
        tab[i] = (pq, pq2, pk, pk2, pv, 0 if bits is None else _ptr(bits, torch.int32, "mask_bits"), out.data_ptr(), ldq, ldq2, ldk, ldk2, ldv,
                  out.stride(0), Lq, Lk)
        nb += lib.sd3d_attention_ws_bytes(Lq, num_heads)
    ws = _WS6.get(nb, jobs[0][0].device)
    _lib.check(lib.sd3d_attention_batch(len(jobs), tab.ctypes.data, num_heads, float(scale), 1 if bf16_decoder_active() else 0,
                                        ws.data_ptr(), ws.numel(), _stream()), "attention_batch")


def attention_parts(jobs, num_heads, scale):
    """`attention_batch` WITHOUT the pass that combines the key splits (sd3d_attention_batch_parts): returns (ws, [(ksplit, part_off)])
    - scene i's rows are final in its `out` where ksplit == 1, else its partial softmax states wait at ws (a uint8 tensor) + part_off
    floats for the consumer that combines them (rowchain MERGE).  jobs as in `attention_batch`."""
    global _ATTN_JOB_DT
    import ctypes as C
    import numpy as np
    lib = _lib.load()
    if _ATTN_JOB_DT is None:
        _ATTN_JOB_DT = np.dtype([("q0", "<u8"), ("q1", "<u8"), ("k0", "<u8"), ("k1", "<u8"), ("v", "<u8"), ("bits", "<u8"), ("out", "<u8"),
                                 ("ldq0", "<i4"), ("ldq1", "<i4"), ("ldk0", "<i4"), ("ldk1", "<i4"), ("ldv", "<i4"), ("ldo", "<i4"),
                                 ("Lq", "<i4"), ("Lk", "<i4")], align=True)
        assert _ATTN_JOB_DT.itemsize == 88
    n = len(jobs)
    tab = np.zeros(n, dtype=_ATTN_JOB_DT)
    nb = 0
    for i, (q, k, v, bits, q2, k2, out) in enumerate(jobs):
        pq, ldq = _rows(q, "q")
        pk, ldk = _rows(k, "k")
        pv, ldv = _rows(v, "v")
        pq2 = pk2 = ldq2 = ldk2 = 0
        if q2 is not None:
            pq2, ldq2 = _rows(q2, "q2")
            pk2, ldk2 = _rows(k2, "k2")
        Lq, Lk = q.shape[0], k.shape[0]
        if q.shape[1] != num_heads * 32 or v.shape[1] != num_heads * 32:
            raise ValueError("attention_parts: head slices must be 32 channels wide")
        if bits is not None and tuple(bits.shape) != (Lq, (Lk + 31) // 32):
            raise ValueError("attention_parts: mask bits shape mismatch")
        po, ldo = _rows(out, "out")
        if out.shape[0] != Lq or out.shape[1] != num_heads * 32:
            raise ValueError("attention_parts: `out` must be fp32 [Lq, H * 32] rows")
        tab[i] = (pq, pq2, pk, pk2, pv, 0 if bits is None else _ptr(bits, torch.int32, "mask_bits"), po, ldq, ldq2, ldk, ldk2, ldv, ldo, Lq, Lk)
        nb += lib.sd3d_attention_ws_bytes(Lq, num_heads)
    ws = _WS6.get(nb, jobs[0][0].device)                 # consumed by the next launch on this stream, before the next attention refills it
    ks = (C.c_int32 * n)()
    off = (C.c_int64 * n)()
    _lib.check(lib.sd3d_attention_batch_parts(n, tab.ctypes.data, num_heads, float(scale), 1 if bf16_decoder_active() else 0,
                                              ws.data_ptr(), ws.numel(), ks, off, _stream()), "attention_parts")
    return ws, [(int(ks[i]), int(off[i])) for i in range(n)]


def mask_bits_batch(logits_list, S_list, thr, out=None):
    """mask_bits for several scenes' logit matrices in one launch -> list of bit tensors (`out`: contiguous int32 [Q_i, ceil(S_i / 32)]
    tensors to write into, e.g. slices of one buffer)."""
    import ctypes as C
    lib = _lib.load()
    n = len(logits_list)
    given, out = out, []
    P, I, L = (C.c_void_p * n), (C.c_int * n), (C.c_int64 * n)
    lp, ld, Q, S, bp, nw = P(), I(), L(), I(), P(), I()
    for i, (lg, s) in enumerate(zip(logits_list, S_list)):
        lp[i], ld[i] = _rows(lg, "logits")
        Q[i], S[i] = lg.shape[0], int(s)
        nw[i] = (int(s) + 31) // 32
        if given is not None:
            bits = given[i]
            if tuple(bits.shape) != (lg.shape[0], nw[i]) or not bits.is_contiguous() or bits.dtype != torch.int32 or not bits.is_cuda:
                raise ValueError("mask_bits_batch: `out` entries must be contiguous int32 [Q, ceil(S / 32)] device tensors")
        else:
            bits = torch.empty(lg.shape[0], nw[i], dtype=torch.int32, device=lg.device)
        bp[i] = bits.data_ptr()
        out.append(bits)
    _lib.check(lib.sd3d_mask_bits_batch(n, lp, ld, Q, S, bp, nw, float(thr), _stream()), "mask_bits_batch")
    return out


def dinox_mask_bits_batch(blocked_list, near_list):
    """dinox_mask_bits for several scenes in one launch -> list of bit tensors."""
    import ctypes as C
    lib = _lib.load()
    n = len(blocked_list)
    P, I, L = (C.c_void_p * n), (C.c_int * n), (C.c_int64 * n)
    bl, nr, nw, Q, Mq, op, nwo = P(), P(), I(), L(), L(), P(), I()
    out = []
    for i, (b, nearb) in enumerate(zip(blocked_list, near_list)):
        bl[i], nr[i] = _ptr(b, torch.int32, "blocked"), _ptr(nearb, torch.int32, "near")
        Q[i], nw[i] = b.shape
        Mq[i] = nearb.shape[0]
        nwo[i] = (nearb.shape[0] + 1 + 31) // 32
        o = torch.empty(b.shape[0], nwo[i], dtype=torch.int32, device=b.device)
        op[i] = o.data_ptr()
        out.append(o)
    _lib.check(lib.sd3d_dinox_mask_bits_batch(n, bl, nr, nw, Q, Mq, op, nwo, _stream()), "dinox_mask_bits_batch")
    return out


def mask_bits(logits, S, thr):
    lib = _lib.load()
    pl, ld = _rows(logits, "logits")
    Q = logits.shape[0]
    nw = (S + 31) // 32
    bits = torch.empty(Q, nw, dtype=torch.int32, device=logits.device)
    _lib.check(lib.sd3d_mask_bits(pl, ld, Q, S, float(thr), _ptr(bits), nw, _stream()), "mask_bits")
    return bits


def near_bits(sp_pos, centers, thr):
    lib = _lib.load()
    S, M = sp_pos.shape[0], centers.shape[0]
    nw = (S + 31) // 32
    near = torch.empty(M, nw, dtype=torch.int32, device=sp_pos.device)
    _lib.check(lib.sd3d_near_bits(_ptr(sp_pos, torch.float32, "sp_pos"), S, _ptr(centers, torch.float32, "centers"), M,
                                  float(thr), _ptr(near), nw, _stream()), "near_bits")
    return near


def dinox_mask_bits(blocked, near):
    lib = _lib.load()
    Q, nw = blocked.shape
    M = near.shape[0]
    nwo = (M + 1 + 31) // 32
    out = torch.empty(Q, nwo, dtype=torch.int32, device=blocked.device)
    _lib.check(lib.sd3d_dinox_mask_bits(_ptr(blocked, torch.int32, "blocked"), _ptr(near, torch.int32, "near"), nw, Q, M,
                                        _ptr(out), nwo, _stream()), "dinox_mask_bits")
    return out


def box_refine(ref_points, d_center, size_prev, d_size, rng, normalize, row_scene=None):
    """row_scene int32 [Q]: rows of several scenes, rng [n_scenes, 6] (sd3d_box_refine_rows)."""
    lib = _lib.load()
    Q = ref_points.shape[0]
    dev = ref_points.device
    center = torch.empty(Q, 3, dtype=torch.float32, device=dev)
    size = size_metric = None
    ps, lds = None, 0
    if d_size is not None:
        size = torch.empty(Q, 3, dtype=torch.float32, device=dev)
        size_metric = torch.empty(Q, 3, dtype=torch.float32, device=dev)
        ps = _ptr(size_prev, torch.float32, "size_prev")
        lds = 0 if size_prev.dim() == 1 else 3
    if row_scene is not None:
        _lib.check(lib.sd3d_box_refine_rows(_ptr(ref_points, torch.float32, "ref_points"), _ptr(d_center, torch.float32, "d_center"),
                                            ps, lds, _ptr(d_size, torch.float32, "d_size"), _ptr(rng, torch.float32, "rng"),
                                            _ptr(row_scene, torch.int32, "row_scene"), int(normalize), Q, _ptr(center), _ptr(size),
                                            _ptr(size_metric), _stream()), "box_refine_rows")
        return center, size, size_metric
    _lib.check(lib.sd3d_box_refine(_ptr(ref_points, torch.float32, "ref_points"), _ptr(d_center, torch.float32, "d_center"),
                                   ps, lds, _ptr(d_size, torch.float32, "d_size"), _ptr(rng, torch.float32, "rng"),
                                   int(normalize), Q, _ptr(center), _ptr(size), _ptr(size_metric), _stream()), "box_refine")
    return center, size, size_metric


# --------------------------------------------------------------------------------------------
# post-processing ops
# --------------------------------------------------------------------------------------------
def class_scores(cls, C, want_scores=True, want_rowmax=False):
    lib = _lib.load()
    pc, ld = _rows(cls, "cls")
    Q = cls.shape[0]
    scores = torch.empty(Q * C, dtype=torch.float32, device=cls.device) if want_scores else None
    rowmax = torch.empty(Q, dtype=torch.float32, device=cls.device) if want_rowmax else None
    _lib.check(lib.sd3d_class_scores(pc, ld, Q, C, _ptr(scores), _ptr(rowmax), _stream()), "class_scores")
    return scores, rowmax


def mask_scores(masks, S, flat_idx, score_in, C, normalize):
    lib = _lib.load()
    pm, ld = _rows(masks, "masks")
    n = flat_idx.numel()
    dev = masks.device
    labels = torch.empty(n, dtype=torch.int32, device=dev)
    qidx = torch.empty(n, dtype=torch.int32, device=dev)
    out = torch.empty(n, dtype=torch.float32, device=dev)
    _lib.check(lib.sd3d_mask_scores(pm, ld, S, _ptr(flat_idx, torch.int32, "flat_idx"), _ptr(score_in, torch.float32, "score_in"),
                                    n, C, int(bool(normalize)), _ptr(labels), _ptr(qidx), _ptr(out), _stream()), "mask_scores")
    return labels, qidx, out


TOPK_SELECT_MAX_N = 40 * 1024    # what the one workgroup of the select holds in registers; above it: the radix sort


def topk_desc(x: torch.Tensor, k: int):
    """int32 [k]: the first k indices of the stable descending sort of the fp32 vector x (`sd3d_topk_desc_f32`: one launch)."""
    lib = _lib.load()
    out = torch.empty(k, dtype=torch.int32, device=x.device)
    _lib.check(lib.sd3d_topk_desc_f32(_ptr(x, torch.float32, "x"), x.numel(), int(k), _ptr(out), _stream()), "topk_desc")
    return out


def take_f32(src, idx):
    """src[idx] for an int32 index vector, one launch (ATen: `.long()` + index)."""
    lib = _lib.load()
    n = idx.numel()
    out = torch.empty(n, dtype=torch.float32, device=src.device)
    _lib.check(lib.sd3d_take_f32(_ptr(src, torch.float32, "src"), _ptr(idx, torch.int32, "idx"), n, _ptr(out), _stream()), "take_f32")
    return out


def select_instances(scores, count, thr0, thr1, npoint_thr):
    """Row selections for the two score thresholds on the device (`sd3d_select_instances`).  Returns (buffers, counts): `buffers` =
    (keep, pkeep, union, keep_u, pkeep_u int32 [k]; score_mask, npoint_mask uint8 [k]), `counts` int32 [4] for one small host read."""
    lib = _lib.load()
    k = scores.numel()
    dev = scores.device
    ints = torch.empty(5, max(k, 1), dtype=torch.int32, device=dev)
    bytes_ = torch.empty(2, max(k, 1), dtype=torch.uint8, device=dev)
    counts = torch.empty(4, dtype=torch.int32, device=dev)
    ip, bp = ints.data_ptr(), bytes_.data_ptr()
    st = 4 * max(k, 1)
    _lib.check(lib.sd3d_select_instances(_ptr(scores, torch.float32, "scores"), _ptr(count, torch.int32, "count"), k, float(thr0), float(thr1),
                                         int(npoint_thr), ip, ip + st, ip + 2 * st, ip + 3 * st, ip + 4 * st, bp, bp + max(k, 1),
                                         _ptr(counts), _stream()), "select_instances")
    return (ints, bytes_), counts


def take_instances(keep, labels, scores, boxes=None):
    """(labels[keep] as int64, scores[keep], boxes[keep] | None) in one launch."""
    lib = _lib.load()
    m = keep.numel()
    dev = keep.device
    lo = torch.empty(m, dtype=torch.int64, device=dev)
    so = torch.empty(m, dtype=torch.float32, device=dev)
    bo = torch.empty(m, 6, dtype=torch.float32, device=dev) if boxes is not None else None
    _lib.check(lib.sd3d_take_instances(_ptr(keep, torch.int32, "keep"), m, _ptr(labels, torch.int32, "labels"), _ptr(scores, torch.float32, "scores"),
                                       _ptr(boxes, torch.float32, "boxes"), _ptr(lo), _ptr(so), _ptr(bo), _stream()), "take_instances")
    return lo, so, bo


def take_pair(order, labels, scores):
    """(labels[order], scores[order]) in one launch."""
    lib = _lib.load()
    n = order.numel()
    lo = torch.empty(n, dtype=torch.int32, device=order.device)
    so = torch.empty(n, dtype=torch.float32, device=order.device)
    _lib.check(lib.sd3d_take_pair(_ptr(order, torch.int32, "order"), _ptr(labels, torch.int32, "labels"), _ptr(scores, torch.float32, "scores"), n,
                                  _ptr(lo), _ptr(so), _stream()), "take_pair")
    return lo, so


def nms_finish(order2, scores2, labels1, order1, qidx, centers=None, sizes=None):
    """-> (final_scores, final_labels int32, record int64, boxes [n, 6] | None): the selections behind matrix-NMS's final sort in one launch."""
    lib = _lib.load()
    n = order2.numel()
    dev = order2.device
    fs = torch.empty(n, dtype=torch.float32, device=dev)
    fl = torch.empty(n, dtype=torch.int32, device=dev)
    rec = torch.empty(n, dtype=torch.int64, device=dev)
    boxes = torch.empty(n, 6, dtype=torch.float32, device=dev) if centers is not None and sizes is not None else None
    _lib.check(lib.sd3d_nms_finish(_ptr(order2, torch.int32, "order2"), _ptr(scores2, torch.float32, "scores2"), _ptr(labels1, torch.int32, "labels1"),
                                   _ptr(order1, torch.int32, "order1"), _ptr(qidx, torch.int32, "qidx"),
                                   _ptr(centers, torch.float32, "centers") if boxes is not None else None,
                                   _ptr(sizes, torch.float32, "sizes") if boxes is not None else None, n, _ptr(fs), _ptr(fl), _ptr(rec), _ptr(boxes),
                                   _stream()), "nms_finish")
    return fs, fl, rec, boxes


def gather_sigmoid(masks, S, qidx, order, ld_out):
    lib = _lib.load()
    pm, ld = _rows(masks, "masks")
    n = order.numel()
    sig = torch.empty(n, ld_out, dtype=torch.float32, device=masks.device)
    area = torch.empty(n, dtype=torch.float32, device=masks.device)
    _lib.check(lib.sd3d_gather_sigmoid(pm, ld, S, _ptr(qidx, torch.int32, "qidx"), _ptr(order, torch.int32, "order"), n,
                                       _ptr(sig), ld_out, _ptr(area), _stream()), "gather_sigmoid")
    return sig, area


def nms_decay(inter, area, labels, score_in, kernel="linear", sigma=2.0):
    lib = _lib.load()
    pi, ld = _rows(inter, "inter")
    n = area.numel()
    comp = torch.empty(n, dtype=torch.float32, device=inter.device)
    out = torch.empty(n, dtype=torch.float32, device=inter.device)
    if kernel not in ("linear", "gaussian"):
        raise NotImplementedError(f"{kernel} kernel is not supported in matrix nms!")
    _lib.check(lib.sd3d_nms_decay(pi, ld, _ptr(area, torch.float32, "area"), _ptr(labels, torch.int32, "labels"), n,
                                  int(kernel == "gaussian"), float(sigma), _ptr(score_in, torch.float32, "score_in"),
                                  _ptr(comp), _ptr(out), _stream()), "nms_decay")
    return out


class MaskBits:
    """Step 1 of the point-mask expansion (`sd3d_mask_rowbits`): the thresholded rows as a bit table over the superpoints + the point
    count of every row.  `rows(list)` expands a list of rows to [m, N] bytes (`sd3d_expand_rows`): only the instances that survive."""

    def __init__(self, sig, src_row, superpoints, points, sp_thr, boxes=None, loose_ratio=1.5):
        lib = _lib.load()
        ps, lds = _rows(sig, "sig")
        self.n, self.N, self.lds = src_row.numel(), superpoints.numel(), lds
        self.superpoints, self.points, self.boxes, self.loose = superpoints, points, boxes, float(loose_ratio)
        # (a tensor of its own, not the per-stream workspace: the table must survive until the rows are expanded, after the host read)
        self.ws = torch.empty(lib.sd3d_expand_masks_ws_bytes(self.n, lds), dtype=torch.uint8, device=sig.device)
        # the point counts sit right behind the table's superpoint sizes: the library zeroes both with one memset launch
        c0 = 4 * lds * ((self.n + 31) // 32 + 1)
        self.count = self.ws[c0:c0 + 4 * self.n].view(torch.int32)
        _lib.check(lib.sd3d_mask_rowbits(ps, lds, _ptr(src_row, torch.int32, "src_row"), self.n, _ptr(superpoints, torch.int64, "superpoints"),
                                         self.N, float(sp_thr), _ptr(self.count), self.ws.data_ptr(), self.ws.numel(), _stream()), "mask_rowbits")

    def rows(self, rows):
        lib = _lib.load()
        m = rows.numel()
        out = torch.empty(m, self.N, dtype=torch.uint8, device=self.ws.device)
        if m:
            pp, ldp = _rows(self.points, "points")
            _lib.check(lib.sd3d_expand_rows(self.ws.data_ptr(), self.n, self.lds, _ptr(rows, torch.int32, "rows"), m,
                                            _ptr(self.superpoints, torch.int64, "superpoints"), pp, ldp, self.N,
                                            _ptr(self.boxes, torch.float32, "boxes"), self.loose, _ptr(out), _stream()), "expand_rows")
        return out


def expand_masks(sig, src_row, superpoints, points, sp_thr, boxes=None, loose_ratio=1.5):
    lib = _lib.load()
    ps, lds = _rows(sig, "sig")
    pp, ldp = _rows(points, "points")
    n, N = src_row.numel(), superpoints.numel()
    out = torch.empty(n, N, dtype=torch.uint8, device=sig.device)
    count = torch.empty(n, dtype=torch.int32, device=sig.device)
    ws = _WS5.get(lib.sd3d_expand_masks_ws_bytes(n, lds), sig.device)
    _lib.check(lib.sd3d_expand_masks(ps, lds, _ptr(src_row, torch.int32, "src_row"), n, _ptr(superpoints, torch.int64, "superpoints"),
                                     pp, ldp, N, float(sp_thr), _ptr(boxes, torch.float32, "boxes"), float(loose_ratio),
                                     _ptr(out), _ptr(count), ws.data_ptr(), ws.numel(), _stream()), "expand_masks")
    return out, count


def pack_mask_rows(masks_u8, rows=None):
    """masks_u8 [n, N] uint8 (0 / 1) -> bit-packed uint8 [len(rows) | n, ceil(N / 8)] of the selected rows (int32 `rows`, or all):
    bit j of byte b = masks[row][8 b + j] (numpy bitorder "little").  The device half of the host-output path."""
    lib = _lib.load()
    n, N = masks_u8.shape
    n_rows = n if rows is None else rows.numel()
    nb = (N + 7) // 8
    out = torch.empty(n_rows, nb, dtype=torch.uint8, device=masks_u8.device)
    _lib.check(lib.sd3d_pack_mask_rows(_ptr(masks_u8, torch.uint8, "masks"), N, _ptr(rows, torch.int32, "rows"), n_rows, _ptr(out), nb,
                                       _stream()), "pack_mask_rows")
    return out


class _HostPool:
    """Recycles the big host arrays of the unpacked instance masks: a fresh 90 MB numpy array per scene is 22 k page faults (20 - 40 ms
    on the evaluation hosts, more than the forward itself); an array handed out here returns its memory to the pool when the LAST
    reference to it (or to a view of it) dies.  A caller that keeps every scene's masks (the reference's evaluator until
    `evaluate()`) simply never gives anything back - then this is `np.empty`."""

    class _Lease:
        __slots__ = ("buf", "pool", "__array_interface__")

        def __init__(self, buf, pool, shape):
            self.buf, self.pool = buf, pool
            self.__array_interface__ = {"shape": shape, "typestr": "|b1", "data": (buf.ctypes.data, False), "version": 3}

        def __del__(self):
            try:
                self.pool._give_back(self.buf)
            except Exception:  # noqa: BLE001 - interpreter shutdown
                pass

    def __init__(self, max_free_bytes=None):
        # SD3D_HOST_POOL_BYTES: most freed host bytes the pool keeps for the life of the process (default 512 MiB = five 90 MB mask
        # arrays, one per scene in flight plus one; 0 switches the recycling off)
        if max_free_bytes is None:
            max_free_bytes = int(_os.environ.get("SD3D_HOST_POOL_BYTES", str(512 << 20)))
        self.free, self.lock, self.max_free = [], threading.Lock(), max_free_bytes

    def _give_back(self, buf):
        with self.lock:
            if sum(b.nbytes for b in self.free) + buf.nbytes <= self.max_free:
                self.free.append(buf)

    def bool_array(self, shape):
        import numpy as np
        need = int(np.prod(shape))
        if need < (1 << 20):                                     # small arrays: the allocator's own free lists do this already
            return np.empty(shape, dtype=np.bool_)
        buf = None
        with self.lock:
            fit = [i for i, b in enumerate(self.free) if need <= b.nbytes <= need + need // 4 + (4 << 20)]
            if fit:
                buf = self.free.pop(min(fit, key=lambda i: self.free[i].nbytes))
        if buf is None:
            buf = np.empty((need + (4 << 20) - 1) // (4 << 20) * (4 << 20), dtype=np.uint8)
        return np.asarray(self._Lease(buf, self, tuple(int(v) for v in shape)))


_HOST_POOL = _HostPool()


def unpack_bits_host(packed, n_points: int):
    """HOST arrays: packed uint8 [n, ceil(N / 8)] (numpy) -> bool [n, N] (pageable; large ones come from a recycling pool, `_HostPool`).
    Runs in the C library with the GIL released (sd3d_unpack_bits_host): the other scene threads keep issuing while this one expands
    its masks."""
    import numpy as np
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    n, nb = packed.shape
    if nb != (n_points + 7) // 8:
        raise ValueError(f"unpack_bits_host: {nb} bytes per row cannot hold {n_points} points")
    out = _HOST_POOL.bool_array((n, n_points))
    _lib.check(_lib.load_nogil().sd3d_unpack_bits_host(packed.ctypes.data, n, n_points, nb, out.ctypes.data), "unpack_bits_host")
    return out


class PinnedStaging:
    """One grow-only pinned host buffer per (host thread): the staging area of a forward's device -> host copies.  The outputs
    handed to the caller are pageable copies made from it, so nothing page-locked outlives the forward (a long evaluation keeps
    every scene's outputs until `evaluate()`: ~100 MB of pinned memory per scene otherwise)."""

    def __init__(self):
        self._tls = threading.local()

    def get(self, nbytes: int):
        buf = getattr(self._tls, "buf", None)
        if buf is None or buf.numel() < nbytes:
            buf = self._tls.buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, pin_memory=True)
        return buf


_STAGING = PinnedStaging()


def to_host_arrays(tensors):
    """Device tensors -> pageable numpy arrays through ONE pinned staging buffer: asynchronous copies, one polled wait (the issue
    baton goes to another scene's thread meanwhile), then host copies out of the staging area."""
    import numpy as np
    offs, total = [], 0
    for t in tensors:
        total = (total + 63) // 64 * 64
        offs.append(total)
        total += t.numel() * t.element_size()
    stage = _STAGING.get(total)
    views = []
    for t, o in zip(tensors, offs):
        nb = t.numel() * t.element_size()
        v = stage[o:o + nb].view(t.dtype).view(t.shape) if nb else torch.empty(t.shape, dtype=t.dtype)
        if nb:
            v.copy_(t.contiguous(), non_blocking=True)
        views.append(v)
    wait_event(stream_event())
    return [np.array(v.numpy(), copy=True) for v in views]


def row_argmax(x, ncols=None, cols=None):
    lib = _lib.load()
    px, ld = _rows(x, "x")
    Q = x.shape[0]
    out = torch.empty(Q, dtype=torch.int64, device=x.device)
    nc = cols.numel() if cols is not None else (ncols if ncols is not None else x.shape[1])
    _lib.check(lib.sd3d_row_argmax(px, ld, Q, _ptr(cols, torch.int32, "cols"), nc, _ptr(out), _stream()), "row_argmax")
    return out


def gather_i64(table, idx, use_index=True):
    lib = _lib.load()
    N = idx.numel()
    out = torch.empty(N, dtype=torch.int64, device=idx.device)
    _lib.check(lib.sd3d_gather_i64(_ptr(table, torch.int64, "table"), _ptr(idx, torch.int64, "idx"), N, int(use_index),
                                   _ptr(out), _stream()), "gather_i64")
    return out


def panoptic(masks_u8, rows_desc, labels_desc, n_stuff, npoint_thr, sem_stuff):
    lib = _lib.load()
    N = masks_u8.shape[1]
    n = rows_desc.numel()
    dev = masks_u8.device
    inst_ws = torch.empty(N, dtype=torch.int32, device=dev)
    hist = torch.empty(n + n_stuff + 1, dtype=torch.int32, device=dev)
    sem_map = torch.empty(N, dtype=torch.int64, device=dev)
    inst_map = torch.empty(N, dtype=torch.int64, device=dev)
    _lib.check(lib.sd3d_panoptic(_ptr(masks_u8, torch.uint8, "masks"), N, _ptr(rows_desc, torch.int32, "rows"),
                                 _ptr(labels_desc, torch.int32, "labels"), n, n_stuff, int(npoint_thr),
                                 _ptr(sem_stuff, torch.int64, "sem_stuff"), _ptr(inst_ws), _ptr(hist), _ptr(sem_map),
                                 _ptr(inst_map), _stream()), "panoptic")
    return sem_map, inst_map


def instance_boxes(points, masks_bool, mode):
    """masks_bool [n_inst, N] torch.bool -> (centers [n,3], sizes [n,3])."""
    lib = _lib.load()
    pp, ld = _rows(points, "points")
    n, N = masks_bool.shape
    m = masks_bool.contiguous().view(torch.uint8)
    centers = torch.zeros(n, 3, dtype=torch.float32, device=points.device)
    sizes = torch.zeros(n, 3, dtype=torch.float32, device=points.device)
    ws = _WS.get(lib.sd3d_instance_boxes_ws_bytes(n), points.device)
    _lib.check(lib.sd3d_instance_boxes(pp, ld, N, _ptr(m, torch.uint8, "masks"), N, n, 0 if mode == "mean" else 1,
                                       _ptr(centers), _ptr(sizes), ws.data_ptr(), ws.numel(), _stream()), "instance_boxes")
    return centers, sizes


def scale_shift_act(x, scale, shift, act=None, x2=None, add=None):
    """act(cat[x, x2] * scale + shift) (+ add, AFTER the activation) -> new contiguous [M, C] tensor."""
    lib = _lib.load()
    p0, ld0 = _rows(x, "x")
    C0 = x.shape[1]
    p1, ld1, C = None, 0, C0
    if x2 is not None:
        p1, ld1 = _rows(x2, "x2")
        C = C0 + x2.shape[1]
    pa, lda = (None, 0) if add is None else _rows(add, "add")
    if add is not None and tuple(add.shape) != (x.shape[0], C):
        raise ValueError(f"scale_shift_act: add must be [{x.shape[0]}, {C}], got {tuple(add.shape)}")
    out = torch.empty(x.shape[0], C, dtype=torch.float32, device=x.device)
    _lib.check(lib.sd3d_scale_shift_act_add(p0, ld0, C0, p1, ld1, _ptr(scale, torch.float32, "scale"),
                                            _ptr(shift, torch.float32, "shift"), ACT[act], x.shape[0], C, pa, lda, _ptr(out), C,
                                            _stream()), "scale_shift_act")
    return out
