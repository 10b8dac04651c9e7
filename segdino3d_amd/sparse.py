"""Device-side sparse-voxel coordinate hierarchy for one scene (host orchestration of csrc/voxel.hip).

Plays the role MinkowskiEngine's CoordinateManager / spconv's indice-pair cache play for the
reference (`minkunet.py:624-631`, `spconvunet.py:283-315, 383-388`): voxelise the points once, derive
the coarser tensor strides, and hand out cached neighbour tables (`nbr[k, v]`) for each
(level, kernel) the network asks for.

HBM layout per scene (V_l voxels at level l, Z-order sorted, so a voxel's children/parents and
spatial neighbours are close in memory):
    keys_l   int64 [V_l]        Z-order key of (coord - origin) >> l  (+ batch bits)
    hash_l   int64/int32 [cap]  open-addressing table key -> voxel id (cap = pow2 >= 2 V_l)
    nbr      int32 [K, V_out]   gather table of one convolution kernel (K-major: coalesced per offset)
    parent_l int32 [V_l]        id of the parent voxel at level l+1
    inverse  int32 [N]          point -> level-0 voxel id
    sidx/seg int32 [N], [V_0+1] points sorted by voxel + segment starts (ascending point order inside)
Exactly one host synchronisation per scene: the voxel counts of all levels (needed to size the
launches) are read back together with the overflow flag and the superpoint count.
"""
from __future__ import annotations

from typing import Sequence, Dict, Optional, Tuple

import os

import numpy as np
import torch

from . import ops

_OFFSET_CACHE: Dict[Tuple[int, str, str], torch.Tensor] = {}


def kernel_offsets_np(ksize: int, order: str) -> np.ndarray:
    """[K,3] offsets in units of the tensor stride.  Odd kernels are centred, even kernels use
    {0..k-1}.  order 'x_fastest' = MinkowskiEngine enumeration, 'z_fastest' = spconv weight layout
    [k0,k1,k2] row-major.  (Same table as oracle.sparse_ref.kernel_offsets; tests compare them.)"""
    r = np.arange(ksize) - (ksize - 1) // 2 if ksize % 2 else np.arange(ksize)
    if order == "x_fastest":
        z, y, x = np.meshgrid(r, r, r, indexing="ij")
    elif order == "z_fastest":
        x, y, z = np.meshgrid(r, r, r, indexing="ij")
    else:
        raise ValueError(order)
    return np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1).astype(np.int8)


def offsets_device(ksize: int, order: str, device) -> torch.Tensor:
    key = (ksize, order, str(device))
    if key not in _OFFSET_CACHE:
        _OFFSET_CACHE[key] = torch.from_numpy(kernel_offsets_np(ksize, order)).to(device)
    return _OFFSET_CACHE[key]


def perm8_device(order: str, device) -> torch.Tensor:
    """`child_perm(order)` on the device, once per (order, device) - not one pageable, i.e. blocking, host-to-device copy per scene"""
    key = ("perm8", order, str(device))
    if key not in _OFFSET_CACHE:
        _OFFSET_CACHE[key] = torch.from_numpy(child_perm(order)).to(device)
    return _OFFSET_CACHE[key]


def child_perm(order: str) -> np.ndarray:
    """Z-order child position (x | y<<1 | z<<2) -> weight index of a 2x2x2 kernel."""
    if order == "x_fastest":
        return np.arange(8, dtype=np.int32)
    p = np.zeros(8, dtype=np.int32)
    for c in range(8):
        x, y, z = c & 1, (c >> 1) & 1, (c >> 2) & 1
        p[c] = (x * 2 + y) * 2 + z
    return p


# SD3D_EXACT_PAIRS=0: do not read the rulebook sizes back (saves the second host synchronisation of a scene) and size
# the pair lists for the worst case instead (K * V entries: 2.2 GB of partial-product scratch per stream at 150 k
# points).  Measured the same scenes/s either way once the GPU is the limiter, so the exact sizes are the default.
# "auto" (default): ONE scene in flight on this GPU (single-scene latency: the synchronisation is 0.24 ms of a 12.4 ms forward) and at most
# WORST_CASE_MAX_VOXELS voxels skips the read-back; several scenes in flight (the other scenes' kernels fill the wait, and the scratch is
# per stream) keep the exact sizes.  The rulebooks, and therefore every output bit, are the same either way (tests/test_gpu_sparse.py).
_ep = os.environ.get("SD3D_EXACT_PAIRS", "auto")
EXACT_PAIR_CAPACITY = "auto" if _ep == "auto" else (_ep == "1")
WORST_CASE_MAX_VOXELS = int(os.environ.get("SD3D_WORST_CASE_MAX_VOXELS", "250000"))


# Footprint of the worst-case sizing (grow-only scratch per (thread, stream)): the 5^3 stem's partial products are 125 x V x 32 x 4 B =
# 16 KB per level-0 voxel (2.2 GB at 138 k voxels, 4 GB at 250 k) plus ~0.1 GB of list entries.  "auto" takes it only while that stays
# under WORST_CASE_MEM_FRACTION of the device's memory (2 %: 5.8 GB of an MI355X's 288 GB - the voxel limit above is the tighter one
# there; on a 64 GB device scenes above 80 k voxels keep the exact sizes and the second read-back).
WORST_CASE_MEM_FRACTION = float(os.environ.get("SD3D_WORST_CASE_MEM_FRACTION", "0.02"))
_TOTAL_MEM = {}


def _device_memory(device) -> int:
    idx = torch.cuda.current_device() if device is None or device.index is None else device.index
    if idx not in _TOTAL_MEM:
        _TOTAL_MEM[idx] = int(torch.cuda.get_device_properties(idx).total_memory)
    return _TOTAL_MEM[idx]


def exact_pair_capacity(n_vox0: int, device=None) -> bool:
    if EXACT_PAIR_CAPACITY == "auto":
        return (ops.scenes_in_flight_now() > 1 or n_vox0 > WORST_CASE_MAX_VOXELS
                or 125 * 32 * 4 * n_vox0 > WORST_CASE_MEM_FRACTION * _device_memory(device))
    return bool(EXACT_PAIR_CAPACITY)


# SD3D_HIER_MAPS=0: kernel maps by hash-table probes (rounds 1-4: one table + one probe launch per level).  Default: every 3^3 map of a
# scene and the stem's 5^3 map from ONE call through the level hierarchy of the sorted keys (csrc/voxel.hip kmap_hier_kernel: two
# cache-friendly loads per probe, no hash tables, no memsets) whenever the levels came from `sd3d_unique_levels` (MinkowskiEngine
# semantics: every parent exists).  Entry for entry the same tables.
HIER_MAPS = os.environ.get("SD3D_HIER_MAPS", "1") != "0"
_INV27 = {}


def inv27_table(order: str) -> np.ndarray:
    """(dx + 1) + 3 (dy + 1) + 9 (dz + 1) -> row of the 3^3 offset table in the weights' enumeration order"""
    if order not in _INV27:
        offs = kernel_offsets_np(3, order).astype(np.int64)
        inv = np.zeros(27, dtype=np.int8)
        for k, (dx, dy, dz) in enumerate(offs):
            inv[(dx + 1) + 3 * (dy + 1) + 9 * (dz + 1)] = k
        _INV27[order] = inv
    return _INV27[order]


# kernel offsets are enumerated symmetrically (kernel_offsets_np: off[K-1-k] == -off[k] for odd k), so a stride-1 table
# needs only half its hash probes.  SD3D_MIRRORED_MAPS=0 probes every offset (cross-check).
MIRRORED_MAPS = os.environ.get("SD3D_MIRRORED_MAPS", "1") != "0"

# SD3D_PAIR_CHAIN=0: plain offset-major lists for the 3^3 tables too.  Default: CHAINED lists in evaluation (csrc/pair_gemm.hip: the
# entries of an output row's mirror offsets {k, K-1-k} and its centre share ONE partial product - 27-44 % fewer partial rows written
# by pass 1 and read by pass 2).  Training keeps the plain lists (the weight-gradient kernels walk them offset by offset).
PAIR_CHAIN = os.environ.get("SD3D_PAIR_CHAIN", "1") != "0"
# Which U-Net levels get chained lists is a property of the LEVEL, never of the scene or the batch (a scene's rows must see the same
# list format whether it runs alone or in a batch: the two formats sum in different orders).  Levels 0-2 (the tables with thousands
# of tiles): measured same box, 4 streams x 4 scenes 118.1 -> 120.5-123 scenes/s, convolutions 7.29 -> 7.06 ms per scene in the
# instrumented replay (0.471 -> 0.486 of the fp32 matrix peak), one scene in flight neutral (7.78 vs 7.76 ms: what the fewer partial
# rows win, the weight-stationary pass-1 variants - which cannot chain - lose).  Levels 3-4 have few hundred to ~1800 tiles: chains
# of three tiles on 3-4 tiles per workgroup unbalance the static ranges and the level-3 layers lose their weight-stationary variant:
# all five levels 124.6 scenes/s with several scenes in flight but 8.27 vs 7.74 ms of convolutions with one.
# SD3D_FORK_JOIN=0: every neighbour table of a scene is built on the scene's own stream before the first convolution.  Default:
# with ONE scene in flight (worst-case list sizes, no second read-back) only the stem's table is; the other levels' hash tables,
# kernel maps and pair lists are built on a side stream while the stem and the first blocks convolve, each table's first layer
# waits for its event (sd3d_run_layers_ev).  Same kernels on the same data: bit-identical outputs.
# Round 5: OFF by default.  The fork bought 0.2 ms when a scene's maps and lists cost 0.96 ms of kernels; alone on the GPU they now take 0.24 ms
# (hierarchical maps, row-block list builders, lean tables), and three cross-stream hand-offs cost the host about what the overlap still
# wins: same-process A/B 10.565 - 10.570 ms without the fork against 10.582 - 10.584 with it (profiles/EXPERIMENTS.md).  SD3D_FORK_JOIN=1 restores it.
FORK_JOIN = os.environ.get("SD3D_FORK_JOIN", "0") != "0"
# SD3D_OPTIMISTIC_SORT=0: a scene's voxel keys / superpoint ids are always sorted over all their bits (7 + 4 radix passes instead of 4 + 2)
OPTIMISTIC_SORT = os.environ.get("SD3D_OPTIMISTIC_SORT", "1") != "0"
# SD3D_LEVELS_AT_ONCE=0: the coarser levels of a scene by one run-length unique per level (four launches each) instead of all of them
# from the level-0 keys in four launches (`sd3d_unique_levels`; MinkowskiEngine semantics only - spconv's extent clip keeps the per-level path)
LEVELS_AT_ONCE = os.environ.get("SD3D_LEVELS_AT_ONCE", "1") != "0"
# SD3D_LEAN_LISTS=0: evaluation tables with their [K, M] position table and -1-filled unused capacity (rounds 1 - 4)
LEAN_LISTS = os.environ.get("SD3D_LEAN_LISTS", "1") != "0"
MORTON_BITS = 48               # SD3D_MORTON_BITS (csrc/common.h): the Z-order code of a voxel; a batch's scene index sits above it
# SD3D_VOXELISE_ONE_CALL=0: the voxelisation chain of a scene as ~10 Python calls (the round 1 - 4 path; the clipped / per-level variants keep it)
VOXELISE_ONE_CALL = os.environ.get("SD3D_VOXELISE_ONE_CALL", "1") != "0"
PAIR_CHAIN_LEVELS = tuple(int(v) for v in os.environ.get("SD3D_PAIR_CHAIN_LEVELS", "0,1,2").split(",") if v.strip() != "")


class SceneMaps:
    """Voxelisation + coordinate levels + neighbour tables of ONE scene, all on the HIP device."""

    def __init__(self, points: torch.Tensor, voxel_size: float, n_levels: int, shift_to_min: bool = False,
                 order: str = "x_fastest", superpoints: Optional[torch.Tensor] = None, clip_min_shape: int = 0, while_waiting=None):
        """clip_min_shape > 0 enables spconv's output-extent rule for the strided levels (needs
        shift_to_min coordinates); 0 = MinkowskiEngine semantics (every parent voxel exists).
        while_waiting(self): called once between issuing the scene's read-back and waiting for it - host work that needs no size of the
        scene (the validity check of the derived weights) costs nothing there, the host would sleep."""
        if not points.is_cuda:
            raise RuntimeError("SceneMaps needs device-resident points (no CPU fallback in the product path)")
        self.order = order
        self.device = points.device
        self.voxel_size = float(voxel_size)
        self.clipped = clip_min_shape > 0
        N = points.shape[0]
        self.n_points = N
        inv = float(np.float32(1.0) / np.float32(voxel_size))
        self.superpoints = superpoints                    # int64 id per point (kept for the pooling backward, train_ops)
        # Optimistic radix passes: the voxel keys are sorted over their low 32 bits (4 passes instead of 7: any scene up to ~20 m at 2 cm)
        # and the superpoint ids over 16 (2 instead of 4); the key kernels raise a flag when a key needs more, the flag rides in the
        # scene's read-back, and the chain is then redone with the full sorts (tests/test_gpu_sparse.py exercises both).
        key_bits, sp_bits = (32, 16) if OPTIMISTIC_SORT else (56, 32)
        at_once = LEVELS_AT_ONCE and clip_min_shape == 0 and 1 < n_levels <= 8
        while True:
            if at_once and VOXELISE_ONE_CALL:
                # the whole chain below from one C call (same kernels, same order): ~28 launches of 4 - 13 us that a Python host issues
                # slower than the GPU runs them, in front of everything else of the scene
                v = ops.voxelise_scene(points, inv, shift_to_min, key_bits, n_levels, superpoints,
                                       (sp_bits if sp_bits < 32 else 64) if superpoints is not None else 64)
                self.stats, self.icoords, self.origin, self.sidx = v["stats"], v["icoords"], v["origin"], v["sidx"]
                self.seg_start, self.inverse, keys_l, parents = v["seg_start"], v["inverse"], v["ukeys"], v["parents"]
                self._rb_dev = v["readback"]
                read = ops.HostRead(v["readback"])           # synchronisation 1 of the scene (polled)
                if superpoints is not None:                  # sorted while the read-back travels
                    self.sp_sorted, self.sp_sidx = ops.sort_pairs(v["sp_keys"], None, 0, sp_bits)
                if while_waiting is not None:
                    while_waiting(self)
                    while_waiting = None
                host = read.wait().tolist()
                flags = int(host[n_levels])
                if (flags & 2 and key_bits < 56) or (flags & 4 and sp_bits < 32):
                    key_bits, sp_bits = 56, 32
                    continue
                break
            # the scene's read-back, written in place by the kernels (no concatenation launch): [n_0 .. n_{L-1}, flags, largest superpoint id]
            rb = torch.zeros(n_levels + 2, dtype=torch.int32, device=self.device)
            err = rb[n_levels:n_levels + 1]
            self.stats = ops.scene_stats(points)
            keys, self.icoords, self.origin, _ = ops.voxel_keys(points, inv, self.stats, shift_to_min, err=err)
            skeys, self.sidx = ops.sort_pairs(keys, None, 0, key_bits)
            ukeys, self.seg_start, self.inverse, n0 = ops.unique_sorted(
                skeys, self.sidx, N, None, 0, want_seg_start=True, want_map=True, map_size=N, nuniq_out=rb[0:1])
            keys_l, parents = [ukeys], []
            cap = N
            if at_once:
                uks, parents, _ = ops.unique_levels(ukeys, cap, n0, n_levels - 1, counts_out=rb[1:n_levels])   # every coarser level in four launches
                keys_l += uks
            else:
                n_prev = n0
                for lvl in range(1, n_levels):
                    clip = (self.stats, inv, lvl, clip_min_shape) if clip_min_shape > 0 else None
                    uk, _, parent, n_prev = ops.unique_sorted(keys_l[-1], None, cap, n_prev, 3, want_seg_start=False, want_map=True,
                                                              clip=clip, nuniq_out=rb[lvl:lvl + 1])
                    keys_l.append(uk)
                    parents.append(parent)
            sp_keys = None
            if superpoints is not None:                  # (the ids' largest value - the superpoint count - does not wait for their sort)
                sp_keys = ops.keys_from_i64(superpoints, check=(sp_bits if sp_bits < 32 else 64, err, 4), max_out=rb[n_levels + 1:])
            self._rb_dev = rb
            read = ops.HostRead(rb)                      # synchronisation 1 of the scene (polled)
            if sp_keys is not None:                      # sorted while the read-back travels
                self.sp_sorted, self.sp_sidx = ops.sort_pairs(sp_keys, None, 0, sp_bits)
            if while_waiting is not None:
                while_waiting(self)
                while_waiting = None
            host = read.wait().tolist()
            flags = int(host[n_levels])
            if (flags & 2 and key_bits < 56) or (flags & 4 and sp_bits < 32):
                key_bits, sp_bits = 56, 32                                   # a key did not fit: full sorts (rare: the chain runs twice)
                continue
            break
        self.n_vox = [int(v) for v in host[:n_levels]]
        if int(host[n_levels]) & 1:
            raise RuntimeError("scene exceeds the 16-bit-per-axis voxel key range (extent > ~1.3 km at 2 cm)")
        if int(host[n_levels]) & 8:
            raise RuntimeError("superpoint ids must lie in [0, 2^31 - 2] (a negative or oversized id was found)")
        self.n_superpoints = int(host[n_levels + 1]) + 1 if superpoints is not None else 0
        self.keys = [k[: self.n_vox[l]] for l, k in enumerate(keys_l)]
        self.parents = [p[: self.n_vox[l]] for l, p in enumerate(parents)]
        self.seg_start = self.seg_start[: self.n_vox[0] + 1]
        self._hash: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
        self._same: Dict[Tuple[int, int], torch.Tensor] = {}
        self._stride: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
        self._perm8 = perm8_device(order, self.device)
        self._sp_start = None
        self.density: Dict[Tuple, float] = {}
        self.pairs: Dict[Tuple, "ops.PairLists"] = {}       # offset-major rulebooks (prepare())
        self.events: Dict[Tuple, torch.cuda.Event] = {}     # tables built on the side stream (prepare(fork=True)) -> their event
        self._join_ev = self._late = self._stem_done = None
        self._fork_ev = ops.stream_event()                   # everything the tables are built from is complete here

    def prepare(self, same=(), strides=(), chained=False, fork=False):
        """Build the listed neighbour tables now and read their rulebook sizes back in ONE copy (the
        second and last synchronisation of a scene): density[key] = pairs / (K * V_out) lets the host
        pick the pair-compacted convolution kernel for sparse maps.  same: [(level, ksize)], strides: [level].
        fork=True (the caller runs the tables through `LayerPlan.run`, which waits per table): with worst-case list sizes (no
        read-back) only the FIRST table of `same` - the stem's - is built on this stream; the others go to the thread's side stream
        in the order the U-Net needs them, `self.events[key]` is recorded behind each group."""
        same = list(dict.fromkeys((lvl, k) for (lvl, k) in same if ("same", lvl, k) not in self.density))
        exact = exact_pair_capacity(self.n_vox[0], self.device) or not ops.PAIR_CONV
        if fork and FORK_JOIN and not exact and len(same) > 1 and not self.pairs:
            return self._prepare_forked(same, list(strides), chained)
        self._build_tables(same, strides, chained, exact)

    def _prepare_forked(self, same, strides, chained):
        """Fork / join inside a scene.  Everything the side stream reads was finished before the scene's host
        synchronisation (keys, parents) or is ordered by an event (the level-0 hash table, built here for the stem)."""
        self._build_tables(same[:1], [], chained, False)                 # stem: this stream, first in line
        stem_done = ops.stream_event()                                   # (its hash table serves the other level-0 table as well)
        lvl0 = same[0][0]
        rest = same[1:]
        # need order: the levels below the stem's downwards (one group per level: the first must be ready when the stem's convolution
        # ends), then the deepest levels together with the stem level's other tables (the last block of the U-Net)
        groups = []
        for d in (1, 2):
            groups.append(([t for t in rest if t[0] == lvl0 + d], [l for l in strides if l == lvl0 + d - 1]))
        taken_s = [t for g in groups for t in g[0]]
        taken_l = [l for g in groups for l in g[1]]
        groups.append(([t for t in rest if t not in taken_s], [l for l in strides if l not in taken_l]))
        self._stem_done = stem_done
        # LayerPlan.run issues the groups between its segments (next_fork): the stem's convolution is enqueued before the first group
        self._late = [(g, chained) for g in groups if g[0] or g[1]]

    def _fork_side(self, group, chained):
        g_same, g_strides = group
        if not g_same and not g_strides:
            return
        side = ops.side_streams(1, self.device)[0]
        self._side_used = True
        with ops.use_stream(side):
            side.wait_event(self._fork_ev)
            if getattr(self, "_hier_built", False) or any(t[0] in self._hash for t in g_same):
                side.wait_event(self._stem_done)                          # maps / a hash table the stem's stream built
            keys = self._build_tables(g_same, g_strides, chained, False)
            ev = ops.stream_event()
            for key in keys:
                self.events[key] = ev
            self._join_ev = ev

    def next_fork(self) -> bool:
        """Enqueue the next pending table group of a forked prepare() on the side stream (LayerPlan.run calls it once the layers that
        need no further table are enqueued).  False: nothing was pending."""
        if not self._late:
            return False
        self._fork_side(*self._late.pop(0))
        return True

    def release_side(self):
        """After the last kernel that reads a side-built table is enqueued (LayerPlan.run calls it): the side stream waits for
        everything the scene's stream holds so far.  The tables come out of the SIDE stream's allocator pool but are read by the
        scene's stream; when this SceneMaps dies their blocks return to that pool at once, and without this edge only the next
        scene's fork event - recorded on whatever stream THAT forward runs on - would keep later side-stream work from overwriting
        them under the U-Net kernels that still read them."""
        if getattr(self, "_side_used", False):
            ops.side_streams(1, self.device)[0].wait_event(ops.stream_event())
            self._side_used = False

    def join(self):
        """The calling stream waits for the side stream's table building (no-op without a fork)."""
        while self.next_fork():
            pass
        if self._join_ev is not None:
            torch.cuda.current_stream().wait_event(self._join_ev)
            self._join_ev = None
            self.events = {}

    def _build_tables(self, same, strides, chained, exact):
        """-> keys of the pair lists built (on the current stream)."""
        counters = torch.zeros(max(1, len(same)), 64, dtype=torch.int32, device=self.device) if exact else None
        L = len(self.keys)
        hier = (HIER_MAPS and same and not self.clipped and not self._same and len(self.parents) == L - 1 and min(self.n_vox) > 0
                and all((k == 3) or (k == 5 and lvl == 0) for lvl, k in same))
        if hier:
            # all maps of the scene now, whatever subset this call asks for (a forked prepare() calls again for the other levels and
            # finds them): the hierarchy runs coarse to fine, and the whole chain costs less than one level's hash probes did
            want5 = (0, 5) in same
            cnt_all = torch.zeros(L + 1, 64, dtype=torch.int32, device=self.device) if exact else None
            nbr3, nbr5, st_maps = ops.kernel_maps_hier(self.keys, self.parents, self.n_vox, offsets_device(3, self.order, self.device),
                                                       offsets_device(5, self.order, self.device) if want5 else None, inv27_table(self.order),
                                                       cnt_all, perm8=self._perm8 if not self._stride else None)
            for l, t in enumerate(st_maps or []):                # the stride-2 maps of every level pair ride in the same launches
                self._stride[l] = t
            for l in range(L):
                self._same[(l, 3)] = nbr3[l]
            if want5:
                self._same[(0, 5)] = nbr5
            self._hier_built = True
            if exact:
                counters = cnt_all[[L if k == 5 else lvl for lvl, k in same]]
        for i, (lvl, k) in enumerate(same):
            if (lvl, k) in self._same:
                if exact and not hier:                           # (built earlier without a counter: count the table's entries)
                    counters[i, 0] = (self._same[(lvl, k)] >= 0).sum().to(torch.int32)
                continue
            offs = offsets_device(k, self.order, self.device)
            self._same[(lvl, k)] = ops.kernel_map(self.keys[lvl], self.n_vox[lvl], self.table(lvl), offs,
                                                  counters[i] if exact else None, mirrored=MIRRORED_MAPS and k % 2 == 1)
        for lvl in strides:
            self._stride_maps(lvl)
        if exact:
            host = ops.HostRead(counters.sum(dim=1)).wait().tolist() if same else []   # synchronisation 2 (polled)
        else:
            # no read-back: size the pair lists for the worst case (every offset of every voxel has a neighbour).  The
            # kernels walk the REAL tile count, which the list builder leaves on the device; only allocations grow.
            host = [k ** 3 * self.n_vox[lvl] for (lvl, k) in same]
        # evaluation (`chained`): nothing reads a [K, M] position table or the lists' unused capacity - pass 2 walks per-row lists, pass 1 the
        # real tiles (csrc/pair_gemm.hip, "plain lists without a position table"); training keeps both (pair_out_rows, the weight gradient)
        lean = bool(chained) and LEAN_LISTS
        todo = []                                               # (key, nbr, pairs): all rulebooks of the scene in one launch set
        for (lvl, k), c in zip(same, host):
            self.density[("same", lvl, k)] = (c / max(1, k ** 3 * self.n_vox[lvl])) if exact else None
            if ops.PAIR_CONV:                                   # stride-1 table of the level onto itself: offset k^3 // 2 pairs every row with itself
                if chained and PAIR_CHAIN and k == 3 and lvl in PAIR_CHAIN_LEVELS:
                    center = ops.PAIR_CHAINED                    # mirror groups + centre share partial products (evaluation)
                else:
                    center = -1
                todo.append((("same", lvl, k), self._same[(lvl, k)], c, center, False, lean))
        for lvl in strides:
            # every fine voxel has exactly one parent: P = V_fine pairs in both directions
            self.density[("down", lvl)] = self.n_vox[lvl] / max(1, 8 * self.n_vox[lvl + 1])
            self.density[("up", lvl)] = 1.0 / 8.0
            if ops.PAIR_CONV and ("down", lvl) not in self.pairs:
                dn, up = self._stride_maps(lvl)
                todo.append((("down", lvl), dn, self.n_vox[lvl], -1, False, lean))
                # transposed convolution: every fine voxel has exactly one parent - one pair per output row.  (Not with spconv's
                # output-extent clip: fine voxels whose parent was dropped have NO pair and must still receive shift / activation.)
                todo.append((("up", lvl), up, self.n_vox[lvl], -1, not self.clipped, lean and not self.clipped))
        if todo:
            for t, pl in zip(todo, ops.pair_lists_batch([t[1:] for t in todo])):
                self.pairs[t[0]] = pl
        return [t[0] for t in todo]

    # ------------------------------------------------------------------------------------------
    def table(self, level: int):
        if level not in self._hash:
            self._hash[level] = ops.hash_build(self.keys[level], self.n_vox[level])
        return self._hash[level]

    def same(self, level: int, ksize: int) -> torch.Tensor:
        """nbr [k^3, V_l] of a stride-1 convolution on level `level`."""
        self.join()
        key = (level, ksize)
        if key not in self._same:
            offs = offsets_device(ksize, self.order, self.device)
            self._same[key] = ops.kernel_map(self.keys[level], self.n_vox[level], self.table(level), offs,
                                             mirrored=MIRRORED_MAPS and ksize % 2 == 1)
        return self._same[key]

    def _stride_maps(self, level: int):
        if level not in self._stride:
            self._stride[level] = ops.stride_maps(self.keys[level], self.parents[level], self.n_vox[level],
                                                  self.n_vox[level + 1], self._perm8)
        return self._stride[level]

    def down(self, level: int) -> torch.Tensor:
        """nbr [8, V_{l+1}] of the k=2 s=2 convolution level -> level+1."""
        self.join()
        return self._stride_maps(level)[0]

    def up(self, level: int) -> torch.Tensor:
        """nbr [8, V_l] of the transposed k=2 s=2 convolution level+1 -> level."""
        self.join()
        return self._stride_maps(level)[1]

    def conv_table(self, kind: str, level: int, ksize: int = 0) -> dict:
        """Keyword arguments of ops.gather_gemm for one neighbour table: nbr, its rulebook density and (after
        prepare()) its offset-major pair lists.  kind: "same" (ksize^3 offsets), "down" / "up" (k=2 s=2)."""
        key = (kind, level, ksize) if kind == "same" else (kind, level)
        nbr = self.same(level, ksize) if kind == "same" else (self.down(level) if kind == "down" else self.up(level))
        return dict(nbr=nbr, density=self.density.get(key), pairs=self.pairs.get(key))

    # ------------------------------------------------------------------------------------------
    def voxel_features(self, points, feats2d, mode: int, ld_out: int, stats=None) -> torch.Tensor:
        """`stats`: scene statistics of `points` when they are not the points the maps were built from (elastic coordinates
        voxelise the scene, the features still come from the undistorted points)."""
        return ops.voxel_mean(points, feats2d, mode, self.stats if stats is None else stats, self.sidx, self.seg_start,
                              self.n_vox[0], ld_out)

    def pool(self, feat: torch.Tensor, C: int):
        """Fused devoxelise + superpoint mean: ([S,C] features, [S,3] quantised mean positions)."""
        if self._sp_start is None:
            self._sp_start = ops.segment_starts(self.sp_sorted, self.n_points, self.n_superpoints)
        return ops.pool_superpoints(feat, C, self.inverse, self.icoords, self.voxel_size, self.sp_sidx,
                                    self._sp_start, self.n_superpoints)

    def rulebook_sizes(self):
        """{(kind, level[, k]): number of (in, out, offset) pairs} of the tables built so far (for the
        roofline accounting of bench.py; costs a sync)."""
        self.join()
        out = {}
        for (lvl, k), t in self._same.items():
            out[("same", lvl, k)] = int((t >= 0).sum())
        for lvl, (d, u) in self._stride.items():
            out[("down", lvl)] = int((d >= 0).sum())
            out[("up", lvl)] = int((u >= 0).sum())
        return out


class BatchSceneMaps(SceneMaps):
    """B scenes voxelised as ONE block-diagonal sparse tensor for the evaluation forward - what ME.utils.batch_sparse_collate
    builds for a batch (`minkunet.py:624-627`, collated by `utils/dataset_utils.py:215-230`): the scene index rides in the key
    bits above the Z-order code, so ONE radix sort, ONE run-length unique per level, ONE hash table and ONE kernel-map launch per
    level serve every scene, the rows of level l are the scenes' rows one after the other (each scene in its own Z-order, the
    order a single-scene SceneMaps gives it), neighbour tables never cross scenes, and every convolution of the U-Net is one
    launch over B times the pairs.  Evaluation BatchNorm is an affine map and every kernel on this path computes an output row
    from that row's own pairs in a fixed order, so each scene's rows come out bit-identical to its single-scene forward
    (tests/test_gpu_batch_eval.py).  Point / voxel / superpoint numbers are batch-global; `sp_off` splits the pooled rows."""

    def __init__(self, points: Sequence[torch.Tensor], voxel_size: float, n_levels: int, shift_to_min: bool = False,
                 order: str = "x_fastest", superpoints: Optional[Sequence[torch.Tensor]] = None, clip_min_shape: int = 0):
        B = len(points)
        if not 1 <= B <= 16:
            raise ValueError("BatchSceneMaps: 1..16 scenes per batch (SD3D_MAX_BATCH)")
        if not all(p.is_cuda for p in points):
            raise RuntimeError("BatchSceneMaps needs device-resident points (no CPU fallback in the product path)")
        self.order = order
        self.device = dev = points[0].device
        self.voxel_size = float(voxel_size)
        self.clipped = clip_min_shape > 0
        self.n_scenes = B
        self.point_off = [0]
        for p in points:
            self.point_off.append(self.point_off[-1] + int(p.shape[0]))
        N = self.point_off[-1]
        self.n_points = N
        inv = float(np.float32(1.0) / np.float32(voxel_size))
        self.stats = torch.empty(B, 9, dtype=torch.float32, device=dev)
        self.superpoints = superpoints
        scene_bits = max(1, (B - 1).bit_length())
        # Optimistic radix passes as for one scene: Morton parts that fit 32 bits are sorted over those and then over the scene bits
        # (5 passes instead of 7), superpoint ids that fit 16 bits likewise (3 instead of 5); the key kernels raise a flag otherwise and
        # the chain runs again with the full sorts.  Either way the arrays are those of the full sorts.
        optimistic = OPTIMISTIC_SORT
        while True:
            # the batch's read-back, written in place by the kernels: [n_0 .. n_{L-1}, flags, largest superpoint id of scene 0 .. B-1]
            rb = torch.zeros(n_levels + 1 + B, dtype=torch.int32, device=dev)
            err = rb[n_levels:n_levels + 1]
            keys = torch.empty(N, dtype=torch.int64, device=dev)
            self.icoords = torch.empty(N, 3, dtype=torch.int32, device=dev)
            for i, p in enumerate(points):
                a, b = self.point_off[i], self.point_off[i + 1]
                ops.scene_stats(p, out=self.stats[i])
                ops.voxel_keys(p, inv, self.stats[i], shift_to_min, batch_index=i, out=(keys[a:b], self.icoords[a:b], err))
            if optimistic:
                skeys, self.sidx = ops.sort_pairs(keys, None, 0, 32)
                skeys, self.sidx = ops.sort_pairs(skeys, self.sidx, MORTON_BITS, MORTON_BITS + scene_bits)
            else:
                skeys, self.sidx = ops.sort_pairs(keys, None, 0, 56)
            if LEVELS_AT_ONCE and VOXELISE_ONE_CALL and clip_min_shape == 0 and 1 < n_levels <= 8:
                # every level from the sorted point keys in four launches (round 5; level 0 and the coarser levels were four each)
                keys_l, self.seg_start, self.inverse, parents = ops.voxel_levels_all(skeys, self.sidx, n_levels, rb[0:n_levels])
            elif LEVELS_AT_ONCE and clip_min_shape == 0 and 1 < n_levels <= 8:
                ukeys, self.seg_start, self.inverse, n0 = ops.unique_sorted(
                    skeys, self.sidx, N, None, 0, want_seg_start=True, want_map=True, map_size=N, nuniq_out=rb[0:1])
                uks, parents, _ = ops.unique_levels(ukeys, N, n0, n_levels - 1, counts_out=rb[1:n_levels])
                keys_l = [ukeys] + uks
            else:
                ukeys, self.seg_start, self.inverse, n0 = ops.unique_sorted(
                    skeys, self.sidx, N, None, 0, want_seg_start=True, want_map=True, map_size=N, nuniq_out=rb[0:1])
                keys_l, parents = [ukeys], []
                n_prev = n0
                for lvl in range(1, n_levels):
                    clip = (self.stats, inv, lvl, clip_min_shape) if clip_min_shape > 0 else None
                    uk, _, parent, n_prev = ops.unique_sorted(keys_l[-1], None, N, n_prev, 3, want_seg_start=False, want_map=True, clip=clip,
                                                              nuniq_out=rb[lvl:lvl + 1])
                    keys_l.append(uk)
                    parents.append(parent)
            sp_keys = None
            if superpoints is not None:                        # (every scene's largest id - its superpoint count - does not wait for the sort)
                sp_keys = torch.empty(N, dtype=torch.int64, device=dev)
                for i, sp in enumerate(superpoints):
                    ops.keys_from_i64_offset(sp.contiguous(), i << 32, sp_keys[self.point_off[i]:self.point_off[i + 1]],
                                             check=(16 if optimistic else 64, err, 4), max_out=rb[n_levels + 1 + i:n_levels + 2 + i])
            read = ops.HostRead(rb)                            # synchronisation 1 of the batch (polled)
            if sp_keys is not None:                            # sorted while the read-back travels
                if optimistic:
                    k1, v1 = ops.sort_pairs(sp_keys, None, 0, 16)
                    self.sp_sorted, self.sp_sidx = ops.sort_pairs(k1, v1, 32, 32 + scene_bits)
                else:
                    self.sp_sorted, self.sp_sidx = ops.sort_pairs(sp_keys, None, 0, 32 + scene_bits)
            host = read.wait().tolist()
            flags = int(host[n_levels])
            if optimistic and flags & 6:
                optimistic = False                             # a key did not fit: full sorts (rare: the chain runs twice)
                continue
            break
        self.n_vox = [int(v) for v in host[:n_levels]]
        if int(host[n_levels]) & 1:                                # (bit 1 = "more than 32 Morton bits": the batch sorts all 56 key bits anyway)
            raise RuntimeError("scene exceeds the 16-bit-per-axis voxel key range (extent > ~1.3 km at 2 cm)")
        if int(host[n_levels]) & 8:
            raise RuntimeError("superpoint ids must lie in [0, 2^31 - 2] (a negative or oversized id was found)")
        self.sp_off = [0]
        if superpoints is not None:
            for v in host[n_levels + 1:n_levels + 1 + B]:
                self.sp_off.append(self.sp_off[-1] + int(v) + 1)
        self.n_superpoints = self.sp_off[-1]
        self.keys = [k[: self.n_vox[l]] for l, k in enumerate(keys_l)]
        self.parents = [p[: self.n_vox[l]] for l, p in enumerate(parents)]
        self.seg_start = self.seg_start[: self.n_vox[0] + 1]
        self._hash, self._same, self._stride = {}, {}, {}
        self._perm8 = perm8_device(order, self.device)
        self._sp_start = None
        self.density, self.pairs = {}, {}
        self.events, self._join_ev, self._fork_ev = {}, None, ops.stream_event()
        self._late = self._stem_done = None

    def voxel_features(self, points: Sequence[torch.Tensor], feats2d, mode: int, ld_out: int, stats=None) -> torch.Tensor:
        """points / feats2d: the scenes' tensors in batch order (feats2d None or a list)."""
        if stats is not None:
            raise NotImplementedError("BatchSceneMaps.voxel_features: per-scene override statistics are a training feature")
        scenes = [(points[i], None if feats2d is None else feats2d[i], self.stats[i], self.point_off[i]) for i in range(self.n_scenes)]
        return ops.voxel_mean_batch(scenes, mode, self.keys[0], self.sidx, self.seg_start, self.n_vox[0], ld_out)

    def pool(self, feat: torch.Tensor, C: int):
        """-> ([S_total, C], [S_total, 3]) over the batch-global superpoint numbering; scene i owns rows sp_off[i]:sp_off[i + 1]."""
        if self._sp_start is None:
            self._sp_start = ops.segment_starts_batch(self.sp_sorted, self.n_points, self.n_superpoints, self.sp_off[:-1])
        return ops.pool_superpoints(feat, C, self.inverse, self.icoords, self.voxel_size, self.sp_sidx, self._sp_start,
                                    self.n_superpoints)


class BatchedMaps:
    """Several scenes as ONE block-diagonal sparse tensor, for the training step: MinkowskiEngine collates the scenes of a batch
    into one tensor (`minkunet.py:624-627`), convolutions never cross scenes, but every BatchNorm takes its statistics over the
    voxels of ALL scenes.  Rows of level l are the scenes' rows one after the other; a neighbour table is the scenes' tables
    side by side with the input indices shifted by the scene's row offset of the INPUT level, and gets its own pair lists.
    Implements the part of SceneMaps the network definitions and `train_ops` use."""

    def __init__(self, maps: Sequence["SceneMaps"]):
        self.maps = list(maps)
        self.device = self.maps[0].device
        n_levels = len(self.maps[0].n_vox)
        self.offsets = [[0] for _ in range(n_levels)]
        for m in self.maps:
            for l in range(n_levels):
                self.offsets[l].append(self.offsets[l][-1] + int(m.n_vox[l]))
        self.n_vox = [self.offsets[l][-1] for l in range(n_levels)]
        self._tables: Dict[Tuple, dict] = {}

    def prepare(self, same=(), strides=(), chained=False, fork=False):     # (training batches: `chained` / `fork` are False by construction)
        for m in self.maps:
            m.prepare(same=same, strides=strides, chained=chained)

    def rows(self, level: int, i: int):
        return self.offsets[level][i], self.offsets[level][i + 1]

    def conv_table(self, kind: str, level: int, ksize: int = 0) -> dict:
        key = (kind, level, ksize) if kind == "same" else (kind, level)
        if key not in self._tables:
            in_level = level if kind == "same" else (level if kind == "down" else level + 1)
            parts, cap = [], 0
            for i, m in enumerate(self.maps):
                t = m.conv_table(kind, level, ksize)
                nbr = t["nbr"]
                off = self.offsets[in_level][i]
                parts.append(torch.where(nbr >= 0, nbr + off, nbr) if off else nbr)
                cap += int(t["pairs"].p_cap) if t.get("pairs") is not None else int(nbr.numel())
            nbr_cat = torch.cat(parts, dim=1).contiguous()
            self._tables[key] = dict(nbr=nbr_cat, density=None, pairs=ops.pair_lists(nbr_cat, cap))
        return self._tables[key]
