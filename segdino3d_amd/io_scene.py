"""Input side of the path: the reference's on-disk scene formats, a packed single-file format,
and a pinned-memory prefetcher that overlaps disk -> host -> device copies with the forward of earlier scenes.

Reference behaviour restated here (validation split; training-only augmentation is out of scope):
  * file layout and decoding: `segdino3d/datasets/dataset/scannet200.py:206-256` - `points/{id}.bin` f32 [N,6]
    (xyz + rgb 0..255), `instance_mask|semantic_mask/{id}.bin` i64 [N], `super_points/{id}.bin` i64 [N],
    `{id}.pth` = list of multi-scale [N,256] feature tensors averaged at load (`:234-235`), `{id}_query_feats.pth`
    [M,256], `{id}_query_3dctr.pth` [M,3];
  * colour normalisation of the val transform: `(rgb - mean) / std` with the constants of
    `datasets/transform/wrappers_3d.py:19-26`, applied as in `point_cloud_transforms.py:380-386`
    (two fp32 steps: subtract, then divide).
One scene is ~157 MB of fp32 (153 MB of it the per-point 2D features); decoding three .pth pickles and averaging
the feature scales costs more host time than the forward costs GPU time, so `pack_scene` writes the averaged,
normalised scene ONCE into one flat file that is read with a single sequential read into pinned memory.

`ScenePrefetcher` keeps `depth` scenes ahead: a reader thread fills pinned staging buffers, the copies are issued on
its own HIP stream, and the consumer only makes its compute stream wait on the copy event - the host never blocks on
a transfer.  Measured by `tools/bench_e2e.py`.
"""
from __future__ import annotations

import os
import struct
import threading
import queue
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np
import torch

from .gtypes import GD3DTarget

COLOR_MEAN = (0.47793125906962 * 255, 0.4303257521323044 * 255, 0.3749598901421883 * 255)
COLOR_STD = (0.2834475483823543 * 255, 0.27566157565723015 * 255, 0.27018971370874995 * 255)

MAGIC = b"SD3DSCN1"
# header: magic, N, M, feat dim, feature dtype code (0 = f32, 1 = f16), has_gt
_HEADER = struct.Struct("<8sqqqqq")
_ALIGN = 256


def normalize_points_color(points: torch.Tensor, mean=COLOR_MEAN, std=COLOR_STD) -> torch.Tensor:
    """In place, like `NormalizePointsColor.__call__`: subtract the mean, then divide by the std (fp32)."""
    assert points.shape[1] == 6, f"points should have 6 channels (xyz rgb), but got {points.shape[1]}"
    if mean is not None:
        points[:, 3:] = points[:, 3:] - points[:, 3:].new_tensor(mean)
    if std is not None:
        points[:, 3:] = points[:, 3:] / points[:, 3:].new_tensor(std)
    return points


def read_reference_scene(root_scenes: str, root_2dfeats: Optional[str], scene_id: str, normalize: bool = True) -> Dict:
    """Decode one scene from the reference's files (validation view of `ScanNet200InstanceSeg3D.__getitem__`)."""
    pts = np.fromfile(os.path.join(root_scenes, "points", f"{scene_id}.bin"), dtype=np.float32).reshape(-1, 6)
    out = {"points": torch.from_numpy(pts.copy())}
    for key, sub in (("instance_mask", "instance_mask"), ("semantic_mask", "semantic_mask"), ("super_points", "super_points")):
        path = os.path.join(root_scenes, sub, f"{scene_id}.bin")
        if os.path.exists(path):
            out[key] = torch.from_numpy(np.fromfile(path, dtype=np.int64).copy())
    if root_2dfeats is not None:
        feats = torch.load(os.path.join(root_2dfeats, f"{scene_id}.pth"))
        out["points_2dfeats"] = torch.stack(list(feats), dim=0).mean(dim=0) if isinstance(feats, (list, tuple)) else feats
        out["query2d_feats"] = torch.load(os.path.join(root_2dfeats, f"{scene_id}_query_feats.pth"))
        out["query2d_pos"] = torch.load(os.path.join(root_2dfeats, f"{scene_id}_query_3dctr.pth"))
    if normalize:
        normalize_points_color(out["points"])
    return out


def _pad(n: int) -> int:
    return (n + _ALIGN - 1) // _ALIGN * _ALIGN


def _sections(N: int, M: int, D: int, f16: bool, has_gt: bool) -> List[Tuple[str, np.dtype, Tuple[int, ...]]]:
    secs = [("points", np.dtype("<f4"), (N, 6)), ("super_points", np.dtype("<i8"), (N,)),
            ("points_2dfeats", np.dtype("<f2") if f16 else np.dtype("<f4"), (N, D)),
            ("query2d_feats", np.dtype("<f4"), (M, D)), ("query2d_pos", np.dtype("<f4"), (M, 3))]
    if has_gt:
        secs += [("instance_mask", np.dtype("<i8"), (N,)), ("semantic_mask", np.dtype("<i8"), (N,))]
    return secs


def pack_scene(path: str, scene: Dict, feats_fp16: bool = False) -> int:
    """Write a decoded scene (already colour-normalised, scales averaged) as one flat file.  feats_fp16 halves the
    file and the H2D copy but rounds the 2D features to half precision (NOT the fp32 configuration; off by default)."""
    pts = scene["points"].contiguous().float()
    N = pts.shape[0]
    f2d = scene["points_2dfeats"].contiguous()
    D = f2d.shape[1]
    qf, qp = scene["query2d_feats"].contiguous().float(), scene["query2d_pos"].contiguous().float()
    M = qf.shape[0]
    has_gt = "instance_mask" in scene and "semantic_mask" in scene
    arrays = {"points": pts.numpy(), "super_points": scene["super_points"].long().numpy(),
              "points_2dfeats": (f2d.half() if feats_fp16 else f2d.float()).numpy(), "query2d_feats": qf.numpy(),
              "query2d_pos": qp.numpy()}
    if has_gt:
        arrays["instance_mask"] = scene["instance_mask"].long().reshape(-1).numpy()
        arrays["semantic_mask"] = scene["semantic_mask"].long().reshape(-1).numpy()
    with open(path, "wb") as f:
        f.write(_HEADER.pack(MAGIC, N, M, D, 1 if feats_fp16 else 0, 1 if has_gt else 0).ljust(_ALIGN, b"\0"))
        for name, dt, shape in _sections(N, M, D, feats_fp16, has_gt):
            a = np.ascontiguousarray(arrays[name], dtype=dt).reshape(shape)
            f.write(a.tobytes())
            f.write(b"\0" * (_pad(a.nbytes) - a.nbytes))
        return f.tell()


def packed_layout(path: str):
    with open(path, "rb") as f:
        magic, N, M, D, fcode, has_gt = _HEADER.unpack(f.read(_HEADER.size))
    if magic != MAGIC:
        raise ValueError(f"{path}: not a packed scene file")
    off, lay = _ALIGN, {}
    for name, dt, shape in _sections(N, M, D, fcode == 1, has_gt == 1):
        nbytes = int(np.prod(shape)) * dt.itemsize
        lay[name] = (off, dt, shape)
        off += _pad(nbytes)
    return lay, off


def load_packed(path: str, pin: bool = False, staging: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """One sequential read of the whole file into a (pinned) byte buffer; the returned tensors are views into it."""
    lay, total = packed_layout(path)
    if staging is None or staging.numel() < total:
        staging = torch.empty(total, dtype=torch.uint8, pin_memory=pin)
    with open(path, "rb", buffering=0) as f:
        got = f.readinto(memoryview(staging.numpy())[:total])
    if got != total:
        raise IOError(f"{path}: short read ({got} of {total} bytes)")
    out = {}
    tdt = {np.dtype("<f4"): torch.float32, np.dtype("<f2"): torch.float16, np.dtype("<i8"): torch.int64}
    for name, (off, dt, shape) in lay.items():
        n = int(np.prod(shape)) * dt.itemsize
        out[name] = staging[off:off + n].view(tdt[dt]).view(*shape)
    out["_staging"] = staging
    return out


def to_device_scene(host: Dict[str, torch.Tensor], device, non_blocking: bool = True):
    """(points [N,6] f32, GD3DTarget) on the device, laid out like the reference dataset output."""
    dev = {k: v.to(device, non_blocking=non_blocking) for k, v in host.items() if not k.startswith("_")}
    f2d = dev["points_2dfeats"]
    if f2d.dtype != torch.float32:
        f2d = f2d.float()
    N = dev["points"].shape[0]
    masks = torch.ones(1, N, 1, dtype=torch.bool, device=device)          # the eval forward only needs the scene range
    extra = {"points_2dfeats": f2d, "query2d_feats": dev["query2d_feats"], "query2d_pos": dev["query2d_pos"],
             "super_point_masks": dev["super_points"]}
    tgt = GD3DTarget(masks=masks, labels=torch.zeros(1, dtype=torch.int64, device=device), extra_features=extra)
    if "instance_mask" in dev:
        tgt["gt_instance_mask"], tgt["gt_semantic_mask"] = dev["instance_mask"], dev["semantic_mask"]
    return dev["points"], tgt


class ScenePrefetcher:
    """Iterates packed scene files as device-resident (points, target) pairs, up to `depth` scenes ahead.

    `readers` threads take file indices from a shared counter, read their file into a pinned staging buffer (one
    sequential read, the GIL is released inside it) and issue the H2D copies on a dedicated stream; `__next__`
    hands scenes out IN ORDER, making the consumer's current stream wait on the scene's copy event (a device-side
    wait: the host never blocks on a transfer).  A staging buffer is recycled when the consumer asks for the scene
    after the one that used it.  Thread-safe for several consumers (the pipelined runner's workers)."""

    def __init__(self, paths: Iterable[str], device, depth: int = 3, readers: int = 2):
        self.paths = list(paths)
        self.device = torch.device(device)
        self.depth = max(1, depth)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._cv = threading.Condition()
        self._ready: Dict[int, tuple] = {}
        self._next_read = 0                                               # next file index a reader may take
        self._next_out = 0                                                # next index handed to a consumer
        self._free: "queue.Queue" = queue.Queue()
        for _ in range(self.depth + max(1, readers)):
            self._free.put(None)                                          # staging buffers are allocated on first use
        self._err: Optional[BaseException] = None
        self._threads = [threading.Thread(target=self._reader, daemon=True) for _ in range(max(1, readers))]
        for t in self._threads:
            t.start()

    def _reader(self):
        try:
            torch.cuda.set_device(self.device)
            while True:
                with self._cv:
                    while self._next_read < len(self.paths) and self._next_read >= self._next_out + self.depth:
                        self._cv.wait()                                   # far enough ahead of the consumer
                    if self._next_read >= len(self.paths):
                        return
                    i = self._next_read
                    self._next_read += 1
                staging = self._free.get()
                host = load_packed(self.paths[i], pin=True, staging=staging)
                with torch.cuda.stream(self.copy_stream):
                    pts, tgt = to_device_scene(host, self.device, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self.copy_stream)
                with self._cv:
                    self._ready[i] = (pts, tgt, ev, host["_staging"])
                    self._cv.notify_all()
        except BaseException as e:  # noqa: BLE001 - surfaced in the consumer
            with self._cv:
                self._err = e
                self._cv.notify_all()

    def __iter__(self):
        return self

    def __len__(self):
        return len(self.paths)

    def __next__(self):
        with self._cv:
            i = self._next_out
            if i >= len(self.paths):
                raise StopIteration
            self._next_out += 1
            self._cv.notify_all()                                         # readers may move ahead
            while i not in self._ready and self._err is None:
                self._cv.wait()
            if self._err is not None:
                raise self._err
            pts, tgt, ev, staging = self._ready.pop(i)
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        # the tensors were allocated on the copy stream: tell the allocator the consumer's stream uses them
        pts.record_stream(cur)
        for v in tgt.extra_features.values():
            if torch.is_tensor(v):
                v.record_stream(cur)
        # the H2D copy out of `staging` is complete once `ev` has fired; recycle it when that is known on the host
        self._recycle(ev, staging)
        return pts, tgt

    def _recycle(self, ev, staging):
        pending = getattr(self, "_pending", None)
        if pending is None:
            pending = self._pending = []
        pending.append((ev, staging))
        while pending and pending[0][0].query():
            self._free.put(pending.pop(0)[1])
        if len(pending) > self.depth:                                     # never starve the readers
            ev0, st0 = pending.pop(0)
            ev0.synchronize()
            self._free.put(st0)
