"""Per-call capture of intermediate results (decoder outputs, the scene's voxel maps) for tests, bench.py and smoke().

The model object is shared by the worker threads of `dist_eval.PipelinedRunner`, so intermediates are never stored on it:
a caller that wants them opens a `capture()` scope in ITS thread and the forward running in that thread records into it.

    with segdino3d_amd.capture() as cap:
        model([pts], [tgt])
    cap.outputs["masks"][0], cap.maps[0].n_vox
"""
from __future__ import annotations

import threading

_TLS = threading.local()


class capture:
    def __init__(self):
        self.outputs = None          # decoder output dict of the most recent forward in the scope
        self.maps = []               # sparse.SceneMaps, one per scene run in the scope
        self.sp_feats = None         # backbone output: per-scene superpoint features / positions
        self.sp_pos = None

    def __enter__(self):
        self._prev = getattr(_TLS, "cap", None)
        _TLS.cap = self
        return self

    def __exit__(self, *exc):
        _TLS.cap = self._prev
        return False


def active():
    return getattr(_TLS, "cap", None)
