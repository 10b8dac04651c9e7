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
    def __init__(self, keep_all: bool = False, keep_arena: bool = False):
        """keep_all=False (default): `maps` holds the voxel maps of the MOST RECENT forward only (a scene's neighbour tables and
        pair lists are > 100 MB at 150 k points; a scope around an evaluation loop must not accumulate them).  keep_all=True keeps
        every forward's maps for the lifetime of the scope."""
        self.keep_all = bool(keep_all)
        # keep_arena=True: `arenas` receives the activation arena of every U-Net plan run in the scope (EVERY layer's output rows back to
        # back, plan.LayerPlan.run) - tests compare them bit for bit between kernel variants
        self.keep_arena = bool(keep_arena)
        self.arenas = []
        self.outputs = None          # decoder output dict of the most recent forward in the scope
        self.maps = []               # sparse.SceneMaps of the most recent forward (all forwards with keep_all)
        self.sp_feats = None         # backbone output: per-scene superpoint features / positions
        self.sp_pos = None

    def __enter__(self):
        self._prev = getattr(_TLS, "cap", None)
        _TLS.cap = self
        return self

    def __exit__(self, *exc):
        _TLS.cap = self._prev
        return False

    def record_maps(self, maps):
        """Called by the backbones' forward_wrapper with the maps of the scenes of ONE forward."""
        if self.keep_all:
            self.maps.extend(maps)
        else:
            self.maps = list(maps)


def active():
    return getattr(_TLS, "cap", None)
