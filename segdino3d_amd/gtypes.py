"""Attribute-dict target container of the reference boundary.

Mirrors the behaviour (not the code) of `segdino3d/gtypes.py:3-94` in the reference: an object
whose fields are reachable both as attributes and by `obj["key"]`, supports `key in obj`,
`.keys()`, and `.to(device)` that moves tensors found directly in the fields, in list fields and
one level down in dict fields (that is how `extra_features` reaches the GPU,
`evaluation/evaluator_3d.py:81-82`).
"""
from __future__ import annotations

import torch


def _move(value, fn):
    if isinstance(value, torch.Tensor):
        return fn(value)
    return value


class GDType:
    def __init__(self, **fields):
        self.__dict__.update(fields)

    # mapping-style access -------------------------------------------------------------
    def __getitem__(self, key):
        return self.__dict__[key]

    def __setitem__(self, key, value):
        self.__dict__[key] = value

    def __contains__(self, key):
        return key in self.__dict__

    def keys(self):
        return self.__dict__.keys()

    def get(self, key, default=None):
        return self.__dict__.get(key, default)

    def __repr__(self):
        return f"{type(self).__name__}({self.__dict__!r})"

    # device movement ------------------------------------------------------------------
    def _apply(self, fn):
        for key, value in self.__dict__.items():
            self.__dict__[key] = _move(value, fn)
        return self

    def to(self, device):
        return self._apply(lambda t: t.to(device))

    def cpu(self):
        return self._apply(lambda t: t.cpu())

    def cuda(self, idx=None):
        return self._apply(lambda t: t.cuda(idx) if idx else t.cuda())

    @property
    def shape(self):
        return {k: v.shape for k, v in self.__dict__.items() if isinstance(v, torch.Tensor)}


class GD3DTarget(GDType):
    """Per-scene annotation + extra features (reference: `gtypes.py:50-94`)."""

    _DEFAULTS = dict(labels=None, size=None, positive_map=None, scene_id=None, data_source=None,
                     prompt_type=None, loss_branch=None, area=None, orig_size=None, iscrowd=0,
                     masks=None)

    def __init__(self, **fields):
        merged = dict(self._DEFAULTS)
        merged.update(fields)
        super().__init__(**merged)

    def _apply(self, fn):
        for key, value in self.__dict__.items():
            if isinstance(value, torch.Tensor):
                self.__dict__[key] = fn(value)
            elif isinstance(value, list):
                for i, item in enumerate(value):
                    value[i] = _move(item, fn)
            elif isinstance(value, dict):
                for k, item in value.items():
                    value[k] = _move(item, fn)
        return self
