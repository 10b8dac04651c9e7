"""Validity of derived weights.

The eval path runs on COPIES of the parameters: permuted / padded convolution kernels, BatchNorm folded from the running
statistics, concatenated projection weights, bf16 roundings, the recorded layer plan that points into them.  The reference
loop alternates training steps and evaluations on the same module (`engine/train_engine_3d.py:75, 171-173`), and
`optimizer.step()`, the training-mode BatchNorm update and `load_state_dict` all write the sources IN PLACE - so every
derived copy carries the version stamp of its sources and is rebuilt when the stamp moves.
"""
from __future__ import annotations

import operator

import torch
import torch.nn as nn

_is = operator.is_
_data_ptr = torch.Tensor.data_ptr


class DerivedWeights(nn.Module):
    """Mixin for modules that cache tensors derived from their own parameters and buffers."""

    def _derived_reset(self):
        """Drop every derived copy (subclasses extend)."""
        self.__dict__["_dw_sources"] = None
        self.__dict__["_dw_stamp"] = None

    def _apply(self, fn, *a, **k):                      # .to() / .cuda() / .float(): storage replaced
        self._derived_reset()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._derived_reset()
        return super().load_state_dict(*a, **k)

    def train(self, mode: bool = True):                 # .train() / .eval(): cheap, and what the reference loop does
        self._derived_reset()
        return super().train(mode)

    def invalidate_packed_weights(self):
        self._derived_reset()

    def _sources_stamp(self):
        """Changes whenever a parameter or buffer of this module tree was written in place (tensor version counters), had its
        storage replaced (`p.data = new`, `vector_to_parameters`: the address moves), or was itself replaced - a parameter,
        buffer or submodule swapped in (every cached tensor is re-checked against the `_parameters` / `_buffers` / `_modules`
        slot it was found in).  The tree is enumerated once per reset (a fresh `parameters()` walk costs ~1 ms per forward on
        this model, the slot check ~0.1 ms).  What it cannot see: writes through `.data` that keep the storage
        (`p.data.copy_()`, `p.data.mul_()`) do not bump the version counter - call `invalidate_packed_weights()` after those."""
        src = self.__dict__.get("_dw_sources")
        if src is not None and not all(map(_is, map(dict.get, src[0], src[1]), src[2])):
            src = None                                   # a slot holds another object now: re-enumerate
        if src is None:
            slots = []
            for m in self.modules():
                slots += [(m._modules, n, c) for n, c in m._modules.items() if c is not None]
                slots += [(m._parameters, n, p) for n, p in m._parameters.items() if p is not None]
                slots += [(m._buffers, n, b) for n, b in m._buffers.items() if b is not None]
            src = ([d for d, _, _ in slots], [n for _, n, _ in slots], [t for _, _, t in slots])
            self.__dict__["_dw_sources"] = src
            self.__dict__["_dw_tensors"] = [t for t in src[2] if not isinstance(t, nn.Module)]
            self.__dict__["_dw_epoch"] = self.__dict__.get("_dw_epoch", 0) + 1
        ts = self.__dict__["_dw_tensors"]
        return hash((self.__dict__["_dw_epoch"], tuple([t._version for t in ts]), tuple(map(_data_ptr, ts))))

    def _derived_valid(self) -> bool:
        """True if the derived copies were built from the sources as they are now; otherwise drops them and records the
        current stamp (the caller rebuilds)."""
        stamp = self._sources_stamp()
        if self.__dict__.get("_dw_stamp") == stamp:
            return True
        self._derived_reset()
        self.__dict__["_dw_stamp"] = self._sources_stamp()    # the reset dropped the enumeration: stamp of the fresh one
        return False
