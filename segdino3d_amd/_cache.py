"""Validity of derived weights.

The eval path runs on COPIES of the parameters: permuted / padded convolution kernels, BatchNorm folded from the running
statistics, concatenated projection weights, bf16 roundings, the recorded layer plan that points into them.  The reference
loop alternates training steps and evaluations on the same module (`engine/train_engine_3d.py:75, 171-173`), and
`optimizer.step()`, the training-mode BatchNorm update and `load_state_dict` all write the sources IN PLACE - so every
derived copy carries the version stamp of its sources and is rebuilt when the stamp moves.
"""
from __future__ import annotations

import torch.nn as nn


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

    def _sources_stamp(self) -> int:
        """Changes whenever a parameter or buffer of this module was written in place (tensor version counters) or the
        set of tensors changed."""
        src = self.__dict__.get("_dw_sources")
        if src is None:
            src = [p for p in self.parameters()] + [b for b in self.buffers()]
            self.__dict__["_dw_sources"] = src
        return sum([t._version for t in src]) + (len(src) << 40)

    def _derived_valid(self) -> bool:
        """True if the derived copies were built from the sources as they are now; otherwise drops them and records the
        current stamp (the caller rebuilds)."""
        stamp = self._sources_stamp()
        if self.__dict__.get("_dw_stamp") == stamp:
            return True
        self._derived_reset()
        self.__dict__["_dw_sources"] = None
        self.__dict__["_dw_stamp"] = self._sources_stamp()
        return False
