"""Validity stamp of the derived (packed / folded / planned) weights, segdino3d_amd/_cache.py - host logic, CPU only."""
import torch
import torch.nn as nn

import segdino3d_amd as seg
from segdino3d_amd.configs import scannet200_model_cfg


def test_stamp_is_stable_and_sees_every_kind_of_change():
    m = seg.build_architecture(scannet200_model_cfg(query_num=200)).eval()
    for mod in (m.backbone, m.decoder):
        assert mod._derived_valid() is False            # first call: nothing derived yet
        assert mod._derived_valid() and mod._derived_valid(), "an unchanged module must stay valid (else every forward re-packs)"
    d = m.decoder
    with torch.no_grad():
        d.out_norm.bias.add_(1)                          # in-place write: version counter
    assert d._derived_valid() is False and d._derived_valid()
    d.out_norm.weight.data = torch.ones(256)             # storage replaced
    assert d._derived_valid() is False and d._derived_valid()
    d.out_norm.bias = nn.Parameter(torch.zeros(256))     # parameter replaced
    assert d._derived_valid() is False and d._derived_valid()
    d.out_cls[2] = nn.Linear(256, 199)                   # submodule replaced
    assert d._derived_valid() is False and d._derived_valid()
    m.backbone.bn0.bn.running_mean.add_(0.5)             # buffer written in place
    assert m.backbone._derived_valid() is False and m.backbone._derived_valid()
    m.eval()
    assert d._derived_valid() is False and d._derived_valid()


def test_split_cache_is_lru_under_a_lock():
    from segdino3d_amd import ops
    ops.clear_split_cache()
    ws = [torch.randn(1, 4, 32) for _ in range(5)]
    old_max = ops._SPLIT_CACHE_MAX
    ops._SPLIT_CACHE_MAX = 3
    try:
        for w in ws[:3]:
            ops._cached_split(w, 1)
        a = ops._cached_split(ws[0], 1)                  # hit: moves to the end, same object back
        assert a is ops._cached_split(ws[0], 1)
        ops._cached_split(ws[3], 1)                      # evicts ws[1] (least recently used), not ws[0]
        keys = [k[0] for k in ops._SPLIT_CACHE]
        assert ws[1].data_ptr() not in keys and ws[0].data_ptr() in keys and len(keys) == 3
    finally:
        ops._SPLIT_CACHE_MAX = old_max
        ops.clear_split_cache()
