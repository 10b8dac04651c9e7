"""Host-output path of `Baseline3D._predict_finish` (baseline3d.py:453-454 -> evaluator_3d.py:178): the [n, N] instance masks travel
bit-packed; the host half (sd3d_unpack_bits_host, PackedMasks) is checked here without a GPU, the device half in
tests/test_gpu_host_outputs.py."""
import numpy as np
import pytest


@pytest.mark.parametrize("n,N", [(0, 17), (1, 1), (3, 8), (5, 13), (7, 64), (4, 1000), (2, 150001)])
def test_unpack_bits_host_matches_numpy(n, N):
    from segdino3d_amd import ops
    rng = np.random.default_rng(n * 1000 + N)
    a = rng.integers(0, 2, (n, N), dtype=np.uint8)
    packed = np.packbits(a, axis=1, bitorder="little") if n else np.zeros((0, (N + 7) // 8), np.uint8)
    u = ops.unpack_bits_host(packed, N)
    assert u.dtype == np.bool_ and u.shape == (n, N) and u.flags["C_CONTIGUOUS"]
    assert np.array_equal(u.view(np.uint8), a)
    with pytest.raises(ValueError):
        ops.unpack_bits_host(np.zeros((2, 3), np.uint8), 100)


def test_packed_masks_behaves_like_the_bool_array():
    import torch
    from segdino3d_amd.architecture import PackedMasks
    rng = np.random.default_rng(3)
    a = rng.integers(0, 2, (6, 77), dtype=np.uint8).astype(bool)
    m = PackedMasks(np.packbits(a, axis=1, bitorder="little"), 77)
    assert m.shape == (6, 77) and len(m) == 6 and m.dtype == np.bool_ and m.nbytes == 6 * 10
    assert np.array_equal(np.asarray(m), a) and np.array_equal(m.unpack(), a)
    assert np.array_equal(m[2], a[2]) and np.array_equal(m[[4, 1]], a[[4, 1]]) and np.array_equal(m[1:3], a[1:3])
    assert np.array_equal(np.asarray(m, dtype=np.int64), a.astype(np.int64))
    # what the reference's evaluator does with the field (evaluator_3d.py:178)
    assert torch.equal(torch.tensor(np.asarray(m)), torch.from_numpy(a))
