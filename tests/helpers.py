"""Shared test helpers (numpy Z-order codec, oracle <-> device voxel matching)."""
import numpy as np
import torch


def _compact3(x):
    x = x & np.uint64(0x1249249249249249)
    x = (x ^ (x >> np.uint64(2))) & np.uint64(0x10c30c30c30c30c3)
    x = (x ^ (x >> np.uint64(4))) & np.uint64(0x100f00f00f00f00f)
    x = (x ^ (x >> np.uint64(8))) & np.uint64(0x001f0000ff0000ff)
    x = (x ^ (x >> np.uint64(16))) & np.uint64(0x001f00000000ffff)
    x = (x ^ (x >> np.uint64(32))) & np.uint64(0x00000000001fffff)
    return x


def morton_decode_np(keys: np.ndarray) -> np.ndarray:
    k = keys.astype(np.uint64) & np.uint64((1 << 48) - 1)
    return np.stack([_compact3(k), _compact3(k >> np.uint64(1)), _compact3(k >> np.uint64(2))], axis=1).astype(np.int64)


def device_level_coords(maps, level: int) -> np.ndarray:
    """Absolute integer coordinates (in units of voxels at stride 1) of the device's level-`level` voxels."""
    keys = maps.keys[level].cpu().numpy().view(np.uint64)
    origin = maps.origin.cpu().numpy().astype(np.int64)
    c = morton_decode_np(keys)
    return (c << level) + origin[None, :]


def match_rows(dev_coords: np.ndarray, ref_coords: np.ndarray) -> np.ndarray:
    """perm with ref_coords[perm[i]] == dev_coords[i]; asserts the two sets are identical."""
    from oracle.sparse_ref import pack
    dk, rk = pack(dev_coords), pack(ref_coords)
    assert len(dk) == len(rk), f"voxel count differs: device {len(dk)} vs oracle {len(rk)}"
    order = np.argsort(rk, kind="stable")
    pos = np.searchsorted(rk[order], dk)
    pos = np.minimum(pos, len(rk) - 1)
    assert (rk[order][pos] == dk).all(), "device voxel set differs from the oracle's"
    return order[pos]


def pairs_from_nbr(nbr: torch.Tensor):
    """device nbr [K, V_out] -> list over k of (in_idx, out_idx) numpy arrays."""
    t = nbr.cpu().numpy()
    out = []
    for k in range(t.shape[0]):
        o = np.nonzero(t[k] >= 0)[0]
        out.append((t[k][o].astype(np.int64), o.astype(np.int64)))
    return out
