"""ORACLE (test infrastructure, not product code): CPU restatement of the sparse-voxel backbones.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.

PARITY UNPINNED against MinkowskiEngine / spconv themselves: their arithmetic lives in CUDA-only
third-party libraries that are neither vendored in /root/reference nor installable here
(MinkowskiEngine: unpinned git master + manual patches, `installation.md:16-33`;
spconv-cu121==2.3.8 / cumm-cu121==0.7.11, `installation.md:80,99`; torch-scatter==2.1.2,
`installation.md:56`), and the reference holds no test or golden vector at this boundary.
What IS followed line by line is the reference's own network code:
  - `segdino3d/models/backbone/minkunet.py:352-529`  topology / channel plan of Res16UNet34C
  - `.../minkunet.py:531-601`   forward (encoder, transposed-conv decoder, skip concatenation)
  - `.../minkunet.py:195-250`   BasicBlock (conv-BN-ReLU-conv-BN, 1x1 downsample, add, ReLU)
  - `.../minkunet.py:603-685`   forward_wrapper (feature assembly, voxelise, slice, superpoint mean)
  - `segdino3d/models/backbone/spconvunet.py:21-99, 116-268, 270-399`  SpConvUNet (ScanNetv2)
and the published library semantics:
  - float coords are floor-quantised to int32; a voxel's feature is the unweighted mean of its points
  - odd kernels are centred, even kernels use offsets {0,1} * tensor_stride
  - out(u) = sum_k W_k . in(u + offset_k * tensor_stride)   (correlation, no flip)
  - stride-2 convolution creates output coords floor(c / 2s) * 2s
  - transposed convolution writes onto the EXISTING coordinate map of the finer stride
  - MinkowskiEngine enumerates kernel offsets with the first spatial axis (x) fastest;
    spconv weights are [C_out, k0, k1, k2, C_in] i.e. the last axis (z) fastest
  - BatchNorm eps 1e-5 (ME wrapper of nn.BatchNorm1d) vs 1e-4 (spconv path, spconvunet.py:35-36)
  - spconv SparseConv3d output extent (D - k) // s + 1 with D = clip(max + 1, min=128)
The kernel-offset order is the one unobservable choice (it only shows through released
checkpoints); it is a single table (`kernel_offsets`) shared with the product's tests.
"""
from __future__ import annotations

import numpy as np
import torch

# ------------------------------------------------------------------------------------------------
# integer coordinate machinery (numpy int64 keys)
# ------------------------------------------------------------------------------------------------
_B = 1 << 20      # per-axis bias so that packed keys are non-negative
_S = 1 << 21


def pack(c: np.ndarray) -> np.ndarray:
    c = c.astype(np.int64) + _B
    return (c[:, 0] * _S + c[:, 1]) * _S + c[:, 2]


def floor_voxel(xyz: torch.Tensor, voxel_size: float) -> np.ndarray:
    """batch_sparse_collate on float coords (minkunet.py:624-626): floor(xyz / voxel_size) as int32.
    The reference evaluates `coords / self.voxel_size` on a CUDA tensor, where ATen's
    div-by-CPU-scalar kernel multiplies by the fp32 reciprocal (1/0.02f rounds to exactly 50.0f); that
    deployed behaviour is what is restated here (a true division differs only for coordinates that are
    exact multiples of the voxel size)."""
    inv = float(np.float32(1.0) / np.float32(voxel_size))
    return torch.floor(xyz * inv).to(torch.int32).numpy()


def unique_voxels(c: np.ndarray):
    """-> (unique coords [V,3] sorted by packed key, inverse [N] point->voxel)."""
    keys = pack(c)
    uk, first, inverse = np.unique(keys, return_index=True, return_inverse=True)
    return c[first].astype(np.int32), inverse.astype(np.int64)


def segment_mean(feats: torch.Tensor, index: np.ndarray, n: int) -> torch.Tensor:
    idx = torch.from_numpy(np.asarray(index, dtype=np.int64))
    out = torch.zeros((n,) + tuple(feats.shape[1:]), dtype=feats.dtype)
    out.index_add_(0, idx, feats)
    cnt = torch.bincount(idx, minlength=n).clamp(min=1).to(feats.dtype)
    return out / cnt[:, None]


def kernel_offsets(ksize: int, order: str = "x_fastest") -> np.ndarray:
    """[K,3] integer offsets in units of the tensor stride.  Odd: centred; even: {0..k-1}."""
    r = np.arange(ksize) - (ksize - 1) // 2 if ksize % 2 else np.arange(ksize)
    if order == "x_fastest":            # MinkowskiEngine: k = ix + K*(iy + K*iz)
        z, y, x = np.meshgrid(r, r, r, indexing="ij")
    elif order == "z_fastest":          # spconv [k0,k1,k2] row-major: k = (ix*K + iy)*K + iz
        x, y, z = np.meshgrid(r, r, r, indexing="ij")
    else:
        raise ValueError(order)
    return np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1).astype(np.int64)


def kernel_map(in_coords: np.ndarray, out_coords: np.ndarray, offsets: np.ndarray, scale: int):
    """For every offset k: (in_idx, out_idx) with in_coords[in_idx] == out_coords[out_idx] + off_k*scale."""
    in_keys = pack(in_coords)
    order = np.argsort(in_keys, kind="stable")
    sk = in_keys[order]
    pairs = []
    for off in offsets:
        q = pack(out_coords.astype(np.int64) + off[None, :] * scale)
        pos = np.searchsorted(sk, q)
        pos_c = np.minimum(pos, len(sk) - 1)
        hit = sk[pos_c] == q
        pairs.append((order[pos_c[hit]], np.nonzero(hit)[0]))
    return pairs


def sparse_conv(feat: torch.Tensor, pairs, weight: torch.Tensor, n_out: int) -> torch.Tensor:
    """weight [K, Cin, Cout]; out[o] += feat[i] @ W[k] over the pairs of offset k."""
    out = torch.zeros(n_out, weight.shape[-1], dtype=feat.dtype)
    for k, (i_idx, o_idx) in enumerate(pairs):
        if len(i_idx) == 0:
            continue
        out.index_add_(0, torch.from_numpy(o_idx), feat[torch.from_numpy(i_idx)] @ weight[k])
    return out


def downsample_coords(coords: np.ndarray, new_stride: int) -> np.ndarray:
    c = np.floor_divide(coords.astype(np.int64), new_stride) * new_stride
    return unique_voxels(c.astype(np.int32))[0]


BN_TRAIN = False    # tests of the training step switch the BatchNorms to batch statistics (biased variance)


def bn_eval(x, sd, name, eps):
    w, b = sd[name + ".weight"], sd[name + ".bias"]
    if BN_TRAIN:
        m, v = x.mean(dim=0), x.var(dim=0, unbiased=False)
    else:
        m, v = sd[name + ".running_mean"], sd[name + ".running_var"]
    return (x - m) / torch.sqrt(v + eps) * w + b


# ------------------------------------------------------------------------------------------------
# Res16UNet34C (MinkowskiEngine semantics)
# ------------------------------------------------------------------------------------------------
MINK_PLANES = (32, 64, 128, 256, 256, 128, 96, 96)     # minkunet.py:694
MINK_LAYERS = (2, 3, 4, 6, 2, 2, 2, 2)                 # minkunet.py:689
MINK_EPS = 1e-5


class MinkLevels:
    """Coordinate maps at tensor strides 1, 2, 4, 8, 16 plus cached kernel maps."""

    def __init__(self, coords1: np.ndarray, order="x_fastest"):
        self.order = order
        self.coords = {1: coords1}
        s = 1
        while s < 16:
            self.coords[2 * s] = downsample_coords(self.coords[s], 2 * s)
            s *= 2
        self._cache = {}

    def n(self, stride):
        return len(self.coords[stride])

    def same(self, stride, ksize):
        key = ("same", stride, ksize)
        if key not in self._cache:
            c = self.coords[stride]
            self._cache[key] = kernel_map(c, c, kernel_offsets(ksize, self.order), stride)
        return self._cache[key]

    def down(self, stride):
        """k=2 s=2 conv from `stride` to 2*stride: in = out + off*stride."""
        key = ("down", stride)
        if key not in self._cache:
            self._cache[key] = kernel_map(self.coords[stride], self.coords[2 * stride],
                                          kernel_offsets(2, self.order), stride)
        return self._cache[key]

    def up(self, stride):
        """transposed k=2 s=2 conv from 2*stride onto the existing `stride` map: the forward
        conv's map with in/out swapped, same kernel index."""
        return [(o, i) for (i, o) in self.down(stride)]


def _mink_block(x, sd, p, lv: MinkLevels, stride):
    """BasicBlock (minkunet.py:234-250)."""
    n = x.shape[0]
    out = sparse_conv(x, lv.same(stride, 3), sd[p + ".conv1.kernel"], n)
    out = torch.relu(bn_eval(out, sd, p + ".norm1.bn", MINK_EPS))
    out = sparse_conv(out, lv.same(stride, 3), sd[p + ".conv2.kernel"], n)
    out = bn_eval(out, sd, p + ".norm2.bn", MINK_EPS)
    if (p + ".downsample.0.kernel") in sd:
        res = bn_eval(x @ sd[p + ".downsample.0.kernel"], sd, p + ".downsample.1.bn", MINK_EPS)
    else:
        res = x
    return torch.relu(out + res)


def _mink_stage(x, sd, p, n_blocks, lv, stride):
    for j in range(n_blocks):
        x = _mink_block(x, sd, f"{p}.{j}", lv, stride)
    return x


def res16unet34c(sd, coords1: np.ndarray, feats: torch.Tensor, prefix="backbone.", conv1_kernel_size=5,
                 order="x_fastest", return_levels=False):
    """Res16UNetBase.forward (minkunet.py:531-601) on one coordinate map.  -> [V1, 96]."""
    lv = MinkLevels(coords1, order)
    P = prefix

    def cbr(x, conv, bn, pairs, n_out):
        return torch.relu(bn_eval(sparse_conv(x, pairs, sd[P + conv + ".kernel"], n_out), sd, P + bn + ".bn", MINK_EPS))

    out_p1 = cbr(feats, "conv0p1s1", "bn0", lv.same(1, conv1_kernel_size), lv.n(1))
    out = cbr(out_p1, "conv1p1s2", "bn1", lv.down(1), lv.n(2))
    out_b1p2 = _mink_stage(out, sd, P + "block1", MINK_LAYERS[0], lv, 2)
    out = cbr(out_b1p2, "conv2p2s2", "bn2", lv.down(2), lv.n(4))
    out_b2p4 = _mink_stage(out, sd, P + "block2", MINK_LAYERS[1], lv, 4)
    out = cbr(out_b2p4, "conv3p4s2", "bn3", lv.down(4), lv.n(8))
    out_b3p8 = _mink_stage(out, sd, P + "block3", MINK_LAYERS[2], lv, 8)
    out = cbr(out_b3p8, "conv4p8s2", "bn4", lv.down(8), lv.n(16))
    out = _mink_stage(out, sd, P + "block4", MINK_LAYERS[3], lv, 16)
    out = cbr(out, "convtr4p16s2", "bntr4", lv.up(8), lv.n(8))
    out = _mink_stage(torch.cat([out, out_b3p8], 1), sd, P + "block5", MINK_LAYERS[4], lv, 8)
    out = cbr(out, "convtr5p8s2", "bntr5", lv.up(4), lv.n(4))
    out = _mink_stage(torch.cat([out, out_b2p4], 1), sd, P + "block6", MINK_LAYERS[5], lv, 4)
    out = cbr(out, "convtr6p4s2", "bntr6", lv.up(2), lv.n(2))
    out = _mink_stage(torch.cat([out, out_b1p2], 1), sd, P + "block7", MINK_LAYERS[6], lv, 2)
    out = cbr(out, "convtr7p2s2", "bntr7", lv.up(1), lv.n(1))
    out = _mink_stage(torch.cat([out, out_p1], 1), sd, P + "block8", MINK_LAYERS[7], lv, 1)
    return (out, lv) if return_levels else out


def mink_state_dict_shapes(in_channels=259, conv1_kernel_size=5):
    """Key -> shape of the reference `backbone.*` state_dict."""
    shapes = {}

    def bn(name, c):
        for leaf, shp in (("weight", (c,)), ("bias", (c,)), ("running_mean", (c,)), ("running_var", (c,)),
                          ("num_batches_tracked", ())):
            shapes[f"{name}.bn.{leaf}"] = shp

    def stage(name, inplanes, planes, nblocks):
        for j in range(nblocks):
            cin = inplanes if j == 0 else planes
            shapes[f"{name}.{j}.conv1.kernel"] = (27, cin, planes)
            bn(f"{name}.{j}.norm1", planes)
            shapes[f"{name}.{j}.conv2.kernel"] = (27, planes, planes)
            bn(f"{name}.{j}.norm2", planes)
            if j == 0 and cin != planes:
                shapes[f"{name}.{j}.downsample.0.kernel"] = (cin, planes)
                bn(f"{name}.{j}.downsample.1", planes)
        return planes

    Pl, Ly = MINK_PLANES, MINK_LAYERS
    shapes["conv0p1s1.kernel"] = (conv1_kernel_size ** 3, in_channels, 32); bn("bn0", 32)
    inpl = 32
    for idx, (cname, bname) in enumerate((("conv1p1s2", "bn1"), ("conv2p2s2", "bn2"), ("conv3p4s2", "bn3"),
                                          ("conv4p8s2", "bn4"))):
        shapes[cname + ".kernel"] = (8, inpl, inpl); bn(bname, inpl)
        inpl = stage(f"block{idx + 1}", inpl, Pl[idx], Ly[idx])
    skips = (Pl[2], Pl[1], Pl[0], 32)
    for idx, (cname, bname) in enumerate((("convtr4p16s2", "bntr4"), ("convtr5p8s2", "bntr5"),
                                          ("convtr6p4s2", "bntr6"), ("convtr7p2s2", "bntr7"))):
        shapes[cname + ".kernel"] = (8, inpl, Pl[4 + idx]); bn(bname, Pl[4 + idx])
        inpl = stage(f"block{5 + idx}", Pl[4 + idx] + skips[idx], Pl[4 + idx], Ly[4 + idx])
    return shapes


def mink_forward_wrapper(sd, points: torch.Tensor, feats2d, superpoints: torch.Tensor, voxel_size=0.02,
                         mode="early_fusion", prefix="backbone.", conv1_kernel_size=5, order="x_fastest", elastic=None):
    """forward_wrapper (minkunet.py:603-685), eval (no elastic coords), ONE scene.
    -> (sp_feats [S,96], sp_pos [S,3], sp_pos_wo_elastic [S,3])."""
    xyz = points[:, :3]
    f = points[:, 3:]
    if mode == "early_fusion":
        f = torch.cat([f, feats2d], dim=1)
    # train-time: the scene is voxelised at its elastically distorted coordinates (voxel units * voxel_size, :606-608)
    c = floor_voxel(xyz if elastic is None else elastic.float() * voxel_size, voxel_size)
    uc, inv = unique_voxels(c)
    vf = segment_mean(f, inv, len(uc))
    x = res16unet34c(sd, uc, vf, prefix, conv1_kernel_size, order)
    x = x[torch.from_numpy(inv)]                                   # .slice(field)
    S = int(superpoints.max()) + 1
    sp_feats = segment_mean(x, superpoints.numpy(), S)
    sp_pos = segment_mean(torch.from_numpy(c).float() * voxel_size, superpoints.numpy(), S)
    if elastic is None:
        return sp_feats, sp_pos, sp_pos.clone()
    c0 = floor_voxel(xyz, voxel_size)                               # :665-682
    return sp_feats, sp_pos, segment_mean(torch.from_numpy(c0).float() * voxel_size, superpoints.numpy(), S)


# ------------------------------------------------------------------------------------------------
# SpConvUNet (spconv semantics) - ScanNetv2 prototype
# ------------------------------------------------------------------------------------------------
SPCONV_EPS = 1e-4


def _spw(w: torch.Tensor) -> torch.Tensor:
    """spconv weight [Cout, k0, k1, k2, Cin] -> [K, Cin, Cout] with K enumerated z-fastest."""
    co = w.shape[0]
    ci = w.shape[-1]
    return w.reshape(co, -1, ci).permute(1, 2, 0).contiguous()


class SpLevels:
    def __init__(self, coords1: np.ndarray, n_levels: int, min_spatial_shape=128):
        self.coords = [coords1]
        self.pairs_down = []
        shape = np.maximum(coords1.max(0) + 1, min_spatial_shape).astype(np.int64)
        for _ in range(n_levels - 1):
            c = self.coords[-1].astype(np.int64)
            out_shape = (shape - 2) // 2 + 1
            o = c // 2
            valid = (o < out_shape[None, :]).all(1)
            oc, _ = unique_voxels(o[valid].astype(np.int32))
            self.pairs_down.append(self._down_pairs(c, valid, oc))
            self.coords.append(oc)
            shape = out_shape
        self._same = {}

    @staticmethod
    def _down_pairs(c, valid, oc):
        """pairs per offset k=(c - 2*o) z-fastest: (in_idx, out_idx)."""
        ok = pack(oc)
        order = np.argsort(ok, kind="stable")
        pairs = []
        o = c // 2
        rel = c - 2 * o
        kidx = (rel[:, 0] * 2 + rel[:, 1]) * 2 + rel[:, 2]
        pos = np.searchsorted(ok[order], pack(o))
        pos = np.minimum(pos, len(ok) - 1)
        for k in range(8):
            sel = np.nonzero(valid & (kidx == k))[0]
            pairs.append((sel, order[pos[sel]]))
        return pairs

    def same(self, level, ksize=3):
        key = (level, ksize)
        if key not in self._same:
            c = self.coords[level]
            self._same[key] = kernel_map(c, c, kernel_offsets(ksize, "z_fastest"), 1)
        return self._same[key]


def _sp_resblock(x, sd, p, pairs, normalize_before=True):
    """ResidualBlock (spconvunet.py:48-81, 82-99): BN, ReLU, conv, BN, ReLU, conv when normalising before; conv, BN, ReLU,
    conv, BN, ReLU otherwise - either way `conv_branch(x) + i_branch(x)`, so the second form adds the identity AFTER its ReLU."""
    n = x.shape[0]
    if (p + ".i_branch.0.weight") in sd:
        ident = x @ _spw(sd[p + ".i_branch.0.weight"])[0]
    else:
        ident = x
    if not normalize_before:
        h = sparse_conv(x, pairs, _spw(sd[p + ".conv_branch.0.weight"]), n)
        h = torch.relu(bn_eval(h, sd, p + ".conv_branch.1", SPCONV_EPS))
        h = sparse_conv(h, pairs, _spw(sd[p + ".conv_branch.3.weight"]), n)
        h = torch.relu(bn_eval(h, sd, p + ".conv_branch.4", SPCONV_EPS))
        return h + ident
    h = torch.relu(bn_eval(x, sd, p + ".conv_branch.0", SPCONV_EPS))
    h = sparse_conv(h, pairs, _spw(sd[p + ".conv_branch.2.weight"]), n)
    h = torch.relu(bn_eval(h, sd, p + ".conv_branch.3", SPCONV_EPS))
    h = sparse_conv(h, pairs, _spw(sd[p + ".conv_branch.5.weight"]), n)
    return h + ident


def _sp_unet(x, sd, p, lv: SpLevels, level, n_levels, block_reps=2, normalize_before=True):
    """SpConvUNet.forward (spconvunet.py:233-268), recursive; `conv` / `deconv` are BN, ReLU, conv or conv, BN, ReLU (:154-201)."""
    pairs = lv.same(level)
    nb = normalize_before
    for r in range(block_reps):
        x = _sp_resblock(x, sd, f"{p}blocks.block{r}", pairs, nb)
    if level < n_levels - 1:
        ident = x
        down = lv.pairs_down[level]
        up = [(o, i) for (i, o) in down]
        if nb:
            h = torch.relu(bn_eval(x, sd, p + "conv.0", SPCONV_EPS))
            h = sparse_conv(h, down, _spw(sd[p + "conv.2.weight"]), len(lv.coords[level + 1]))
            h = _sp_unet(h, sd, p + "u.", lv, level + 1, n_levels, block_reps, nb)
            h = torch.relu(bn_eval(h, sd, p + "deconv.0", SPCONV_EPS))
            h = sparse_conv(h, up, _spw(sd[p + "deconv.2.weight"]), x.shape[0])
        else:
            h = sparse_conv(x, down, _spw(sd[p + "conv.0.weight"]), len(lv.coords[level + 1]))
            h = torch.relu(bn_eval(h, sd, p + "conv.1", SPCONV_EPS))
            h = _sp_unet(h, sd, p + "u.", lv, level + 1, n_levels, block_reps, nb)
            h = sparse_conv(h, up, _spw(sd[p + "deconv.0.weight"]), x.shape[0])
            h = torch.relu(bn_eval(h, sd, p + "deconv.1", SPCONV_EPS))
        x = torch.cat([ident, h], dim=1)
        for r in range(block_reps):
            x = _sp_resblock(x, sd, f"{p}blocks_tail.block{r}", pairs, nb)
    return x


def spconv_state_dict_shapes(num_planes=(32, 64, 96, 128, 160), in_channels=262, block_reps=2, normalize_before=True):
    shapes = {}
    nb = normalize_before

    def bn(name, c):
        for leaf, shp in (("weight", (c,)), ("bias", (c,)), ("running_mean", (c,)), ("running_var", (c,)),
                          ("num_batches_tracked", ())):
            shapes[f"{name}.{leaf}"] = shp

    def resblock(name, cin, cout):
        if cin != cout:
            shapes[name + ".i_branch.0.weight"] = (cout, 1, 1, 1, cin)
        if nb:
            bn(name + ".conv_branch.0", cin)
            shapes[name + ".conv_branch.2.weight"] = (cout, 3, 3, 3, cin)
            bn(name + ".conv_branch.3", cout)
            shapes[name + ".conv_branch.5.weight"] = (cout, 3, 3, 3, cout)
        else:
            shapes[name + ".conv_branch.0.weight"] = (cout, 3, 3, 3, cin)
            bn(name + ".conv_branch.1", cout)
            shapes[name + ".conv_branch.3.weight"] = (cout, 3, 3, 3, cout)
            bn(name + ".conv_branch.4", cout)

    def unet(p, planes):
        for r in range(block_reps):
            resblock(f"{p}blocks.block{r}", planes[0], planes[0])
        if len(planes) > 1:
            if nb:
                bn(p + "conv.0", planes[0])
                shapes[p + "conv.2.weight"] = (planes[1], 2, 2, 2, planes[0])
            else:
                shapes[p + "conv.0.weight"] = (planes[1], 2, 2, 2, planes[0])
                bn(p + "conv.1", planes[1])
            unet(p + "u.", planes[1:])
            if nb:
                bn(p + "deconv.0", planes[1])
                shapes[p + "deconv.2.weight"] = (planes[0], 2, 2, 2, planes[1])
            else:
                shapes[p + "deconv.0.weight"] = (planes[0], 2, 2, 2, planes[1])
                bn(p + "deconv.1", planes[0])
            for r in range(block_reps):
                resblock(f"{p}blocks_tail.block{r}", planes[0] * (2 - r), planes[0])

    shapes["input_conv.0.weight"] = (32, 3, 3, 3, in_channels)
    unet("", tuple(num_planes))
    bn("output_layer.0", 32)
    return shapes


def spconv_forward_wrapper(sd, points, feats2d, superpoints, voxel_size=0.02, num_planes=(32, 64, 96, 128, 160),
                           prefix="backbone.", min_spatial_shape=128, elastic=None, normalize_before=True):
    """forward_wrapper + collate (spconvunet.py:364-399, 270-362), eval, early_fusion, ONE scene."""
    sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    xyz = points[:, :3]
    f = torch.cat([points[:, 3:], xyz - xyz.mean(0), feats2d], dim=1)
    if elastic is None:
        c = floor_voxel(xyz - xyz.min(0)[0], voxel_size)
    else:                                                           # already in voxel units (spconvunet.py:291-294)
        el = elastic.float()
        c = torch.floor(el - el.min(0)[0]).to(torch.int32).numpy()
    uc, inv = unique_voxels(c)
    vf = segment_mean(f, inv, len(uc))
    lv = SpLevels(uc, len(num_planes), min_spatial_shape)
    x = sparse_conv(vf, lv.same(0), _spw(sd["input_conv.0.weight"]), len(uc))
    x = _sp_unet(x, sd, "", lv, 0, len(num_planes), normalize_before=normalize_before)
    x = torch.relu(bn_eval(x, sd, "output_layer.0", SPCONV_EPS))
    S = int(superpoints.max()) + 1
    sp_feats = segment_mean(x[torch.from_numpy(inv)], superpoints.numpy(), S)
    cq = floor_voxel(xyz, voxel_size)
    sp_pos = segment_mean(torch.from_numpy(cq).float() * voxel_size, superpoints.numpy(), S)
    if elastic is None:
        return sp_feats, sp_pos, sp_pos.clone()
    ce = torch.floor(elastic.float()).to(torch.int32)               # :337-352
    return sp_feats, segment_mean(ce.float() * voxel_size, superpoints.numpy(), S), sp_pos
