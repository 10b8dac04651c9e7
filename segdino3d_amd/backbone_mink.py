"""Res16UNet34C sparse-voxel U-Net on the MI355X gather-GEMM (host side).

Mirrors the reference operator interface `segdino3d/models/backbone/minkunet.py`:
  - constructor kwargs `in_channels, out_channels, config{dilations, conv1_kernel_size, bn_momentum},
    D, voxel_size, mode_fuse_2d_feat, add_positional_embedding, **ignored` (`minkunet.py:277-298`)
  - `forward_wrapper(samples, targets, return_sp_mean_pos) -> (sp_feats[], sp_pos[], sp_pos_wo_elastic[])`
    (`minkunet.py:603-685`)
  - the same parameter / buffer names and shapes as the reference `state_dict`
    (`backbone.conv0p1s1.kernel [125,259,32]`, `backbone.bn0.bn.weight`, `backbone.block1.0.conv1.kernel`,
    `...downsample.0.kernel [Cin,Cout]`, `...downsample.1.bn.*`, `backbone.convtr4p16s2.kernel`, ...),
    so released checkpoints load with `load_state_dict`.
The nn.Module tree only HOLDS parameters; all arithmetic runs in libsegdino3d_hip.so:
conv + folded BatchNorm + ReLU (+ residual) = one `gather_gemm` launch per convolution, skip
concatenations are never materialised (two input pointers), devoxelise + superpoint mean is one
fused kernel.  `.train()` runs the same network definition on train_ops.TrainBackend (batch-statistics BatchNorm,
autograd nodes over HIP kernels).
"""
from __future__ import annotations

import math
import os
from typing import List

import torch
import torch.nn as nn

from . import _trace, ops, plan
from ._cache import DerivedWeights
from .builder import BACKBONES
from .sparse import SceneMaps

BN_EPS = 1e-5      # MinkowskiBatchNorm wraps nn.BatchNorm1d with the default eps
# SD3D_BATCH_EVAL=0: an evaluation forward of several scenes runs them one after the other (A/B switch; default = one
# block-diagonal sparse tensor, sparse.BatchSceneMaps)
import os as _os
BATCH_EVAL = _os.environ.get("SD3D_BATCH_EVAL", "1") != "0"


# SD3D_NATIVE_TRAIN_WEIGHTS=0: the round-2 training path (a permuted autograd copy per convolution and step, flipped / transposed copies in backward)
NATIVE_TRAIN_WEIGHTS = os.environ.get("SD3D_NATIVE_TRAIN_WEIGHTS", "1") != "0"


class MinkConv(nn.Module):
    """Parameter holder named like ME.MinkowskiConvolution: `kernel` [K, Cin, Cout] ([Cin, Cout] for k=1)."""

    def __init__(self, cin: int, cout: int, ksize: int):
        super().__init__()
        K = ksize ** 3
        shape = (cin, cout) if K == 1 else (K, cin, cout)
        self.kernel = nn.Parameter(torch.empty(*shape))
        std = 1.0 / math.sqrt(cin * K)
        with torch.no_grad():
            self.kernel.uniform_(-std, std)
        self.ksize = ksize


class MinkBN(nn.Module):
    """Named like ME.MinkowskiBatchNorm: the BatchNorm1d lives under `.bn` (`minkunet.py:302-304`)."""

    def __init__(self, c: int, momentum: float):
        super().__init__()
        self.bn = nn.BatchNorm1d(c, eps=BN_EPS, momentum=momentum)


class BasicBlock(nn.Module):
    def __init__(self, inplanes, planes, momentum, downsample=None):
        super().__init__()
        self.conv1 = MinkConv(inplanes, planes, 3)
        self.norm1 = MinkBN(planes, momentum)
        self.conv2 = MinkConv(planes, planes, 3)
        self.norm2 = MinkBN(planes, momentum)
        self.downsample = downsample


def _pack_conv(conv: MinkConv, pad_cin_to: int = 0) -> torch.Tensor:
    """[K, Cin, Cout] -> [K, Cout, Cin(+pad)] contiguous (the gather-GEMM weight layout)."""
    w = conv.kernel.detach()
    if w.dim() == 2:
        w = w.unsqueeze(0)
    w = w.permute(0, 2, 1)
    if pad_cin_to and pad_cin_to > w.shape[2]:
        w = torch.nn.functional.pad(w, (0, pad_cin_to - w.shape[2]))
    return w.contiguous().float()


def _fold_bn(bn: MinkBN):
    b = bn.bn
    scale = (b.weight.detach() / torch.sqrt(b.running_var + b.eps)).float().contiguous()
    shift = (b.bias.detach() - b.running_mean * scale).float().contiguous()
    return scale, shift


def _round32(c: int) -> int:
    return (c + 31) // 32 * 32


class Res16UNetBase(DerivedWeights):
    PLANES = (32, 64, 128, 256, 256, 256, 256, 256)
    LAYERS = (2, 2, 2, 2, 2, 2, 2, 2)
    INIT_DIM = 32
    KERNEL_ORDER = "x_fastest"       # MinkowskiEngine offset enumeration (see oracle/sparse_ref.py header)

    def __init__(self, in_channels, out_channels=None, config=None, D=3, voxel_size=0.02,
                 mode_fuse_2d_feat="early_fusion", add_positional_embedding=False, **kwargs):
        super().__init__()
        if D != 3:
            raise NotImplementedError("only 3-D sparse tensors are supported")
        if mode_fuse_2d_feat == "only_rgb":
            in_channels = 3
        elif mode_fuse_2d_feat != "early_fusion":
            raise NotImplementedError(f"Mode fuse 2d feat {mode_fuse_2d_feat} not implemented")
        config = config or {}
        self.voxel_size = voxel_size
        self.mode_fuse_2d_feat = mode_fuse_2d_feat
        self.add_positional_embedding = add_positional_embedding
        self.in_channels = in_channels
        self.conv1_kernel_size = int(config.get("conv1_kernel_size", 5))
        mom = float(config.get("bn_momentum", 0.1))
        P, Ly = self.PLANES, self.LAYERS

        self.inplanes = self.INIT_DIM
        self.conv0p1s1 = MinkConv(in_channels, self.inplanes, self.conv1_kernel_size)
        self.bn0 = MinkBN(self.inplanes, mom)
        for i, (cname, bname) in enumerate((("conv1p1s2", "bn1"), ("conv2p2s2", "bn2"), ("conv3p4s2", "bn3"),
                                            ("conv4p8s2", "bn4"))):
            setattr(self, cname, MinkConv(self.inplanes, self.inplanes, 2))
            setattr(self, bname, MinkBN(self.inplanes, mom))
            setattr(self, f"block{i + 1}", self._make_layer(P[i], Ly[i], mom))
        skips = (P[2], P[1], P[0], self.INIT_DIM)
        for i, (cname, bname) in enumerate((("convtr4p16s2", "bntr4"), ("convtr5p8s2", "bntr5"),
                                            ("convtr6p4s2", "bntr6"), ("convtr7p2s2", "bntr7"))):
            setattr(self, cname, MinkConv(self.inplanes, P[4 + i], 2))
            setattr(self, bname, MinkBN(P[4 + i], mom))
            self.inplanes = P[4 + i] + skips[i]
            setattr(self, f"block{5 + i}", self._make_layer(P[4 + i], Ly[4 + i], mom))
        self.out_planes = P[7]
        self._packed = None
        self._plan = None
        self._train_plan = None

    def _make_layer(self, planes, blocks, mom):
        down = None
        if self.inplanes != planes:
            down = nn.Sequential(MinkConv(self.inplanes, planes, 1), MinkBN(planes, mom))
        layers = [BasicBlock(self.inplanes, planes, mom, down)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(BasicBlock(planes, planes, mom))
        return nn.Sequential(*layers)

    # ---- weight packing ------------------------------------------------------------------------
    def _derived_reset(self):
        super()._derived_reset()
        self._packed = None
        self._plan = None
        self._train_plan = None

    def _load_from_state_dict(self, *a, **k):
        self._derived_reset()
        return super()._load_from_state_dict(*a, **k)

    def packed(self):
        """name -> (wt [K,Cout,Cin], scale, shift) in the device layout; rebuilt whenever a parameter or a BatchNorm
        running statistic changed (segdino3d_amd/_cache.py)."""
        if not self._derived_valid() or self._packed is None:
            pk = {}
            convs = {n: m for n, m in self.named_modules() if isinstance(m, MinkConv)}
            bns = {n: m for n, m in self.named_modules() if isinstance(m, MinkBN)}
            for n, m in convs.items():
                pad = _round32(self.in_channels) if n == "conv0p1s1" else 0
                pk[n] = _pack_conv(m, pad)
            for n, m in bns.items():
                pk[n] = _fold_bn(m)
            self._packed = pk
        return self._packed

    def packed_train(self):
        """name -> [K, Cout, Cin] autograd view of the live convolution parameter / the nn.BatchNorm1d module: what
        `_network` runs on with train_ops.TrainBackend (training step)."""
        from . import train_ops
        pk, kernels, pad = {}, {}, {}
        for n, m in self.named_modules():
            if isinstance(m, MinkConv):
                kernels[n] = m.kernel
                if n == "conv0p1s1" and _round32(self.in_channels) > m.kernel.shape[-2]:
                    pad[n] = _round32(self.in_channels)
            elif isinstance(m, MinkBN):
                pk[n] = m.bn
        if NATIVE_TRAIN_WEIGHTS:
            # every [K, Cin, Cout] parameter -> the [K, Cout, Cin] copy the forward kernels read, in ONE launch; the backward reads the parameters as they lie
            pk.update(train_ops.transpose_all(kernels, pad))
        else:
            for n, k in kernels.items():
                w = k if k.dim() == 3 else k.unsqueeze(0)
                w = w.permute(0, 2, 1)
                if n in pad:
                    w = torch.nn.functional.pad(w, (0, pad[n] - w.shape[2]))
                pk[n] = w.contiguous()
        return pk

    # ---- network ---------------------------------------------------------------------------------
    def _cbr(self, be, pk, x, conv, bn, key, x2=None):
        return be.conv(x, pk[conv], pk[bn], key, x2=x2, act="relu")

    def _stage(self, be, pk, name, nblocks, x, key, x2=None):
        for j in range(nblocks):
            p = f"{name}.{j}"
            h = be.conv(x, pk[p + ".conv1"], pk[p + ".norm1"], key, x2=x2, act="relu")
            if (p + ".downsample.0") in pk:
                res = be.dense(x, pk[p + ".downsample.0"], pk[p + ".downsample.1"], x2=x2)
            else:
                res = x
            x = be.conv(h, pk[p + ".conv2"], pk[p + ".norm2"], key, res=res, act="relu")
            x2 = None
        return x

    def _network(self, be, pk, x):
        """`Res16UNetBase.forward` (`minkunet.py:531-601`) against a plan backend (segdino3d_amd.plan)."""
        Ly = self.LAYERS
        k1 = self.conv1_kernel_size
        k3 = [("same", l, 3) for l in range(5)]
        dn = [("down", l) for l in range(4)]
        up = [("up", l) for l in range(4)]
        out_p1 = self._cbr(be, pk, x, "conv0p1s1", "bn0", ("same", 0, k1))
        out = self._cbr(be, pk, out_p1, "conv1p1s2", "bn1", dn[0])
        out_b1p2 = self._stage(be, pk, "block1", Ly[0], out, k3[1])
        out = self._cbr(be, pk, out_b1p2, "conv2p2s2", "bn2", dn[1])
        out_b2p4 = self._stage(be, pk, "block2", Ly[1], out, k3[2])
        out = self._cbr(be, pk, out_b2p4, "conv3p4s2", "bn3", dn[2])
        out_b3p8 = self._stage(be, pk, "block3", Ly[2], out, k3[3])
        out = self._cbr(be, pk, out_b3p8, "conv4p8s2", "bn4", dn[3])
        out = self._stage(be, pk, "block4", Ly[3], out, k3[4])
        out = self._cbr(be, pk, out, "convtr4p16s2", "bntr4", up[3])
        out = self._stage(be, pk, "block5", Ly[4], out, k3[3], x2=out_b3p8)
        out = self._cbr(be, pk, out, "convtr5p8s2", "bntr5", up[2])
        out = self._stage(be, pk, "block6", Ly[5], out, k3[2], x2=out_b2p4)
        out = self._cbr(be, pk, out, "convtr6p4s2", "bntr6", up[1])
        out = self._stage(be, pk, "block7", Ly[6], out, k3[1], x2=out_b1p2)
        out = self._cbr(be, pk, out, "convtr7p2s2", "bntr7", up[0])
        out = self._stage(be, pk, "block8", Ly[7], out, k3[0], x2=out_p1)
        return out

    def forward_sparse(self, maps: SceneMaps, vox_feats: torch.Tensor, pk=None) -> torch.Tensor:
        """`Res16UNetBase.forward` (`minkunet.py:531-601`): [V0, Cin_padded] -> [V0, 96].  pk: `self.packed()` of this forward when
        the caller already has it (`_scene_inputs` validates the derived weights while the scene's read-back travels)."""
        k1 = self.conv1_kernel_size
        use_plan = (not self.training and plan.USE_PLAN and ops.PAIR_CONV and ops.GEMM_MODE is None and ops.GG_FORCE_NT is None
                    and ops.GG_HOOK is None)
        maps.prepare(same=[(0, k1)] + [(l, 3) for l in range(5)], strides=[0, 1, 2, 3], chained=not self.training, fork=use_plan)
        if self.training:                                        # batch-statistics BatchNorm; backward through HIP kernels
            from . import train_ops, train_plan
            pk = self.packed_train()
            if (train_plan.USE_TRAIN_PLAN and NATIVE_TRAIN_WEIGHTS and torch.is_grad_enabled() and not train_ops.TrainBackend.IGNORE_ACT
                    and train_plan.supported([v for v in pk.values() if isinstance(v, nn.BatchNorm1d)])):
                # the whole U-Net as ONE autograd node over two C calls (csrc/train_plan.hip)
                if self._train_plan is None:
                    rec = train_plan.TrainRecorder(vox_feats.shape[1])
                    self._train_plan = rec.finish(self._network(rec, pk, rec.input))
                by_param = {id(v.param): v.fwd for v in pk.values() if isinstance(v, train_ops.TrainWeight)}
                return train_plan.run(self._train_plan, maps, vox_feats, [by_param[id(p)] for p in self._train_plan.params])
            return self._network(train_ops.TrainBackend(maps), pk, vox_feats)
        if pk is None:
            pk = self.packed()
        if use_plan:
            if self._plan is None:                               # one C call per scene instead of ~110
                rec = plan.Recorder(vox_feats.shape[1])
                self._plan = rec.finish(self._network(rec, pk, rec.input))
            return self._plan.run(maps, vox_feats)
        return self._network(plan.EagerBackend(maps), pk, vox_feats)

    def _scene_inputs(self, pts, tgt):
        """-> (maps, voxel features, undistorted points, superpoints, elastic?, packed weights | None) of one scene (`:604-630`)."""
        ef = tgt["extra_features"]
        pts = pts.float().contiguous()
        sp = ef["super_point_masks"].contiguous()
        if self.mode_fuse_2d_feat == "early_fusion":
            f2d, mode = ef["points_2dfeats"].float().contiguous(), 0
        else:
            f2d, mode = None, 1
        elastic = tgt["elastic_coords"] if "elastic_coords" in tgt else None
        geo = pts
        if elastic is not None:                                  # voxelise the elastically distorted scene (:606-608), colours as they are
            geo = pts.clone()
            geo[:, :3] = elastic.to(pts.device).float() * self.voxel_size
        # evaluation: the validity check of the derived weights (~0.1 ms of host time per forward, segdino3d_amd/_cache.py) runs while the
        # host would otherwise sleep on the scene's read-back, not between the table building and the first convolution
        held = []
        cin = _round32(self.in_channels)

        def hook(m):
            held.append(self.packed())
        maps = SceneMaps(geo, self.voxel_size, 5, shift_to_min=False, order=self.KERNEL_ORDER, superpoints=sp,
                         while_waiting=None if self.training else hook)
        vf = maps.voxel_features(geo, f2d, mode, cin)
        return maps, vf, pts, sp, elastic, (held[0] if held else None)

    def _positions_wo_elastic(self, pts, sp, x):
        """superpoint means of the undistorted voxel coordinates (:665-682)"""
        plain = SceneMaps(pts, self.voxel_size, 1, shift_to_min=False, order=self.KERNEL_ORDER, superpoints=sp)
        return plain.pool(x.new_zeros((plain.n_vox[0], 4)), 4)[1]

    def _forward_wrapper_batched(self, samples, targets, return_sp_mean_pos):
        """Evaluation forward of several scenes as ONE block-diagonal sparse tensor (sparse.BatchSceneMaps): one voxelisation,
        one set of neighbour tables, every convolution one launch over all scenes' pairs, one pooling launch.  Each scene's
        outputs are bit-identical to its single-scene forward."""
        from .sparse import BatchSceneMaps
        pts = [p.float().contiguous() for p in samples]
        sps = [t["extra_features"]["super_point_masks"].contiguous() for t in targets]
        if self.mode_fuse_2d_feat == "early_fusion":
            f2d, mode = [t["extra_features"]["points_2dfeats"].float().contiguous() for t in targets], 0
        else:
            f2d, mode = None, 1
        maps = BatchSceneMaps(pts, self.voxel_size, 5, shift_to_min=False, order=self.KERNEL_ORDER, superpoints=sps)
        cap = _trace.active()
        if cap is not None:
            cap.record_maps([maps])
        vf = maps.voxel_features(pts, f2d, mode, _round32(self.in_channels))
        x = self.forward_sparse(maps, vf)
        f_all, p_all = maps.pool(x, self.out_planes)
        cuts = list(zip(maps.sp_off[:-1], maps.sp_off[1:]))
        feats = [f_all[a:b] for a, b in cuts]
        pos = [p_all[a:b] for a, b in cuts]
        pos_wo = [p.clone() for p in pos]
        sp_pos = pos if self.add_positional_embedding else None
        return (feats, sp_pos, pos_wo) if return_sp_mean_pos else (feats, sp_pos, None)

    @ops.bound_stream
    def forward_wrapper(self, samples: List[torch.Tensor], targets, return_sp_mean_pos=False):
        if (not self.training and 1 < len(samples) <= 16 and BATCH_EVAL
                and not any("elastic_coords" in t for t in targets)):
            return self._forward_wrapper_batched(samples, targets, return_sp_mean_pos)
        feats, pos, pos_wo = [], [], []
        scenes = [self._scene_inputs(p, t) for p, t in zip(samples, targets)]
        cap = _trace.active()
        if cap is not None:
            cap.record_maps([s[0] for s in scenes])
        if self.training and len(scenes) > 1:
            # one block-diagonal tensor for the whole batch, as ME's batch_sparse_collate builds (:624-627): convolutions stay
            # within their scene, every BatchNorm sees the voxels of all scenes
            from . import train_ops
            from .sparse import BatchedMaps
            batch = BatchedMaps([s[0] for s in scenes])
            x_all = self.forward_sparse(batch, torch.cat([s[1] for s in scenes], dim=0))
            outs = [x_all[slice(*batch.rows(0, i))] for i in range(len(scenes))]
        else:
            outs = []
            for maps, vf, _, _, _, pk in scenes:
                outs.append(self.forward_sparse(maps, vf, pk))
        for (maps, _, pts, sp, elastic, _), x in zip(scenes, outs):
            if self.training:
                from . import train_ops
                f, p = train_ops.pool_superpoints(x.contiguous(), maps, self.out_planes)
            else:
                f, p = maps.pool(x, self.out_planes)
            feats.append(f)
            pos.append(p)
            if elastic is None:                                  # no distortion: the "without elastic" positions are the same values
                pos_wo.append(p.clone())
            elif return_sp_mean_pos:
                pos_wo.append(self._positions_wo_elastic(pts, sp, x))
        sp_pos = pos if self.add_positional_embedding else None
        if return_sp_mean_pos:
            return feats, sp_pos, pos_wo
        return feats, sp_pos, None


class Res16UNet34(Res16UNetBase):
    LAYERS = (2, 3, 4, 6, 2, 2, 2, 2)


@BACKBONES.register_module()
class Res16UNet34C(Res16UNet34):
    PLANES = (32, 64, 128, 256, 256, 128, 96, 96)
