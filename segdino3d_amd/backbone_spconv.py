"""SpConvUNet (ScanNetv2 prototype) on the MI355X gather-GEMM (host side).

Mirrors the reference operator interface `segdino3d/models/backbone/spconvunet.py`:
  - constructor kwargs `num_planes, norm_fn, block_reps, block, indice_key_id, normalize_before,
    return_blocks, voxel_size, mode_fuse_2d_feat, main_model, min_spatial_shape,
    add_positional_embedding` (:116-129);
  - `forward_wrapper(samples, targets, return_sp_mean_pos) -> (sp_feats[], sp_pos[], sp_pos_wo_elastic[])`
    (:364-399);
  - the reference `state_dict` keys: `input_conv.0.weight [32,3,3,3,262]`,
    `blocks.block{r}.conv_branch.{0,3}.*` (BatchNorm1d), `.conv_branch.{2,5}.weight`,
    `.i_branch.0.weight`, `conv.{0,2}`, `u.<recursive>`, `deconv.{0,2}`, `blocks_tail.block{r}...`,
    `output_layer.0.*` (spconv 2.x weight layout [C_out, k0, k1, k2, C_in]).
Sub-manifold / strided / inverse convolutions all run as `gather_gemm` over neighbour tables of the
Z-order coordinate hierarchy; the pre-activation BatchNorm+ReLU is one elementwise pass, the BN+ReLU
between the two convolutions of a block is fused into the first convolution's epilogue and the
residual add into the second's.  Eval mode, `early_fusion`, no elastic coordinates.

spconv's output-extent rule `(D - 2) // 2 + 1` (the trailing slice of an odd-sized grid has no parent
voxel; `spatial_shape = clip(max + 1, min_spatial_shape)`, :309-310) is applied when the coarser
levels are created (`SceneMaps(clip_min_shape=...)`).
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from . import _trace, ops, plan
from ._cache import DerivedWeights
from .builder import BACKBONES
from .sparse import SceneMaps

BN_EPS = 1e-4       # spconvunet.py:35-36, 228


class SpConv(nn.Module):
    """Parameter holder for spconv SubMConv3d / SparseConv3d / SparseInverseConv3d: `weight` [Cout,k,k,k,Cin]."""

    def __init__(self, cin, cout, ksize):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, ksize, ksize, ksize, cin))
        nn.init.kaiming_uniform_(self.weight.view(cout, -1), a=5 ** 0.5)
        self.ksize = ksize


def _bn(c):
    return nn.BatchNorm1d(c, eps=BN_EPS, momentum=0.1)


class ResidualBlock(nn.Module):
    """Parameter holder with the reference's Sequential indices (`spconvunet.py:41-81`): normalize_before=True is
    BN, ReLU, conv, BN, ReLU, conv (BatchNorms at 0 / 3, convolutions at 2 / 5); False is conv, BN, ReLU, conv, BN, ReLU
    (convolutions at 0 / 3, BatchNorms at 1 / 4)."""

    def __init__(self, cin, cout, normalize_before=True):
        super().__init__()
        self.i_branch = nn.Sequential(nn.Identity() if cin == cout else SpConv(cin, cout, 1))
        if normalize_before:
            self.conv_branch = nn.Sequential(_bn(cin), nn.ReLU(), SpConv(cin, cout, 3), _bn(cout), nn.ReLU(), SpConv(cout, cout, 3))
        else:
            self.conv_branch = nn.Sequential(SpConv(cin, cout, 3), _bn(cout), nn.ReLU(), SpConv(cout, cout, 3), _bn(cout), nn.ReLU())


def _pack(conv: SpConv) -> torch.Tensor:
    w = conv.weight.detach()
    co, ci = w.shape[0], w.shape[-1]
    w = w.reshape(co, -1, ci).permute(1, 0, 2)                 # [K (z fastest), Cout, Cin]
    pad = (ci + 31) // 32 * 32 - ci
    if pad:
        w = torch.nn.functional.pad(w, (0, pad))
    return w.contiguous().float()


def _fold(bn: nn.BatchNorm1d):
    scale = (bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)).float().contiguous()
    shift = (bn.bias.detach() - bn.running_mean * scale).float().contiguous()
    return scale, shift


@BACKBONES.register_module()
class SpConvUNet(DerivedWeights):
    KERNEL_ORDER = "z_fastest"

    def __init__(self, num_planes, norm_fn=None, block_reps=2, block=None, indice_key_id=1, normalize_before=True,
                 return_blocks=False, voxel_size=0.02, mode_fuse_2d_feat="early_fusion", main_model=True,
                 min_spatial_shape=128, add_positional_embedding=False):
        super().__init__()
        if block not in (None, "residual") and getattr(block, "__name__", "") != "ResidualBlock":
            raise NotImplementedError(f"segdino3d_amd SpConvUNet: block {block!r} is not built (the reference ships 'residual' only)")
        self.normalize_before = bool(normalize_before)
        nb = self.normalize_before
        self.return_blocks = return_blocks
        self.num_planes = list(num_planes)
        self.block_reps = block_reps
        p = self.num_planes
        self.blocks = nn.ModuleDict({f"block{i}": ResidualBlock(p[0], p[0], nb) for i in range(block_reps)})
        if len(p) > 1:
            # spconvunet.py:154-201: BN, ReLU, conv when normalising before; conv, BN, ReLU otherwise
            self.conv = (nn.Sequential(_bn(p[0]), nn.ReLU(), SpConv(p[0], p[1], 2)) if nb else
                         nn.Sequential(SpConv(p[0], p[1], 2), _bn(p[1]), nn.ReLU()))
            self.u = SpConvUNet(p[1:], block_reps=block_reps, indice_key_id=indice_key_id + 1, normalize_before=nb,
                                return_blocks=return_blocks, main_model=False)
            self.deconv = (nn.Sequential(_bn(p[1]), nn.ReLU(), SpConv(p[1], p[0], 2)) if nb else
                           nn.Sequential(SpConv(p[1], p[0], 2), _bn(p[0]), nn.ReLU()))
            self.blocks_tail = nn.ModuleDict({f"block{i}": ResidualBlock(p[0] * (2 - i), p[0], nb) for i in range(block_reps)})
        self.mode_fuse_2d_feat = mode_fuse_2d_feat
        self.voxel_size = voxel_size
        self.min_spatial_shape = min_spatial_shape
        self.main_model = main_model
        if main_model:
            if not mode_fuse_2d_feat.startswith("early_fusion"):
                raise NotImplementedError("segdino3d_amd SpConvUNet: only early_fusion is supported (SURVEY q21)")
            self.in_channels = 256 + 6
            self.input_conv = nn.Sequential(SpConv(self.in_channels, 32, 3))
            self.output_layer = nn.Sequential(_bn(32), nn.ReLU(inplace=True))
        self.add_positional_embedding = add_positional_embedding
        self._packed = None
        self._plan = None

    # ---- packing ---------------------------------------------------------------------------------
    def _derived_reset(self):
        super()._derived_reset()
        self._packed = None
        self._plan = None

    def packed(self):
        if not self._derived_valid() or self._packed is None:      # segdino3d_amd/_cache.py
            pk = {}
            for n, m in self.named_modules():
                if isinstance(m, SpConv):
                    pk[n] = _pack(m)
                elif isinstance(m, nn.BatchNorm1d):
                    pk[n] = _fold(m)
            self._packed = pk
        return self._packed

    def packed_train(self):
        """name -> [K, Cout, Cin(+pad)] autograd view of the live SpConv weight / the nn.BatchNorm1d module, for
        train_ops.TrainBackend (training step)."""
        pk = {}
        for n, m in self.named_modules():
            if isinstance(m, SpConv):
                w = m.weight
                co, ci = w.shape[0], w.shape[-1]
                w = w.reshape(co, -1, ci).permute(1, 0, 2)
                pad = (ci + 31) // 32 * 32 - ci
                if pad:
                    w = torch.nn.functional.pad(w, (0, pad))
                pk[n] = w.contiguous()
            elif isinstance(m, nn.BatchNorm1d):
                pk[n] = m
        return pk

    # ---- network -----------------------------------------------------------------------------------
    def _resblock(self, be, pk, p, x, key, x2=None):
        """ResidualBlock.forward (:82-99): conv_branch(x) + i_branch(x); x may be the concat [x | x2]."""
        if not self.normalize_before:
            # conv, BN, ReLU, conv, BN, ReLU (:66-81): the first BN + ReLU rides in its convolution's epilogue; the second
            # ReLU comes BEFORE the identity add, so that BN + ReLU + add is one elementwise pass over the raw convolution
            h = be.conv(x, pk[p + ".conv_branch.0"], pk[p + ".conv_branch.1"], key, x2=x2, act="relu")
            h = be.conv(h, pk[p + ".conv_branch.3"], None, key)
            ident = be.dense(x, pk[p + ".i_branch.0"], None, x2=x2) if (p + ".i_branch.0") in pk else x
            assert (p + ".i_branch.0") in pk or x2 is None
            return be.affine(h, pk[p + ".conv_branch.4"], act="relu", add=ident)
        h = be.affine(x, pk[p + ".conv_branch.0"], x2=x2, act="relu")
        h = be.conv(h, pk[p + ".conv_branch.2"], pk[p + ".conv_branch.3"], key, act="relu")
        if (p + ".i_branch.0") in pk:
            ident = be.dense(x, pk[p + ".i_branch.0"], None, x2=x2)
        else:
            assert x2 is None
            ident = x
        return be.conv(h, pk[p + ".conv_branch.5"], None, key, res=ident)

    def _unet(self, be, pk, prefix, n_levels, level, x):
        key = ("same", level, 3)
        for r in range(self.block_reps):
            x = self._resblock(be, pk, f"{prefix}blocks.block{r}", x, key)
        if level < n_levels - 1:
            ident = x
            if self.normalize_before:
                h = be.affine(x, pk[prefix + "conv.0"], act="relu")
                h = be.conv(h, pk[prefix + "conv.2"], None, ("down", level))
                h = self._unet(be, pk, prefix + "u.", n_levels, level + 1, h)
                h = be.affine(h, pk[prefix + "deconv.0"], act="relu")
                h = be.conv(h, pk[prefix + "deconv.2"], None, ("up", level))
            else:                                                # conv, BN, ReLU (:166-174, 194-201): all epilogue
                h = be.conv(x, pk[prefix + "conv.0"], pk[prefix + "conv.1"], ("down", level), act="relu")
                h = self._unet(be, pk, prefix + "u.", n_levels, level + 1, h)
                h = be.conv(h, pk[prefix + "deconv.0"], pk[prefix + "deconv.1"], ("up", level), act="relu")
            x = self._resblock(be, pk, f"{prefix}blocks_tail.block0", ident, key, x2=h)
            for r in range(1, self.block_reps):
                x = self._resblock(be, pk, f"{prefix}blocks_tail.block{r}", x, key)
        return x

    def _network(self, be, pk, x, n_levels):
        """`SpConvUNet.forward` (:233-268) against a plan backend (segdino3d_amd.plan)."""
        x = be.conv(x, pk["input_conv.0"], None, ("same", 0, 3))
        x = self._unet(be, pk, "", n_levels, 0, x)
        return be.affine(x, pk["output_layer.0"], act="relu")

    def forward_sparse(self, maps: SceneMaps, vox_feats: torch.Tensor) -> torch.Tensor:
        nl = len(self.num_planes)
        use_plan = (not self.training and plan.USE_PLAN and ops.PAIR_CONV and ops.GEMM_MODE is None and ops.GG_FORCE_NT is None
                    and ops.GG_HOOK is None)
        maps.prepare(same=[(l, 3) for l in range(nl)], strides=list(range(nl - 1)), chained=not self.training, fork=use_plan)
        if self.training:                                        # batch-statistics BatchNorm, autograd nodes over HIP kernels
            from . import train_ops
            return self._network(train_ops.TrainBackend(maps), self.packed_train(), vox_feats, nl)
        pk = self.packed()
        if use_plan:
            if self._plan is None:                               # one C call per scene instead of ~130
                rec = plan.Recorder(vox_feats.shape[1])
                self._plan = rec.finish(self._network(rec, pk, rec.input, nl))
            return self._plan.run(maps, vox_feats)
        return self._network(plan.EagerBackend(maps), pk, vox_feats, nl)

    @ops.bound_stream
    def forward_wrapper(self, samples: List[torch.Tensor], targets, return_sp_mean_pos=True):
        feats, pos, pos_wo = [], [], []
        scenes = []
        for pts, tgt in zip(samples, targets):
            ef = tgt["extra_features"]
            pts = pts.float().contiguous()
            f2d = ef["points_2dfeats"].float().contiguous()
            sp = ef["super_point_masks"].contiguous()
            # network coordinates are shifted to start at 0 (:286); superpoint positions are NOT (:344-353)
            elastic = tgt["elastic_coords"] if "elastic_coords" in tgt else None
            el = None
            if elastic is None:
                maps = SceneMaps(pts, self.voxel_size, len(self.num_planes), shift_to_min=True, order=self.KERNEL_ORDER,
                                 superpoints=sp, clip_min_shape=self.min_spatial_shape)
            else:                                                # distorted coordinates are already in voxel units (:291-294)
                el = elastic.to(pts.device).float().contiguous()
                maps = SceneMaps(el, 1.0, len(self.num_planes), shift_to_min=True, order=self.KERNEL_ORDER,
                                 superpoints=sp, clip_min_shape=self.min_spatial_shape)
            cin_pad = (self.in_channels + 31) // 32 * 32
            vf = maps.voxel_features(pts, f2d, 2, cin_pad, stats=None if elastic is None else ops.scene_stats(pts))
            scenes.append((maps, vf, pts, sp, el))
        cap = _trace.active()
        if cap is not None:
            cap.record_maps([s[0] for s in scenes])
        if self.training and len(scenes) > 1:
            # the batch as one block-diagonal tensor (spconv's batched SparseConvTensor, :378-379): BatchNorm over all scenes
            from .sparse import BatchedMaps
            batch = BatchedMaps([s[0] for s in scenes])
            x_all = self.forward_sparse(batch, torch.cat([s[1] for s in scenes], dim=0))
            outs = [x_all[slice(*batch.rows(0, i))] for i in range(len(scenes))]
        else:
            outs = []
            for maps, vf, _, _, _ in scenes:
                outs.append(self.forward_sparse(maps, vf))
        for (maps, _, pts, sp, el), x in zip(scenes, outs):
            if self.training:
                from . import train_ops
                f, _ = train_ops.pool_superpoints(x.contiguous(), maps, x.shape[1])
            else:
                f, _ = maps.pool(x, x.shape[1])
            # positions: mean of floor(xyz / voxel) * voxel with the UN-shifted coordinates
            pos_maps = SceneMaps(pts, self.voxel_size, 1, shift_to_min=False, order=self.KERNEL_ORDER, superpoints=sp)
            _, p = pos_maps.pool(x.new_zeros((pos_maps.n_vox[0], 4)), 4)
            feats.append(f)
            pos_wo.append(p)
            if el is None:
                pos.append(p.clone())
            else:                                                # mean of floor(elastic) * voxel_size (:337-352)
                el_maps = SceneMaps(el, 1.0, 1, shift_to_min=False, order=self.KERNEL_ORDER, superpoints=sp)
                pos.append(el_maps.pool(x.new_zeros((el_maps.n_vox[0], 4)), 4)[1] * self.voxel_size)
        sp_pos = pos if self.add_positional_embedding else None
        if return_sp_mean_pos:
            return feats, sp_pos, pos_wo
        return feats, sp_pos
