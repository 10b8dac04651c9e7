"""ScanNetQueryDecoder on the MI355X kernels (host side).

Mirrors the reference operator interface `segdino3d/models/decoder/instance_seg_3d_decoder.py`:
  - constructor kwargs of `QueryDecoder.__init__` (:212-216) + `ScanNetQueryDecoder.__init__` (:441-445),
    unknown kwargs tolerated;
  - `forward(x, sp_pos, sp_pos_wo_elastic, queries, queries_pos, dinox_queries, dinox_query_pos,
    scene_range) -> dict(cls_preds, sem_preds, masks, scores, centers, sizes[, hidden_states,
    aux_outputs])` of per-scene lists (:417-435, :786-799);
  - attributes the architecture touches: `add_dinox_query_ca`, `add_box_size_pred`, `query_proj`,
    `out_norm`, `out_cls`, `return_hidden_states`, `return_aux_outputs` (baseline3d.py:180,198,233-235,321-322);
  - the same `state_dict` key names / shapes (SURVEY.md 8(b) "Checkpoint names"): torch.nn modules are
    used as PARAMETER HOLDERS only - their forward is never called.
All arithmetic goes through segdino3d_amd.ops (libsegdino3d_hip.so): Linear = fp32-MFMA GEMM with
fused bias/activation/residual, attention = fused masked MFMA kernel on bit-packed masks, the
layer-invariant key-side projections of all layers are hoisted into two GEMMs per scene, the
boolean (mask . distance) product is an AND/any over bit words.

Supported configuration = the SegDINO3D prototypes (sine positional embedding, iterative prediction,
mask attention, superpoint queries).  Eval mode only.
"""
from __future__ import annotations

import copy

import os
import threading

import torch
import torch.nn as nn

from . import ops
from ._cache import DerivedWeights
from .builder import DECODERS


class MLP(nn.Module):
    """Parameter holder named like the reference MLP (`utils.py:167-179`): `layers.{i}`."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))


class _OutProjOnly(nn.Module):
    """The reference's projection-free MultiheadAttention owns just `out_proj` (`attention.py:100-101`)."""

    def __init__(self, vdim):
        super().__init__()
        self.out_proj = nn.Linear(vdim, vdim)
        nn.init.constant_(self.out_proj.bias, 0.0)


class _CrossAttentionHolder(nn.Module):
    """`CrossAttentionLayer` (:36-58): nn.MultiheadAttention (packed in-proj) + LayerNorm."""

    def __init__(self, d_model, num_heads, dropout, fix):
        super().__init__()
        self.fix = fix
        self.attn = nn.MultiheadAttention(d_model, num_heads, dropout=dropout, batch_first=True)
        self.norm = nn.LayerNorm(d_model)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)


class _SelfAttentionHolder(nn.Module):
    """`SelfAttentionLayer` (:117-131): nn.MultiheadAttention + LayerNorm (default init)."""

    def __init__(self, d_model, num_heads, dropout):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, num_heads, dropout=dropout, batch_first=True)
        self.norm = nn.LayerNorm(d_model)


def _range6(lo: torch.Tensor, hi: torch.Tensor) -> torch.Tensor:
    """[lo | hi] as one fp32 row of six.  `Baseline3D.get_extra_instance_data` hands both as views of ONE statistics row (min xyz, max xyz,
    sum xyz: `ops.scene_stats`) - then the row's first six entries ARE the answer and no concatenation launch is needed."""
    if (lo.dtype == torch.float32 and hi.dtype == torch.float32 and lo.numel() == 3 and hi.numel() == 3 and lo.is_contiguous() and hi.is_contiguous()
            and lo.untyped_storage().data_ptr() == hi.untyped_storage().data_ptr() and hi.storage_offset() == lo.storage_offset() + 3):
        return torch.as_strided(lo, (6,), (1,), lo.storage_offset())
    return torch.cat([lo.reshape(3), hi.reshape(3)]).float().contiguous()


class _FFNHolder(nn.Module):
    def __init__(self, d_model, hidden_dim, dropout, activation_fn):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(d_model, hidden_dim), nn.ReLU() if activation_fn == "relu" else nn.GELU(),
                                 nn.Dropout(dropout), nn.Linear(hidden_dim, d_model), nn.Dropout(dropout))
        self.norm = nn.LayerNorm(d_model)


class _FourierPE(nn.Module):
    """Buffer holder named like the reference's `position_embedding` for `pos_type="fourier"` (utils.py:45-51)."""

    def __init__(self, d_pos, gauss_scale):
        super().__init__()
        assert d_pos % 2 == 0
        self.register_buffer("gauss_B", torch.empty(3, d_pos // 2).normal_() * gauss_scale)


class _EvalF:
    """The decoder's differentiable building blocks in eval mode: straight calls into the forward kernels."""

    @staticmethod
    def linear(x, w, b=None, act=None, res=None, x2=None):
        return ops.gather_gemm(x, w, x2=x2, shift=b, act=act, res=res)

    @staticmethod
    def linear_group(jobs):                                      # independent small Linears: one launch (ops.linear_group)
        return ops.linear_group(jobs)

    layernorm = staticmethod(ops.layernorm)

    @staticmethod
    def linear_ln(x, w, b, ln_w, ln_b, res=None):               # projection + residual + LayerNorm: one launch for few rows
        return ops.linear_layernorm(x, w, b, ln_w, ln_b, res=res)

    attention = staticmethod(ops.attention)
    box_refine = staticmethod(ops.box_refine)

    @staticmethod
    def dropout(x):                                             # nn.Dropout is the identity in evaluation
        return x

    @staticmethod
    def split_cols(t, width):                                   # equal column blocks of a packed projection's output (views)
        return [t[:, c:c + width] for c in range(0, t.shape[1], width)]

    @staticmethod
    def sine_pe_mod(xyz, rng, dim_t, axis, num, den):
        return ops.sine_pe(xyz, rng, dim_t, axis, mod_num=num, mod_den=den)

    @staticmethod
    def mask_logits(nq, mask_feats):                            # einsum('nd,md->nm'); fp32 in every mode: it feeds thresholds
        return ops.gather_gemm(nq, mask_feats, exact=True)


class _TrainF:
    """Training mode: the same kernels as autograd nodes with HIP backward passes (segdino3d_amd/train_dec.py)."""

    @staticmethod
    def linear(x, w, b=None, act=None, res=None, x2=None):
        from . import train_dec
        return train_dec.linear(x, w, b, act=act, res=res, x2=x2)

    @staticmethod
    def linear_group(jobs):                                      # training: every Linear its own autograd node
        from . import train_dec
        return [train_dec.linear(x, w, b, act=act, res=res, x2=x2) for (x, w, b, act, res, x2) in jobs]

    @staticmethod
    def layernorm(x, w, b, res=None, act=None, eps=1e-5):
        from . import train_dec
        return train_dec.layernorm(x, w, b, res=res, act=act, eps=eps)

    @staticmethod
    def split_cols(t, width):
        from . import train_dec
        return train_dec.split_cols(t, width)

    @staticmethod
    def dropout(x):
        """nn.Dropout(p) of the reference's layers (`instance_seg_3d_decoder.py:51, 131, 168-170, 499, 515`); the rate of the
        running forward is thread-local (set by ScanNetQueryDecoder.forward), 0.0 in every shipped config."""
        p = getattr(_F_TLS, "p", 0.0)
        return torch.nn.functional.dropout(x, p, training=True) if p > 0.0 else x

    @staticmethod
    def linear_ln(x, w, b, ln_w, ln_b, res=None):               # training: two autograd nodes, the layer's Dropout between them
        from . import train_dec
        return train_dec.layernorm(_TrainF.dropout(train_dec.linear(x, w, b)), ln_w, ln_b, res=res)

    @staticmethod
    def attention(q, k, v, num_heads, scale, mask_bits=None, q2=None, k2=None):
        from . import train_dec
        p = getattr(_F_TLS, "p", 0.0)
        if p > 0.0:                                             # nn.MultiheadAttention(dropout=p): dropout on the probabilities
            return train_dec.attention_dropout(q, k, v, num_heads, scale, mask_bits=mask_bits, q2=q2, k2=k2, p=p)
        return train_dec.attention(q, k, v, num_heads, scale, mask_bits=mask_bits, q2=q2, k2=k2)

    @staticmethod
    def box_refine(ref_points, d_center, size_prev, d_size, rng, normalize):
        from . import train_dec
        return train_dec.box_refine(ref_points, d_center, size_prev, d_size, rng, normalize)

    @staticmethod
    def sine_pe_mod(xyz, rng, dim_t, axis, num, den):
        from . import train_dec
        return train_dec.sine_pe_modulated(xyz, rng, dim_t, axis, num, den)

    @staticmethod
    def mask_logits(nq, mask_feats):
        from . import train_dec
        return train_dec.linear(nq, mask_feats, exact=True)


_F_TLS = threading.local()
# SD3D_BATCH_DECODER=0: the decoder of a multi-scene evaluation forward runs scene by scene (A/B switch; default: the row-wise work of
# all scenes in one pass, `ScanNetQueryDecoder._forward_batch`)
BATCH_DECODER = os.environ.get("SD3D_BATCH_DECODER", "1") != "0"
# SD3D_FUSED_DECODER=0: evaluation runs the op-by-op decoder of rounds 1-3 (`_forward_scene` / `_forward_batch`) instead of the
# row-chain launches (`_forward_fused`); training always runs op by op (autograd nodes)
# "auto" (default): a scene takes the row-chain path when it has more than FUSED_MIN_ROWS query rows (one query per superpoint,
# the reference's evaluation mode: 15.7 -> 15.0 ms per scene); a scene with a few hundred queries stays op by op (12.6 vs 12.85 ms:
# thirteen 16-row workgroups are bound by ONE CU's fp32 matrix rate, 3.9 us per 256 x 256 Linear, and cannot outrun ~27 small
# launches that each use 50 - 200 CUs - profiles/EXPERIMENTS.md, round 4).  "1" / "0" force it for every scene.
_fd = os.environ.get("SD3D_FUSED_DECODER", "auto")
FUSED_DECODER = "auto" if _fd == "auto" else (_fd != "0")
FUSED_MIN_ROWS = int(os.environ.get("SD3D_FUSED_MIN_ROWS", "512"))
FUSED_SA_MAX_KEYS = int(os.environ.get("SD3D_FUSED_SA_MAX_KEYS", "1024"))
# SD3D_FUSED_NARROW=1: scenes with at most FUSED_MIN_ROWS query rows take the row-chain path too, on 4-row tiles (csrc/rowchain_narrow.hip)
FUSED_NARROW = os.environ.get("SD3D_FUSED_NARROW", "0") == "1"


def _F():
    return getattr(_F_TLS, "f", _EvalF)


def _lin(x, layer: nn.Linear, act=None, res=None):
    return _F().linear(x, layer.weight, layer.bias, act=act, res=res)


def _mlp(x, mlp: MLP, final_act=None, res=None):
    n = len(mlp.layers)
    for i, layer in enumerate(mlp.layers):
        last = i == n - 1
        x = _lin(x, layer, act=(final_act if last else "relu"), res=(res if last else None))
    return x


@DECODERS.register_module()
class ScanNetQueryDecoder(DerivedWeights):
    def __init__(self, num_layers, num_instance_queries, num_semantic_queries, num_instance_classes,
                 num_semantic_classes, num_semantic_linears, in_channels, d_model, num_heads, hidden_dim, dropout,
                 activation_fn, iter_pred, attn_mask, fix_attention, objectness_flag, add_dinox_query_ca=False,
                 add_dinox_query_ca_mask=False, dinox_query_ca_mask_threshold=0.2, mask_attention_threshold=0.5,
                 add_positional_embedding=False, pos_type="fourier", temperature=10000, gauss_scale=1.0,
                 add_box_size_pred=False, box_modulate_ca=False, normalize_box_prediction=False,
                 use_activation_checkpoint=False, **kwargs):
        super().__init__()
        assert num_semantic_linears in [1, 2]
        unsupported = []
        if add_positional_embedding and pos_type not in ("sine", "fourier"):
            raise AssertionError(f"pos_type must be 'sine' or 'fourier', got {pos_type!r}")        # utils.py:41
        if not add_positional_embedding and (add_dinox_query_ca or add_box_size_pred or box_modulate_ca):
            unsupported.append("2D-query attention / box heads need add_positional_embedding=True")
        if not iter_pred or not attn_mask:
            unsupported.append("iter_pred=True and attn_mask=True are required")
        if num_instance_queries != 0:
            # the reference drops `query_proj` then (`:235-238`) while Baseline3D always passes superpoint queries (`:339-345`
            # of baseline3d.py call `_get_queries` -> `self.query_proj`): not runnable there either
            unsupported.append("num_instance_queries > 0 (the reference's own forward cannot run it behind Baseline3D)")
        if num_semantic_queries != 0 and add_positional_embedding:
            unsupported.append("learned queries with add_positional_embedding (the reference has no positions for them, :631-640)")
        if d_model != num_heads * 32:
            unsupported.append("attention heads must be 32 channels wide (d_model == 32 * num_heads)")
        if add_dinox_query_ca and not add_dinox_query_ca_mask:
            unsupported.append("add_dinox_query_ca requires add_dinox_query_ca_mask")
        if box_modulate_ca:
            assert pos_type == "sine", "Only implemented for sine positional embedding now."        # `:528`
            assert add_positional_embedding and add_box_size_pred, \
                " If you want to use box to modulate cross attention, you should set add_positional_embedding and add_box_size_pred to True."
        if unsupported:
            raise NotImplementedError("segdino3d_amd ScanNetQueryDecoder: " + "; ".join(unsupported))
        L, d = num_layers, d_model
        self.return_hidden_states = True
        self.return_aux_outputs = True
        self.num_layers, self.d_model, self.num_heads, self.in_channels = L, d, num_heads, in_channels
        self.iter_pred, self.attn_mask, self.objectness_flag = iter_pred, attn_mask, objectness_flag
        self.activation_fn = activation_fn
        self.add_dinox_query_ca = add_dinox_query_ca
        self.add_dinox_query_ca_mask = add_dinox_query_ca_mask
        self.dinox_query_ca_mask_threshold = dinox_query_ca_mask_threshold
        self.mask_attention_threshold = mask_attention_threshold
        self.num_semantic_classes = num_semantic_classes
        self.num_instance_classes = num_instance_classes
        self.add_positional_embedding = add_positional_embedding
        self.add_box_size_pred = add_box_size_pred
        self.box_modulate_ca = box_modulate_ca
        self.normalize_box_prediction = normalize_box_prediction
        self.temperature = float(temperature)
        self.num_queries = num_instance_queries + num_semantic_queries
        if self.num_queries > 0:
            self.query = nn.Embedding(self.num_queries, d_model)       # learned queries, prepended to the projected ones (:302-307)
        if objectness_flag:
            self.out_score = nn.Sequential(nn.Linear(d_model, d_model), nn.ReLU(), nn.Linear(d_model, 1))
        self.pos_type = pos_type
        # nn.Dropout is the identity in evaluation; training with p > 0 (no shipped config) drops between the autograd nodes with
        # torch's generator and forms the attention probabilities explicitly (train_dec.attention_dropout)
        self.dropout = float(dropout)
        # "fp32" (default, BASELINE config #2) or "bf16" (config #3: projections and both attention contractions on the bf16
        # MFMA with fp32 accumulation; LayerNorm, softmax, positional encodings, mask logits and thresholds stay fp32).  Not a
        # key of the reference's config surface - there bf16 comes from autocast(cfg.amp), train_engine_3d.py:88-100.
        self.compute_dtype = str(kwargs.get("compute_dtype", os.environ.get("SD3D_DECODER_DTYPE", "fp32")))
        if self.compute_dtype not in ("fp32", "bf16"):
            raise ValueError(f"decoder compute_dtype must be 'fp32' or 'bf16', got {self.compute_dtype!r}")

        self.input_proj = nn.Sequential(nn.Linear(in_channels, d), nn.LayerNorm(d), nn.ReLU())
        self.query_proj = nn.Sequential(nn.Linear(in_channels, d), nn.ReLU(), nn.Linear(d, d))
        if add_positional_embedding:
            self.cross_attn_layers = nn.ModuleList(_OutProjOnly(d) for _ in range(L))
            self.self_attn_layers = nn.ModuleList(_OutProjOnly(d) for _ in range(L))
        else:   # QueryDecoder.__init__ :244-251: CrossAttentionLayer / SelfAttentionLayer with nn.MultiheadAttention
            self.cross_attn_layers = nn.ModuleList(_CrossAttentionHolder(d, num_heads, dropout, fix_attention) for _ in range(L))
            self.self_attn_layers = nn.ModuleList(_SelfAttentionHolder(d, num_heads, dropout) for _ in range(L))
        self.ffn_layers = nn.ModuleList(_FFNHolder(d, hidden_dim, dropout, activation_fn) for _ in range(L))
        self.out_norm = nn.LayerNorm(d)
        self.out_cls = nn.Sequential(nn.Linear(d, d), nn.ReLU(), nn.Linear(d, num_instance_classes + 1))
        self.x_mask = nn.Sequential(nn.Linear(in_channels, d), nn.ReLU(), nn.Linear(d, d))
        if num_semantic_linears == 2:
            self.out_sem = nn.Sequential(nn.Linear(d, d), nn.ReLU(), nn.Linear(d, num_semantic_classes + 1))
        else:
            self.out_sem = nn.Linear(d, num_semantic_classes + 1)
        if add_dinox_query_ca:
            self.dinox_query_cross_attn_layers = nn.ModuleList(
                _CrossAttentionHolder(d, num_heads, dropout, fix_attention) for _ in range(L))
        if add_positional_embedding:
            if pos_type == "fourier":                          # checkpoint key `decoder.position_embedding.gauss_B` (utils.py:45-51)
                self.position_embedding = _FourierPE(d, gauss_scale)
            self.ref_point_head = MLP(d, d, d, 2)
            bbox = MLP(d, d, 3, 3)
            nn.init.constant_(bbox.layers[-1].weight.data, 0)
            nn.init.constant_(bbox.layers[-1].bias.data, 0)
            self.bbox_embed = nn.ModuleList(copy.deepcopy(bbox) for _ in range(L))
            mk = lambda: nn.ModuleList(nn.Linear(d, d) for _ in range(L))  # noqa: E731
            self.ca_qcontent_proj = mk()
            self.ca_qpos_proj = nn.Linear(d, d)
            self.ca_kcontent_proj, self.ca_kpos_proj, self.ca_v_proj, self.ca_qpos_sine_proj = mk(), mk(), mk(), mk()
            self.norm1 = nn.ModuleList(nn.LayerNorm(d) for _ in range(L))
            self.sa_qcontent_proj, self.sa_qpos_proj, self.sa_kcontent_proj = mk(), mk(), mk()
            self.sa_kpos_proj, self.sa_v_proj = mk(), mk()
            self.norm2 = nn.ModuleList(nn.LayerNorm(d) for _ in range(L))
            if add_box_size_pred:
                self.bbox_size_embed = nn.ModuleList(copy.deepcopy(bbox) for _ in range(L))
        if box_modulate_ca:
            self.ref_anchor_head = MLP(d, d, 3, 2)
        self._packed = None
        self._pe_tables = {}

    # ---- derived weights (hoisted / concatenated / bf16-rounded), rebuilt when their sources change (_cache.py) ----
    def _derived_reset(self):
        super()._derived_reset()
        self._packed = None
        self._pe_tables = {}
        ops.clear_split_cache()          # bf16 roundings of the packed weights die with them
        from . import rowchain
        rowchain.clear_pack_cache()      # and their MFMA-fragment-order copies (row-chain LINEAR)

    def packed(self, live=False):
        """Packed projection weights; `live=True` keeps them attached to the parameters (training: rebuilt every step, the
        concatenations are autograd views of the live weights)."""
        det = (lambda t: t) if live else (lambda t: t.detach())  # noqa: E731
        if live or not self._derived_valid() or self._packed is None:
            d, L = self.d_model, self.num_layers
            cat = lambda mods, attr: torch.cat([det(getattr(m, attr)) for m in mods]).contiguous()  # noqa: E731
            if not self.add_positional_embedding:
                pk = {}
                for nm, mods in (("ca", self.cross_attn_layers), ("sa", self.self_attn_layers)):
                    ws = [det(m.attn.in_proj_weight) for m in mods]
                    bs = [det(m.attn.in_proj_bias) for m in mods]
                    pk[nm + "_q_w"] = [w[:d].contiguous() for w in ws]
                    pk[nm + "_q_b"] = [b[:d].contiguous() for b in bs]
                    pk[nm + "_kv_w"] = [w[d:].contiguous() for w in ws]
                    pk[nm + "_kv_b"] = [b[d:].contiguous() for b in bs]
                # all layers' key/value projections of the (layer-invariant) superpoint features in one GEMM
                wca = [det(m.attn.in_proj_weight) for m in self.cross_attn_layers]
                bca = [det(m.attn.in_proj_bias) for m in self.cross_attn_layers]
                pk["ca_kv_all_w"] = torch.cat([w[d:2 * d] for w in wca] + [w[2 * d:] for w in wca]).contiguous()
                pk["ca_kv_all_b"] = torch.cat([b[d:2 * d] for b in bca] + [b[2 * d:] for b in bca]).contiguous()
                if not live:
                    self._packed = pk
                return pk
            pk = {
                # all layers' key-content and value projections of the superpoint features: [2*L*d, d]
                "kv_w": torch.cat([cat(self.ca_kcontent_proj, "weight"), cat(self.ca_v_proj, "weight")]).contiguous(),
                "kv_b": torch.cat([cat(self.ca_kcontent_proj, "bias"), cat(self.ca_v_proj, "bias")]).contiguous(),
                "kp_w": cat(self.ca_kpos_proj, "weight"), "kp_b": cat(self.ca_kpos_proj, "bias"),
            }
            if self.add_dinox_query_ca:
                ws = [det(m.attn.in_proj_weight) for m in self.dinox_query_cross_attn_layers]
                bs = [det(m.attn.in_proj_bias) for m in self.dinox_query_cross_attn_layers]
                pk["q2d_w"] = [w[:d].contiguous() for w in ws]
                pk["q2d_b"] = [b[:d].contiguous() for b in bs]
                pk["kv2d_w"] = torch.cat([w[d:2 * d] for w in ws] + [w[2 * d:] for w in ws]).contiguous()
                pk["kv2d_b"] = torch.cat([b[d:2 * d] for b in bs] + [b[2 * d:] for b in bs]).contiguous()
            # self-attention q / k / v of one layer as ONE GEMM over the concatenated input [queries | query_pos]:
            #   q = Wqc q + Wqp p,  k = Wkc q + Wkp p,  v = Wv q            (5 launches -> 1, :695-700)
            sa_w, sa_b = [], []
            for i in range(L):
                wq = torch.cat([self.sa_qcontent_proj[i].weight, self.sa_qpos_proj[i].weight], dim=1)
                wk = torch.cat([self.sa_kcontent_proj[i].weight, self.sa_kpos_proj[i].weight], dim=1)
                wv = torch.cat([self.sa_v_proj[i].weight, torch.zeros_like(self.sa_v_proj[i].weight)], dim=1)
                sa_w.append(det(torch.cat([wq, wk, wv])).contiguous())
                sa_b.append(det(torch.cat([self.sa_qcontent_proj[i].bias + self.sa_qpos_proj[i].bias,
                                       self.sa_kcontent_proj[i].bias + self.sa_kpos_proj[i].bias,
                                       self.sa_v_proj[i].bias])).contiguous())
            pk["sa_qkv_w"], pk["sa_qkv_b"] = sa_w, sa_b
            # first layer's content query also takes the positional query (:672): [queries | query_pos] again
            pk["ca_q0_w"] = det(torch.cat([self.ca_qcontent_proj[0].weight, self.ca_qpos_proj.weight], dim=1)).contiguous()
            pk["ca_q0_b"] = det(self.ca_qcontent_proj[0].bias + self.ca_qpos_proj.bias).contiguous()
            if live:
                return pk
            self._packed = pk
        return self._packed

    def packed_train(self):
        return self.packed(live=True)

    def pe_tables(self, device):
        """(dim_t [d] fp32, axis [d] int8): per-channel divisor and coordinate axis of the sine PE
        (`utils.py:64-86`): channel split 86/86/84 for d=256, dim_t = T^(2*(i//2)/cdim)."""
        key = str(device)
        if key not in self._pe_tables:
            d_pos, d_in = self.d_model, 3
            ndim = d_pos // d_in
            if ndim % 2:
                ndim -= 1
            rems = d_pos - ndim * d_in
            dim_t, axis = [], []
            for a in range(d_in):
                cdim = ndim
                if rems > 0:
                    cdim += 2
                    rems -= 2
                i = torch.arange(cdim, dtype=torch.float32)
                dim_t.append(self.temperature ** (2 * (i // 2) / cdim))
                axis.append(torch.full((cdim,), a, dtype=torch.int8))
            self._pe_tables[key] = (torch.cat(dim_t).to(device).contiguous(), torch.cat(axis).to(device).contiguous())
        return self._pe_tables[key]

    # ---- prediction head (:532-577) ----------------------------------------------------------------
    def _head(self, queries, mask_feats, last_flag, defer_cls=False):
        """-> (class logits | the normalised queries when `defer_cls`, semantic logits | None, mask logits, mask bits,
        objectness score | None).  The class
        MLP feeds nothing inside the decoder, so the positional variant runs it inside the NEXT layer's first launches."""
        S = mask_feats.shape[0]
        nq = _F().layernorm(queries, self.out_norm.weight, self.out_norm.bias)
        cls = nq if defer_cls else _lin(_lin(nq, self.out_cls[0], act="relu"), self.out_cls[2])
        sem = None
        if last_flag:
            if isinstance(self.out_sem, nn.Linear):
                sem = _lin(nq, self.out_sem)
            else:
                sem = _lin(_lin(nq, self.out_sem[0], act="relu"), self.out_sem[2])
        logits = _F().mask_logits(nq, mask_feats)
        bits = ops.mask_bits(logits.detach(), S, self.mask_attention_threshold)
        score = _lin(_lin(nq, self.out_score[0], act="relu"), self.out_score[2]) if self.objectness_flag else None   # [Q, 1] (:548-550)
        return cls, sem, logits, bits, score

    def select_scores(self, x):
        """max_c softmax(out_cls(out_norm(query_proj(x))))[:-1] per superpoint (baseline3d.py:233-238)."""
        q = _lin(_lin(x, self.query_proj[0], act="relu"), self.query_proj[2])
        nq = ops.layernorm(q, self.out_norm.weight, self.out_norm.bias)
        cls = _lin(_lin(nq, self.out_cls[0], act="relu"), self.out_cls[2])
        return ops.class_scores(cls, self.num_instance_classes, want_scores=False, want_rowmax=True)[1]

    # ---- one scene, non-positional variant (Baseline_ScanNet200 prototype; :693, :711, :733) ---------
    def _forward_scene_plain(self, x, q_in):
        d, H, L = self.d_model, self.num_heads, self.num_layers
        F = _F()
        pk = self.packed_train() if self.training else self.packed()
        x, q_in = x.contiguous(), q_in.contiguous()
        inst = F.layernorm(_lin(x, self.input_proj[0]), self.input_proj[1].weight, self.input_proj[1].bias, act="relu")
        mask_feats = _lin(_lin(x, self.x_mask[0], act="relu"), self.x_mask[2])
        queries = _lin(_lin(q_in, self.query_proj[0], act="relu"), self.query_proj[2])
        if self.num_queries > 0:                               # learned queries first, then the projected ones (`_get_queries` :302-307)
            queries = torch.cat([self.query.weight if self.training else self.query.weight.detach(), queries]).contiguous()
        cls, sem, logits, bits, score = self._head(queries, mask_feats, False)
        aux = [dict(cls_preds=cls, sem_preds=None, masks=logits, centers=None, sizes=None, scores=score)]
        kv = F.split_cols(F.linear(inst, pk["ca_kv_all_w"], pk["ca_kv_all_b"]), d)     # [S, 2*L*d]: k_0..k_{L-1} | v_0..v_{L-1}
        scale = (d // H) ** -0.5
        dropping = self.training and self.dropout > 0.0
        for i in range(L):
            ca, sa, ffn = self.cross_attn_layers[i], self.self_attn_layers[i], self.ffn_layers[i]
            q = F.linear(queries, pk["ca_q_w"][i], pk["ca_q_b"][i])
            a = F.attention(q, kv[i], kv[L + i], H, scale, mask_bits=bits)
            if ca.fix:                                          # dropout only on the `fix` path (:80-84)
                queries = F.layernorm(F.dropout(_lin(a, ca.attn.out_proj)), ca.norm.weight, ca.norm.bias, res=queries)
            else:
                queries = _lin(a, ca.attn.out_proj, res=queries)
            qkv = F.split_cols(F.linear(queries, sa.attn.in_proj_weight, sa.attn.in_proj_bias), d)      # [Q, 3d]
            a = F.attention(qkv[0], qkv[1], qkv[2], H, scale)
            queries = F.layernorm(F.dropout(_lin(a, sa.attn.out_proj)), sa.norm.weight, sa.norm.bias, res=queries)
            hdn = F.dropout(_lin(queries, ffn.net[0], act=("relu" if self.activation_fn == "relu" else "gelu")))
            if dropping:                                        # net(y) ends in a Dropout, then + y, then the norm (:166-190)
                queries = F.layernorm(F.dropout(_lin(hdn, ffn.net[3])), ffn.norm.weight, ffn.norm.bias, res=queries)
            else:
                hdn = _lin(hdn, ffn.net[3], res=queries)
                queries = F.layernorm(hdn, ffn.norm.weight, ffn.norm.bias)
            cls, sem, logits, bits, score = self._head(queries, mask_feats, i == L - 1)
            aux.append(dict(cls_preds=cls, sem_preds=sem, masks=logits, centers=None, sizes=None, scores=score))
        final = aux.pop()
        final["hidden_states"] = queries
        final["attn_mask_bits"] = bits
        # the reference's aux list is one entry short in this variant (pred_centers misses the layer-0
        # placeholder, :653-655, and the zip at :781-783 truncates): keep the same length
        return final, aux[:-1]

    # ---- one scene -----------------------------------------------------------------------------------
    def _forward_scene(self, x, sp_pos, sp_pos_wo, q_in, q_pos, q2d_feat, q2d_pos, lo, hi):
        dev = x.device
        d, H, L = self.d_model, self.num_heads, self.num_layers
        F = _F()
        pk = self.packed_train() if self.training else self.packed()
        dim_t, axis = self.pe_tables(dev)
        x, sp_pos, q_in, q_pos = x.contiguous(), sp_pos.contiguous(), q_in.contiguous(), q_pos.contiguous()
        Q = q_in.shape[0]
        rng = _range6(lo, hi)
        pe = (lambda xyz: ops.fourier_pe(xyz, rng, self.position_embedding.gauss_B, d)) if self.pos_type == "fourier" \
            else (lambda xyz: ops.sine_pe(xyz, rng, dim_t, axis))
        memory_emb = pe(sp_pos)
        if self.normalize_box_prediction:
            # one row, broadcast over queries; 0.5 / d == (1 / d) * 0.5 bit for bit (a power-of-two factor commutes with the rounding): one launch less
            size_q = (0.5 / (hi - lo)).float().reshape(3).contiguous()
        else:
            size_q = torch.full((Q, 3), 0.5, dtype=torch.float32, device=dev)
        inst = F.layernorm(_lin(x, self.input_proj[0]), self.input_proj[1].weight, self.input_proj[1].bias, act="relu")
        mask_feats = _lin(_lin(x, self.x_mask[0], act="relu"), self.x_mask[2])
        queries = _lin(_lin(q_in, self.query_proj[0], act="relu"), self.query_proj[2])
        # Independent Linears on the few hundred query rows go out in ONE launch each time (F.linear_group): the class MLP of
        # a prediction head rides with the next layer's first projections, the two box MLPs run side by side, projections that
        # share an input are batched.  Every Linear is still the same fp32 product; only the dispatch count changes.
        J = lambda x, layer, act=None, res=None, x2=None: (x, layer.weight, layer.bias, act, res, x2)  # noqa: E731
        nq_pending, sem, logits, bits, score = self._head(queries, mask_feats, False, defer_cls=True)
        aux = [dict(cls_preds=None, sem_preds=None, masks=logits, centers=None, sizes=None, scores=score)]

        # layer-invariant key side, hoisted out of the loop (the reference recomputes it per layer, :669-671)
        kv_all = F.split_cols(F.linear(inst, pk["kv_w"], pk["kv_b"]), d)             # [S, 2*L*d]: kc_0..kc_{L-1} | v_0..v_{L-1}
        kp_all = F.split_cols(F.linear(memory_emb, pk["kp_w"], pk["kp_b"]), d)       # [S, L*d]
        if self.add_dinox_query_ca:
            if not isinstance(q2d_pos, torch.Tensor):
                q2d_pos = q2d_pos.tensor.type(sp_pos_wo.dtype).to(dev)
            keys2d = torch.cat([q2d_feat.float(), q2d_feat.new_ones(1, q2d_feat.shape[1], dtype=torch.float32)]).contiguous()
            kv2d_all = F.split_cols(F.linear(keys2d, pk["kv2d_w"], pk["kv2d_b"]), d)     # [M+1, 2*L*d]
            near = ops.near_bits(sp_pos_wo.float().contiguous(), q2d_pos.float().contiguous(),
                                 self.dinox_query_ca_mask_threshold)

        ref_points = q_pos.float().contiguous()
        ref_sizes = size_q
        for i in range(L):
            ops.baton_yield()
            # ---- launch A: first layer of the box-modulation MLP, this layer's content query (layers >= 1), and the first
            #      layer of the previous head's class MLP
            jobs, what = [], []
            if self.box_modulate_ca:
                jobs.append(J(queries, self.ref_anchor_head.layers[0], "relu")); what.append("anchor")
            if i > 0:
                jobs.append(J(queries, self.ca_qcontent_proj[i])); what.append("qc")
            jobs.append(J(nq_pending, self.out_cls[0], "relu")); what.append("cls")
            outs = dict(zip(what, F.linear_group(jobs)))
            # ---- launch B: second layers of both
            jobs, what = [J(outs["cls"], self.out_cls[2])], ["cls"]
            if self.box_modulate_ca:
                jobs.append(J(outs["anchor"], self.ref_anchor_head.layers[1], "sigmoid")); what.append("hwl")
            outs2 = dict(zip(what, F.linear_group(jobs)))
            aux[-1]["cls_preds"] = outs2["cls"]
            # ---- box-modulated positional query (:659-666)
            if self.box_modulate_ca:
                pq_emb = F.sine_pe_mod(ref_points, rng, dim_t, axis, outs2["hwl"], ref_sizes)
            else:
                pq_emb = pe(ref_points)
            # ---- launch C: first layer of the positional-query MLP next to the sine projection of the cross-attention
            h, qs = F.linear_group([J(pq_emb, self.ref_point_head.layers[0], "relu"), J(pq_emb, self.ca_qpos_sine_proj[i])])
            query_pos = _lin(h, self.ref_point_head.layers[1])
            # ---- masked cross-attention to the superpoints (:668-691)
            kc, v, kp = kv_all[i], kv_all[L + i], kp_all[i]
            if i == 0:
                qc = F.linear(queries, pk["ca_q0_w"], pk["ca_q0_b"], x2=query_pos)
                kc = _lin(inst, self.ca_kcontent_proj[0], res=kp)
            else:
                qc = outs["qc"]
            a = F.attention(qc, kc, v, H, (2 * d // H) ** -0.5, mask_bits=bits, q2=qs, k2=kp)
            op = self.cross_attn_layers[i].out_proj
            queries = F.linear_ln(a, op.weight, op.bias, self.norm1[i].weight, self.norm1[i].bias, res=queries)
            # ---- self-attention (:695-709)
            qkv = F.split_cols(F.linear(queries, pk["sa_qkv_w"][i], pk["sa_qkv_b"][i], x2=query_pos), d)     # [Q, 3d]
            a = F.attention(qkv[0], qkv[1], qkv[2], H, (d // H) ** -0.5)
            op = self.self_attn_layers[i].out_proj
            queries = F.linear_ln(a, op.weight, op.bias, self.norm2[i].weight, self.norm2[i].bias, res=queries)
            # ---- cross-attention to the cached DINO-X 2D object queries (:713-731, :60-86)
            if self.add_dinox_query_ca:
                layer = self.dinox_query_cross_attn_layers[i]
                bits2d = ops.dinox_mask_bits(bits, near)
                q = F.linear(queries, pk["q2d_w"][i], pk["q2d_b"][i])
                a = F.attention(q, kv2d_all[i], kv2d_all[L + i], H,
                                  (d // H) ** -0.5, mask_bits=bits2d)
                if layer.fix:
                    op = layer.attn.out_proj
                    queries = F.linear_ln(a, op.weight, op.bias, layer.norm.weight, layer.norm.bias, res=queries)
                else:
                    queries = _lin(a, layer.attn.out_proj, res=queries)
            # ---- FFN (:173-190)
            ffn = self.ffn_layers[i]
            hdn = F.dropout(_lin(queries, ffn.net[0], act=("relu" if self.activation_fn == "relu" else "gelu")))
            queries = F.linear_ln(hdn, ffn.net[3].weight, ffn.net[3].bias, ffn.norm.weight, ffn.norm.bias, res=queries)
            # ---- iterative box refinement (:735-759): the centre and the size MLP side by side, three launches for six Linears
            if self.add_box_size_pred:
                be, se = self.bbox_embed[i].layers, self.bbox_size_embed[i].layers
                c1, s1 = F.linear_group([J(queries, be[0], "relu"), J(queries, se[0], "relu")])
                c2, s2 = F.linear_group([J(c1, be[1], "relu"), J(s1, se[1], "relu")])
                dc, ds = F.linear_group([J(c2, be[2]), J(s2, se[2])])
            else:
                dc, ds = _mlp(queries, self.bbox_embed[i]), None
            center, size, size_metric = F.box_refine(ref_points, dc, size_q, ds, rng, self.normalize_box_prediction)
            ref_points = center.detach()                       # `:740`
            if self.add_box_size_pred:
                ref_sizes = size_q = size.detach()             # `:753`
            last = i == L - 1
            nq_pending, sem, logits, bits, score = self._head(queries, mask_feats, last, defer_cls=True)
            aux.append(dict(cls_preds=None, sem_preds=sem, masks=logits, centers=center, sizes=size_metric, scores=score))
        aux[-1]["cls_preds"] = _lin(_lin(nq_pending, self.out_cls[0], act="relu"), self.out_cls[2])
        final = aux.pop()
        final["hidden_states"] = queries
        final["attn_mask_bits"] = bits
        return final, aux

    # ---- several scenes at once (evaluation, positional variant) ---------------------------------------------------------
    def _row_shapes(self):
        """(Cin, Cout) of every plain Linear of the positional decoder, by the rows it runs on: queries / superpoints / 2D keys."""
        d, L, c = self.d_model, self.num_layers, self.in_channels
        q = [(c, d), (d, d), (2 * d, d), (2 * d, 3 * d), (d, self.ffn_layers[0].net[0].out_features), (self.ffn_layers[0].net[0].out_features, d),
             (d, self.num_instance_classes + 1), (d, self.num_semantic_classes + 1), (d, 3), (d, 1)]
        s = [(c, d), (d, d), (d, 2 * L * d), (d, L * d)]
        return q, s, [(d, 2 * L * d)]

    def _batchable(self, x, queries, dinox_queries):
        """The batched path keeps every row on the kernel its scene's own forward would use (bit-identical outputs): it is taken when
        every scene gets the same tiling code (ops.dense_code) for every Linear of its three row families, the query tensors have few
        hundred rows (grouped launches, fused Linear + LayerNorm) and no instrumentation / opt-in arithmetic mode is active;
        anything else runs scene by scene."""
        if self.training or not self.add_positional_embedding or not 2 <= len(x) <= 16 or not BATCH_DECODER or ops.GG_HOOK is not None:
            return False                                        # (the batched launches hold at most SD3D_MAX_BATCH = 16 scenes)
        lim = ops.LINEAR_LN_MAX_ROWS
        if lim <= 0 or ops.GEMM_MODE is not None or ops.GG_FORCE_NT is not None:
            return False
        if not all(0 < q.shape[0] <= min(512, lim) for q in queries):
            return False
        if ops.bf16_decoder_active() and not (all(t.shape[0] >= ops.BF16_MIN_ROWS for t in x) or all(t.shape[0] < ops.BF16_MIN_ROWS for t in x)):
            return False
        fams = [[q.shape[0] for q in queries], [t.shape[0] for t in x]]
        if self.add_dinox_query_ca:
            fams.append([t.shape[0] + 1 for t in dinox_queries])
        # the mask-logit products nq_b . mask_feats_b^T run grouped on the split-contraction kernel: only where ONE scene's own launch
        # would take it too (ops.gather_gemm(nq, mask_feats, exact=True) -> dense_code(Q_b, d, S_b); ~320 < Q <= 512 at S ~ 3000 does not)
        if any(ops.dense_code(q.shape[0], self.d_model, t.shape[0]) != -1 for q, t in zip(queries, x)):
            return False
        for rows, shapes in zip(fams, self._row_shapes()):
            for cin, cout in shapes:
                codes = {ops.dense_code(r, cin, cout) for r in rows}
                if len(codes) != 1 or None in codes:
                    return False
        return True

    def _forward_batch(self, xs, sp_pos, sp_pos_wo, q_in, q_pos, q2d_feat, q2d_pos, ranges):
        """`_forward_scene` for B scenes in one pass: everything row-wise (every Linear, LayerNorm, positional encoding, box
        refinement) runs ONCE over the scenes' rows back to back - B x fewer launches per scene - with the tiling codes one scene's
        rows would get; what couples the rows of a scene (the three attentions, mask bits, the 2D-query masks) runs per scene on
        row slices; the B mask-logit products are one grouped launch.  Returns [(final, aux)] per scene, the tensors being row
        slices of the batch tensors; bit-identical to B calls of `_forward_scene`."""
        dev = xs[0].device
        d, H, L = self.d_model, self.num_heads, self.num_layers
        B = len(xs)
        pk = self.packed()
        dim_t, axis = self.pe_tables(dev)
        s_off, q_off = [0], [0]
        for b in range(B):
            s_off.append(s_off[-1] + xs[b].shape[0])
            q_off.append(q_off[-1] + q_in[b].shape[0])
        S_tot, Q_tot = s_off[-1], q_off[-1]
        X = torch.cat([t.contiguous() for t in xs]).contiguous()
        SP = torch.cat([t.float() for t in sp_pos]).contiguous()
        Qin = torch.cat(list(q_in)).contiguous()
        rng = torch.stack([_range6(lo, hi) for lo, hi in ranges]).contiguous()      # [B, 6]
        # scene index of every row, made ON the device from host-known sizes (B fills + a concatenation): no host -> device copy and no
        # synchronising op here - a blocking call inside the issue baton stalls the other scenes' threads (measured: 114 -> 96 scenes/s)
        scene_of = lambda offs: torch.cat([torch.full((offs[b + 1] - offs[b],), b, dtype=torch.int32, device=dev) for b in range(B)])  # noqa: E731
        rs_s, rs_q = scene_of(s_off), scene_of(q_off)
        # every Linear with the tiling code ONE scene's rows get (`_batchable` made sure all scenes agree on it); superpoint-side rows
        # may take the bf16 kernel in the bf16 mode exactly when one scene's rows would (>= BF16_MIN_ROWS), query-side rows never do
        rows_q, rows_s = q_in[0].shape[0], xs[0].shape[0]
        s_exact = rows_s < ops.BF16_MIN_ROWS

        def lin_rows(rows, exact):
            def f(x, w, b=None, act=None, res=None, x2=None):
                return ops.gather_gemm(x, w, x2=x2, shift=b, act=act, res=res, nt=ops.dense_code(rows, w.shape[-1], w.shape[0]), exact=exact)
            return f
        qlin, slin = lin_rows(rows_q, True), lin_rows(rows_s, s_exact)
        QL = lambda x, layer, act=None, res=None: qlin(x, layer.weight, layer.bias, act, res)  # noqa: E731
        SL = lambda x, layer, act=None, res=None: slin(x, layer.weight, layer.bias, act, res)  # noqa: E731
        J = lambda x, layer, act=None, res=None, x2=None: (x, layer.weight, layer.bias, act, res, x2)  # noqa: E731
        group = lambda jobs: ops.linear_group(jobs, force_small=True)  # noqa: E731
        lin_ln = lambda x, w, b, ln_w, ln_b, res: ops.linear_layernorm(x, w, b, ln_w, ln_b, res=res, max_rows=Q_tot)  # noqa: E731
        if self.pos_type == "fourier":
            pe = lambda xyz, rs: ops.fourier_pe(xyz, rng, self.position_embedding.gauss_B, d, row_scene=rs)  # noqa: E731
        else:
            pe = lambda xyz, rs: ops.sine_pe(xyz, rng, dim_t, axis, row_scene=rs)  # noqa: E731

        memory_emb = pe(SP, rs_s)
        if self.normalize_box_prediction:
            size_q = torch.cat([(0.5 / (hi - lo)).float().reshape(1, 3).expand(q_off[b + 1] - q_off[b], 3)
                                for b, (lo, hi) in enumerate(ranges)]).contiguous()
        else:
            size_q = torch.full((Q_tot, 3), 0.5, dtype=torch.float32, device=dev)
        inst = ops.layernorm(SL(X, self.input_proj[0]), self.input_proj[1].weight, self.input_proj[1].bias, act="relu")
        mask_feats = SL(SL(X, self.x_mask[0], act="relu"), self.x_mask[2])
        queries = QL(QL(Qin, self.query_proj[0], act="relu"), self.query_proj[2])

        def head(queries, last):
            nq = ops.layernorm(queries, self.out_norm.weight, self.out_norm.bias)
            sem = None
            if last:
                sem = QL(nq, self.out_sem) if isinstance(self.out_sem, nn.Linear) else QL(QL(nq, self.out_sem[0], act="relu"), self.out_sem[2])
            # the B mask-logit products nq_b . mask_feats_b^T: one grouped launch (exact fp32 in every mode: they feed thresholds)
            logits = group([(nq[q_off[b]:q_off[b + 1]], mask_feats[s_off[b]:s_off[b + 1]], None, None, None, None) for b in range(B)])
            bits = ops.mask_bits_batch(logits, [s_off[b + 1] - s_off[b] for b in range(B)], self.mask_attention_threshold)
            score = QL(QL(nq, self.out_score[0], act="relu"), self.out_score[2]) if self.objectness_flag else None
            return nq, sem, logits, bits, score

        nq_pending, sem, logits, bits, score = head(queries, False)
        aux = [dict(cls_preds=None, sem_preds=None, masks=logits, centers=None, sizes=None, scores=score)]
        kv_all = slin(inst, pk["kv_w"], pk["kv_b"])                      # [S_tot, 2*L*d]
        kp_all = slin(memory_emb, pk["kp_w"], pk["kp_b"])                # [S_tot, L*d]
        if self.add_dinox_query_ca:
            m_off, keys2d, near = [0], [], []
            for b in range(B):
                qp = q2d_pos[b]
                if not isinstance(qp, torch.Tensor):
                    qp = qp.tensor.type(sp_pos_wo[b].dtype).to(dev)
                f = q2d_feat[b]
                keys2d.append(torch.cat([f.float(), f.new_ones(1, f.shape[1], dtype=torch.float32)]))
                m_off.append(m_off[-1] + f.shape[0] + 1)
                near.append(ops.near_bits(sp_pos_wo[b].float().contiguous(), qp.float().contiguous(), self.dinox_query_ca_mask_threshold))
            rows_m = q2d_feat[0].shape[0] + 1
            kv2d_all = lin_rows(rows_m, rows_m < ops.BF16_MIN_ROWS)(torch.cat(keys2d).contiguous(), pk["kv2d_w"], pk["kv2d_b"])    # [sum(M_b + 1), 2*L*d]

        ref_points = torch.cat([t.float() for t in q_pos]).contiguous()
        ref_sizes = size_q
        for i in range(L):
            ops.baton_yield()
            jobs, what = [], []
            if self.box_modulate_ca:
                jobs.append(J(queries, self.ref_anchor_head.layers[0], "relu")); what.append("anchor")
            if i > 0:
                jobs.append(J(queries, self.ca_qcontent_proj[i])); what.append("qc")
            jobs.append(J(nq_pending, self.out_cls[0], "relu")); what.append("cls")
            outs = dict(zip(what, group(jobs)))
            jobs, what = [J(outs["cls"], self.out_cls[2])], ["cls"]
            if self.box_modulate_ca:
                jobs.append(J(outs["anchor"], self.ref_anchor_head.layers[1], "sigmoid")); what.append("hwl")
            outs2 = dict(zip(what, group(jobs)))
            aux[-1]["cls_preds"] = outs2["cls"]
            if self.box_modulate_ca:
                pq_emb = ops.sine_pe(ref_points, rng, dim_t, axis, mod_num=outs2["hwl"], mod_den=ref_sizes, row_scene=rs_q)
            else:
                pq_emb = pe(ref_points, rs_q)
            h, qs = group([J(pq_emb, self.ref_point_head.layers[0], "relu"), J(pq_emb, self.ca_qpos_sine_proj[i])])
            query_pos = QL(h, self.ref_point_head.layers[1])
            kc = kv_all[:, i * d:(i + 1) * d]
            v = kv_all[:, (L + i) * d:(L + i + 1) * d]
            kp = kp_all[:, i * d:(i + 1) * d]
            if i == 0:
                qc = qlin(queries, pk["ca_q0_w"], pk["ca_q0_b"], x2=query_pos)
                kc = SL(inst, self.ca_kcontent_proj[0], res=kp)
            else:
                qc = outs["qc"]
            a = torch.empty(Q_tot, d, dtype=torch.float32, device=dev)
            QS = [(q_off[b], q_off[b + 1], s_off[b], s_off[b + 1]) for b in range(B)]
            ops.attention_batch([(qc[q0:q1], kc[k0:k1], v[k0:k1], bits[b], qs[q0:q1], kp[k0:k1], a[q0:q1]) for b, (q0, q1, k0, k1) in enumerate(QS)],
                                H, (2 * d // H) ** -0.5)
            op = self.cross_attn_layers[i].out_proj
            queries = lin_ln(a, op.weight, op.bias, self.norm1[i].weight, self.norm1[i].bias, queries)
            qkv = qlin(queries, pk["sa_qkv_w"][i], pk["sa_qkv_b"][i], x2=query_pos)
            a = torch.empty(Q_tot, d, dtype=torch.float32, device=dev)
            ops.attention_batch([(qkv[q0:q1, :d], qkv[q0:q1, d:2 * d], qkv[q0:q1, 2 * d:], None, None, None, a[q0:q1]) for (q0, q1, _, _) in QS],
                                H, (d // H) ** -0.5)
            op = self.self_attn_layers[i].out_proj
            queries = lin_ln(a, op.weight, op.bias, self.norm2[i].weight, self.norm2[i].bias, queries)
            if self.add_dinox_query_ca:
                layer = self.dinox_query_cross_attn_layers[i]
                q = qlin(queries, pk["q2d_w"][i], pk["q2d_b"][i])
                a = torch.empty(Q_tot, d, dtype=torch.float32, device=dev)
                bits2d = ops.dinox_mask_bits_batch(bits, near)
                ops.attention_batch([(q[q_off[b]:q_off[b + 1]], kv2d_all[m_off[b]:m_off[b + 1], i * d:(i + 1) * d],
                                      kv2d_all[m_off[b]:m_off[b + 1], (L + i) * d:(L + i + 1) * d], bits2d[b], None, None, a[q_off[b]:q_off[b + 1]])
                                     for b in range(B)], H, (d // H) ** -0.5)
                op = layer.attn.out_proj
                if layer.fix:
                    queries = lin_ln(a, op.weight, op.bias, layer.norm.weight, layer.norm.bias, queries)
                else:
                    queries = qlin(a, op.weight, op.bias, res=queries)
            ffn = self.ffn_layers[i]
            hdn = QL(queries, ffn.net[0], act=("relu" if self.activation_fn == "relu" else "gelu"))
            queries = ops.layernorm(qlin(hdn, ffn.net[3].weight, ffn.net[3].bias, res=queries), ffn.norm.weight, ffn.norm.bias)
            if self.add_box_size_pred:
                be, se = self.bbox_embed[i].layers, self.bbox_size_embed[i].layers
                c1, s1 = group([J(queries, be[0], "relu"), J(queries, se[0], "relu")])
                c2, s2 = group([J(c1, be[1], "relu"), J(s1, se[1], "relu")])
                dc, ds = group([J(c2, be[2]), J(s2, se[2])])
            else:
                dc, ds = queries, None
                for li, layer_ in enumerate(self.bbox_embed[i].layers):
                    dc = QL(dc, layer_, act=None if li == len(self.bbox_embed[i].layers) - 1 else "relu")
            center, size, size_metric = ops.box_refine(ref_points, dc, size_q, ds, rng, self.normalize_box_prediction, row_scene=rs_q)
            ref_points = center
            if self.add_box_size_pred:
                ref_sizes = size_q = size
            last = i == L - 1
            nq_pending, sem, logits, bits, score = head(queries, last)
            aux.append(dict(cls_preds=None, sem_preds=sem, masks=logits, centers=center, sizes=size_metric, scores=score))
        aux[-1]["cls_preds"] = QL(QL(nq_pending, self.out_cls[0], act="relu"), self.out_cls[2])

        def scene_view(entry, b):
            q0, q1 = q_off[b], q_off[b + 1]
            row = lambda t: None if t is None else t[q0:q1]  # noqa: E731
            return dict(cls_preds=row(entry["cls_preds"]), sem_preds=row(entry["sem_preds"]), masks=entry["masks"][b],
                        centers=row(entry["centers"]), sizes=row(entry["sizes"]), scores=row(entry["scores"]))
        results = []
        for b in range(B):
            views = [scene_view(e, b) for e in aux]
            final = views.pop()
            final["hidden_states"] = queries[q_off[b]:q_off[b + 1]]
            final["attn_mask_bits"] = bits[b]
            results.append((final, views))
        return results

    # ---- evaluation, positional variant: the row-local work of a layer as row-chain launches (csrc/rowchain.hip) ----------------
    def _fusable(self, rows=1 << 30):
        """Tile rows (16 or 4) of the fused (row-chain) path a scene with `rows` query rows takes, 0 = op by op.  The path covers the
        SegDINO3D prototypes in evaluation: sine positional embedding, 256 channels in 8 heads."""
        want = rows > FUSED_MIN_ROWS if FUSED_DECODER == "auto" else bool(FUSED_DECODER)
        tile = 16
        if FUSED_NARROW and rows <= FUSED_MIN_ROWS and FUSED_DECODER is not False:
            want, tile = True, 4
        return tile if (want and not self.training and self.add_positional_embedding and self.pos_type == "sine" and self.d_model == 256
                and self.num_heads == 8 and self.num_queries == 0 and ops.GEMM_MODE is None and ops.GG_FORCE_NT is None
                and self.in_channels % 16 == 0 and self.ffn_layers[0].net[0].out_features <= 1024
                and self.ffn_layers[0].net[0].out_features % 16 == 0) else 0

    def _forward_fused(self, xs, sp_pos, sp_pos_wo, q_in, q_pos, q2d_feat, q2d_pos, ranges, tile=16):
        """`_forward_scene` for B >= 1 scenes with every query-row-local stretch of a layer as ONE launch (rowchain.Program):
             A  positional query: anchor MLP -> box-modulated sine PE -> ref_point_head, the two cross-attention query projections
                (+ the class head of the previous layer as a second program of the same launch)
             -  masked cross-attention to the superpoints (dense.hip attention kernel, key-split pass only)
             B  combine the key splits -> out-projection + residual + norm1 -> packed self-attention q / k / v projection
             C  self-attention over the scene's queries -> out-projection + norm2 -> 2D-query mask bits -> 2D cross-attention -> FFN
                -> box MLPs + refinement -> head norm (+ semantic head on the last layer)
             -  mask logits (GEMM against the superpoints' mask features) and their attention-mask bits
        6 launches per layer instead of ~27.  Per row the arithmetic is independent of the other rows of a launch: B scenes in one
        call give every scene the bits of its own call."""
        from .rowchain import Program as _Program
        Program = lambda n, r=None: _Program(n, r, rows=tile)  # noqa: E731
        dev = xs[0].device
        d, H, L = self.d_model, self.num_heads, self.num_layers
        B = len(xs)
        pk = self.packed()
        dim_t, axis = self.pe_tables(dev)
        s_off, q_off = [0], [0]
        for b in range(B):
            s_off.append(s_off[-1] + xs[b].shape[0])
            q_off.append(q_off[-1] + q_in[b].shape[0])
        S_tot, Q_tot = s_off[-1], q_off[-1]
        S_b = [s_off[b + 1] - s_off[b] for b in range(B)]
        Q_b = [q_off[b + 1] - q_off[b] for b in range(B)]
        cat = (lambda ts: ts[0].contiguous()) if B == 1 else (lambda ts: torch.cat([t.contiguous() for t in ts]).contiguous())
        X, SP, Qin = cat(list(xs)), cat([t.float() for t in sp_pos]), cat(list(q_in))
        rng = torch.stack([_range6(lo, hi) for lo, hi in ranges]).contiguous()      # [B, 6]

        # ---- superpoint side: big Linears on the existing GEMM kernels, each scene's rows on the tiling its own call would get
        def s_linear(x, w, b=None, act=None, res=None, rows=S_b, offs=s_off, exact=False):
            if B == 1:
                return ops.gather_gemm(x, w, shift=b, act=act, res=res, exact=exact)
            codes = {ops.dense_code(r, w.shape[-1], w.shape[0]) for r in rows}
            small = all(r < ops.BF16_MIN_ROWS for r in rows)
            if len(codes) == 1 and None not in codes and (not ops.bf16_decoder_active() or small or all(r >= ops.BF16_MIN_ROWS for r in rows)):
                return ops.gather_gemm(x, w, shift=b, act=act, res=res, nt=codes.pop(), exact=exact or small)
            out = torch.empty(x.shape[0], w.shape[0], dtype=torch.float32, device=dev)
            for i in range(len(rows)):
                ops.gather_gemm(x[offs[i]:offs[i + 1]], w, shift=b, act=act, res=None if res is None else res[offs[i]:offs[i + 1]],
                                out=out[offs[i]:offs[i + 1]], exact=exact)
            return out
        SL = lambda x, layer, act=None, res=None: s_linear(x, layer.weight, layer.bias, act, res)  # noqa: E731
        if B == 1:
            memory_emb = ops.sine_pe(SP, rng[0], dim_t, axis)
        else:
            rs_s = torch.cat([torch.full((S_b[b],), b, dtype=torch.int32, device=dev) for b in range(B)])
            memory_emb = ops.sine_pe(SP, rng, dim_t, axis, row_scene=rs_s)
        inst = ops.layernorm(SL(X, self.input_proj[0]), self.input_proj[1].weight, self.input_proj[1].bias, act="relu")
        mask_feats = SL(SL(X, self.x_mask[0], act="relu"), self.x_mask[2])
        kv_all = s_linear(inst, pk["kv_w"], pk["kv_b"])                      # [S_tot, 2*L*d]: kc_0..kc_{L-1} | v_0..v_{L-1}
        kp_all = s_linear(memory_emb, pk["kp_w"], pk["kp_b"])                # [S_tot, L*d]
        kc0 = SL(inst, self.ca_kcontent_proj[0], res=kp_all[:, :d])          # layer 0: content + positional key (:669-672)
        m_off, nw_b, near_off = [0], [(s + 31) // 32 for s in S_b], [0]
        kv2d_all = near_all = None
        if self.add_dinox_query_ca:
            keys2d, nears = [], []
            for b in range(B):
                qp = q2d_pos[b]
                if not isinstance(qp, torch.Tensor):
                    qp = qp.tensor.type(sp_pos_wo[b].dtype).to(dev)
                f = q2d_feat[b]
                keys2d.append(torch.cat([f.float(), f.new_ones(1, f.shape[1], dtype=torch.float32)]))
                m_off.append(m_off[-1] + f.shape[0] + 1)
                nears.append(ops.near_bits(sp_pos_wo[b].float().contiguous(), qp.float().contiguous(), self.dinox_query_ca_mask_threshold).reshape(-1))
                near_off.append(near_off[-1] + nears[-1].numel())
            M_b = [m_off[b + 1] - m_off[b] for b in range(B)]
            kv2d_all = s_linear(cat(keys2d), pk["kv2d_w"], pk["kv2d_b"], rows=M_b, offs=m_off)      # [sum(M_b + 1), 2*L*d]
            near_all = nears[0] if B == 1 else torch.cat(nears)
            if near_all.numel() == 0:                              # no 2D query in any scene: only the dummy keys, the table is never read
                near_all = torch.zeros(1, dtype=torch.int32, device=dev)
        nw_max = max(nw_b)
        nw2_max = max(((m_off[b + 1] - m_off[b]) + 31) // 32 for b in range(B)) if self.add_dinox_query_ca else 0

        def scene_table(bits_off=None, ca=None):
            tab = []
            for b in range(B):
                sc = dict(q0=q_off[b], nq=Q_b[b], nw=nw_b[b])
                if self.add_dinox_query_ca:
                    sc.update(m0=m_off[b], nm=m_off[b + 1] - m_off[b], near_off=near_off[b])
                if bits_off is not None:
                    sc["bits_off"] = bits_off[b]
                if ca is not None:
                    sc.update(ksplit=ca[b][0], part_off=ca[b][1])
                tab.append(sc)
            return tab

        def new(cols, rows=Q_tot):
            return torch.empty(rows, cols, dtype=torch.float32, device=dev)

        # ---- the prediction head's mask branch: logits = nq . mask_feats^T per scene, thresholded into one bit buffer ------------
        logit_codes = [ops.dense_code(Q_b[b], d, S_b[b]) for b in range(B)]

        def mask_head(nq):
            pairs = [(nq[q_off[b]:q_off[b + 1]], mask_feats[s_off[b]:s_off[b + 1]]) for b in range(B)]
            if B > 1 and all(c == -1 for c in logit_codes) and ops.GG_HOOK is None:
                logits = ops.linear_group([(a, m, None, None, None, None) for a, m in pairs], force_small=True)
            else:
                logits = [ops.gather_gemm(a, m, exact=True) for a, m in pairs]
            sizes = [Q_b[b] * nw_b[b] for b in range(B)]
            buf = torch.empty(sum(sizes), dtype=torch.int32, device=dev)
            offs = [0]
            for z in sizes:
                offs.append(offs[-1] + z)
            views = [buf[offs[b]:offs[b + 1]].view(Q_b[b], nw_b[b]) for b in range(B)]
            ops.mask_bits_batch(logits, S_b, self.mask_attention_threshold, out=views)
            return logits, views, buf, offs

        def cls_program(P, nq, out):
            """class head of a prediction (feeds nothing inside the decoder): its own program next to the main chain"""
            P.load(0, nq)
            P.linear(1, 0, self.out_cls[0].weight, self.out_cls[0].bias, act="relu")
            P.linear(None, 1, self.out_cls[2].weight, self.out_cls[2].bias, gout=out)

        # ---- query side ---------------------------------------------------------------------------------------------------------
        queries, nq = new(d), new(d)
        P = Program(2)
        P.load(0, Qin)
        P.linear(1, 0, self.query_proj[0].weight, self.query_proj[0].bias, act="relu")
        P.linear(0, 1, self.query_proj[2].weight, self.query_proj[2].bias, gout=queries)
        P.ln(1, 0, self.out_norm.weight, self.out_norm.bias, gout=nq)
        score = None
        if self.objectness_flag:
            score = new(1)
            P.linear(0, 1, self.out_score[0].weight, self.out_score[0].bias, act="relu")
            P.linear(None, 0, self.out_score[2].weight, self.out_score[2].bias, gout=score)
        P.launch(scene_table())
        logits, bits, bits_buf, bits_off = mask_head(nq)
        aux = [dict(cls_preds=None, sem_preds=None, masks=logits, centers=None, sizes=None, scores=score)]

        ref_points = cat([t.float() for t in q_pos])
        if self.normalize_box_prediction:
            size_q = cat([(0.5 / (hi - lo)).float().reshape(1, 3).expand(Q_b[b], 3) for b, (lo, hi) in enumerate(ranges)])
        else:
            size_q = torch.full((Q_tot, 3), 0.5, dtype=torch.float32, device=dev)
        ref_sizes = size_q
        ncls = self.out_cls[2].out_features
        # self-attention inside chain C (a workgroup walks all keys of its scene for its 16 queries: 13 us at 200 keys, 194 us at 3000)
        # or as its own launch (130 us at 3000 x 3000 on the whole chip)
        sa_in_chain = max(Q_b) <= FUSED_SA_MAX_KEYS
        act_ffn = "relu" if self.activation_fn == "relu" else "gelu"
        for i in range(L):
            ops.baton_yield()
            # ---- A: positional query + cross-attention query projections | class head of the previous prediction
            query_pos, qs, qc, cls_prev = new(d), new(d), new(d), new(ncls)
            P = Program(4, rng)
            P.load(0, queries)
            if self.box_modulate_ca:
                ah = self.ref_anchor_head.layers
                P.linear(1, 0, ah[0].weight, ah[0].bias, act="relu")
                P.linear(2, 1, ah[1].weight, ah[1].bias, act="sigmoid")
                P.pe(3, ref_points, dim_t, axis, num_slot=2, den=ref_sizes)
            else:
                P.pe(3, ref_points, dim_t, axis)
            rp = self.ref_point_head.layers
            P.linear(1, 3, rp[0].weight, rp[0].bias, act="relu")
            P.linear(2, 1, rp[1].weight, rp[1].bias, gout=query_pos)
            P.linear(None, 3, self.ca_qpos_sine_proj[i].weight, self.ca_qpos_sine_proj[i].bias, gout=qs)
            if i == 0:
                P.linear(None, 0, pk["ca_q0_w"], pk["ca_q0_b"], src1=2, gout=qc)
            else:
                P.linear(None, 0, self.ca_qcontent_proj[i].weight, self.ca_qcontent_proj[i].bias, gout=qc)
            P.begin()
            cls_program(P, nq, cls_prev)
            P.launch(scene_table())
            aux[-1]["cls_preds"] = cls_prev
            # ---- masked cross-attention to the superpoints (:668-691): key-split pass only, chain B combines
            a = new(d)
            kc = kc0 if i == 0 else kv_all[:, i * d:(i + 1) * d]
            v, kp = kv_all[:, (L + i) * d:(L + i + 1) * d], kp_all[:, i * d:(i + 1) * d]
            jobs = [(qc[q_off[b]:q_off[b + 1]], kc[s_off[b]:s_off[b + 1]], v[s_off[b]:s_off[b + 1]], bits[b], qs[q_off[b]:q_off[b + 1]],
                     kp[s_off[b]:s_off[b + 1]], a[q_off[b]:q_off[b + 1]]) for b in range(B)]
            ws, ca = ops.attention_parts(jobs, H, (2 * d // H) ** -0.5)
            # ---- B: out-projection + norm1, packed self-attention projections (:690-700)
            queries1, qkv = new(d), new(3 * d)
            op = self.cross_attn_layers[i].out_proj
            P = Program(3)
            P.merge(1, ws, a)
            P.load(0, queries)
            P.linear(2, 1, op.weight, op.bias, res=0)
            P.ln(0, 2, self.norm1[i].weight, self.norm1[i].bias, gout=queries1)
            P.load(1, query_pos)
            P.linear(None, 0, pk["sa_qkv_w"][i], pk["sa_qkv_b"][i], src1=1, gout=qkv)
            P.launch(scene_table(ca=ca))
            # ---- C: self-attention, 2D-query cross-attention, FFN, box refinement, head norm
            queries, nq = new(d), new(d)
            center = new(3)
            size = size_metric = None
            P = Program(8, rng)
            P.load(0, queries1)
            if sa_in_chain:
                P.load(1, qkv[:, :d])
                P.attn(2, 1, qkv[:, d:2 * d], qkv[:, 2 * d:], (d // H) ** -0.5, aux=6)
            else:                                                # thousands of keys: the stand-alone kernel spreads them over the chip
                a_sa = new(d)
                ops.attention_batch([(qkv[q_off[b]:q_off[b + 1], :d], qkv[q_off[b]:q_off[b + 1], d:2 * d], qkv[q_off[b]:q_off[b + 1], 2 * d:],
                                      None, None, None, a_sa[q_off[b]:q_off[b + 1]]) for b in range(B)], H, (d // H) ** -0.5)
                P.load(2, a_sa)
            op = self.self_attn_layers[i].out_proj
            P.linear(3, 2, op.weight, op.bias, res=0)
            P.ln(0, 3, self.norm2[i].weight, self.norm2[i].bias)
            if self.add_dinox_query_ca:
                layer = self.dinox_query_cross_attn_layers[i]
                P.bits2d(bits_buf, near_all)
                P.linear(1, 0, pk["q2d_w"][i], pk["q2d_b"][i])
                P.attn(2, 1, kv2d_all[:, i * d:(i + 1) * d], kv2d_all[:, (L + i) * d:(L + i + 1) * d], (d // H) ** -0.5, aux=6, keys_2d=True, masked=True)
                op = layer.attn.out_proj
                if layer.fix:
                    P.linear(3, 2, op.weight, op.bias, res=0)
                    P.ln(0, 3, layer.norm.weight, layer.norm.bias)
                else:
                    P.linear(0, 2, op.weight, op.bias, res=0)
            ffn = self.ffn_layers[i]
            P.linear(4, 0, ffn.net[0].weight, ffn.net[0].bias, act=act_ffn)
            P.linear(3, 4, ffn.net[3].weight, ffn.net[3].bias, res=0)
            P.ln(0, 3, ffn.norm.weight, ffn.norm.bias, gout=queries)
            be = self.bbox_embed[i].layers
            P.linear(1, 0, be[0].weight, be[0].bias, act="relu")
            P.linear(2, 1, be[1].weight, be[1].bias, act="relu")
            P.linear(3, 2, be[2].weight, be[2].bias)
            if self.add_box_size_pred:
                se = self.bbox_size_embed[i].layers
                size, size_metric = new(3), new(3)
                P.linear(1, 0, se[0].weight, se[0].bias, act="relu")
                P.linear(2, 1, se[1].weight, se[1].bias, act="relu")
                P.linear(4, 2, se[2].weight, se[2].bias)
                P.box(ref_points, 3, center, size_prev=size_q, ds_slot=4, size=size, size_metric=size_metric, normalize=self.normalize_box_prediction)
            else:
                P.box(ref_points, 3, center)
            P.ln(1, 0, self.out_norm.weight, self.out_norm.bias, gout=nq)
            last = i == L - 1
            sem = score = None
            if last:
                sem = new(self.num_semantic_classes + 1)
                if isinstance(self.out_sem, nn.Linear):
                    P.linear(None, 1, self.out_sem.weight, self.out_sem.bias, gout=sem)
                else:
                    P.linear(2, 1, self.out_sem[0].weight, self.out_sem[0].bias, act="relu")
                    P.linear(None, 2, self.out_sem[2].weight, self.out_sem[2].bias, gout=sem)
            if self.objectness_flag:
                score = new(1)
                P.linear(2, 1, self.out_score[0].weight, self.out_score[0].bias, act="relu")
                P.linear(None, 2, self.out_score[2].weight, self.out_score[2].bias, gout=score)
            P.launch(scene_table(bits_off=bits_off), nw_max=nw_max, nw2_max=nw2_max)
            ref_points = center
            if self.add_box_size_pred:
                ref_sizes = size_q = size
            logits, bits, bits_buf, bits_off = mask_head(nq)
            aux.append(dict(cls_preds=None, sem_preds=sem, masks=logits, centers=center, sizes=size_metric, scores=score))
        cls_last = new(ncls)
        P = Program(2)
        cls_program(P, nq, cls_last)
        P.launch(scene_table())
        aux[-1]["cls_preds"] = cls_last

        def scene_view(entry, b):
            q0, q1 = q_off[b], q_off[b + 1]
            row = lambda t: None if t is None else t[q0:q1]  # noqa: E731
            return dict(cls_preds=row(entry["cls_preds"]), sem_preds=row(entry["sem_preds"]), masks=entry["masks"][b],
                        centers=row(entry["centers"]), sizes=row(entry["sizes"]), scores=row(entry["scores"]))
        results = []
        for b in range(B):
            views = [scene_view(e, b) for e in aux]
            final = views.pop()
            final["hidden_states"] = queries[q_off[b]:q_off[b + 1]]
            final["attn_mask_bits"] = bits[b]
            results.append((final, views))
        return results

    # ---- reference-shaped entry point (:417-435) --------------------------------------------------
    @ops.bound_stream
    def forward(self, x, sp_pos=None, sp_pos_wo_elastic=None, queries=None, queries_pos=None, dinox_queries=None,
                dinox_query_pos=None, scene_range=None):
        prev, prev_p = getattr(_F_TLS, "f", _EvalF), getattr(_F_TLS, "p", 0.0)
        # train mode drops under no_grad too (nn.Dropout looks at .training only): the autograd nodes run fine without a graph
        _F_TLS.f = _TrainF if (self.training and (torch.is_grad_enabled() or self.dropout > 0.0)) else _EvalF
        _F_TLS.p = self.dropout if _F_TLS.f is _TrainF else 0.0
        try:
            with ops.bf16_decoder_scope(self.compute_dtype == "bf16"):      # training: bf16 forward and backward products of the projections
                return self._forward(x, sp_pos, sp_pos_wo_elastic, queries, queries_pos, dinox_queries, dinox_query_pos, scene_range)
        finally:
            _F_TLS.f, _F_TLS.p = prev, prev_p

    def _forward(self, x, sp_pos, sp_pos_wo_elastic, queries, queries_pos, dinox_queries, dinox_query_pos, scene_range):
        n = len(x)
        finals, auxes = [None] * n, [None] * n
        if not self.add_positional_embedding:
            for j in range(n):
                finals[j], auxes[j] = self._forward_scene_plain(x[j], queries[j])
        else:
            assert (sp_pos is not None) and (queries_pos is not None) and (scene_range is not None)
            wo = sp_pos_wo_elastic if sp_pos_wo_elastic is not None else sp_pos
            pick = lambda lst, ids: None if lst is None else [lst[j] for j in ids]  # noqa: E731
            # Which path a scene takes depends on ITS OWN query rows only (`_fusable`), never on what else is in the call: a scene of a
            # batch gets the bits of its single-scene forward.  The scenes of a kind run together.
            kind = [self._fusable(queries[j].shape[0]) if queries[j].shape[0] > 0 else 0 for j in range(n)]
            plain_ids = [j for j in range(n) if kind[j] == 0]
            for tile in (16, 4):
                fused_ids = [j for j in range(n) if kind[j] == tile]
                for c0 in range(0, len(fused_ids), 16):
                    ids = fused_ids[c0:c0 + 16]
                    out = self._forward_fused(pick(x, ids), pick(sp_pos, ids), pick(wo, ids), pick(queries, ids), pick(queries_pos, ids),
                                              pick(dinox_queries, ids), pick(dinox_query_pos, ids), pick(scene_range, ids), tile=tile)
                    for j, (f, a) in zip(ids, out):
                        finals[j], auxes[j] = f, a
            if plain_ids and self._batchable(pick(x, plain_ids), pick(queries, plain_ids), pick(dinox_queries, plain_ids)):
                out = self._forward_batch(pick(x, plain_ids), pick(sp_pos, plain_ids), pick(wo, plain_ids), pick(queries, plain_ids),
                                          pick(queries_pos, plain_ids), pick(dinox_queries, plain_ids), pick(dinox_query_pos, plain_ids),
                                          pick(scene_range, plain_ids))
                for j, (f, a) in zip(plain_ids, out):
                    finals[j], auxes[j] = f, a
            else:
                for j in plain_ids:
                    finals[j], auxes[j] = self._forward_scene(
                        x[j], sp_pos[j], wo[j], queries[j], queries_pos[j], dinox_queries[j] if dinox_queries is not None else None,
                        dinox_query_pos[j] if dinox_query_pos is not None else None, scene_range[j][0], scene_range[j][1])
        B = len(finals)
        result = dict(cls_preds=[f["cls_preds"] for f in finals], sem_preds=[f["sem_preds"] for f in finals],
                      masks=[f["masks"] for f in finals], scores=[f.get("scores") for f in finals], centers=[f["centers"] for f in finals],
                      sizes=[f["sizes"] for f in finals])
        if getattr(self, "return_hidden_states", True):
            result["hidden_states"] = [f["hidden_states"] for f in finals]
        if getattr(self, "return_aux_outputs", True):
            result["aux_outputs"] = [
                dict(cls_preds=[a[li]["cls_preds"] for a in auxes],
                     sem_preds=None if auxes[0][li]["sem_preds"] is None else [a[li]["sem_preds"] for a in auxes],
                     masks=[a[li]["masks"] for a in auxes], scores=[a[li].get("scores") for a in auxes],
                     centers=[a[li]["centers"] for a in auxes], sizes=[a[li]["sizes"] for a in auxes])
                for li in range(len(auxes[0]))]
        return result
