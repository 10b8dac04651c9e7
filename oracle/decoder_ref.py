"""ORACLE (test infrastructure, not product code): CPU restatement of the SegDINO3D query decoder.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.

Restates, for ONE scene and without per-scene python lists, what the reference computes in
  - `segdino3d/models/decoder/instance_seg_3d_decoder.py:606-799`  (forward_iter_pred)
  - `.../instance_seg_3d_decoder.py:532-577`                       (_forward_head)
  - `.../instance_seg_3d_decoder.py:60-86`                         (CrossAttentionLayer.forward)
  - `.../instance_seg_3d_decoder.py:173-190`                       (FFN.forward)
  - `segdino3d/models/module/attention.py:186-395`                 (multi_head_attention_forward)
  - `segdino3d/models/module/utils.py:53-105, 167-179`             (sine PE, MLP)
  - `segdino3d/models/module/pc_util.py:48-76`                     (shift_scale_points)
Weights are read from a flat dict keyed exactly like the reference `state_dict`
(prefix "decoder.").  Parity of this file is PINNED by tests/test_oracle_golden.py against golden
vectors produced by the imported reference (tests/golden/make_golden.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class DecoderCfg:
    num_layers: int = 6
    d_model: int = 256
    num_heads: int = 8
    temperature: float = 20.0
    mask_attention_threshold: float = 0.5
    dinox_query_ca_mask_threshold: float = 0.2
    add_dinox_query_ca: bool = True
    add_dinox_query_ca_mask: bool = True
    add_box_size_pred: bool = True
    box_modulate_ca: bool = True
    normalize_box_prediction: bool = True
    activation_fn: str = "gelu"
    add_positional_embedding: bool = True
    pos_type: str = "sine"                 # "fourier": Gaussian random features, buffer position_embedding.gauss_B


class W:
    """state_dict view with a key prefix."""

    def __init__(self, sd, prefix="decoder."):
        self.sd, self.p = sd, prefix

    def __call__(self, name):
        return self.sd[self.p + name]

    def lin(self, x, name):
        return F.linear(x, self(name + ".weight"), self(name + ".bias"))

    def ln(self, x, name, eps=1e-5):
        return F.layer_norm(x, (x.shape[-1],), self(name + ".weight"), self(name + ".bias"), eps)

    def mlp(self, x, name, n_layers):
        for i in range(n_layers):
            x = self.lin(x, f"{name}.layers.{i}")
            if i < n_layers - 1:
                x = torch.relu(x)
        return x


# ------------------------------------------------------------------------------------------------
def pe_channel_plan(d_pos: int = 256, d_in: int = 3):
    """Per-axis channel counts of the sine PE: utils.py:64-82 -> (86, 86, 84) for d_pos=256."""
    ndim = d_pos // d_in
    if ndim % 2:
        ndim -= 1
    rems = d_pos - ndim * d_in
    plan = []
    for _ in range(d_in):
        c = ndim
        if rems > 0:
            c += 2
            rems -= 2
        plan.append(c)
    return plan


def pe_dim_t(cdim: int, temperature: float) -> torch.Tensor:
    """utils.py:85-86 evaluated the same way (fp32 arange, python-float base)."""
    i = torch.arange(cdim, dtype=torch.float32)
    return temperature ** (2 * (i // 2) / cdim)


def sine_pe(xyz, lo, hi, temperature=20.0, d_pos=256, modulated=None):
    """xyz [n,3] metric coords -> [n,d_pos]; lo/hi [3] scene range (utils.py:53-105)."""
    x = (xyz - lo) * 1.0 / (hi - lo) + 0.0          # shift_scale_points with dst = [0, 1]
    out = []
    for d, cdim in enumerate(pe_channel_plan(d_pos, xyz.shape[1])):
        dim_t = pe_dim_t(cdim, temperature)
        raw = x[:, d] * (2 * math.pi)
        pos = raw[:, None] / dim_t
        emb = torch.stack((pos[:, 0::2].sin(), pos[:, 1::2].cos()), dim=2).flatten(1)
        if modulated is not None:
            emb = emb * modulated[:, d:d + 1]
        out.append(emb)
    return torch.cat(out, dim=1)


def inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(0, 1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def attention_core(q, k, v, num_heads, blocked=None):
    """softmax(q k^T * hd^-1/2 [blocked -> -inf]) v per head; q [Lq,E], k [Lk,E], v [Lk,Ev]."""
    Lq, E = q.shape
    Lk, Ev = v.shape
    hd, hv = E // num_heads, Ev // num_heads
    qh = (q * (float(hd) ** -0.5)).view(Lq, num_heads, hd).transpose(0, 1)
    kh = k.view(Lk, num_heads, hd).transpose(0, 1)
    vh = v.view(Lk, num_heads, hv).transpose(0, 1)
    s = torch.bmm(qh, kh.transpose(1, 2))
    if blocked is not None:
        s = s.masked_fill(blocked.unsqueeze(0), float("-inf"))
    s = torch.softmax(s - s.max(dim=-1, keepdim=True)[0], dim=-1)
    return torch.bmm(s, vh).transpose(0, 1).reshape(Lq, Ev)


def forward_head(w: W, queries, mask_feats, last: bool, thr: float):
    """_forward_head (:532-577): cls / sem logits, mask logits, next-layer attention mask."""
    nq = w.ln(queries, "out_norm")
    cls = w.lin(torch.relu(w.lin(nq, "out_cls.0")), "out_cls.2")
    if last:
        if (w.p + "out_sem.weight") in w.sd:
            sem = w.lin(nq, "out_sem")
        else:
            sem = w.lin(torch.relu(w.lin(nq, "out_sem.0")), "out_sem.2")
    else:
        sem = None
    logits = nq @ mask_feats.t()
    blocked = torch.sigmoid(logits) < thr
    dead = blocked.all(dim=1)
    blocked[dead] = False
    return cls, sem, logits, blocked


def objectness_score(w: W, queries):
    """`out_score(out_norm(queries))` (:548-550), present when the decoder was built with objectness_flag; else None."""
    if (w.p + "out_score.0.weight") not in w.sd:
        return None
    nq = w.ln(queries, "out_norm")
    return w.lin(torch.relu(w.lin(nq, "out_score.0")), "out_score.2")


def dinox_blocked_mask(open_mask, pos_wo, q2d_pos, thr):
    """:721-726 - True = query may NOT look at 2D query m; last column (dummy key) always open."""
    dist = torch.cdist(pos_wo, q2d_pos, p=1)
    hits = open_mask.float() @ (dist < thr).float()
    blocked = hits == 0
    return torch.cat([blocked, blocked.new_zeros(blocked.shape[0], 1)], dim=1)


def _packed_mha(w: W, name, q_in, kv_in, num_heads, blocked):
    """nn.MultiheadAttention with packed in-projection on unbatched inputs (decoder :79, :146)."""
    d = q_in.shape[1]
    wi, bi = w(name + ".in_proj_weight"), w(name + ".in_proj_bias")
    q = F.linear(q_in, wi[:d], bi[:d])
    k = F.linear(kv_in, wi[d:2 * d], bi[d:2 * d])
    v = F.linear(kv_in, wi[2 * d:], bi[2 * d:])
    return w.lin(attention_core(q, k, v, num_heads, blocked), name + ".out_proj")


def fourier_pe(xyz, lo, hi, gauss_b, d_pos=256):
    """get_fourier_embeddings (utils.py:107-142): [sin | cos] of (2 pi * normalised xyz) @ gauss_B[:, :d_pos / 2]."""
    x = ((xyz - lo) * 1.0 / (hi - lo) + 0.0) * (2 * math.pi)
    proj = torch.mm(x, gauss_b[:, : d_pos // 2])
    return torch.cat([proj.sin(), proj.cos()], dim=1)


def decoder_forward_plain(sd, cfg: DecoderCfg, x, q_in, prefix="decoder."):
    """Non-positional variant (Baseline_ScanNet200 prototype: add_positional_embedding=False, no 2D-query
    attention): forward_iter_pred :693, :711, :733 with CrossAttentionLayer (:60-86, fix=True),
    SelfAttentionLayer (:133-150) and FFN; no box heads."""
    w = W(sd, prefix)
    H, L = cfg.num_heads, cfg.num_layers
    inst = torch.relu(w.ln(w.lin(x, "input_proj.0"), "input_proj.1"))
    mask_feats = w.lin(torch.relu(w.lin(x, "x_mask.0")), "x_mask.2")
    queries = w.lin(torch.relu(w.lin(q_in, "query_proj.0")), "query_proj.2")
    if (prefix + "query.weight") in sd:                      # learned queries first (`_get_queries` :302-307)
        queries = torch.cat([w("query.weight"), queries])
    cls, sem, logits, blocked = forward_head(w, queries, mask_feats, False, cfg.mask_attention_threshold)
    aux = [dict(cls_preds=cls, masks=logits, centers=None, sizes=None, scores=objectness_score(w, queries))]
    act = F.gelu if cfg.activation_fn == "gelu" else torch.relu
    for i in range(L):
        a = _packed_mha(w, f"cross_attn_layers.{i}.attn", queries, inst, H, blocked)
        queries = w.ln(a + queries, f"cross_attn_layers.{i}.norm")
        a = _packed_mha(w, f"self_attn_layers.{i}.attn", queries, queries, H, None)
        queries = w.ln(a + queries, f"self_attn_layers.{i}.norm")
        h = w.lin(act(w.lin(queries, f"ffn_layers.{i}.net.0")), f"ffn_layers.{i}.net.3")
        queries = w.ln(h + queries, f"ffn_layers.{i}.norm")
        cls, sem, logits, blocked = forward_head(w, queries, mask_feats, i == L - 1, cfg.mask_attention_threshold)
        aux.append(dict(cls_preds=cls, masks=logits, centers=None, sizes=None, scores=objectness_score(w, queries)))
    final = aux[-1]
    return dict(cls_preds=final["cls_preds"], sem_preds=sem, masks=final["masks"], centers=None, sizes=None, scores=final["scores"],
                hidden_states=queries, aux=aux[:-1], attn_blocked=blocked)


def decoder_forward(sd, cfg: DecoderCfg, x, sp_pos, sp_pos_wo, q_in, q_pos, q2d_feat, q2d_pos, lo, hi,
                    prefix="decoder.", trace=None):
    """Returns dict(cls_preds, sem_preds, masks, centers, sizes, hidden_states, aux=[...])."""
    if not cfg.add_positional_embedding:
        return decoder_forward_plain(sd, cfg, x, q_in, prefix)
    w = W(sd, prefix)
    H, L = cfg.num_heads, cfg.num_layers
    d = cfg.d_model
    hd = d // H
    extent = hi - lo
    if cfg.pos_type == "fourier":
        pe = lambda p: fourier_pe(p, lo, hi, w("position_embedding.gauss_B"), d)  # noqa: E731
    else:
        pe = lambda p: sine_pe(p, lo, hi, cfg.temperature, d)  # noqa: E731
    memory_emb = pe(sp_pos)
    Q = q_in.shape[0]
    if cfg.normalize_box_prediction:
        size_q = (1.0 / extent * 0.5).expand(Q, 3)
    else:
        size_q = torch.full((Q, 3), 0.5)
    ref_sizes = size_q
    inst = torch.relu(w.ln(w.lin(x, "input_proj.0"), "input_proj.1"))
    mask_feats = w.lin(torch.relu(w.lin(x, "x_mask.0")), "x_mask.2")
    queries = w.lin(torch.relu(w.lin(q_in, "query_proj.0")), "query_proj.2")
    cls, sem, logits, blocked = forward_head(w, queries, mask_feats, False, cfg.mask_attention_threshold)
    aux = [dict(cls_preds=cls, masks=logits, centers=None, sizes=None)]
    ref_points = q_pos
    act = F.gelu if cfg.activation_fn == "gelu" else torch.relu
    for i in range(L):
        # ---- box-modulated positional query (:659-666)
        if cfg.box_modulate_ca:
            hwl = torch.sigmoid(w.mlp(queries, "ref_anchor_head", 2))
            pq_emb = sine_pe(ref_points, lo, hi, cfg.temperature, d, modulated=hwl / ref_sizes)
        else:
            pq_emb = pe(ref_points)
        query_pos = w.mlp(pq_emb, "ref_point_head", 2)
        # ---- masked cross-attention to superpoints (:668-691)
        qc = w.lin(queries, f"ca_qcontent_proj.{i}")
        kc = w.lin(inst, f"ca_kcontent_proj.{i}")
        v = w.lin(inst, f"ca_v_proj.{i}")
        kp = w.lin(memory_emb, f"ca_kpos_proj.{i}")
        if i == 0:
            qc = qc + w.lin(query_pos, "ca_qpos_proj")
            kc = kc + kp
        qs = w.lin(pq_emb, f"ca_qpos_sine_proj.{i}")
        q_cat = torch.cat([qc.view(Q, H, hd), qs.view(Q, H, hd)], dim=2).reshape(Q, 2 * d)
        k_cat = torch.cat([kc.view(-1, H, hd), kp.view(-1, H, hd)], dim=2).reshape(-1, 2 * d)
        a = attention_core(q_cat, k_cat, v, H, blocked)
        a = w.lin(a, f"cross_attn_layers.{i}.out_proj")
        queries = w.ln(queries + a, f"norm1.{i}")
        if trace is not None:
            trace[f"l{i}.after_ca"] = queries
        # ---- self-attention (:695-709)
        q = w.lin(queries, f"sa_qcontent_proj.{i}") + w.lin(query_pos, f"sa_qpos_proj.{i}")
        k = w.lin(queries, f"sa_kcontent_proj.{i}") + w.lin(query_pos, f"sa_kpos_proj.{i}")
        v = w.lin(queries, f"sa_v_proj.{i}")
        a = attention_core(q, k, v, H)
        a = w.lin(a, f"self_attn_layers.{i}.out_proj")
        queries = w.ln(queries + a, f"norm2.{i}")
        if trace is not None:
            trace[f"l{i}.after_sa"] = queries
        # ---- cross-attention to the cached DINO-X 2D object queries (:713-731, :60-86)
        if cfg.add_dinox_query_ca:
            name = f"dinox_query_cross_attn_layers.{i}"
            keys = torch.cat([q2d_feat, q2d_feat.new_ones(1, q2d_feat.shape[1])], dim=0)
            if cfg.add_dinox_query_ca_mask:
                blk2d = dinox_blocked_mask(~blocked, sp_pos_wo, q2d_pos, cfg.dinox_query_ca_mask_threshold)
            else:
                keys, blk2d = q2d_feat, None
            wi, bi = w(name + ".attn.in_proj_weight"), w(name + ".attn.in_proj_bias")
            q = F.linear(queries, wi[:d], bi[:d])
            k = F.linear(keys, wi[d:2 * d], bi[d:2 * d])
            v = F.linear(keys, wi[2 * d:], bi[2 * d:])
            a = attention_core(q, k, v, H, blk2d)
            a = w.lin(a, name + ".attn.out_proj")
            queries = w.ln(a + queries, name + ".norm")
            if trace is not None:
                trace[f"l{i}.after_2d"] = queries
        # ---- FFN (:173-190)
        h = w.lin(act(w.lin(queries, f"ffn_layers.{i}.net.0")), f"ffn_layers.{i}.net.3")
        queries = w.ln(h + queries, f"ffn_layers.{i}.norm")
        # ---- iterative box refinement (:735-759)
        center = ref_points + w.mlp(queries, f"bbox_embed.{i}", 3)
        ref_points = center.detach()                    # :740 (matters for gradients only)
        if cfg.add_box_size_pred:
            delta = w.mlp(queries, f"bbox_size_embed.{i}", 3)
            size = torch.sigmoid(inverse_sigmoid(size_q) + delta) if cfg.normalize_box_prediction \
                else size_q + delta
            ref_sizes = size.detach()                   # :753
            size_q = ref_sizes
        else:
            size = None
        last = i == L - 1
        cls, sem, logits, blocked = forward_head(w, queries, mask_feats, last, cfg.mask_attention_threshold)
        aux.append(dict(cls_preds=cls, masks=logits, centers=center, sizes=size))
    if cfg.normalize_box_prediction:
        for a_ in aux:
            if a_["sizes"] is not None:
                a_["sizes"] = a_["sizes"] * extent
    final = aux[-1]
    return dict(cls_preds=final["cls_preds"], sem_preds=sem, masks=final["masks"], centers=final["centers"],
                sizes=final["sizes"], hidden_states=queries, aux=aux[:-1], attn_blocked=blocked)


def select_queries(sd, x, x_pos, query_num, prefix="decoder."):
    """Baseline3D._select_queries eval branch with query_num > 0 (baseline3d.py:231-249)."""
    w = W(sd, prefix)
    if query_num <= 0:
        ids = torch.arange(x.shape[0])
        return x, x_pos, ids
    q = w.lin(torch.relu(w.lin(x, "query_proj.0")), "query_proj.2")
    nq = w.ln(q, "out_norm")
    cls = w.lin(torch.relu(w.lin(nq, "out_cls.0")), "out_cls.2")
    score = torch.softmax(cls, dim=-1)[:, :-1].max(dim=1)[0]
    if score.shape[0] > query_num:
        ids = torch.topk(score, query_num, largest=True)[1]
    else:
        ids = torch.arange(score.shape[0])
    return x[ids], x_pos[ids], ids
