// Row-chain executor: the decoder's row-local work as a few fat launches instead of ~25 thin ones per layer
// (reference: segdino3d/models/decoder/instance_seg_3d_decoder.py:606-799 - per layer {positional query, masked cross-attention,
//  self-attention, 2D-query cross-attention, FFN, box refinement, head}; VERDICT r1-r3 "one persistent kernel per decoder layer").
//
// Everything in a decoder layer except (a) the superpoint cross-attention and (b) the mask-logit product couples only the
// channels of ONE query row (Linear, LayerNorm, positional encoding, box refinement) or one query row with a few hundred
// layer-invariant keys (self-attention over the scene's queries, attention over the 2D object queries, the boolean
// mask x distance product).  A workgroup of 16 waves therefore OWNS 16 consecutive query rows of one scene, keeps their
// activations in LDS "slots" ([16][260] fp32, row stride padded by 4 floats: conflict-free ds_read_b128) and interprets a
// small program (RCOp list, passed by value in the kernel arguments) over them:
//
//   LOAD / STORE   slot <-> global rows
//   LINEAR         dst = act(src . W^T + b (+ res)), src = one or two slots (concatenated channels), W straight from L2:
//                  wave w owns the 16-column tiles {w, w + 16, ...}; v_mfma_f32_16x16x4_f32, A = the 16 rows (LDS), B = W rows
//                  (global, requested a 16-MFMA block ahead); exact fp32, fixed summation order (channel ascending)
//   LN             dst = act(LayerNorm(src (+ res)) * g + b), wave w = row w, two-pass statistics
//   PE             (box-modulated) sine positional encoding of the rows' reference points (utils.py:53-105)
//   BOX            iterative box refinement (:735-759)
//   MERGE          gathers the rows of the key-split superpoint cross-attention (dense.hip attention_body partial states)
//   BITS2D         blocked2d[q][m] = no superpoint both open for q and near 2D query m (:722-726), into LDS bit rows
//   ATTN           multi-head attention of the 16 rows over <= a few thousand keys of their scene (K / V from global, L2-resident):
//                  wave = (head, key half); S^T = K Q^T per 16-key tile puts a query in a lane column, online softmax in the
//                  log2 domain, O^T += V^T P^T with the probability registers as B operand; the two key halves meet in LDS
//
// One barrier after every op.  A row's arithmetic never depends on which other rows share its tile or its launch, so the
// rows of several scenes in one launch (a batched evaluation forward) get the bits of their own single-scene launch.
// gridDim.y selects one of up to 4 independent programs over the same rows (e.g. the class head of the previous layer next
// to the positional-query chain of this one): they run on different CUs at the same time.

#define RC_R 16
#define RC_WAVES 16
#include "rowchain_ops.h"

// ------------------------------------------------------------------------------------------------ LINEAR
// acc[t][i] = out[row 4 kq + i][col (wave + 16 t) * 16 + c16]
// NB > 0: the number of 16-MFMA blocks is a compile-time constant and the block loop is unrolled completely - straight-line code, so
// the waitcnt pass counts the outstanding requests exactly (around a loop back-edge it falls back to vmcnt(0): every block then waits
// for the requests issued for the NEXT two, 10 us per 256 x 256 Linear instead of 4).  NB == 0: any K, runtime loop.
template <int NT, int GB, int NB>
__device__ __forceinline__ void rc_linear_body(const RCOp& op, const RCCtx& cx, f32x4 (&acc)[NT]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c16 = lane & 15, kq = lane >> 4;
    const int K0 = op.k0, K = op.k0 + op.k1, Cout = op.cout;
    const float* __restrict__ W = (const float*)op.p0;   // packed
    const float* a0 = cx.lds + op.src0 * RC_SLOT + c16 * rc_ld(op.k0) + 4 * kq;
    const float* a1 = cx.lds + op.src1 * RC_SLOT + c16 * rc_ld(op.k1) + 4 * kq - K0;     // indexed by the global channel
    // W is PRE-PACKED in MFMA-fragment order (rowchain.pack_weight): [16-column tile][16-channel group][lane][4] - the 64 lanes of a
    // request read 1 KB back to back (8 whole cache lines).  From the nn.Linear layout [cout, K] the same request touches 16 rows x
    // 64 bytes - 16 half lines - and cost 2.3 us more per 256 x 256 Linear (8.5 vs 6.2 us, measured).
    const float* wrow[NT];
    const int ngroups = K >> 4;
    const int ntiles = (Cout + 15) >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int tile = wave + 16 * t;
        tile = tile < ntiles ? tile : ntiles - 1;
        wrow[t] = W + ((int64_t)tile * ngroups * 64 + lane) * 4;
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int nb = NB > 0 ? NB : (K >> 4) / GB;
    // W fragments of three 16-MFMA blocks in registers: block b multiplies while b + 1 and b + 2 are in flight (one CU streams the
    // whole W [cout, K] for its 16 rows: 64 KB per block and workgroup, 1 - 2 us of L2 / HBM latency to cover).
    f32x4 w[3][GB][NT];
    auto load_w = [&](f32x4 (&wb)[GB][NT], int blk) {
#pragma unroll
        for (int g = 0; g < GB; ++g)
#pragma unroll
            for (int t = 0; t < NT; ++t) wb[g][t] = *(const f32x4*)(wrow[t] + (blk * GB + g) * 256);
    };
    f32x4 part[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) part[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mma = [&](const f32x4 (&wb)[GB][NT], int blk) {
        f32x4 a[GB];
#pragma unroll
        for (int g = 0; g < GB; ++g) {
            const int ch = (blk * GB + g) * 16;
            a[g] = *(const f32x4*)((ch < K0 ? a0 : a1) + ch);
        }
        // NT = 1: one output tile per wave would be ONE chain of dependent MFMAs (K / 4 of them, ~4x the issue interval apart: a lone
        // wave needs 4.9 us for K = 256); the contraction is therefore dealt to four accumulators (channel e of every quad -> chain e)
        // that are added in a fixed order at the end.
#pragma unroll
        for (int g = 0; g < GB; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (NT == 1) part[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][e], wb[g][t][e], part[e], 0, 0, 0);
                    else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][e], wb[g][t][e], acc[t], 0, 0, 0);
                }
    };
    if (NB > 0) {
        load_w(w[0], 0);
        if (NB > 1) load_w(w[1], 1);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b + 2 < NB) load_w(w[(b + 2) % 3], b + 2);
            __builtin_amdgcn_sched_barrier(0);                // keep the requests ABOVE the block they overlap (the scheduler sinks them to save registers)
            mma(w[b % 3], b);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // (requests past the end are clamped, not skipped: a conditional load costs an s_waitcnt vmcnt(0))
        load_w(w[0], 0);
        load_w(w[1], nb > 1 ? 1 : 0);
        int b = 0;
        while (true) {
            load_w(w[2], b + 2 < nb ? b + 2 : nb - 1); mma(w[0], b); if (++b >= nb) break;
            load_w(w[0], b + 2 < nb ? b + 2 : nb - 1); mma(w[1], b); if (++b >= nb) break;
            load_w(w[1], b + 2 < nb ? b + 2 : nb - 1); mma(w[2], b); if (++b >= nb) break;
        }
    }
    if (NT == 1) acc[0] = (part[0] + part[1]) + (part[2] + part[3]);
}

template <int NT, int GB, int NB>
__device__ __forceinline__ void rc_linear(const RCOp& op, const RCCtx& cx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c16 = lane & 15, kq = lane >> 4;
    const int Cout = op.cout;
    const float* __restrict__ bias = (const float*)op.p1;
    float bv[NT];                                             // requested before the contraction: its latency hides behind the MFMAs
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = (wave + 16 * t) * 16 + c16;
        bv[t] = bias ? bias[col < Cout ? col : Cout - 1] : 0.f;
    }
    f32x4 acc[NT];
    if (wave * 16 < Cout) rc_linear_body<NT, GB, NB>(op, cx, acc);      // (a wave without a column tile - the 3- and 199-column heads - sits the op out)
    if (op.flag & SD3D_RC_F_INPLACE) __syncthreads();         // dst overlaps a source: every wave must have read its inputs first
    const int ldd = rc_ld(Cout);
    float* dst = cx.lds + op.dst * RC_SLOT;
    const float* res = op.res != 0xFF ? cx.lds + op.res * RC_SLOT : nullptr;
    float* gout = (float*)op.p2;                              // optional global copy of the result (rows of this tile)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = (wave + 16 * t) * 16 + c16;
        if (col >= Cout || wave * 16 >= Cout) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * kq + i;
            float y = acc[t][i] + bv[t];
            if (res) y += res[r * ldd + col];
            y = rc_act(y, op.act);
            if (!(op.flag & SD3D_RC_F_NO_LDS_DST)) dst[r * ldd + col] = y;
            if (gout && r < cx.nrows) gout[(int64_t)(cx.row0 + r) * op.ld + col] = y;
        }
    }
}

__device__ __forceinline__ void rc_linear_dispatch(const RCOp& op, const RCCtx& cx) {
    const int ct = (op.cout + 15) >> 4;
    const int nt = (ct + 15) >> 4;
    const int ng = (op.k0 + op.k1) >> 4;
    if (nt <= 1) {                                            // <= 256 output columns: the decoder's d_model-wide Linears and the small heads
        if (ng == 16) rc_linear<1, 4, 4>(op, cx);             // K = 256
        else if (ng == 32) rc_linear<1, 4, 8>(op, cx);        // K = 512 ([queries | query_pos])
        else if (ng == 64) rc_linear<1, 4, 16>(op, cx);       // K = 1024 (second FFN Linear)
        else if (ng == 6) rc_linear<1, 2, 3>(op, cx);         // K = 96 (backbone features)
        else if ((ng & 3) == 0) rc_linear<1, 4, 0>(op, cx);
        else if ((ng & 1) == 0) rc_linear<1, 2, 0>(op, cx);
        else rc_linear<1, 1, 0>(op, cx);
    } else if (nt == 2) {
        if ((ng & 1) == 0) rc_linear<2, 2, 0>(op, cx); else rc_linear<2, 1, 0>(op, cx);
    } else if (nt == 3) {
        if (ng == 32) rc_linear<3, 1, 32>(op, cx);            // packed self-attention q / k / v projection: K = 512 -> 768
        else rc_linear<3, 1, 0>(op, cx);
    } else {
        if (ng == 16) rc_linear<4, 1, 16>(op, cx);            // first FFN Linear: K = 256 -> 1024
        else rc_linear<4, 1, 0>(op, cx);
    }
}

// ------------------------------------------------------------------------------------------------ ATTN
// 16 query rows x Lk keys x 8 heads of 32 channels.  wave = (head = wave & 7, key half = wave >> 3).
// lane (c16, kq):  S tile: S[key 4 kq + v][query c16];  O: O[dv 4 kq + v (+ 16 dt)][query c16]
__device__ __forceinline__ void rc_attn(const RCOp& op, const RCCtx& cx, const RCScene& sc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c16 = lane & 15, kq = lane >> 4;
    const int head = wave & 7, half = wave >> 3;
    const bool keys_2d = (op.flag & SD3D_RC_F_KEYS_2D) != 0;
    const int key0 = keys_2d ? sc.m0 : sc.q0, Lk = keys_2d ? sc.nm : sc.nq;
    const float* __restrict__ Kp = (const float*)op.p0;
    const float* __restrict__ Vp = (const float*)op.p1;
    const int ldk = op.ld;
    const bool masked = (op.flag & SD3D_RC_F_MASK_BITS2D) != 0;
    // queries, pre-scaled into the log2 domain: B operand of S^T = K Q^T:  B[k = channel][j = query c16]
    f32x4 qf[2];
    {
        const float* q = cx.lds + op.src0 * RC_SLOT + c16 * RC_LDW + head * 32 + 4 * kq;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            qf[g] = *(const f32x4*)(q + 16 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) qf[g][e] *= op.f0 * RC_LOG2E;
        }
    }
    f32x4 O[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float m = -INFINITY, l = 0.f;
    const int ntiles = (Lk + 15) >> 4;
    const int t_mid = (ntiles + 1) >> 1;
    const int t_begin = half ? t_mid : 0, t_end = half ? ntiles : t_mid;
    // K rows: A[i = key c16][k = channel];  V: A[i = dv c16 (+16)][k = key 4 kq + e].  The next tile's rows are requested before this
    // tile multiplies (a tile past the end re-reads the last one: unconditional requests).
    auto load_tile = [&](int t, f32x4& k0, f32x4& k1, float (&v0)[4], float (&v1)[4]) {
        const int kt0 = (t < t_end ? t : t_end - 1) * 16;
        const int krow = key0 + min(kt0 + c16, Lk - 1);
        const float* ks = Kp + (int64_t)krow * ldk + head * 32 + 4 * kq;
        k0 = *(const f32x4*)ks;
        k1 = *(const f32x4*)(ks + 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int vrow = key0 + min(kt0 + 4 * kq + e, Lk - 1);
            const float* vs = Vp + (int64_t)vrow * ldk + head * 32 + c16;
            v0[e] = vs[0];
            v1[e] = vs[16];
        }
    };
    f32x4 k0, k1, k0n, k1n;
    float v0[4], v1[4], v0n[4], v1n[4];
    if (t_begin < t_end) load_tile(t_begin, k0, k1, v0, v1);
    for (int t = t_begin; t < t_end; ++t) {
        const int kt0 = t * 16;
        load_tile(t + 1, k0n, k1n, v0n, v1n);
        f32x4 S = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) S = __builtin_amdgcn_mfma_f32_16x16x4f32(k0[e], qf[0][e], S, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) S = __builtin_amdgcn_mfma_f32_16x16x4f32(k1[e], qf[1][e], S, 0, 0, 0);
        // S[v] = log2-score(key kt0 + 4 kq + v, query c16)
        uint32_t wbits = 0u;
        if (masked) {
            const uint32_t w = cx.bits2d[c16 * cx.nw2_max + (t >> 1)];
            wbits = (w >> ((t & 1) * 16 + 4 * kq)) & 0xFu;
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const bool off = ((wbits >> v) & 1u) || (kt0 + 4 * kq + v >= Lk);
            S[v] = off ? -INFINITY : S[v];
            tmax = fmaxf(tmax, S[v]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float mn = fmaxf(m, tmax);
        f32x4 pr = f32x4{0.f, 0.f, 0.f, 0.f};
        if (mn != -INFINITY) {
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            float ls = 0.f;
#pragma unroll
            for (int v = 0; v < 4; ++v) { pr[v] = __builtin_amdgcn_exp2f(S[v] - mn); ls += pr[v]; }
            l = l * alpha + ls;
            m = mn;
#pragma unroll
            for (int v = 0; v < 4; ++v) { O[0][v] *= alpha; O[1][v] *= alpha; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            O[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v0[e], pr[e], O[0], 0, 0, 0);
            O[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v1[e], pr[e], O[1], 0, 0, 0);
        }
        k0 = k0n; k1 = k1n;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v0[e] = v0n[e]; v1[e] = v1n[e]; }
    }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    // the second key half hands (m, l, O) over through the scratch slots; the first combines (fixed order: half 0, half 1)
    float* scratch = cx.lds + op.aux * RC_SLOT + head * (64 * 10);
    if (half == 1) {
        float* s = scratch + lane * 10;
        s[0] = m; s[1] = l;
#pragma unroll
        for (int v = 0; v < 4; ++v) { s[2 + v] = O[0][v]; s[6 + v] = O[1][v]; }
    }
    __syncthreads();
    if (half == 0) {
        const float* s = scratch + lane * 10;
        const float m1 = s[0], l1 = s[1];
        const float M = fmaxf(m, m1);
        const float f0 = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - M);
        const float f1 = (m1 == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m1 - M);
        const float L = l * f0 + l1 * f1;
        float* dst = cx.lds + op.dst * RC_SLOT + c16 * RC_LDW + head * 32 + 4 * kq;
        f32x4 o0, o1;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            o0[v] = (O[0][v] * f0 + s[2 + v] * f1) / L;
            o1[v] = (O[1][v] * f0 + s[6 + v] * f1) / L;
        }
        *(f32x4*)dst = o0;
        *(f32x4*)(dst + 16) = o1;
    }
}

// ------------------------------------------------------------------------------------------------ the interpreter
__global__ __launch_bounds__(1024) void row_chain_kernel(const RCProgram P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int si = 0;
    for (int k = 1; k < P.n_scenes; ++k) if ((int)blockIdx.x >= P.tile0[k]) si = k;
    const RCScene& sc = P.scenes[si];
    RCCtx cx;
    cx.lds = smem;
    cx.bits2d = (uint32_t*)(smem + P.n_slots * RC_SLOT);
    cx.nw2_max = P.nw2_max;
    cx.nw_max = P.nw_max;
    cx.open_w = cx.bits2d + RC_R * P.nw2_max;
    const int tile = blockIdx.x - P.tile0[si];
    cx.row0 = sc.q0 + tile * RC_R;
    cx.nrows = min(RC_R, sc.nq - tile * RC_R);
    cx.scene = si;
    cx.scratch = nullptr;
    const int o_begin = P.prog_begin[blockIdx.y], o_end = P.prog_begin[blockIdx.y + 1];
    for (int o = o_begin; o < o_end; ++o) {
        const RCOp& op = P.ops[o];
        switch (op.type) {
            case SD3D_RC_LOAD: rc_load(op, cx); break;
            case SD3D_RC_STORE: rc_store(op, cx); break;
            case SD3D_RC_LINEAR: rc_linear_dispatch(op, cx); break;
            case SD3D_RC_LN: rc_layernorm(op, cx); break;
            case SD3D_RC_PE: rc_pe(op, cx, P); break;
            case SD3D_RC_BOX: rc_box(op, cx, P); break;
            case SD3D_RC_MERGE: rc_merge(op, cx, sc); break;
            case SD3D_RC_BITS2D: rc_bits2d(op, cx, sc); break;
            case SD3D_RC_ATTN: rc_attn(op, cx, sc); break;
            default: break;
        }
        __syncthreads();
    }
}

// R = rows of a tile (16: this file, 4: rowchain_narrow.hip); slots are R x 260 floats
const char* rc_check(const RCProgram& P, int R) {
    const int SLOT = R * RC_LDW;
    if (P.n_scenes < 1 || P.n_scenes > SD3D_MAX_BATCH) return "row_chain: 1..16 scenes";
    if (P.n_programs < 1 || P.n_programs > SD3D_RC_MAX_PROGRAMS) return "row_chain: 1..4 programs";
    if (P.n_slots < 1 || P.n_slots > 9) return "row_chain: 1..9 LDS slots";
    if (P.prog_begin[0] != 0) return "row_chain: programs start at op 0";
    for (int i = 0; i < P.n_programs; ++i)
        if (P.prog_begin[i + 1] < P.prog_begin[i] || P.prog_begin[i + 1] > SD3D_RC_MAX_OPS) return "row_chain: bad program bounds";
    int tiles = 0;
    for (int s = 0; s < P.n_scenes; ++s) {
        if (P.tile0[s] != tiles) return "row_chain: tile0 must be the prefix sums of ceil(nq / tile_rows)";
        if (P.scenes[s].nq <= 0) return "row_chain: a scene without query rows";
        tiles += (P.scenes[s].nq + R - 1) / R;
    }
    if (P.tile0[P.n_scenes] != tiles) return "row_chain: tile0[n_scenes] != number of tiles";
    auto slot_ok = [&](int slot, int width) { return slot >= 0 && slot * SLOT + R * (width <= 256 ? RC_LDW : width + 4) <= P.n_slots * SLOT; };
    bool needs_rng = false;
    for (int o = 0; o < P.prog_begin[P.n_programs]; ++o) {
        const RCOp& op = P.ops[o];
        switch (op.type) {
            case SD3D_RC_LOAD:
                if (!op.p0 || (op.cout & 3) || (op.ld & 3) || op.cout < 4 || !slot_ok(op.dst, op.cout)) return "row_chain: bad LOAD";
                break;
            case SD3D_RC_STORE:
                if (!op.p0 || op.cout < 1 || !slot_ok(op.src0, op.cout)) return "row_chain: bad STORE";
                break;
            case SD3D_RC_LINEAR:
                if (!op.p0 || op.k0 < 16 || (op.k0 & (R == 16 ? 15 : 3)) || (op.k1 & (R == 16 ? 15 : 3)) || op.cout < 1 || op.cout > 1024 || !slot_ok(op.src0, op.k0) ||
                    (op.k1 && !slot_ok(op.src1, op.k1)) || (!(op.flag & SD3D_RC_F_NO_LDS_DST) && !slot_ok(op.dst, op.cout)) ||
                    (op.res != 0xFF && !slot_ok(op.res, op.cout)) || ((op.flag & SD3D_RC_F_NO_LDS_DST) && !op.p2))
                    return "row_chain: bad LINEAR";
                break;
            case SD3D_RC_LN:
                if (!op.p0 || !op.p1 || !slot_ok(op.src0, 256) || !slot_ok(op.dst, 256) || (op.res != 0xFF && !slot_ok(op.res, 256)) || (op.p2 && (op.ld & 3)))
                    return "row_chain: bad LN";
                break;
            case SD3D_RC_PE:
                needs_rng = true;
                if (!op.p0 || !op.p1 || !op.p2 || !slot_ok(op.dst, 256) || (op.src0 != 0xFF && (!slot_ok(op.src0, 256) || !op.p3))) return "row_chain: bad PE";
                break;
            case SD3D_RC_BOX:
                needs_rng = true;
                if (!op.p0 || !op.p2 || !slot_ok(op.src0, 256) || (op.src1 != 0xFF && (!slot_ok(op.src1, 256) || !op.p1 || !op.p3 || !op.p4))) return "row_chain: bad BOX";
                break;
            case SD3D_RC_MERGE:
                if (!slot_ok(op.dst, 256)) return "row_chain: bad MERGE";
                for (int s = 0; s < P.n_scenes; ++s)
                    if ((P.scenes[s].ksplit > 1 && !op.p0) || (P.scenes[s].ksplit <= 1 && (!op.p1 || (op.ld & 3)))) return "row_chain: MERGE without its source";
                break;
            case SD3D_RC_BITS2D:
                if (!op.p0 || !op.p1) return "row_chain: bad BITS2D";
                for (int s = 0; s < P.n_scenes; ++s)
                    if (P.scenes[s].nw > P.nw_max || (P.scenes[s].nm + 31) / 32 > P.nw2_max || P.scenes[s].nm < 1) return "row_chain: BITS2D sizes exceed the LDS areas";
                break;
            case SD3D_RC_ATTN:
                if (!op.p0 || !op.p1 || (op.ld & 3) || !slot_ok(op.src0, 256) || !slot_ok(op.dst, 256) || (R == 16 && op.aux * SLOT + 8 * 640 > P.n_slots * SLOT))
                    return "row_chain: bad ATTN";
                break;
            default: return "row_chain: unknown op";
        }
    }
    if (needs_rng && !P.rng) return "row_chain: PE / BOX need the scene ranges";
    return nullptr;
}

size_t row_chain_lds_bytes(const RCProgram& P) {
    return ((size_t)P.n_slots * RC_SLOT + (size_t)RC_R * (P.nw2_max + P.nw_max)) * sizeof(float);
}

int launch_row_chain_narrow(const RCProgram* P, hipStream_t st);       // rowchain_narrow.hip

int launch_row_chain(const RCProgram* P, hipStream_t st) {
    if (!P) return sd3d_set_error(SD3D_ERR_ARG, "row_chain: null program");
    if (P->tile_rows == 4) return launch_row_chain_narrow(P, st);
    if (P->tile_rows != 0 && P->tile_rows != 16) return sd3d_set_error(SD3D_ERR_ARG, "row_chain: tile_rows must be 16 (or 0) or 4");
    const char* err = rc_check(*P, RC_R);
    if (err) return sd3d_set_error(SD3D_ERR_ARG, err);
    const size_t sm = row_chain_lds_bytes(*P);
    if (sm > 160 * 1024) return sd3d_set_error(SD3D_ERR_ARG, "row_chain: more than 160 KB of LDS");
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)row_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    const int tiles = P->tile0[P->n_scenes];
    hipLaunchKernelGGL(row_chain_kernel, dim3((unsigned)tiles, (unsigned)P->n_programs), dim3(1024), sm, st, *P);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
