// Row-chain executor, NARROW tiles: 4 query rows per 8-wave workgroup (program.tile_rows == 4).
// (reference: segdino3d/models/decoder/instance_seg_3d_decoder.py:606-799; the ops, the program and the slot numbering are those of
//  rowchain.hip - a slot is [4][260] fp32 here.)
//
// Why a second tile shape: at the benchmark's 200 queries the 16-row tiles are thirteen workgroups, and a 256 x 256 Linear on 16 rows
// is bound by ONE CU's fp32 matrix rate (3.9 us floor, 5.8 measured; profiles/EXPERIMENTS.md round 4) - the chain lost to ~27 small
// launches per layer that each spread one Linear over 50-200 CUs.  Four rows per workgroup put the same rows on 50 CUs: the matrix
// work of a Linear drops to 0.85 us per CU and what remains is streaming W (256 KB per Linear and workgroup, L2-resident after the
// first workgroup of an XCD touched it).  A row tile this thin has no use for the 16 x 16 MFMA shapes:
//
//   LINEAR  v_mfma_f32_4x4x1_16B_f32: 16 independent 4 x 4 blocks = ONE 4-row x 64-column tile per instruction and channel.
//           lane l supplies A = x[row l & 3][ch] (LDS) and B = W[col 64 tile + l][ch] (global, packed so that a wave's request for four
//           channels is 1 KB back to back); D register i of lane l = out[row i][col 64 tile + l].  The (column tile, K part) tasks of a
//           Linear are dealt to the 8 waves, every wave requests ALL the weights of its task round (32 channel quads = 32 KB per wave,
//           256 KB per CU in flight) before it multiplies: no software pipeline to keep alive across a loop back-edge.  The K parts meet
//           in LDS and are added in a fixed order (part 0, 1, ...) by the epilogue, which also applies bias / residual / activation.
//   ATTN    wave = head.  S^T = K Q^T with the same instruction: block b of a 64-key tile holds keys 4b..4b+3 (A = K rows, one key per
//           lane), B = the 4 query rows; online softmax in the log2 domain per query (lanes l & 3); O^T += V^T P^T block by block (every
//           block accumulates its own 4 keys: 8 channel groups x 4 keys = 32 instructions per tile), the 16 blocks are added across lanes
//           once at the end.
//
// Per row the arithmetic is independent of the other rows of the tile and of the launch, as in rowchain.hip; the summation ORDER differs
// from the 16-row tiles, so which tile shape a scene's decoder runs on is decided by that scene's own query count (decoder._fusable).
#define RC_R 4
#define RC_WAVES 8
#include "rowchain_ops.h"

#define RCN_THREADS (RC_WAVES * 64)
#define RCN_SCRATCH_FLOATS 4096          // K parts x 4 rows x (64-column tiles x 64): <= 16 KB for every LINEAR the check admits

// ------------------------------------------------------------------------------------------------ LINEAR
// one round of a task: RQ channel quads.  All RQ requests leave before the first multiply (sched_barrier: the scheduler would sink them
// next to their use to save registers).  Channel e of every quad feeds accumulator chain e (four independent chains: a lone chain of
// dependent MFMAs issues at a fraction of the pipe rate).
template <int RQ>
__device__ __forceinline__ void rcn_round(const float* __restrict__ wq, const float* a0, const float* a1, int K0, int ch0, f32x4 (&part)[4]) {
    f32x4 w[RQ];
#pragma unroll
    for (int q = 0; q < RQ; ++q) w[q] = *(const f32x4*)(wq + q * 256);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
        const int ch = ch0 + 4 * q;                           // wave-uniform
        const f32x4 a = *(const f32x4*)((ch < K0 ? a0 : a1) + ch);
#pragma unroll
        for (int e = 0; e < 4; ++e) part[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[e], w[q][e], part[e], 0, 0, 0);
    }
}

__device__ __forceinline__ void rcn_linear(const RCOp& op, const RCCtx& cx) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K0 = op.k0, K = op.k0 + op.k1, Cout = op.cout;
    const int nq = K >> 2, ntiles = (Cout + 63) >> 6, CW = ntiles * 64;
    int kp = ntiles >= 8 ? 1 : (ntiles >= 4 ? 2 : (ntiles >= 2 ? 4 : 8));     // K parts: >= 8 tasks for the 8 waves
    while (nq % kp) kp >>= 1;
    const int pq = nq / kp;
    const float* __restrict__ W = (const float*)op.p0;          // packed [tile][quad][lane][4] (rowchain.pack_weight, rows = 4)
    const float* __restrict__ bias = (const float*)op.p1;
    // the epilogue's columns of this thread (tid, tid + 512): bias requested before the contraction
    float bv[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int col = tid + RCN_THREADS * m;
        bv[m] = (bias && col < Cout) ? bias[col] : 0.f;
    }
    const int j = lane & 3;
    const float* a0 = cx.lds + op.src0 * RC_SLOT + j * rc_ld(op.k0);
    const float* a1 = cx.lds + op.src1 * RC_SLOT + j * rc_ld(op.k1) - K0;     // indexed by the global channel
    for (int task = wave; task < ntiles * kp; task += RC_WAVES) {
        const int tile = task / kp, part_i = task - tile * kp;
        const int q0 = part_i * pq;
        const float* wq = W + ((int64_t)(tile * nq + q0) * 64 + lane) * 4;
        f32x4 part[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) part[e] = f32x4{0.f, 0.f, 0.f, 0.f};
        int q = 0;
        for (; pq - q >= 32; q += 32) rcn_round<32>(wq + q * 256, a0, a1, K0, 4 * (q0 + q), part);
        if (pq - q >= 16) { rcn_round<16>(wq + q * 256, a0, a1, K0, 4 * (q0 + q), part); q += 16; }
        if (pq - q >= 8) { rcn_round<8>(wq + q * 256, a0, a1, K0, 4 * (q0 + q), part); q += 8; }
        if (pq - q >= 4) { rcn_round<4>(wq + q * 256, a0, a1, K0, 4 * (q0 + q), part); q += 4; }
        for (; q < pq; ++q) rcn_round<1>(wq + q * 256, a0, a1, K0, 4 * (q0 + q), part);
        const f32x4 acc = (part[0] + part[1]) + (part[2] + part[3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) cx.scratch[(part_i * 4 + i) * CW + tile * 64 + lane] = acc[i];
    }
    __syncthreads();                                           // (also separates every read of the sources from the stores: in-place ops are safe)
    const int ldd = rc_ld(Cout);
    float* dst = cx.lds + op.dst * RC_SLOT;
    const float* res = op.res != 0xFF ? cx.lds + op.res * RC_SLOT : nullptr;
    float* gout = (float*)op.p2;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int col = tid + RCN_THREADS * m;
        if (col >= Cout) continue;
#pragma unroll
        for (int r = 0; r < RC_R; ++r) {
            float y = cx.scratch[r * CW + col];
            for (int p = 1; p < kp; ++p) y += cx.scratch[(p * 4 + r) * CW + col];
            y += bv[m];
            if (res) y += res[r * ldd + col];
            y = rc_act(y, op.act);
            if (!(op.flag & SD3D_RC_F_NO_LDS_DST)) dst[r * ldd + col] = y;
            if (gout && r < cx.nrows) gout[(int64_t)(cx.row0 + r) * op.ld + col] = y;
        }
    }
}

// ------------------------------------------------------------------------------------------------ ATTN
// 4 query rows x Lk keys, wave = head (32 channels).  lane l = (block b = l >> 2, j = l & 3):
//   S tile (64 keys): register v of lane (b, j) = log2-score(key 64 t + 4 b + v, query j)
//   O: register i of accumulator g at lane (b, j) = sum over block b's keys of V[key][8 i + g] * P[key][query j]
__device__ __forceinline__ void rcn_attn(const RCOp& op, const RCCtx& cx, const RCScene& sc) {
    const int lane = threadIdx.x & 63, head = threadIdx.x >> 6;
    const int b = lane >> 2, j = lane & 3;
    const bool keys_2d = (op.flag & SD3D_RC_F_KEYS_2D) != 0;
    const int key0 = keys_2d ? sc.m0 : sc.q0, Lk = keys_2d ? sc.nm : sc.nq;
    const float* __restrict__ Kp = (const float*)op.p0;
    const float* __restrict__ Vp = (const float*)op.p1;
    const int ldk = op.ld;
    const bool masked = (op.flag & SD3D_RC_F_MASK_BITS2D) != 0;
    f32x4 qf[8];                                               // B operand of S^T = K Q^T: query j, pre-scaled into the log2 domain
    {
        const float* q = cx.lds + op.src0 * RC_SLOT + j * RC_LDW + head * 32;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            qf[c] = *(const f32x4*)(q + 4 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) qf[c][e] *= op.f0 * RC_LOG2E;
        }
    }
    f32x4 O[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) O[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;
    const int ntiles = (Lk + 63) >> 6;
    for (int t = 0; t < ntiles; ++t) {
        const int kt0 = t * 64;
        f32x4 kr[8], vr[4][2];
        {
            const float* ks = Kp + (int64_t)(key0 + min(kt0 + lane, Lk - 1)) * ldk + head * 32;
#pragma unroll
            for (int c = 0; c < 8; ++c) kr[c] = *(const f32x4*)(ks + 4 * c);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float* vs = Vp + (int64_t)(key0 + min(kt0 + 4 * b + kk, Lk - 1)) * ldk + head * 32 + 8 * j;
                vr[kk][0] = *(const f32x4*)vs;
                vr[kk][1] = *(const f32x4*)(vs + 4);
            }
        }
        f32x4 Sc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) Sc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) Sc[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(kr[c][e], qf[c][e], Sc[e], 0, 0, 0);
        f32x4 S = (Sc[0] + Sc[1]) + (Sc[2] + Sc[3]);
        uint32_t wbits = 0u;
        if (masked) {
            const uint32_t w = cx.bits2d[j * cx.nw2_max + t * 2 + (b >> 3)];
            wbits = (w >> ((4 * b) & 31)) & 0xFu;
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const bool off = ((wbits >> v) & 1u) || (kt0 + 4 * b + v >= Lk);
            S[v] = off ? -INFINITY : S[v];
            tmax = fmaxf(tmax, S[v]);
        }
#pragma unroll
        for (int d = 4; d <= 32; d <<= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, d));
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
            for (int g = 0; g < 8; ++g) O[g] *= alpha;
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int g = 0; g < 8; ++g) O[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(vr[kk][g >> 2][g & 3], pr[kk], O[g], 0, 0, 0);
    }
    // the 16 blocks of a query meet: fixed butterfly order over the lane bits 2..5
#pragma unroll
    for (int d = 4; d <= 32; d <<= 1) {
        l += __shfl_xor(l, d);
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) O[g][i] += __shfl_xor(O[g][i], d);
    }
    if (b == 0) {
        float* dst = cx.lds + op.dst * RC_SLOT + j * RC_LDW + head * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 o0, o1;
#pragma unroll
            for (int g = 0; g < 4; ++g) { o0[g] = O[g][i] / l; o1[g] = O[4 + g][i] / l; }
            *(f32x4*)(dst + 8 * i) = o0;
            *(f32x4*)(dst + 8 * i + 4) = o1;
        }
    }
}

// ------------------------------------------------------------------------------------------------ the interpreter
__global__ __launch_bounds__(RCN_THREADS) void row_chain_narrow_kernel(const RCProgram P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int si = 0;
    for (int k = 1; k < P.n_scenes; ++k) if ((int)blockIdx.x >= P.tile0[k]) si = k;
    const RCScene& sc = P.scenes[si];
    RCCtx cx;
    cx.lds = smem;
    cx.scratch = smem + P.n_slots * RC_SLOT;
    cx.bits2d = (uint32_t*)(cx.scratch + RCN_SCRATCH_FLOATS);
    cx.nw2_max = P.nw2_max;
    cx.nw_max = P.nw_max;
    cx.open_w = cx.bits2d + RC_R * P.nw2_max;
    const int tile = blockIdx.x - P.tile0[si];
    cx.row0 = sc.q0 + tile * RC_R;
    cx.nrows = min(RC_R, sc.nq - tile * RC_R);
    cx.scene = si;
    const int o_begin = P.prog_begin[blockIdx.y], o_end = P.prog_begin[blockIdx.y + 1];
    for (int o = o_begin; o < o_end; ++o) {
        const RCOp& op = P.ops[o];
        switch (op.type) {
            case SD3D_RC_LOAD: rc_load(op, cx); break;
            case SD3D_RC_STORE: rc_store(op, cx); break;
            case SD3D_RC_LINEAR: rcn_linear(op, cx); break;
            case SD3D_RC_LN: rc_layernorm(op, cx); break;
            case SD3D_RC_PE: rc_pe(op, cx, P); break;
            case SD3D_RC_BOX: rc_box(op, cx, P); break;
            case SD3D_RC_MERGE: rc_merge(op, cx, sc); break;
            case SD3D_RC_BITS2D: rc_bits2d(op, cx, sc); break;
            case SD3D_RC_ATTN: rcn_attn(op, cx, sc); break;
            default: break;
        }
        __syncthreads();
    }
}

const char* rc_check(const RCProgram& P, int R);                 // rowchain.hip

int launch_row_chain_narrow(const RCProgram* P, hipStream_t st) {
    const char* err = rc_check(*P, RC_R);
    if (err) return sd3d_set_error(SD3D_ERR_ARG, err);
    const size_t sm = ((size_t)P->n_slots * RC_SLOT + RCN_SCRATCH_FLOATS + (size_t)RC_R * (P->nw2_max + P->nw_max)) * sizeof(float);
    if (sm > 160 * 1024) return sd3d_set_error(SD3D_ERR_ARG, "row_chain: more than 160 KB of LDS");
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)row_chain_narrow_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    const int tiles = P->tile0[P->n_scenes];
    hipLaunchKernelGGL(row_chain_narrow_kernel, dim3((unsigned)tiles, (unsigned)P->n_programs), dim3(RCN_THREADS), sm, st, *P);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
