// Backward of the fused multi-head attention; the reference differentiates bmm + masked_fill + softmax
// + bmm (attention.py:361-385) and nn.MultiheadAttention (instance_seg_3d_decoder.py:79) with torch autograd.
//
// With P = softmax(S), S = (scale q) . k (+ (scale q2) . k2), O = P V and the row statistic lse = log sum exp(S) kept by
// the forward pass (sd3d_attention_lse):
//     D[q]  = dO[q] . O[q]                      dP = dO V^T                dS = P (dP - D)
//     dQ    = scale dS K      (dQ2 = scale dS K2)
//     dK    = dS^T (scale Q)  (dK2 likewise)     dV = P^T dO
// Two kernels, both recomputing P tile by tile from lse (nothing of size Lq x Lk is ever stored), both on the exact fp32
// matrix cores (v_mfma_f32_32x32x2_f32) with the forward kernel's register layout:
//   attn_bwd_q_kernel : one workgroup per (32-query tile, head); waves split the key tiles; S^T and dP^T tiles come out
//                       with lane = query, so the softmax algebra is lane-local; dQ^T += K^T dS^T like the forward's O^T += V^T P^T.
//   attn_bwd_kv_kernel: one workgroup per (32-key tile, head); waves split the query tiles; the operand roles are swapped
//                       so that S and dP come out with lane = key; dV^T += dO^T P and dK^T += Q^T dS.
// Partial sums of the waves are added through LDS in a fixed order: bit-reproducible, no atomics.
#include "common.h"
#include "../../include/segdino3d_hip.h"
#include <math.h>
#include <stdlib.h>

struct AttnBwdParams {
    const float* q[2]; int ldq[2];
    const float* k[2]; int ldk[2];
    const float* v; int ldv;
    const uint32_t* bits; int nwords;
    const float* o; int ldo;
    const float* d_o; int ld_do;
    const float* lse;                         // [H][Lq]
    float* dsum;                              // [H][Lq]: D = dO . O (written by the q kernel, read by the kv kernel)
    float* dq[2]; int ld_dq[2];
    float* dk[2]; int ld_dk[2];
    float* dv; int ld_dv;
    int Lq, Lk, H;
    float scale;
};

__device__ __forceinline__ int tile_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// P = exp(S - lse) as v_exp_f32 of (S - lse) * log2(e): one fused multiply-add and the hardware exponential instead of the library
// expf's range handling (the forward kernel forms its probabilities the same way)
#define ABW_LOG2E 1.4426950408889634f
__device__ __forceinline__ float prob_of(float s, float lse_log2) { return __builtin_amdgcn_exp2f(__builtin_fmaf(s, ABW_LOG2E, -lse_log2)); }

template <int NSRC>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [nw][NSRC][32 ch][32 q]
    const int nw = blockDim.x >> 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int head = blockIdx.y;
    const int q0 = blockIdx.x * 32;
    const int qi = min(q0 + i, p.Lq - 1);
    const int hc = head * 32 + h * 16;

    float qreg[NSRC][16], doreg[16];
#pragma unroll
    for (int s = 0; s < NSRC; ++s) {
        const float* src = p.q[s] + (int64_t)qi * p.ldq[s] + hc;
#pragma unroll
        for (int e = 0; e < 16; ++e) qreg[s][e] = src[e] * p.scale;
    }
    float dsum = 0.f;
    {
        const float* dsrc = p.d_o + (int64_t)qi * p.ld_do + hc;
        const float* osrc = p.o + (int64_t)qi * p.ldo + hc;
#pragma unroll
        for (int e = 0; e < 16; ++e) { doreg[e] = dsrc[e]; dsum += dsrc[e] * osrc[e]; }
    }
    dsum += __shfl_xor(dsum, 32);
    if (wave == 0 && h == 0 && q0 + i < p.Lq) p.dsum[(int64_t)head * p.Lq + q0 + i] = dsum;
    const float lse = p.lse[(int64_t)head * p.Lq + qi] * ABW_LOG2E;

    f32x16 dQ[NSRC];
#pragma unroll
    for (int s = 0; s < NSRC; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) dQ[s][r] = 0.f;

    const int ntiles = (p.Lk + 31) >> 5;
    for (int t = wave; t < ntiles; t += nw) {
        const int kt0 = t * 32;
        const int kr = min(kt0 + i, p.Lk - 1);
        f32x16 S, dP;
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < NSRC; ++s) {
            const float* src = p.k[s] + (int64_t)kr * p.ldk[s] + hc;
#pragma unroll
            for (int e = 0; e < 16; ++e) S = __builtin_amdgcn_mfma_f32_32x32x2f32(src[e], qreg[s][e], S, 0, 0, 0);
        }
        {
            const float* src = p.v + (int64_t)kr * p.ldv + hc;
#pragma unroll
            for (int e = 0; e < 16; ++e) dP = __builtin_amdgcn_mfma_f32_32x32x2f32(src[e], doreg[e], dP, 0, 0, 0);
        }
        const uint32_t word = p.bits ? p.bits[(int64_t)qi * p.nwords + t] : 0u;
        float ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kb = tile_row(r, h);
            const bool blocked = ((word >> kb) & 1u) || (kt0 + kb >= p.Lk);
            const float pr = blocked ? 0.f : prob_of(S[r], lse);
            ds[r] = pr * (dP[r] - dsum);
        }
        // dQ^T[c][query] += sum_key K[key][c] dS[query][key];  A = K^T (row = channel i), B = dS^T
#pragma unroll
        for (int s = 0; s < NSRC; ++s) {
            const float* kcol = p.k[s] + head * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = min(kt0 + tile_row(r, h), p.Lk - 1);
                dQ[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(kcol[(int64_t)key * p.ldk[s]], ds[r], dQ[s], 0, 0, 0);
            }
        }
    }
    float* mine = smem + wave * NSRC * 1024;
#pragma unroll
    for (int s = 0; s < NSRC; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[s * 1024 + tile_row(r, h) * 32 + i] = dQ[s][r];
    __syncthreads();
    for (int e = threadIdx.x; e < NSRC * 1024; e += blockDim.x) {
        const int s = e >> 10, qq = (e >> 5) & 31, c = e & 31;          // consecutive threads -> consecutive channels
        float a = 0.f;
        for (int w = 0; w < nw; ++w) a += smem[w * NSRC * 1024 + s * 1024 + c * 32 + qq];
        if (q0 + qq < p.Lq) p.dq[s][(int64_t)(q0 + qq) * p.ld_dq[s] + head * 32 + c] = a * p.scale;
    }
}

template <int NSRC>
__global__ __launch_bounds__(512) void attn_bwd_kv_kernel(const AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [nw][NSRC + 1][32 ch][32 keys]
    const int nw = blockDim.x >> 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int head = blockIdx.y;
    const int k0 = blockIdx.x * 32;
    const int ktile = blockIdx.x;
    const int kj = min(k0 + j, p.Lk - 1);
    const bool key_ok = k0 + j < p.Lk;
    const int hc = head * 32 + h * 16;

    float kreg[NSRC][16], vreg[16];
#pragma unroll
    for (int s = 0; s < NSRC; ++s) {
        const float* src = p.k[s] + (int64_t)kj * p.ldk[s] + hc;
#pragma unroll
        for (int e = 0; e < 16; ++e) kreg[s][e] = src[e];
    }
    {
        const float* src = p.v + (int64_t)kj * p.ldv + hc;
#pragma unroll
        for (int e = 0; e < 16; ++e) vreg[e] = src[e];
    }
    f32x16 dK[NSRC], dV;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        dV[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NSRC; ++s) dK[s][r] = 0.f;
    }

    const int ntiles = (p.Lq + 31) >> 5;
    for (int t = wave; t < ntiles; t += nw) {
        const int qt0 = t * 32;
        const int qr = min(qt0 + j, p.Lq - 1);          // A-operand row = query
        f32x16 S, dP;
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < NSRC; ++s) {
            const float* src = p.q[s] + (int64_t)qr * p.ldq[s] + hc;
#pragma unroll
            for (int e = 0; e < 16; ++e) S = __builtin_amdgcn_mfma_f32_32x32x2f32(src[e] * p.scale, kreg[s][e], S, 0, 0, 0);
        }
        {
            const float* src = p.d_o + (int64_t)qr * p.ld_do + hc;
#pragma unroll
            for (int e = 0; e < 16; ++e) dP = __builtin_amdgcn_mfma_f32_32x32x2f32(src[e], vreg[e], dP, 0, 0, 0);
        }
        // S[r], dP[r]: (query = qt0 + tile_row(r, h), key = k0 + j).  The per-query values (mask word, lse, D) are loaded once per
        // tile by the lane whose index is the query's row in the tile and handed round with readlane - not 48 loads per lane and tile
        const uint32_t my_word = p.bits ? p.bits[(int64_t)qr * p.nwords + ktile] : 0u;
        const float my_lse = p.lse[(int64_t)head * p.Lq + qr] * ABW_LOG2E, my_dsum = p.dsum[(int64_t)head * p.Lq + qr];
        float pr[16], ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row0 = tile_row(r, 0), row1 = tile_row(r, 1);             // compile-time lane numbers
            const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)my_word, row0), w1 = (uint32_t)__builtin_amdgcn_readlane((int)my_word, row1);
            const float l0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_lse), row0));
            const float l1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_lse), row1));
            const float d0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_dsum), row0));
            const float d1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_dsum), row1));
            const uint32_t word = h ? w1 : w0;
            const float lse = h ? l1 : l0, dsum = h ? d1 : d0;
            const int qq = qt0 + (h ? row1 : row0);
            const bool blocked = !key_ok || qq >= p.Lq || ((word >> j) & 1u);
            pr[r] = blocked ? 0.f : prob_of(S[r], lse);
            ds[r] = pr[r] * (dP[r] - dsum);
        }
        // dV^T[dv][key] += sum_q dO[q][dv] P[q][key];  A = dO^T (row = dv = j), B = P
        {
            const float* dcol = p.d_o + head * 32 + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qc = min(qt0 + tile_row(r, h), p.Lq - 1);
                dV = __builtin_amdgcn_mfma_f32_32x32x2f32(dcol[(int64_t)qc * p.ld_do], pr[r], dV, 0, 0, 0);
            }
        }
        // dK^T[c][key] += sum_q (scale Q[q][c]) dS[q][key]
#pragma unroll
        for (int s = 0; s < NSRC; ++s) {
            const float* qcol = p.q[s] + head * 32 + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qc = min(qt0 + tile_row(r, h), p.Lq - 1);
                dK[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(qcol[(int64_t)qc * p.ldq[s]] * p.scale, ds[r], dK[s], 0, 0, 0);
            }
        }
    }
    constexpr int NA = NSRC + 1;
    float* mine = smem + wave * NA * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = tile_row(r, h);
#pragma unroll
        for (int s = 0; s < NSRC; ++s) mine[s * 1024 + row * 32 + j] = dK[s][r];
        mine[NSRC * 1024 + row * 32 + j] = dV[r];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NA * 1024; e += blockDim.x) {
        const int s = e >> 10, kk = (e >> 5) & 31, c = e & 31;
        float a = 0.f;
        for (int w = 0; w < nw; ++w) a += smem[w * NA * 1024 + s * 1024 + c * 32 + kk];
        if (k0 + kk < p.Lk) {
            if (s < NSRC) p.dk[s][(int64_t)(k0 + kk) * p.ld_dk[s] + head * 32 + c] = a;
            else p.dv[(int64_t)(k0 + kk) * p.ld_dv + head * 32 + c] = a;
        }
    }
}

#define ST ((hipStream_t)stream)
extern "C" {

size_t sd3d_attention_backward_ws_bytes(int Lq, int H) { return align_up((size_t)Lq * H * sizeof(float), 256); }

int sd3d_attention_backward(const float* q0, int ldq0, const float* q1, int ldq1, const float* k0, int ldk0, const float* k1, int ldk1,
                            const float* v, int ldv, const uint32_t* mask_bits, int Lq, int Lk, int H, float scale, const float* out, int ldo,
                            const float* lse, const float* d_out, int ld_do, float* dq0, int ld_dq0, float* dq1, int ld_dq1, float* dk0,
                            int ld_dk0, float* dk1, int ld_dk1, float* dv, int ld_dv, void* ws, size_t ws_bytes, void* stream) {
    if ((q1 == nullptr) != (k1 == nullptr) || (q1 && (!dq1 || !dk1))) return sd3d_set_error(SD3D_ERR_ARG, "attention_backward: second source incomplete");
    if (Lq <= 0 || Lk <= 0 || !lse || !out || !d_out) return sd3d_set_error(SD3D_ERR_ARG, "attention_backward: missing forward state");
    if (ws_bytes < sd3d_attention_backward_ws_bytes(Lq, H)) return sd3d_set_error(SD3D_ERR_WS, "attention_backward: workspace too small");
    AttnBwdParams p;
    p.q[0] = q0; p.ldq[0] = ldq0; p.q[1] = q1; p.ldq[1] = ldq1; p.k[0] = k0; p.ldk[0] = ldk0; p.k[1] = k1; p.ldk[1] = ldk1;
    p.v = v; p.ldv = ldv; p.bits = mask_bits; p.nwords = (Lk + 31) / 32; p.o = out; p.ldo = ldo; p.d_o = d_out; p.ld_do = ld_do;
    p.lse = lse; p.dsum = (float*)ws;
    p.dq[0] = dq0; p.ld_dq[0] = ld_dq0; p.dq[1] = dq1; p.ld_dq[1] = ld_dq1; p.dk[0] = dk0; p.ld_dk[0] = ld_dk0; p.dk[1] = dk1; p.ld_dk[1] = ld_dk1;
    p.dv = dv; p.ld_dv = ld_dv; p.Lq = Lq; p.Lk = Lk; p.H = H; p.scale = scale;
    const int nsrc = q1 ? 2 : 1;
    const int kt = (Lk + 31) / 32, qt = (Lq + 31) / 32;
    // kv kernel: 8 waves per workgroup when there are >= 32 query tiles and a 128-register budget (launch bounds 512): at the
    // training shapes (2441 queries x 3000 keys) 752 workgroups x 4 waves at two per SIMD ran two rounds of 19 query tiles each,
    // 611 us; 8 waves x 4 per SIMD: 396 us (4 waves at the same budget: 439).  The q kernel keeps 4 waves and its 256-register
    // budget (302 us; 324 / 362 with the kv kernel's settings).
    const int nwq = kt >= 8 ? 4 : (kt >= 2 ? 2 : 1), nwk = qt >= 32 ? 8 : (qt >= 8 ? 4 : (qt >= 2 ? 2 : 1));
    const dim3 gq((unsigned)qt, (unsigned)H), gk((unsigned)kt, (unsigned)H);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_kv_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 2 * 4096);
        (void)hipFuncSetAttribute((const void*)attn_bwd_kv_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 3 * 4096);
        attr_done = true;
    }
    if (nsrc == 1) {
        attn_bwd_q_kernel<1><<<gq, 64 * nwq, (size_t)nwq * 1 * 4096, ST>>>(p);
        attn_bwd_kv_kernel<1><<<gk, 64 * nwk, (size_t)nwk * 2 * 4096, ST>>>(p);
    } else {
        attn_bwd_q_kernel<2><<<gq, 64 * nwq, (size_t)nwq * 2 * 4096, ST>>>(p);
        attn_bwd_kv_kernel<2><<<gk, 64 * nwk, (size_t)nwk * 3 * 4096, ST>>>(p);
    }
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

}  // extern "C"
