// Row-chain executor, the ops both tile shapes share (rowchain.hip: 16 query rows per 16-wave workgroup; rowchain_narrow.hip: 4 rows per
// 8-wave workgroup).  The including file defines RC_R (rows of a tile) and RC_WAVES (waves of a workgroup, >= RC_R) first.  Every op
// here is local to a row (or to a row and its scene's layer-invariant tables), so its arithmetic - and its bits - are the same in both.
#pragma once
#include "common.h"
#include "../../include/segdino3d_hip.h"

#define RC_LDW 260
#define RC_SLOT (RC_R * RC_LDW)
#define RC_LOG2E 1.4426950408889634f

typedef sd3d_rc_op RCOp;
typedef sd3d_rc_scene RCScene;
typedef sd3d_rc_program RCProgram;

__device__ __forceinline__ int rc_ld(int width) { return width <= 256 ? RC_LDW : width + 4; }

__device__ __forceinline__ float rc_wsum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

__device__ __forceinline__ float rc_act(float y, int act) {
    if (act == 1) return fmaxf(y, 0.f);
    if (act == 2) return 0.5f * y * (1.f + erff(y * 0.70710678118654752440f));
    if (act == 3) return 1.f / (1.f + expf(-y));
    return y;
}

struct RCCtx {
    float* lds;            // slots
    uint32_t* bits2d;      // [RC_R][nw2_max]
    uint32_t* open_w;      // [RC_R][nw_max]
    int nw2_max, nw_max;
    int row0;              // first global query row of this tile
    int nrows;             // valid rows of the tile (1..RC_R)
    int scene;             // scene index
    float* scratch;        // (narrow tiles) partial sums of the K parts of a LINEAR
};


// ------------------------------------------------------------------------------------------------ LN (D = 256)
__device__ __forceinline__ void rc_layernorm(const RCOp& op, const RCCtx& cx) {
    const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
    if (RC_WAVES > RC_R && r >= RC_R) return;
    const float* src = cx.lds + op.src0 * RC_SLOT + r * RC_LDW + 4 * lane;
    f32x4 v = *(const f32x4*)src;
    if (op.res != 0xFF) v += *(const f32x4*)(cx.lds + op.res * RC_SLOT + r * RC_LDW + 4 * lane);
    const float mean = rc_wsum(v[0] + v[1] + v[2] + v[3]) / 256.0f;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(rc_wsum(q) / 256.0f + op.f0);
    const f32x4 g = *(const f32x4*)((const float*)op.p0 + 4 * lane), b = *(const f32x4*)((const float*)op.p1 + 4 * lane);
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        y[e] = (v[e] - mean) * rstd * g[e] + b[e];
        if (op.act == 1) y[e] = fmaxf(y[e], 0.f);
    }
    *(f32x4*)(cx.lds + op.dst * RC_SLOT + r * RC_LDW + 4 * lane) = y;
    float* gout = (float*)op.p2;
    if (gout && r < cx.nrows) *(f32x4*)(gout + (int64_t)(cx.row0 + r) * op.ld + 4 * lane) = y;
}

// ------------------------------------------------------------------------------------------------ LOAD / STORE
__device__ __forceinline__ void rc_load(const RCOp& op, const RCCtx& cx) {
    const int C = op.cout, cv = C >> 2, ldw = rc_ld(C);
    const float* __restrict__ src = (const float*)op.p0;
    float* dst = cx.lds + op.dst * RC_SLOT;
    for (int e = threadIdx.x; e < RC_R * cv; e += blockDim.x) {
        const int r = e / cv, c = (e - r * cv) * 4;
        const int rr = r < cx.nrows ? r : cx.nrows - 1;       // rows past the scene's end repeat its last row (never stored)
        *(f32x4*)(dst + r * ldw + c) = *(const f32x4*)(src + (int64_t)(cx.row0 + rr) * op.ld + c);
    }
}
__device__ __forceinline__ void rc_store(const RCOp& op, const RCCtx& cx) {
    const int C = op.cout, ldw = rc_ld(C);
    float* __restrict__ out = (float*)op.p0;
    const float* src = cx.lds + op.src0 * RC_SLOT;
    if ((C & 3) == 0) {
        const int cv = C >> 2;
        for (int e = threadIdx.x; e < cx.nrows * cv; e += blockDim.x) {
            const int r = e / cv, c = (e - r * cv) * 4;
            *(f32x4*)(out + (int64_t)(cx.row0 + r) * op.ld + c) = *(const f32x4*)(src + r * ldw + c);
        }
    } else {
        for (int e = threadIdx.x; e < cx.nrows * C; e += blockDim.x) {
            const int r = e / C, c = e - r * C;
            out[(int64_t)(cx.row0 + r) * op.ld + c] = src[r * ldw + c];
        }
    }
}

// ------------------------------------------------------------------------------------------------ PE (d = 256)
// the arithmetic of dense.hip sine_pe_kernel, operation for operation
__device__ __forceinline__ void rc_pe(const RCOp& op, const RCCtx& cx, const RCProgram& P) {
    const int r = threadIdx.x >> 6, c4 = (threadIdx.x & 63) * 4;
    if (RC_WAVES > RC_R && r >= RC_R) return;
    const int rr = r < cx.nrows ? r : cx.nrows - 1;
    const float* xyz = (const float*)op.p0 + (int64_t)(cx.row0 + rr) * 3;
    const float* __restrict__ dim_t = (const float*)op.p1;
    const int8_t* __restrict__ axis = (const int8_t*)op.p2;
    const float* den = op.p3 ? (const float*)op.p3 + (int64_t)(cx.row0 + rr) * 3 : nullptr;
    const float* num = op.src0 != 0xFF ? cx.lds + op.src0 * RC_SLOT + r * RC_LDW : nullptr;
    const float* rng = P.rng + 6 * cx.scene;
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = c4 + e;
        const int a = axis[c];
        const float lo = rng[a], hi = rng[3 + a];
        float p = ((xyz[a] - lo) * 1.0f) / (hi - lo) + 0.0f;
        p = p * 6.283185307179586f;
        p = p / dim_t[c];
        float v = (c & 1) ? cosf(p) : sinf(p);
        if (num) v *= num[a] / den[a];
        y[e] = v;
    }
    *(f32x4*)(cx.lds + op.dst * RC_SLOT + r * RC_LDW + c4) = y;
}

// ------------------------------------------------------------------------------------------------ BOX (dense.hip box_refine_kernel)
__device__ __forceinline__ void rc_box(const RCOp& op, const RCCtx& cx, const RCProgram& P) {
    const int t = threadIdx.x;
    if (t >= cx.nrows * 3) return;
    const int r = t / 3, a = t - r * 3;
    const int64_t g = (int64_t)(cx.row0 + r) * 3 + a;
    const float* rng = P.rng + 6 * cx.scene;
    const float dc = cx.lds[op.src0 * RC_SLOT + r * RC_LDW + a];
    ((float*)op.p2)[g] = ((const float*)op.p0)[g] + dc;
    if (op.src1 == 0xFF) return;
    const float ds = cx.lds[op.src1 * RC_SLOT + r * RC_LDW + a];
    const float sp = ((const float*)op.p1)[g];
    float s, so;
    if (op.flag & SD3D_RC_F_NORMALIZE) {
        const float eps = 1e-5f;
        const float x = fminf(fmaxf(sp, 0.f), 1.f);
        const float x1 = fmaxf(x, eps), x2 = fmaxf(1.f - x, eps);
        const float z = logf(x1 / x2) + ds;
        s = 1.0f / (1.0f + expf(-z));
        so = s * (rng[3 + a] - rng[a]);
    } else {
        s = sp + ds;
        so = s;
    }
    ((float*)op.p3)[g] = s;
    ((float*)op.p4)[g] = so;
}

// ------------------------------------------------------------------------------------------------ MERGE
// rows of the superpoint cross-attention: either its finished output (key split 1) or the combination of its partial softmax
// states (dense.hip attention_merge_body: the same expression, the same order over the splits)
__device__ __forceinline__ void rc_merge(const RCOp& op, const RCCtx& cx, const RCScene& sc) {
    float* dst = cx.lds + op.dst * RC_SLOT;
    const int H = 8;
    if (sc.ksplit <= 1) {
        const float* src = (const float*)op.p1;
        for (int e = threadIdx.x; e < RC_R * 64; e += blockDim.x) {
            const int r = e >> 6, c = (e & 63) * 4;
            const int rr = r < cx.nrows ? r : cx.nrows - 1;
            *(f32x4*)(dst + r * RC_LDW + c) = *(const f32x4*)(src + (int64_t)(cx.row0 + rr) * op.ld + c);
        }
        return;
    }
    const float* part = (const float*)op.p0 + sc.part_off;
    const int lrow0 = cx.row0 - sc.q0;                        // row inside the scene
    for (int e = threadIdx.x; e < RC_R * 256; e += blockDim.x) {
        const int r = e >> 8, col = e & 255;
        const int head = col >> 5, dv = col & 31;
        const int lr = lrow0 + (r < cx.nrows ? r : cx.nrows - 1);
        const int bx = lr >> 5, qq = lr & 31;
        const float* base = part + ((int64_t)bx * H + head) * sc.ksplit * (64 + 1024);
        float M = -INFINITY;
        for (int z = 0; z < sc.ksplit; ++z) M = fmaxf(M, base[z * (64 + 1024) + qq]);
        float L = 0.f, acc = 0.f;
        for (int z = 0; z < sc.ksplit; ++z) {
            const float* b = base + z * (64 + 1024);
            const float mz = b[qq];
            const float f = (mz == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mz - M);
            L = __builtin_fmaf(b[32 + qq], f, L);                 // (explicit fused multiply-adds here and in dense.hip attention_merge_body:
            acc = __builtin_fmaf(b[64 + dv * 32 + qq], f, acc);   //  the two must agree bit for bit whatever the compiler would contract)
        }
        dst[r * RC_LDW + col] = acc / L;
    }
}

// ------------------------------------------------------------------------------------------------ BITS2D
// blocked2d[q] bit m = 1 <=> no superpoint is both open for query q and near 2D query m; key Mq (the appended dummy key) is
// always open; bits beyond it are blocked (dense.hip dinox_mask_bits_body).  Wave w makes the 16-key units w, w + RC_WAVES, ...
__device__ __forceinline__ void rc_bits2d(const RCOp& op, const RCCtx& cx, const RCScene& sc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = sc.nw, Mq = sc.nm - 1;
    const uint32_t* __restrict__ blocked = (const uint32_t*)op.p0 + sc.bits_off;
    const uint32_t* __restrict__ near = (const uint32_t*)op.p1 + sc.near_off;
    const int lrow0 = cx.row0 - sc.q0;
    for (int e = threadIdx.x; e < RC_R * nw; e += blockDim.x) {
        const int r = e / nw, w = e - r * nw;
        cx.open_w[r * cx.nw_max + w] = r < cx.nrows ? ~blocked[(int64_t)(lrow0 + r) * nw + w] : 0u;
    }
    __syncthreads();
    const int nw2 = (sc.nm + 31) >> 5, nhalf = nw2 * 2;
    uint16_t* out16 = (uint16_t*)cx.bits2d;
    for (int unit = wave; unit < nhalf; unit += RC_WAVES) {
        uint32_t word[RC_R];
#pragma unroll
        for (int q = 0; q < RC_R; ++q) word[q] = 0u;
        if (nw <= 128) {
            // <= 4096 superpoints: a lane holds two words of every near row; the rows of the unit's 16 keys are requested together
            // (one key at a time the unit costs 16 dependent L2 round trips)
            uint32_t nb[16][2];
#pragma unroll
            for (int jb = 0; jb < 16; ++jb) {
                const int m = unit * 16 + jb;
                const int mc = m < Mq ? m : (Mq > 0 ? Mq - 1 : 0);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int w = lane + 64 * i;
                    nb[jb][i] = (Mq > 0 && w < nw) ? near[(int64_t)mc * nw + w] : 0u;
                }
            }
            uint32_t o[2];
#pragma unroll 1
            for (int q = 0; q < RC_R; ++q) {
                o[0] = lane < nw ? cx.open_w[q * cx.nw_max + lane] : 0u;
                o[1] = lane + 64 < nw ? cx.open_w[q * cx.nw_max + lane + 64] : 0u;
                uint32_t wq = 0u;
#pragma unroll
                for (int jb = 0; jb < 16; ++jb) {
                    const int m = unit * 16 + jb;
                    const bool hit = __ballot(((o[0] & nb[jb][0]) | (o[1] & nb[jb][1])) != 0u) != 0ull;
                    const uint32_t blk = m < Mq ? (hit ? 0u : 1u) : (m == Mq ? 0u : 1u);
                    wq |= blk << jb;
                }
                if (lane == 0) out16[q * (cx.nw2_max * 2) + unit] = (uint16_t)wq;
            }
            continue;
        }
        for (int jb = 0; jb < 16; ++jb) {
            const int m = unit * 16 + jb;                      // wave-uniform
            uint32_t acc[RC_R];
#pragma unroll
            for (int q = 0; q < RC_R; ++q) acc[q] = 0u;
            if (m < Mq) {
                for (int w = lane; w < nw; w += 64) {
                    const uint32_t nbw = near[(int64_t)m * nw + w];
#pragma unroll
                    for (int q = 0; q < RC_R; ++q) acc[q] |= cx.open_w[q * cx.nw_max + w] & nbw;
                }
            }
#pragma unroll
            for (int q = 0; q < RC_R; ++q) {
                const bool hit = __ballot(acc[q] != 0u) != 0ull;
                const uint32_t blk = m < Mq ? (hit ? 0u : 1u) : (m == Mq ? 0u : 1u);
                word[q] |= blk << jb;
            }
        }
        if (lane < RC_R) {
            uint32_t mine = 0u;
#pragma unroll
            for (int q = 0; q < RC_R; ++q) mine = (lane == q) ? word[q] : mine;
            out16[lane * (cx.nw2_max * 2) + unit] = (uint16_t)mine;
        }
    }
}

