// Gather-GEMM with fp32 products evaluated as split-bf16 MFMA sums (opt-in, SD3D_GEMM_MODE=bf16x3|bf16x6).
//
// On gfx950 the fp32-input MFMA runs at the fp32 VECTOR rate (157 TFLOP/s) while the bf16 MFMA is 16x
// faster.  An fp32 number is the exact sum of three bf16 numbers (8 + 8 + 8 mantissa bits), and a
// product of two bf16 values is exact in fp32, so
//     a * b  =  (ah + am + al) * (bh + bm + bl)
//  bf16x3 :  ah*bh + ah*bm + am*bh                          error ~ 2^-16 |a b|   (3 MFMAs,  5.3x the fp32 rate)
//  bf16x6 :  + am*bm + ah*bl + al*bh                        error ~ 2^-24 |a b|   (6 MFMAs,  2.7x the fp32 rate)
// with fp32 accumulation in the matrix core.  bf16x6 matches (slightly beats) the rounding error of an
// fp32 FMA chain of the same length; bf16x3 is ~13x larger (rms 4e-6 relative on a 6912-term reduction).
// Weights are split once on the host into [terms][K][Cout][Cin] bf16; the gathered activations are
// split in registers (v_cvt_pk_bf16_f32).  Structure = the lock-step LDS-shared-weights kernel.
#include "gg_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define SB_LD 40            // bf16 elements per LDS row (32 + 8 pad = 80 bytes: conflict-free ds_read_b128)

template <int NS>
__device__ __forceinline__ void split8(const f32x4& v0, const f32x4& v1, bf16x8 (&t)[NS]) {
    float r[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = v0[i]; r[4 + i] = v1[i]; }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const __bf16 b = (__bf16)r[i];
            t[s][i] = b;
            r[i] -= (float)b;
        }
    }
}

// NS = number of split terms kept per operand: 1 (plain bf16 operands, fp32 accumulate), 2 (bf16x3) or 3 (bf16x6)
template <int NT, int NS>
__global__ __launch_bounds__(256) void gather_gemm_split_kernel(const GGParams p, const __bf16* __restrict__ wsplit) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* Bs = (__bf16*)smem_raw;                            // [2 buffers][NS][NT*32 rows][SB_LD]
    __shared__ unsigned long long wmask[4][2];
    constexpr int ROWS = NT * 32;
    constexpr int BUF = NS * ROWS * SB_LD;
    constexpr int PIECES = NS * ROWS * 4;                      // 16-byte pieces per step
    constexpr int PPT = (PIECES + 255) / 256;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t row_tile = (int64_t)blockIdx.x * 4 + wv;
    const int cg = blockIdx.y;
    const int64_t row0 = row_tile * 32;
    const int64_t row = row0 + j;
    const bool row_ok = row < p.M;
    const int ncol0 = cg * ROWS;
    const int nchunks = p.Cin >> 5;
    const int64_t term_stride = (int64_t)p.K * p.Cout * p.Cin;

    uint64_t m0 = 0, m1 = 0;
    if (p.nbr) {
        for (int k = 0; k < p.K; ++k) {
            const int id = row_ok ? p.nbr[(int64_t)k * p.M + row] : -1;
            const bool any = __ballot(id >= 0) != 0ull;
            if (any) { if (k < 64) m0 |= 1ull << k; else m1 |= 1ull << (k - 64); }
        }
    } else {
        m0 = (row0 < p.M) ? 1ull : 0ull;
    }
    if (lane == 0) { wmask[wv][0] = m0; wmask[wv][1] = m1; }
    __syncthreads();
    uint64_t b0 = wmask[0][0] | wmask[1][0] | wmask[2][0] | wmask[3][0];
    uint64_t b1 = wmask[0][1] | wmask[1][1] | wmask[2][1] | wmask[3][1];
    if (p.ksplit > 1) {
        uint64_t s0 = 0, s1 = 0;
        for (int k = blockIdx.z; k < p.K; k += p.ksplit) { if (k < 64) s0 |= 1ull << k; else s1 |= 1ull << (k - 64); }
        b0 &= s0; b1 &= s1;
    }

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    uint4 bst[PPT];
    auto stage_load = [&](int k, int chunk) {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int f = tid + i * 256;
            if (f < PIECES) {
                const int s = f / (ROWS * 4);
                const int rem = f - s * ROWS * 4;
                int n = ncol0 + (rem >> 2);
                n = n < p.Cout ? n : p.Cout - 1;
                bst[i] = *(const uint4*)(wsplit + s * term_stride + ((int64_t)k * p.Cout + n) * p.Cin + chunk * 32 + (rem & 3) * 8);
            }
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int f = tid + i * 256;
            if (f < PIECES) {
                const int s = f / (ROWS * 4);
                const int rem = f - s * ROWS * 4;
                *(uint4*)(Bs + buf * BUF + (s * ROWS + (rem >> 2)) * SB_LD + (rem & 3) * 8) = bst[i];
            }
        }
    };
    auto load_a = [&](f32x4 (&a)[4], int idx, int chunk) {
        const int c = chunk * 32 + h * 16;
        if (idx >= 0) {
            const float* src = (c < p.C0) ? (p.in0 + (int64_t)idx * p.ld0 + c) : (p.in1 + (int64_t)idx * p.ld1 + (c - p.C0));
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = *(const f32x4*)(src + q * 4);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto row_idx = [&](int k) -> int {
        return p.nbr ? (row_ok ? p.nbr[(int64_t)k * p.M + row] : -1) : (row_ok ? (int)row : -1);
    };

    int k = next_active(b0, b1, -1);
    if (k >= 0) {
        int chunk = 0, buf = 0;
        int idx = row_idx(k);
        f32x4 acur[4];
        load_a(acur, idx, 0);
        stage_load(k, 0);
        stage_store(0);
        __syncthreads();
        while (true) {
            int nk = k, nchunk = chunk + 1, nidx = idx;
            if (nchunk == nchunks) {
                nchunk = 0;
                nk = next_active(b0, b1, k);
                if (nk >= 0) nidx = row_idx(nk);
            }
            const bool has_next = nk >= 0;
            f32x4 anxt[4];
            if (has_next) {
                stage_load(nk, nchunk);
                load_a(anxt, nidx, nchunk);
            }
            const bool mine = (k < 64) ? ((m0 >> k) & 1ull) : ((m1 >> (k - 64)) & 1ull);
            if (mine) {
                // lane half h holds channels h*16 .. h*16+15 of the chunk; 16-channel MFMA group g takes
                // h*16 + g*8 .. +7 from every lane (any bijection works as long as A and B agree)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    bf16x8 at[NS];
                    split8<NS>(acur[2 * g], acur[2 * g + 1], at);
                    const __bf16* bb = Bs + buf * BUF + j * SB_LD + h * 16 + g * 8;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        bf16x8 bt[NS];
#pragma unroll
                        for (int s = 0; s < NS; ++s) bt[s] = *(const bf16x8*)(bb + (s * ROWS + t * 32) * SB_LD);
                        // smallest terms first
                        if (NS == 3) {
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[NS == 3 ? 2 : 0], bt[0], acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[0], bt[NS == 3 ? 2 : 0], acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[NS == 3 ? 1 : 0], bt[NS == 3 ? 1 : 0], acc[t], 0, 0, 0);
                        }
                        if (NS >= 2) {
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[NS >= 2 ? 1 : 0], bt[0], acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[0], bt[NS >= 2 ? 1 : 0], acc[t], 0, 0, 0);
                        }
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[0], bt[0], acc[t], 0, 0, 0);
                    }
                }
            }
            if (!has_next) break;
            stage_store(buf ^ 1);
            __syncthreads();
            buf ^= 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) acur[q] = anxt[q];
            k = nk; chunk = nchunk; idx = nidx;
        }
    }
    if (row0 >= p.M) return;
    if (p.ksplit > 1) {
        float* wsz = p.ws + (int64_t)blockIdx.z * p.M * p.Cout;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = ncol0 + t * 32 + j;
            if (n >= p.Cout) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t rr = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (rr < p.M) wsz[rr * p.Cout + n] = acc[t][r];
            }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = ncol0 + t * 32 + j;
        if (n >= p.Cout) continue;
        const float sc = p.scale ? p.scale[n] : 1.f;
        const float sh = p.shift ? p.shift[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t rr = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (rr >= p.M) continue;
            float y = acc[t][r] * sc + sh;
            if (p.res) y += p.res[rr * p.ld_res + n];
            if (p.act == 1) y = fmaxf(y, 0.f);
            else if (p.act == 2) y = 0.5f * y * (1.f + erff(y * 0.70710678118654752440f));
            else if (p.act == 3) y = 1.f / (1.f + expf(-y));
            p.out[rr * p.ld_out + n] = y;
        }
    }
}

template <int NT, int NS>
static void launch_one(const GGParams& p, const __bf16* w, dim3 grid, hipStream_t st) {
    const size_t sm = (size_t)2 * NS * NT * 32 * SB_LD * sizeof(__bf16);
    static bool attr_set = false;
    if (!attr_set && sm > 65536) {
        (void)hipFuncSetAttribute((const void*)gather_gemm_split_kernel<NT, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        attr_set = true;
    }
    hipLaunchKernelGGL((gather_gemm_split_kernel<NT, NS>), grid, dim3(256), sm, st, p, w);
}

void launch_splitk_epilogue(const GGParams& p, hipStream_t st);

// terms: 1 (plain bf16), 3 (bf16x3) or 6 (bf16x6).  wsplit: [1, 2 or 3][K][Cout][Cin] bf16.
int launch_gather_gemm_split(const GGParams& p_in, int nt, int terms, const void* wsplit, void* ws, size_t ws_bytes,
                             hipStream_t st) {
    GGParams p = p_in;
    p.ksplit = 1;
    p.ws = nullptr;
   
    if (p.M <= 0 || p.Cout <= 0) return SD3D_OK;
    if (terms != 1 && terms != 3 && terms != 6) return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm_split: terms must be 1, 3 or 6");
    if (p.Cin <= 0 || (p.Cin & 31)) return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm_split: Cin must be a positive multiple of 32");
    if (p.in1 && ((p.C0 & 31) || p.C0 > p.Cin)) return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm_split: concat split must be a multiple of 32");
    if (!p.in1) p.C0 = p.Cin;
    if ((p.ld0 & 3) || (p.in1 && (p.ld1 & 3))) return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm_split: input row stride must be a multiple of 4 floats");
    if (p.K > 128) return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm_split: at most 128 kernel offsets");
    if (!p.nbr && p.K != 1) return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm_split: identity gather needs K == 1");
    const int sub = (p.Cout + 31) / 32;
    const int64_t tiles = cdiv(p.M, 32);
    if (nt <= 0) {
        nt = sub >= 4 ? 4 : sub;
        while (nt > 2 && cdiv(tiles, 4) * cdiv(sub, nt) < 256) --nt;
        if (sub % nt) { for (int c = nt; c >= 1; --c) if (sub % c == 0) { nt = c; break; } }
    }
    if (nt > 4) return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm_split: nt must be 1..4");
    p.col_groups = (int)cdiv(p.Cout, 32 * nt);
    const int64_t wgs = cdiv(tiles, 4) * p.col_groups;
    if (p.nbr && p.K >= 8 && wgs < 384) {
        int ksp = wgs < 96 ? 8 : (wgs < 256 ? 4 : 2);
        if ((size_t)ksp * p.M * p.Cout * sizeof(float) <= ws_bytes && ws) { p.ksplit = ksp; p.ws = (float*)ws; }
    }
    const dim3 grid((unsigned)cdiv(tiles, 4), (unsigned)p.col_groups, (unsigned)p.ksplit);
    const __bf16* w = (const __bf16*)wsplit;
    if (terms == 1) {
        switch (nt) {
            case 1: launch_one<1, 1>(p, w, grid, st); break;
            case 2: launch_one<2, 1>(p, w, grid, st); break;
            case 3: launch_one<3, 1>(p, w, grid, st); break;
            default: launch_one<4, 1>(p, w, grid, st); break;
        }
    } else if (terms == 3) {
        switch (nt) {
            case 1: launch_one<1, 2>(p, w, grid, st); break;
            case 2: launch_one<2, 2>(p, w, grid, st); break;
            case 3: launch_one<3, 2>(p, w, grid, st); break;
            default: launch_one<4, 2>(p, w, grid, st); break;
        }
    } else {
        switch (nt) {
            case 1: launch_one<1, 3>(p, w, grid, st); break;
            case 2: launch_one<2, 3>(p, w, grid, st); break;
            case 3: launch_one<3, 3>(p, w, grid, st); break;
            default: launch_one<4, 3>(p, w, grid, st); break;
        }
    }
    if (p.ksplit > 1) launch_splitk_epilogue(p, st);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
