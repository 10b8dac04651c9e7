// Gather-GEMM on the fp32 matrix cores: the one kernel behind every sparse convolution of the
// backbone (reference: ME.MinkowskiConvolution / ConvolutionTranspose, spconv SubMConv3d /
// SparseConv3d / SparseInverseConv3d) and every dense Linear of the
// decoder (K = 1, identity gather).
//
//   out[r][n] = act( scale[n] * ( sum_k sum_c  A[nbr[k][r]][c] * Wt[k][n][c] ) + shift[n] + res[r][n] )
//
// Output-stationary, no atomics, no scatter: one wave64 owns a 32-row x (32*NT)-column output tile
// and keeps it in NT 32x32 accumulators (v_mfma_f32_32x32x2_f32, exact fp32 FMA chains).  For every
// kernel offset k that has at least one neighbour among the wave's 32 rows (rows are consecutive
// voxels on the Z-order curve, so the active offset set is small and coherent) it streams 32-channel
// chunks: each lane loads 16 consecutive channels of its gathered input row straight from HBM/L2
// into registers (one 128-byte line per row per chunk) and 16 channels of NT weight rows (L2
// resident), then issues 16*NT MFMAs.  The next chunk's loads are issued before the current chunk's
// MFMAs.  BatchNorm (folded scale/shift), residual add, ReLU/GELU are fused into the epilogue, so
// each output row is written exactly once.
//
// K-dimension trick: one 32x32x2 MFMA consumes k-slices {0,1}; lane half h = lane>>5 supplies slice
// h.  Any bijection between (instruction index kk, half h) and the 32 channels of a chunk is valid
// as long as A and B agree, so half h takes channels h*16 + kk: 16 contiguous floats per lane,
// loaded as 4 x dwordx4.
#include "gg_common.h"
#include <stdlib.h>

template <int NT>
struct Frag {
    f32x4 a[4];
    f32x4 b[NT][4];
};

template <int NT>
__device__ __forceinline__ void load_frag(Frag<NT>& f, const GGParams& p, int k, int chunk, int idx, int ncol0, int j, int h) {
    const int c = chunk * 32 + h * 16;
    if (idx >= 0) {
        const float* src = (c < p.C0) ? (p.in0 + (int64_t)idx * p.ld0 + c) : (p.in1 + (int64_t)idx * p.ld1 + (c - p.C0));
#pragma unroll
        for (int q = 0; q < 4; ++q) f.a[q] = *(const f32x4*)(src + q * 4);
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) f.a[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int n = ncol0 + t * 32 + j;
        n = n < p.Cout ? n : p.Cout - 1;
        const float* w = p.wt + ((int64_t)k * p.Cout + n) * p.Cin + c;
#pragma unroll
        for (int q = 0; q < 4; ++q) f.b[t][q] = *(const f32x4*)(w + q * 4);
    }
}

template <int NT>
__device__ __forceinline__ void mma_frag(f32x16 (&acc)[NT], const Frag<NT>& f) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[q][e], f.b[t][q][e], acc[t], 0, 0, 0);
}

// KS = 1: every wave owns a (row tile, column group) and walks all (offset, chunk) steps.
// KS = 4: the four waves of a workgroup share one (row tile, column group) and take the steps
//         round-robin (split-K); partial accumulators are reduced through LDS in a fixed order by
//         wave 0.  Used when the launch would otherwise leave most of the 1024 SIMDs idle (coarse
//         levels of the U-Net, decoder Linears with a few hundred rows).
template <int NT, int KS>
__global__ __launch_bounds__(256) void gather_gemm_kernel(const GGParams p) {
    __shared__ float red[(KS > 1) ? (KS - 1) * NT * 16 * 64 : 1];
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int wv = threadIdx.x >> 6;
    const int ks = (KS > 1) ? wv : 0;
    const int64_t unit = (KS > 1) ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * 4 + wv;
    const int64_t row_tile = unit / p.col_groups;
    const int cg = (int)(unit - row_tile * p.col_groups);
    const int64_t row0 = row_tile * 32;
    if (row0 >= p.M) return;                 // uniform per workgroup when KS > 1
    const int64_t row = row0 + j;
    const bool row_ok = row < p.M;
    const int ncol0 = cg * 32 * NT;
    const int nchunks = p.Cin >> 5;

    // which kernel offsets have at least one neighbour among this tile's rows?
    uint64_t m0 = 0, m1 = 0;
    if (p.nbr) {
        for (int k = 0; k < p.K; ++k) {
            const int id = row_ok ? p.nbr[(int64_t)k * p.M + row] : -1;
            const bool any = __ballot(id >= 0) != 0ull;
            if (any) { if (k < 64) m0 |= 1ull << k; else m1 |= 1ull << (k - 64); }
        }
    } else {
        m0 = 1ull;
    }

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // step cursor: (k, chunk) in lexicographic order over the active offsets; this wave owns the
    // steps whose running number is congruent to ks modulo KS.
    int k = next_active(m0, m1, -1);
    int chunk = 0, t_run = 0;
    while (k >= 0 && (t_run % KS) != ks) {
        ++t_run;
        if (++chunk == nchunks) { chunk = 0; k = next_active(m0, m1, k); }
    }
    if (k >= 0) {
        int idx = p.nbr ? (row_ok ? p.nbr[(int64_t)k * p.M + row] : -1) : (row_ok ? (int)row : -1);
        Frag<NT> cur;
        load_frag<NT>(cur, p, k, chunk, idx, ncol0, j, h);
        while (true) {
            int nk = k, nchunk = chunk, nidx = idx;
            do {
                ++t_run;
                if (++nchunk == nchunks) { nchunk = 0; nk = next_active(m0, m1, nk); }
            } while (nk >= 0 && (t_run % KS) != ks);
            const bool has_next = nk >= 0;
            if (has_next && nk != k) nidx = row_ok ? p.nbr[(int64_t)nk * p.M + row] : -1;
            Frag<NT> nxt;
            if (has_next) load_frag<NT>(nxt, p, nk, nchunk, nidx, ncol0, j, h);
            mma_frag<NT>(acc, cur);
            if (!has_next) break;
            cur = nxt;
            k = nk; chunk = nchunk; idx = nidx;
        }
    }

    if (KS > 1) {
        if (ks > 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(((ks - 1) * NT + t) * 16 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (ks > 0) return;
#pragma unroll 1
        for (int s = 0; s < KS - 1; ++s) {          // not unrolled: keeps the live LDS reads to one partial tile
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += red[((s * NT + t) * 16 + r) * 64 + lane];
        }
    }

    // epilogue: acc[t][r] is (row = (r&3) + 8*(r>>2) + 4*h, col = j) of subtile t
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

// ---------------------------------------------------------------------------------------------
// Lock-step variant with LDS-shared weights.  The four waves of a workgroup own four consecutive
// row tiles (128 output rows) of the same column group and walk the (offset, chunk) steps together;
// each step's weight slice W[k][cols][chunk] (32*NT rows x 128 B) is fetched from L2 ONCE per
// workgroup, written to LDS (rows padded to 144 B: conflict-free ds_read_b128) and read by all four
// waves - 4x less L2->CU weight traffic than the private-fragment kernel and 16*NT fewer live VGPRs.
// A rows stay private to the wave (gathered straight into registers, prefetched one step ahead).
// One barrier per step; the next step's global loads are in flight during the current step's MFMAs.
// ---------------------------------------------------------------------------------------------
#define BS_LD 36
template <int NT>
__global__ __launch_bounds__(256) void gather_gemm_lds_kernel(const GGParams p) {
    __shared__ __attribute__((aligned(16))) float Bs[2][NT * 32 * BS_LD];
    __shared__ unsigned long long wmask[4][2];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t row_tile = (int64_t)blockIdx.x * 4 + wv;
    const int cg = blockIdx.y;
    const int64_t row0 = row_tile * 32;
    const int64_t row = row0 + j;
    const bool row_ok = row < p.M;
    const int ncol0 = cg * 32 * NT;
    const int nchunks = p.Cin >> 5;

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
    if (p.ksplit > 1) {            // this workgroup only walks the offsets k with k % ksplit == blockIdx.z
        uint64_t s0 = 0, s1 = 0;
        for (int k = blockIdx.z; k < p.K; k += p.ksplit) { if (k < 64) s0 |= 1ull << k; else s1 |= 1ull << (k - 64); }
        b0 &= s0; b1 &= s1;
    }

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // cooperative staging map: NT*32 rows x 8 float4; thread handles float4 number tid + i*256
    f32x4 bst[NT];
    auto stage_load = [&](int k, int chunk) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int f = tid + i * 256;
            int n = ncol0 + (f >> 3);
            n = n < p.Cout ? n : p.Cout - 1;
            bst[i] = *(const f32x4*)(p.wt + ((int64_t)k * p.Cout + n) * p.Cin + chunk * 32 + (f & 7) * 4);
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int f = tid + i * 256;
            *(f32x4*)(&Bs[buf][(f >> 3) * BS_LD + (f & 7) * 4]) = bst[i];
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
    if (k >= 0) {                                   // uniform over the workgroup
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
            // does this wave have any neighbour at offset k?  (wave-uniform)
            const bool mine = (k < 64) ? ((m0 >> k) & 1ull) : ((m1 >> (k - 64)) & 1ull);
            if (mine) {
                const float* bb = &Bs[buf][j * BS_LD + h * 16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 bq[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) bq[t] = *(const f32x4*)(bb + t * 32 * BS_LD + q * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[q][e], bq[t][e], acc[t], 0, 0, 0);
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
    if (p.ksplit > 1) {            // raw partial sums; splitk_epilogue_kernel reduces them in slice order
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

// out = act(scale * sum_z ws[z] + shift + res): fixed-order reduction of the split-K partials.
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const GGParams p) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= p.M * p.Cout) return;
    const int64_t rr = e / p.Cout;
    const int n = (int)(e - rr * p.Cout);
    float a = 0.f;
    for (int z = 0; z < p.ksplit; ++z) a += p.ws[(int64_t)z * p.M * p.Cout + e];
    float y = a * (p.scale ? p.scale[n] : 1.f) + (p.shift ? p.shift[n] : 0.f);
    if (p.res) y += p.res[rr * p.ld_res + n];
    if (p.act == 1) y = fmaxf(y, 0.f);
    else if (p.act == 2) y = 0.5f * y * (1.f + erff(y * 0.70710678118654752440f));
    else if (p.act == 3) y = 1.f / (1.f + expf(-y));
    p.out[rr * p.ld_out + n] = y;
}

// ---------------------------------------------------------------------------------------------
// Pair-compacted variant for LOW-DENSITY kernel maps (fine levels: only 6-30 % of the (row, offset)
// slots are real pairs, so the output-stationary kernels above multiply mostly zeros).
// A workgroup owns CB_ROWS = 256 consecutive output rows x (32*NT) columns, accumulated in LDS.
// Its four waves take the kernel offsets round-robin; for its offset a wave
//   1. compacts the rows that have a neighbour (4 ballots over the 256 rows) into an LDS list,
//   2. runs the MFMA chunk loop over 32 PAIRS at a time (gathered input rows x W[k], private
//      fragments, next chunk prefetched) - every MFMA row is a real pair,
//   3. adds the 32 x (32*NT) result into the owning output rows of the LDS accumulator (ds_add_f32).
// Weights are read once per 256 rows instead of once per 32.  The accumulation order across
// offsets is not fixed (LDS float atomics), i.e. results may differ in the last bits from run to run.
// ---------------------------------------------------------------------------------------------
#define CB_ROWS 256
template <int NT, int NW>
__global__ __launch_bounds__(64 * NW) void gather_gemm_compact_kernel(const GGParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CW = 32 * NT;
    float* accL = smem;                                              // [CB_ROWS][CW]
    int* l_idx = (int*)(accL + CB_ROWS * CW);                         // [NW][CB_ROWS]
    unsigned short* l_row = (unsigned short*)(l_idx + NW * CB_ROWS);  // [NW][CB_ROWS]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t row_base = (int64_t)blockIdx.x * CB_ROWS;
    const int ncol0 = blockIdx.y * CW;
    const int nchunks = p.Cin >> 5;
    for (int e = tid; e < CB_ROWS * CW; e += 64 * NW) accL[e] = 0.f;
    __syncthreads();
    int* my_idx = l_idx + wv * CB_ROWS;
    unsigned short* my_row = l_row + wv * CB_ROWS;
    const uint64_t lt = (1ull << lane) - 1ull;
    for (int k = wv; k < p.K; k += NW) {
        // all four 64-row slices of the neighbour column are requested before the first ballot, so the
        // list costs one memory round trip instead of four
        int ids[CB_ROWS / 64];
#pragma unroll
        for (int ps = 0; ps < CB_ROWS / 64; ++ps) {
            const int64_t row = row_base + ps * 64 + lane;
            ids[ps] = (row < p.M) ? p.nbr[(int64_t)k * p.M + row] : -1;
        }
        int cnt = 0;
#pragma unroll
        for (int ps = 0; ps < CB_ROWS / 64; ++ps) {
            const uint64_t bal = __ballot(ids[ps] >= 0);
            if (ids[ps] >= 0) {
                const int pos = cnt + __popcll(bal & lt);
                my_idx[pos] = ids[ps];
                my_row[pos] = (unsigned short)(ps * 64 + lane);
            }
            cnt += __popcll(bal);
        }
        __builtin_amdgcn_wave_barrier();
        for (int pc = 0; pc * 32 < cnt; ++pc) {
            const int pp = pc * 32 + j;
            const int pidx = pp < cnt ? my_idx[pp] : -1;
            // output rows of the 16 accumulator rows this lane will hold: fetched from LDS now (off the
            // critical path) so that the scatter after the MFMAs is 16 back-to-back ds_add_f32
            int orow[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int prow = pc * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                orow[r] = prow < cnt ? (int)my_row[prow] : -1;
            }
            f32x16 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            if (!(p.dbg & 2)) {
            Frag<NT> cur;
            load_frag<NT>(cur, p, k, 0, pidx, ncol0, j, h);
            for (int c = 0; c < nchunks; ++c) {
                Frag<NT> nxt;
                if (c + 1 < nchunks) load_frag<NT>(nxt, p, k, c + 1, pidx, ncol0, j, h);
                mma_frag<NT>(acc, cur);
                if (c + 1 < nchunks) cur = nxt;
            }
            }
            if (!(p.dbg & 1)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (orow[r] >= 0) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) atomicAdd(&accL[orow[r] * CW + t * 32 + j], acc[t][r]);
                }
            }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (int e = tid; e < CB_ROWS * CW; e += 64 * NW) {
        const int r = e / CW, nn = e - r * CW;
        const int64_t rr = row_base + r;
        const int n = ncol0 + nn;
        if (rr >= p.M || n >= p.Cout) continue;
        float y = accL[e] * (p.scale ? p.scale[n] : 1.f) + (p.shift ? p.shift[n] : 0.f);
        if (p.res) y += p.res[rr * p.ld_res + n];
        if (p.act == 1) y = fmaxf(y, 0.f);
        else if (p.act == 2) y = 0.5f * y * (1.f + erff(y * 0.70710678118654752440f));
        else if (p.act == 3) y = 1.f / (1.f + expf(-y));
        p.out[rr * p.ld_out + n] = y;
    }
}

// ---------------------------------------------------------------------------------------------
// Pair-compacted kernel, single pass over ALL output columns (Cout = 32*NT <= 128): the gathered
// input rows - the dominant memory traffic of a sparse convolution (4*P*Cin bytes) - are read once
// instead of once per 32-column group.  ROWS output rows per workgroup are accumulated in LDS
// ([ROWS][32*NT] fp32: 48 KB for 128 x 96).  Per pair-chunk the A fragment of a 32-channel chunk is
// loaded once and reused for the NT weight subtiles; B fragments are streamed per (chunk, subtile)
// step with one-step prefetch, which keeps the kernel at ~130 VGPRs.
// ---------------------------------------------------------------------------------------------
template <int NT, int ROWS>
__global__ __launch_bounds__(256) void gather_gemm_compact2_kernel(const GGParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CW = 32 * NT;
    float* accL = smem;                                               // [ROWS][CW]
    int* l_idx = (int*)(accL + ROWS * CW);                            // [4][ROWS]
    unsigned short* l_row = (unsigned short*)(l_idx + 4 * ROWS);      // [4][ROWS]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t row_base = (int64_t)blockIdx.x * ROWS;
    const int nchunks = p.Cin >> 5;
    for (int e = tid; e < ROWS * CW; e += 256) accL[e] = 0.f;
    __syncthreads();
    int* my_idx = l_idx + wv * ROWS;
    unsigned short* my_row = l_row + wv * ROWS;
    const uint64_t lt = (1ull << lane) - 1ull;
    for (int k = wv; k < p.K; k += 4) {
        int ids[ROWS / 64];
#pragma unroll
        for (int ps = 0; ps < ROWS / 64; ++ps) {
            const int64_t row = row_base + ps * 64 + lane;
            ids[ps] = (row < p.M) ? p.nbr[(int64_t)k * p.M + row] : -1;
        }
        int cnt = 0;
#pragma unroll
        for (int ps = 0; ps < ROWS / 64; ++ps) {
            const uint64_t bal = __ballot(ids[ps] >= 0);
            if (ids[ps] >= 0) {
                const int pos = cnt + __popcll(bal & lt);
                my_idx[pos] = ids[ps];
                my_row[pos] = (unsigned short)(ps * 64 + lane);
            }
            cnt += __popcll(bal);
        }
        __builtin_amdgcn_wave_barrier();
        for (int pc = 0; pc * 32 < cnt; ++pc) {
            const int pp = pc * 32 + j;
            const int pidx = pp < cnt ? my_idx[pp] : -1;
            f32x16 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            auto load_a = [&](f32x4 (&a)[4], int chunk) {
                const int c = chunk * 32 + h * 16;
                if (pidx >= 0) {
                    const float* src = (c < p.C0) ? (p.in0 + (int64_t)pidx * p.ld0 + c) : (p.in1 + (int64_t)pidx * p.ld1 + (c - p.C0));
#pragma unroll
                    for (int q = 0; q < 4; ++q) a[q] = *(const f32x4*)(src + q * 4);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) a[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            };
            auto load_b = [&](f32x4 (&b)[4], int chunk, int t) {
                int n = t * 32 + j;
                n = n < p.Cout ? n : p.Cout - 1;
                const float* w = p.wt + ((int64_t)k * p.Cout + n) * p.Cin + chunk * 32 + h * 16;
#pragma unroll
                for (int q = 0; q < 4; ++q) b[q] = *(const f32x4*)(w + q * 4);
            };
            f32x4 acur[4], anxt[4], bcur[4], bnxt[4];
            load_a(acur, 0);
            load_b(bcur, 0, 0);
            for (int c = 0; c < nchunks; ++c) {
                if (c + 1 < nchunks) load_a(anxt, c + 1);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bool last = (t == NT - 1) && (c + 1 == nchunks);
                    if (!last) load_b(bnxt, (t == NT - 1) ? c + 1 : c, (t == NT - 1) ? 0 : t + 1);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[q][e], bcur[q][e], acc[t], 0, 0, 0);
                    if (!last) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) bcur[q] = bnxt[q];
                    }
                }
                if (c + 1 < nchunks) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) acur[q] = anxt[q];
                }
            }
#pragma unroll 4
            for (int r = 0; r < 16; ++r) {
                const int prow = pc * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (prow < cnt) {
                    const int orow = my_row[prow];
#pragma unroll
                    for (int t = 0; t < NT; ++t) atomicAdd(&accL[orow * CW + t * 32 + j], acc[t][r]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (int e = tid; e < ROWS * CW; e += 256) {
        const int r = e / CW, n = e - r * CW;
        const int64_t rr = row_base + r;
        if (rr >= p.M || n >= p.Cout) continue;
        float y = accL[e] * (p.scale ? p.scale[n] : 1.f) + (p.shift ? p.shift[n] : 0.f);
        if (p.res) y += p.res[rr * p.ld_res + n];
        if (p.act == 1) y = fmaxf(y, 0.f);
        else if (p.act == 2) y = 0.5f * y * (1.f + erff(y * 0.70710678118654752440f));
        else if (p.act == 3) y = 1.f / (1.f + expf(-y));
        p.out[rr * p.ld_out + n] = y;
    }
}

template <int NT, int ROWS>
static int launch_compact2(const GGParams& p, hipStream_t st) {
    const size_t sm = (size_t)ROWS * 32 * NT * sizeof(float) + 4 * ROWS * (sizeof(int) + sizeof(unsigned short));
    static bool attr_set = false;
    if (!attr_set && sm > 65536) {
        (void)hipFuncSetAttribute((const void*)gather_gemm_compact2_kernel<NT, ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        attr_set = true;
    }
    hipLaunchKernelGGL((gather_gemm_compact2_kernel<NT, ROWS>), dim3((unsigned)cdiv(p.M, ROWS)), dim3(256), sm, st, p);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Pair-compacted kernel, pipelined across pair-chunks and offsets (32 output columns per pass).
// Same decomposition as gather_gemm_compact_kernel, but a wave first builds the pair lists of a BATCH
// of C3_OB of its offsets (all neighbour-column loads in flight together) and then walks ONE flat
// stream of (offset, pair-chunk, channel-chunk) steps, always loading the next step's fragments before
// the current step's 16 MFMAs - the gather latency, which dominates this kernel, is overlapped across
// pair-chunk and offset boundaries instead of being paid once per pair-chunk.
// ---------------------------------------------------------------------------------------------
#define C3_ROWS 256
#define C3_OB 4
__global__ __launch_bounds__(256) void gather_gemm_compact3_kernel(const GGParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* accL = smem;                                                   // [C3_ROWS][32]
    int* l_idx = (int*)(accL + C3_ROWS * 32);                             // [4 waves][C3_OB][C3_ROWS]
    unsigned char* l_row = (unsigned char*)(l_idx + 4 * C3_OB * C3_ROWS); // [4 waves][C3_OB][C3_ROWS]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t row_base = (int64_t)blockIdx.x * C3_ROWS;
    const int ncol0 = blockIdx.y * 32;
    const int nchunks = p.Cin >> 5;
    for (int e = tid; e < C3_ROWS * 32; e += 256) accL[e] = 0.f;
    __syncthreads();
    int* my_idx = l_idx + wv * C3_OB * C3_ROWS;
    unsigned char* my_row = l_row + wv * C3_OB * C3_ROWS;
    const uint64_t lt = (1ull << lane) - 1ull;
    int n = ncol0 + j;
    n = n < p.Cout ? n : p.Cout - 1;

    for (int kb = wv; kb < p.K; kb += 4 * C3_OB) {          // this wave's offsets: kb, kb+4, ..., kb+4*(C3_OB-1)
        // ---- lists of the batch
        int ids[C3_OB][C3_ROWS / 64];
#pragma unroll
        for (int b = 0; b < C3_OB; ++b) {
            const int k = kb + 4 * b;
#pragma unroll
            for (int ps = 0; ps < C3_ROWS / 64; ++ps) {
                const int64_t row = row_base + ps * 64 + lane;
                ids[b][ps] = (k < p.K && row < p.M) ? p.nbr[(int64_t)k * p.M + row] : -1;
            }
        }
        int cnt[C3_OB];
#pragma unroll
        for (int b = 0; b < C3_OB; ++b) {
            int c = 0;
#pragma unroll
            for (int ps = 0; ps < C3_ROWS / 64; ++ps) {
                const uint64_t bal = __ballot(ids[b][ps] >= 0);
                if (ids[b][ps] >= 0) {
                    const int pos = c + __popcll(bal & lt);
                    my_idx[b * C3_ROWS + pos] = ids[b][ps];
                    my_row[b * C3_ROWS + pos] = (unsigned char)(ps * 64 + lane);
                }
                c += __popcll(bal);
            }
            cnt[b] = c;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- flat step stream over (b, pc, chunk)
        auto first_group = [&](int& b, int& pc) {           // first (b, pc) with pc*32 < cnt[b], b from `b`
            while (b < C3_OB) {
                int cb = 0;
#pragma unroll
                for (int q = 0; q < C3_OB; ++q) cb = (q == b) ? cnt[q] : cb;
                if (pc * 32 < cb) return true;
                ++b; pc = 0;
            }
            return false;
        };
        auto load = [&](Frag<1>& f, int b, int pc, int chunk) {
            const int pp = pc * 32 + j;
            int cb = 0;
#pragma unroll
            for (int q = 0; q < C3_OB; ++q) cb = (q == b) ? cnt[q] : cb;
            const int pidx = pp < cb ? my_idx[b * C3_ROWS + pp] : -1;
            const int c = chunk * 32 + h * 16;
            if (pidx >= 0) {
                const float* src = (c < p.C0) ? (p.in0 + (int64_t)pidx * p.ld0 + c) : (p.in1 + (int64_t)pidx * p.ld1 + (c - p.C0));
#pragma unroll
                for (int q = 0; q < 4; ++q) f.a[q] = *(const f32x4*)(src + q * 4);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) f.a[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const float* w = p.wt + ((int64_t)(kb + 4 * b) * p.Cout + n) * p.Cin + c;
#pragma unroll
            for (int q = 0; q < 4; ++q) f.b[0][q] = *(const f32x4*)(w + q * 4);
        };
        int b = 0, pc = 0, chunk = 0;
        if (!first_group(b, pc)) continue;
        f32x16 acc[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
        Frag<1> cur;
        load(cur, b, pc, 0);
        while (true) {
            int nb = b, npc = pc, nchunk = chunk + 1;
            bool has_next = true;
            if (nchunk == nchunks) { nchunk = 0; ++npc; has_next = first_group(nb, npc); }
            Frag<1> nxt;
            if (has_next) load(nxt, nb, npc, nchunk);
            mma_frag<1>(acc, cur);
            if (nchunk == 0) {                              // the (b, pc) group is complete: scatter it
                int cb = 0;
#pragma unroll
                for (int q = 0; q < C3_OB; ++q) cb = (q == b) ? cnt[q] : cb;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int prow = pc * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (prow < cb) atomicAdd(&accL[(int)my_row[b * C3_ROWS + prow] * 32 + j], acc[0][r]);
                    acc[0][r] = 0.f;
                }
            }
            if (!has_next) break;
            cur = nxt;
            b = nb; pc = npc; chunk = nchunk;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (int e = tid; e < C3_ROWS * 32; e += 256) {
        const int r = e >> 5, nn = e & 31;
        const int64_t rr = row_base + r;
        const int nc = ncol0 + nn;
        if (rr >= p.M || nc >= p.Cout) continue;
        float y = accL[e] * (p.scale ? p.scale[nc] : 1.f) + (p.shift ? p.shift[nc] : 0.f);
        if (p.res) y += p.res[rr * p.ld_res + nc];
        if (p.act == 1) y = fmaxf(y, 0.f);
        else if (p.act == 2) y = 0.5f * y * (1.f + erff(y * 0.70710678118654752440f));
        else if (p.act == 3) y = 1.f / (1.f + expf(-y));
        p.out[rr * p.ld_out + nc] = y;
    }
}

// ---------------------------------------------------------------------------------------------
// Pair-compacted kernel without LDS atomics.  ds_add_f32 turned out to be the bottleneck of the
// compacted kernels above (ablation tools/gg_quick.py: 225 of 275 us for a 32->32 level-1 conv), so
// here every wave accumulates into its OWN LDS copy of the 128 x 32 output tile with plain
// read-add-write (one wave = in-order LDS queue, distinct addresses per instruction) and the four
// copies are summed in a fixed order in the epilogue: deterministic, no atomics, 64 KB of accumulators.
// ---------------------------------------------------------------------------------------------
#define C4_OB 4
template <int C4_ROWS>
__global__ __launch_bounds__(256) void gather_gemm_compact4_kernel(const GGParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* accL = smem;                                                   // [4 waves][C4_ROWS][32]: one private copy per wave
    int* l_idx = (int*)(accL + 4 * C4_ROWS * 32);                         // [4 waves][C4_OB][C4_ROWS]
    unsigned char* l_row = (unsigned char*)(l_idx + 4 * C4_OB * C4_ROWS); // [4 waves][C4_OB][C4_ROWS]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t row_base = (int64_t)blockIdx.x * C4_ROWS;
    const int ncol0 = blockIdx.y * 32;
    const int nchunks = p.Cin >> 5;
    for (int e = tid; e < 4 * C4_ROWS * 32; e += 256) accL[e] = 0.f;
    __syncthreads();
    float* my_acc = accL + wv * C4_ROWS * 32;
    int* my_idx = l_idx + wv * C4_OB * C4_ROWS;
    unsigned char* my_row = l_row + wv * C4_OB * C4_ROWS;
    const uint64_t lt = (1ull << lane) - 1ull;
    int n = ncol0 + j;
    n = n < p.Cout ? n : p.Cout - 1;

    for (int kb = wv; kb < p.K; kb += 4 * C4_OB) {          // this wave's offsets: kb, kb+4, ..., kb+4*(C4_OB-1)
        // ---- lists of the batch
        int ids[C4_OB][C4_ROWS / 64];
#pragma unroll
        for (int b = 0; b < C4_OB; ++b) {
            const int k = kb + 4 * b;
#pragma unroll
            for (int ps = 0; ps < C4_ROWS / 64; ++ps) {
                const int64_t row = row_base + ps * 64 + lane;
                ids[b][ps] = (k < p.K && row < p.M) ? p.nbr[(int64_t)k * p.M + row] : -1;
            }
        }
        int cnt[C4_OB];
#pragma unroll
        for (int b = 0; b < C4_OB; ++b) {
            int c = 0;
#pragma unroll
            for (int ps = 0; ps < C4_ROWS / 64; ++ps) {
                const uint64_t bal = __ballot(ids[b][ps] >= 0);
                if (ids[b][ps] >= 0) {
                    const int pos = c + __popcll(bal & lt);
                    my_idx[b * C4_ROWS + pos] = ids[b][ps];
                    my_row[b * C4_ROWS + pos] = (unsigned char)(ps * 64 + lane);
                }
                c += __popcll(bal);
            }
            cnt[b] = c;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- flat step stream over (b, pc, chunk)
        auto first_group = [&](int& b, int& pc) {           // first (b, pc) with pc*32 < cnt[b], b from `b`
            while (b < C4_OB) {
                int cb = 0;
#pragma unroll
                for (int q = 0; q < C4_OB; ++q) cb = (q == b) ? cnt[q] : cb;
                if (pc * 32 < cb) return true;
                ++b; pc = 0;
            }
            return false;
        };
        auto load = [&](Frag<1>& f, int b, int pc, int chunk) {
            const int pp = pc * 32 + j;
            int cb = 0;
#pragma unroll
            for (int q = 0; q < C4_OB; ++q) cb = (q == b) ? cnt[q] : cb;
            const int pidx = pp < cb ? my_idx[b * C4_ROWS + pp] : -1;
            const int c = chunk * 32 + h * 16;
            if (pidx >= 0) {
                const float* src = (c < p.C0) ? (p.in0 + (int64_t)pidx * p.ld0 + c) : (p.in1 + (int64_t)pidx * p.ld1 + (c - p.C0));
#pragma unroll
                for (int q = 0; q < 4; ++q) f.a[q] = *(const f32x4*)(src + q * 4);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) f.a[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const float* w = p.wt + ((int64_t)(kb + 4 * b) * p.Cout + n) * p.Cin + c;
#pragma unroll
            for (int q = 0; q < 4; ++q) f.b[0][q] = *(const f32x4*)(w + q * 4);
        };
        int b = 0, pc = 0, chunk = 0;
        if (!first_group(b, pc)) continue;
        f32x16 acc[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
        Frag<1> cur;
        load(cur, b, pc, 0);
        while (true) {
            int nb = b, npc = pc, nchunk = chunk + 1;
            bool has_next = true;
            if (nchunk == nchunks) { nchunk = 0; ++npc; has_next = first_group(nb, npc); }
            Frag<1> nxt;
            if (has_next) load(nxt, nb, npc, nchunk);
            mma_frag<1>(acc, cur);
            if (nchunk == 0) {                              // the (b, pc) group is complete: scatter it
                int cb = 0;
#pragma unroll
                for (int q = 0; q < C4_OB; ++q) cb = (q == b) ? cnt[q] : cb;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int prow = pc * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (prow < cb) {                        // wave-private rows: plain read-add-write, no atomics
                        float* dst = &my_acc[(int)my_row[b * C4_ROWS + prow] * 32 + j];
                        *dst += acc[0][r];
                    }
                    acc[0][r] = 0.f;
                }
            }
            if (!has_next) break;
            cur = nxt;
            b = nb; pc = npc; chunk = nchunk;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (int e = tid; e < C4_ROWS * 32; e += 256) {
        const int r = e >> 5, nn = e & 31;
        const int64_t rr = row_base + r;
        const int nc = ncol0 + nn;
        if (rr >= p.M || nc >= p.Cout) continue;
        const float a4 = ((accL[e] + accL[C4_ROWS * 32 + e]) + accL[2 * C4_ROWS * 32 + e]) + accL[3 * C4_ROWS * 32 + e];
        float y = a4 * (p.scale ? p.scale[nc] : 1.f) + (p.shift ? p.shift[nc] : 0.f);
        if (p.res) y += p.res[rr * p.ld_res + nc];
        if (p.act == 1) y = fmaxf(y, 0.f);
        else if (p.act == 2) y = 0.5f * y * (1.f + erff(y * 0.70710678118654752440f));
        else if (p.act == 3) y = 1.f / (1.f + expf(-y));
        p.out[rr * p.ld_out + nc] = y;
    }
}

// nt > 0            : private-fragment kernel, nt subtiles per wave, no split-K
// nt == 0           : heuristic (see below)
// nt == -1          : private-fragment kernel, split-K with one subtile (tests)
// nt in [-14, -11]  : lock-step LDS-shared-weights kernel with (-nt - 10) subtiles (tests / tuning)
// nt in [-23, -21]  : pair-compacted kernel, 4 waves per workgroup, (-nt - 20) subtiles (needs a neighbour table)
// nt in [-33, -31]  : pair-compacted kernel, 8 waves per workgroup, (-nt - 30) subtiles
// nt == -51          : pipelined pair-compacted kernel (32 columns per pass)
// nt == -61          : pipelined pair-compacted kernel with per-wave private LDS accumulators (no atomics)
// nt == -41 / -42    : single-pass pair-compacted kernel over all Cout <= 128 columns, 128 / 256 rows per workgroup
void launch_splitk_epilogue(const GGParams& p, hipStream_t st) {
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)cdiv(p.M * p.Cout, 256)), dim3(256), 0, st, p);
}

int launch_gather_gemm(const GGParams& p_in, int nt, void* ws, size_t ws_bytes, hipStream_t st) {
    GGParams p = p_in;
    p.ksplit = 1;
    p.ws = nullptr;
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("SD3D_GG_DBG"); dbg = e ? atoi(e) : 0; } p.dbg = dbg; }
    if (p.M <= 0 || p.Cout <= 0) return SD3D_OK;
    if (p.Cin <= 0 || (p.Cin & 31)) return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm: Cin must be a positive multiple of 32");
    if (p.in1 && ((p.C0 & 31) || p.C0 > p.Cin)) return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm: concat split must be a multiple of 32");
    if (!p.in1) p.C0 = p.Cin;
    if ((p.ld0 & 3) || (p.in1 && (p.ld1 & 3))) return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm: input row stride must be a multiple of 4 floats");
    if (p.K > 128) return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm: at most 128 kernel offsets");
    if (!p.nbr && p.K != 1) return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm: identity gather needs K == 1");
    const int sub = (p.Cout + 31) / 32;
    const int64_t tiles = cdiv(p.M, 32);
    int ks = 1;
    bool lds = false;
    if (nt == -61 || nt == -62) {
        if (!p.nbr) return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm: the compacted kernel needs a neighbour table");
        p.col_groups = (int)cdiv(p.Cout, 32);
        const int rows = nt == -61 ? 128 : 256;
        const size_t sm = (size_t)4 * rows * 32 * sizeof(float) + (size_t)4 * C4_OB * rows * (sizeof(int) + 1);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)gather_gemm_compact4_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)gather_gemm_compact4_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
        const dim3 grid((unsigned)cdiv(p.M, rows), (unsigned)p.col_groups);
        if (rows == 128) hipLaunchKernelGGL(gather_gemm_compact4_kernel<128>, grid, dim3(256), sm, st, p);
        else hipLaunchKernelGGL(gather_gemm_compact4_kernel<256>, grid, dim3(256), sm, st, p);
        SD3D_CHECK_LAUNCH();
        return SD3D_OK;
    }
    if (nt == -51) {
        if (!p.nbr) return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm: the compacted kernel needs a neighbour table");
        p.col_groups = (int)cdiv(p.Cout, 32);
        const size_t sm = (size_t)C3_ROWS * 32 * sizeof(float) + (size_t)4 * C3_OB * C3_ROWS * (sizeof(int) + 1);
        hipLaunchKernelGGL(gather_gemm_compact3_kernel, dim3((unsigned)cdiv(p.M, C3_ROWS), (unsigned)p.col_groups), dim3(256), sm, st, p);
        SD3D_CHECK_LAUNCH();
        return SD3D_OK;
    }
    if (nt == -41 || nt == -42) {
        if (!p.nbr) return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm: the compacted kernel needs a neighbour table");
        if (p.Cout > 128) return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm: single-pass compacted kernel needs Cout <= 128");
        const int ntc = (p.Cout + 31) / 32;
        p.col_groups = 1;
        if (nt == -41) {
            switch (ntc) {
                case 1: launch_compact2<1, 128>(p, st); break;
                case 2: launch_compact2<2, 128>(p, st); break;
                case 3: launch_compact2<3, 128>(p, st); break;
                default: launch_compact2<4, 128>(p, st); break;
            }
        } else {
            switch (ntc) {
                case 1: launch_compact2<1, 256>(p, st); break;
                case 2: launch_compact2<2, 256>(p, st); break;
                case 3: launch_compact2<3, 256>(p, st); break;
                default: launch_compact2<4, 256>(p, st); break;
            }
        }
        SD3D_CHECK_LAUNCH();
        return SD3D_OK;
    }
    if ((nt <= -21 && nt >= -23) || (nt <= -31 && nt >= -33)) {
        const int nw = nt <= -31 ? 8 : 4;
        nt = nt <= -31 ? -nt - 30 : -nt - 20;
        if (!p.nbr) return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm: the compacted kernel needs a neighbour table");
        p.col_groups = (int)cdiv(p.Cout, 32 * nt);
        const dim3 grid((unsigned)cdiv(p.M, CB_ROWS), (unsigned)p.col_groups);
        const size_t sm = (size_t)CB_ROWS * 32 * nt * sizeof(float) + (size_t)nw * CB_ROWS * (sizeof(int) + sizeof(unsigned short));
#define GC_LAUNCH(NT_, NW_)                                                                                              \
    do {                                                                                                                 \
        static bool attr_set = false;                                                                                    \
        if (!attr_set && sm > 65536) {                                                                                   \
            (void)hipFuncSetAttribute((const void*)gather_gemm_compact_kernel<NT_, NW_>,                                 \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);                              \
            attr_set = true;                                                                                             \
        }                                                                                                                \
        hipLaunchKernelGGL((gather_gemm_compact_kernel<NT_, NW_>), grid, dim3(64 * NW_), sm, st, p);                     \
    } while (0)
        if (nw == 4) {
            switch (nt) {
                case 1: GC_LAUNCH(1, 4); break;
                case 2: GC_LAUNCH(2, 4); break;
                case 3: GC_LAUNCH(3, 4); break;
                default: return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm: compact nt must be 1..3");
            }
        } else {
            switch (nt) {
                case 1: GC_LAUNCH(1, 8); break;
                case 2: GC_LAUNCH(2, 8); break;
                case 3: GC_LAUNCH(3, 8); break;
                default: return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm: compact nt must be 1..3");
            }
        }
#undef GC_LAUNCH
        SD3D_CHECK_LAUNCH();
        return SD3D_OK;
    } else if (nt <= -11 && nt >= -14) {
        lds = true;
        nt = -nt - 10;
    } else if (nt <= 0) {
        const bool force_split = nt < 0;
        const int64_t steps = (int64_t)p.K * (p.Cin / 32);
        if (!force_split && tiles >= 64 && steps >= 2) {
            // enough rows for 4-tile workgroups: share the weights through LDS
            lds = true;
            nt = sub >= 4 ? 4 : sub;
            while (nt > 2 && cdiv(tiles, 4) * cdiv(sub, nt) < 256) --nt;
            if (nt == 2 && cdiv(tiles, 4) * cdiv(sub, 2) < 192) nt = 1;
            if (sub % nt) { for (int c = nt; c >= 1; --c) if (sub % c == 0) { nt = c; break; } }
        } else {
            nt = sub >= 4 ? 4 : sub;
            while (nt > 1 && tiles * cdiv(sub, nt) < 2048) --nt;
            if (sub % nt) { for (int c = nt; c >= 1; --c) if (sub % c == 0) { nt = c; break; } }
            if (force_split || (tiles * cdiv(sub, nt) < 1024 && steps >= 8)) ks = 4;
            if (force_split) nt = 1;
            if (ks == 4 && nt > 2) nt = (sub % 2 == 0) ? 2 : 1;
        }
    }
    p.col_groups = (int)cdiv(p.Cout, 32 * nt);
    const dim3 block(256);
    if (lds) {
        // few workgroups + many offsets (coarse U-Net levels): slice the offsets over gridDim.z
        const int64_t wgs = cdiv(tiles, 4) * p.col_groups;
        if (p.nbr && p.K >= 8 && wgs < 384) {
            int ksp = wgs < 96 ? 8 : (wgs < 256 ? 4 : 2);
            if ((size_t)ksp * p.M * p.Cout * sizeof(float) <= ws_bytes && ws) { p.ksplit = ksp; p.ws = (float*)ws; }
        }
        const dim3 grid((unsigned)cdiv(tiles, 4), (unsigned)p.col_groups, (unsigned)p.ksplit);
        switch (nt) {
            case 1: hipLaunchKernelGGL(gather_gemm_lds_kernel<1>, grid, block, 0, st, p); break;
            case 2: hipLaunchKernelGGL(gather_gemm_lds_kernel<2>, grid, block, 0, st, p); break;
            case 3: hipLaunchKernelGGL(gather_gemm_lds_kernel<3>, grid, block, 0, st, p); break;
            case 4: hipLaunchKernelGGL(gather_gemm_lds_kernel<4>, grid, block, 0, st, p); break;
            default: return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm: nt must be 1..4");
        }
        if (p.ksplit > 1)
            hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)cdiv(p.M * p.Cout, 256)), dim3(256), 0, st, p);
        SD3D_CHECK_LAUNCH();
        return SD3D_OK;
    }
    const int64_t units = tiles * p.col_groups;
    const dim3 grid((unsigned)(ks == 4 ? units : cdiv(units, 4)));
#define GG_LAUNCH(NT_, KS_) hipLaunchKernelGGL((gather_gemm_kernel<NT_, KS_>), grid, block, 0, st, p)
    if (ks == 1) {
        switch (nt) {
            case 1: GG_LAUNCH(1, 1); break;
            case 2: GG_LAUNCH(2, 1); break;
            case 3: GG_LAUNCH(3, 1); break;
            case 4: GG_LAUNCH(4, 1); break;
            default: return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm: nt must be 1..4");
        }
    } else {
        switch (nt) {
            case 1: GG_LAUNCH(1, 4); break;
            case 2: GG_LAUNCH(2, 4); break;
            default: return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm: split-K supports nt 1..2");
        }
    }
#undef GG_LAUNCH
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
