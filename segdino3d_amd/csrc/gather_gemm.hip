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
__device__ __forceinline__ void gather_gemm_body(const GGParams& p, const int64_t block, float* red) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int wv = threadIdx.x >> 6;
    const int ks = (KS > 1) ? wv : 0;
    const int64_t unit = (KS > 1) ? block : block * 4 + wv;
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

template <int NT, int KS>
__global__ __launch_bounds__(256) void gather_gemm_kernel(const GGParams p) {
    __shared__ float red[(KS > 1) ? (KS - 1) * NT * 16 * 64 : 1];
    gather_gemm_body<NT, KS>(p, (int64_t)blockIdx.x, red);
}

// Several INDEPENDENT small Linears in one launch (blockIdx.y = job): the decoder on a few hundred queries is a chain of
// ~140 Linears of 26-100 MFLOP each, every one a separate ~10 us launch; the ones that do not depend on each other (the two
// box MLPs, the projections that share an input, the class head of layer i next to the first projections of layer i + 1)
// run side by side here.  Same code path as the single launch (split-K over the four waves of a workgroup).
#define GG_GROUP_MAX 8
struct GGGroup { int n; GGParams job[GG_GROUP_MAX]; };
__global__ __launch_bounds__(256) void gather_gemm_group_kernel(const GGGroup g) {
    __shared__ float red[3 * 16 * 64];
    const GGParams& p = g.job[blockIdx.y];
    const int64_t units = ((p.M + 31) / 32) * p.col_groups;
    if ((int64_t)blockIdx.x >= units) return;               // uniform per workgroup
    gather_gemm_body<1, 4>(p, (int64_t)blockIdx.x, red);
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
__device__ __forceinline__ void gather_gemm_lds_body(const GGParams& p, const int bx, const int cg, const int bz, float (*Bs)[NT * 32 * BS_LD],
                                                     unsigned long long (*wmask)[2]) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t row_tile = (int64_t)bx * 4 + wv;
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
        for (int k = bz; k < p.K; k += p.ksplit) { if (k < 64) s0 |= 1ull << k; else s1 |= 1ull << (k - 64); }
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
        float* wsz = p.ws + (int64_t)bz * p.M * p.Cout;
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

template <int NT>
__global__ __launch_bounds__(256) void gather_gemm_lds_kernel(const GGParams p) {
    __shared__ __attribute__((aligned(16))) float Bs[2][NT * 32 * BS_LD];
    __shared__ unsigned long long wmask[4][2];
    gather_gemm_lds_body<NT>(p, blockIdx.x, blockIdx.y, blockIdx.z, Bs, wmask);
}
// Several INDEPENDENT plain Linears on a few thousand rows each in one launch (blockIdx.z = job): the decoder with one query per
// superpoint runs ~100 Linears of 3000 x 256 -> 256 per scene, each 192 workgroups for 256 CUs and ~13 us of dependent steps;
// the ones that do not depend on each other share a launch.  Same body, same tiling per job as the single launch: same bits.
template <int NT>
__global__ __launch_bounds__(256) void gather_gemm_lds_group_kernel(const GGGroup g) {
    __shared__ __attribute__((aligned(16))) float Bs[2][NT * 32 * BS_LD];
    __shared__ unsigned long long wmask[4][2];
    const GGParams& p = g.job[blockIdx.z];
    if ((int64_t)blockIdx.x * 128 >= p.M || (int)blockIdx.y >= p.col_groups) return;      // uniform per workgroup
    gather_gemm_lds_body<NT>(p, blockIdx.x, blockIdx.y, 0, Bs, wmask);
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

// (The pair-compacted variants of this output-stationary kernel - LDS accumulators fed with ds_add_f32, single-pass,
// pipelined, per-wave private copies - lived here through round 1.  The pair-major convolution of pair_gemm.hip
// replaced all of them: they were bound by LDS float atomics or LDS capacity.  DESIGN.md section 6 keeps the numbers.)

// nt > 0            : private-fragment kernel, nt subtiles per wave, no split-K
// nt == 0           : heuristic (see below)
// nt == -1          : private-fragment kernel, split-K with one subtile (tests)
// nt in [-14, -11]  : lock-step LDS-shared-weights kernel with (-nt - 10) subtiles (tests / tuning)
void launch_splitk_epilogue(const GGParams& p, hipStream_t st) {
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)cdiv(p.M * p.Cout, 256)), dim3(256), 0, st, p);
}

// The tiling code that makes launch_gather_gemm pick, for ANY number of rows, the kernel its heuristic picks for a plain Linear
// (identity rows, K = 1) on `rows` rows: 0 = lock-step kernel (>= 64 row tiles; its summation order does not depend on the
// row count, so the heuristic stays free), -1 = one column tile per wave with the contraction split over the four waves,
// n > 0 = n column tiles per wave without split.  The batched decoder passes it so that the rows of several scenes run on the
// same kernel - same summation order, same bits - as one scene's rows.  Mirrors the heuristic below; keep them together.
int dense_plan_code(int64_t rows, int Cin, int Cout) {
    const int sub = (Cout + 31) / 32;
    const int64_t tiles = cdiv(rows, 32);
    const int64_t steps = Cin / 32;
    if (tiles >= 64 && steps >= 2) return 0;
    int nt = sub >= 4 ? 4 : sub;
    while (nt > 1 && tiles * cdiv(sub, nt) < 2048) --nt;
    if (sub % nt) { for (int c = nt; c >= 1; --c) if (sub % c == 0) { nt = c; break; } }
    if (tiles * cdiv(sub, nt) < 1024 && steps >= 8) return -1;        // (split-K then runs with one tile per wave when nt > 2; nt <= 2 here)
    return nt;
}

// column tiles per workgroup of the lock-step kernel
static int lds_column_tiles(int sub, int64_t tiles, bool gathered) {
    int nt = sub >= 4 ? 4 : sub;
    while (nt > 2 && cdiv(tiles, 4) * cdiv(sub, nt) < 256) --nt;
    if (nt == 2 && cdiv(tiles, 4) * cdiv(sub, 2) < 192) nt = 1;
    if (sub % nt) { for (int c = nt; c >= 1; --c) if (sub % c == 0) { nt = c; break; } }
    // plain Linears on a few thousand rows (the decoder with one query per superpoint, the superpoint-side projections):
    // measured on M = 3000, 256 -> 1024 / 3072 columns: 1-2 column tiles per workgroup beat 3-4 (24 vs 30 us, 61 vs 88 us)
    if (!gathered && tiles < 1024) nt = (sub % 2 == 0 && cdiv(tiles, 4) * (sub / 2) >= 512) ? 2 : 1;
    return nt;
}

int launch_pair_dense(const GGParams&, hipStream_t);             // pair_gemm.hip
#define GG_PAIR_DENSE_MIN_ROWS 16384

int launch_gather_gemm(const GGParams& p_in, int nt, void* ws, size_t ws_bytes, hipStream_t st) {
    GGParams p = p_in;
    p.ksplit = 1;
    p.ws = nullptr;
   
    if (p.M <= 0 || p.Cout <= 0) return SD3D_OK;
    if (p.Cin <= 0 || (p.Cin & 31)) return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm: Cin must be a positive multiple of 32");
    if (p.in1 && ((p.C0 & 31) || p.C0 > p.Cin)) return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm: concat split must be a multiple of 32");
    if (!p.in1) p.C0 = p.Cin;
    if ((p.ld0 & 3) || (p.in1 && (p.ld1 & 3))) return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm: input row stride must be a multiple of 4 floats");
    if (p.K > 128) return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm: at most 128 kernel offsets");
    if (!p.nbr && p.K != 1) return sd3d_set_error(SD3D_ERR_ARG, "gather_gemm: identity gather needs K == 1");
    // dense products on tens of thousands of rows (the U-Net's 1x1 convolutions; no decoder Linear reaches this many rows per scene, and
    // the batched decoder passes its tiling code explicitly): the persistent pass-1 kernel with the identity rulebook
    static const int pd_env = [] { const char* e = getenv("SD3D_PAIR_DENSE"); return e ? atoi(e) : 1; }();
    if (pd_env && nt == 0 && !p.nbr && p.K == 1 && p.M >= GG_PAIR_DENSE_MIN_ROWS && !(p.Cout & 3) && !(p.ld_out & 3) && (!p.res || !(p.ld_res & 3)))
        return launch_pair_dense(p, st);
    const int sub = (p.Cout + 31) / 32;
    const int64_t tiles = cdiv(p.M, 32);
    int ks = 1;
    bool lds = false;
    if (nt <= -11 && nt >= -14) {
        lds = true;
        nt = -nt - 10;
    } else if (nt <= 0) {
        const bool force_split = nt < 0;
        const int64_t steps = (int64_t)p.K * (p.Cin / 32);
        if (!force_split && tiles >= 64 && steps >= 2) {
            // enough rows for 4-tile workgroups: share the weights through LDS
            lds = true;
            nt = lds_column_tiles(sub, tiles, p.nbr != nullptr);
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

// n <= 8 independent plain Linears (identity rows, K = 1): out_i = act_i(x_i W_i^T + shift_i + res_i)
int launch_linear_group(int n, const GGParams* jobs, hipStream_t st) {
    if (n <= 0) return SD3D_OK;
    if (n > GG_GROUP_MAX) return sd3d_set_error(SD3D_ERR_ARG, "linear_group: at most 8 jobs per launch");
    // every job on enough rows for the lock-step kernel (dense_plan_code 0): one launch per column-tile count, each job on the
    // tiling its own launch would get
    bool all_lds = true;
    for (int i = 0; i < n; ++i) all_lds = all_lds && jobs[i].Cin > 0 && !(jobs[i].Cin & 31) && dense_plan_code(jobs[i].M, jobs[i].Cin, jobs[i].Cout) == 0;
    if (all_lds) {
        for (int want = 1; want <= 4; ++want) {
            GGGroup g;
            g.n = 0;
            unsigned gx = 0, gy = 0;
            for (int i = 0; i < n; ++i) {
                GGParams p = jobs[i];
                if (p.M <= 0 || p.Cout <= 0) return sd3d_set_error(SD3D_ERR_ARG, "linear_group: empty job");
                if (p.in1 && ((p.C0 & 31) || p.C0 > p.Cin)) return sd3d_set_error(SD3D_ERR_ARG, "linear_group: concat split must be a multiple of 32");
                if (!p.in1) p.C0 = p.Cin;
                if ((p.ld0 & 3) || (p.in1 && (p.ld1 & 3))) return sd3d_set_error(SD3D_ERR_ARG, "linear_group: input row stride must be a multiple of 4 floats");
                const int sub = (p.Cout + 31) / 32;
                const int64_t tiles = cdiv(p.M, 32);
                if (lds_column_tiles(sub, tiles, false) != want) continue;
                p.nbr = nullptr; p.K = 1; p.ksplit = 1; p.ws = nullptr;
                p.col_groups = (int)cdiv(p.Cout, 32 * want);
                gx = (unsigned)cdiv(tiles, 4) > gx ? (unsigned)cdiv(tiles, 4) : gx;
                gy = (unsigned)p.col_groups > gy ? (unsigned)p.col_groups : gy;
                g.job[g.n++] = p;
            }
            if (!g.n) continue;
            const dim3 grid(gx, gy, (unsigned)g.n);
            switch (want) {
                case 1: hipLaunchKernelGGL(gather_gemm_lds_group_kernel<1>, grid, dim3(256), 0, st, g); break;
                case 2: hipLaunchKernelGGL(gather_gemm_lds_group_kernel<2>, grid, dim3(256), 0, st, g); break;
                case 3: hipLaunchKernelGGL(gather_gemm_lds_group_kernel<3>, grid, dim3(256), 0, st, g); break;
                default: hipLaunchKernelGGL(gather_gemm_lds_group_kernel<4>, grid, dim3(256), 0, st, g); break;
            }
        }
        SD3D_CHECK_LAUNCH();
        return SD3D_OK;
    }
    GGGroup g;
    g.n = n;
    int64_t max_units = 0;
    for (int i = 0; i < n; ++i) {
        GGParams p = jobs[i];
        if (p.M <= 0 || p.Cout <= 0 || p.Cin <= 0 || (p.Cin & 31)) return sd3d_set_error(SD3D_ERR_ARG, "linear_group: Cin must be a positive multiple of 32");
        if (p.in1 && ((p.C0 & 31) || p.C0 > p.Cin)) return sd3d_set_error(SD3D_ERR_ARG, "linear_group: concat split must be a multiple of 32");
        if (!p.in1) p.C0 = p.Cin;
        if ((p.ld0 & 3) || (p.in1 && (p.ld1 & 3))) return sd3d_set_error(SD3D_ERR_ARG, "linear_group: input row stride must be a multiple of 4 floats");
        p.nbr = nullptr; p.K = 1; p.ksplit = 1; p.ws = nullptr;
        p.col_groups = (int)cdiv(p.Cout, 32);
        const int64_t units = cdiv(p.M, 32) * p.col_groups;
        max_units = units > max_units ? units : max_units;
        g.job[i] = p;
    }
    hipLaunchKernelGGL(gather_gemm_group_kernel, dim3((unsigned)max_units, (unsigned)n), dim3(256), 0, st, g);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
