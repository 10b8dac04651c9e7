// Output-stationary sparse convolution ("slab" kernel), exact fp32 on the matrix cores.
//
// pair_gemm.hip evaluates a sparse convolution pair-major: every (in, out, offset) pair is an MFMA row, the products
// leave the chip as part[P][Cout] and a second kernel adds them up per output row - 2 x 4 x P x Cout bytes of extra
// HBM traffic (2/3 of everything the convolutions move) and 20-25 % of their time.  Here a workgroup OWNS a slab of R
// consecutive output rows (voxels are Morton-sorted, so a slab is a compact patch) and keeps their fp32 sums in LDS for
// the whole convolution:
//
//   prologue   the slab's rulebook, straight from the neighbour table nbr[K][M] (coalesced row segments): per offset k
//              the rows with a neighbour are compacted (ballot / popcount) into (in row, local out row) lists, padded to
//              32 (padding gathers row 0 and adds into a sink row), and cut into UNITS of <= 32 pairs.  No pair lists, no `pos` table, no per-table list kernels.
//   main loop  one step = one unit x one chunk of <= 128 input channels: the gathered rows are staged ONCE in LDS
//              (double buffer, requested one step ahead, indices two steps ahead) and every wave multiplies them with ITS
//              16 output columns of W[k], held in registers for all units of the offset (v_mfma_f32_16x16x4_f32: units of
//              <= 16 pairs cost half).  A wave adds its 32 x 16 product tile into the slab with plain LDS read-add-write:
//              waves own disjoint columns and inside one offset every output row occurs once, so there are no atomics
//              and the order of the adds is fixed - k ascending, exactly the order pair_reduce_kernel used.
//   epilogue   out = act(scale * slab + shift + res), one coalesced HBM write per output row.
//
// HBM-side traffic is the algorithmic 4 P Cin (gathers, mostly L2 / MALL hits) + 4 M Cout.  Levels with few rows
// (stride 8 / 16: 14 k / 3 k voxels) get their parallelism from splitting the OFFSETS over gridDim.y workgroups that write
// partial slabs (ksplit x M x Cout floats, ~1/8 of what the pair-major partial products were), summed in k order by
// slab_reduce_kernel together with the epilogue.
#include "../gg_common.h"
#include <stdlib.h>
#include <type_traits>

typedef int int2v __attribute__((ext_vector_type(2)));
typedef int int4v __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct SlabParams {
    const float* in0; int ld0; int C0;        // input features, first C0 channels
    const float* in1; int ld1;                // optional second source (skip concatenation), channels C0..Cin-1
    const int32_t* nbr; int K; int64_t M;     // neighbour table [K][M], -1 = no neighbour
    const float* wt; int Cin, Cout;           // [K][Cout][Cin]
    const float* scale; const float* shift;   // per output column, optional
    const float* res; int ld_res;
    float* out; int ld_out; int act;
    int R;                                    // output rows per slab
    int lcap;                                 // list capacity per offset (entries)
    int ucap;                                 // unit-table capacity
    int ksplit;                               // gridDim.y; > 1: raw partial slabs go to `part`
    int2v* lists;                             // scratch [gridDim.x * gridDim.y][koffs][lcap]
    float* part;                              // [ksplit][M][Cout]
    int col0;                                 // first output column of this launch (column groups)
    int ncols;                                // columns of this launch (16 * waves)
};

__device__ __forceinline__ float slab_act(float t, int act) {
    if (act == 1) return fmaxf(t, 0.f);
    if (act == 2) return 0.5f * t * (1.f + erff(t * 0.70710678118654752440f));
    if (act == 3) return 1.f / (1.f + expf(-t));
    return t;
}

// CK = channels per step, NCH = Cin / CK, NCB = waves = 16-column blocks of this workgroup
template <int CK, int NCH, int NCB, bool WPF>
__device__ __forceinline__ void slab_conv_body(const SlabParams& p, float* smem) {
    constexpr int NT = 64 * NCB;
    constexpr int CKP = CK + 4;                                // stage row stride (floats)
    constexpr int C4 = CK / 4;                                 // float4 items per staged row
    constexpr int ITEMS = 32 * C4;
    constexpr int NI = (ITEMS + NT - 1) / NT;
    constexpr int NQ = CK / 16;                                // 16-channel groups per step
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int R = p.R, SLD = p.ncols;
    const int64_t row0 = (int64_t)blockIdx.x * R;
    const int nrows = (int)((p.M - row0) < R ? (p.M - row0) : R);
    const int kb = (int)((int64_t)blockIdx.y * p.K / gridDim.y), ke = (int)((int64_t)(blockIdx.y + 1) * p.K / gridDim.y);
    const int KR = ke - kb;
    const int col0 = blockIdx.z * p.ncols;                     // column group of this workgroup

    // ---- LDS carve-up
    float* slab = smem;                                        // [(R + 1)][SLD], row R = sink of the padding entries
    float* stage = slab + (size_t)(R + 1) * SLD;               // [2][32][CKP]
    int* ol = (int*)(stage + 2 * 32 * CKP);                    // [4][32] local output rows of the staged units (ring)
    int* cnt = ol + 128;                                        // [KR] pairs per offset
    int* meta = cnt + 128;                                     // [0] = number of units
    int4v* otbl = (int4v*)(meta + 4);                          // [KR] non-empty offsets: (k, first unit, end unit, 0)
    int2v* tbl = (int2v*)(otbl + 128);                         // [ucap] (list position, k << 8 | rows)
    int2v* lst = p.lists + (((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * ((int64_t)p.K * p.lcap + 32);

    for (int i = tid * 4; i < (R + 1) * SLD; i += NT * 4) *(f32x4_t*)(slab + i) = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: compact the slab's rows per offset
    for (int kk = wv; kk < KR; kk += NCB) {
        const int32_t* src = p.nbr + (int64_t)(kb + kk) * p.M + row0;
        int2v* dst = lst + (int64_t)kk * p.lcap;
        int base = 0;
        for (int it0 = 0; it0 < R; it0 += 256) {
            int v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = it0 + u * 64 + lane;
                v[u] = r < nrows ? src[r] : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint64_t m = __ballot(v[u] >= 0);
                if (v[u] >= 0) dst[base + __popcll(m & ((1ull << lane) - 1ull))] = int2v{v[u], it0 + u * 64 + lane};
                base += __popcll(m);
            }
        }
        const int padded = (base + 31) & ~31;                   // a unit always stages 32 rows
        if (lane < padded - base) dst[base + lane] = int2v{0, R};
        if (lane == 0) cnt[kk] = base;
    }
    if (wv == 0 && lane < 32) lst[(int64_t)KR * p.lcap + lane] = int2v{0, R};
    __syncthreads();
    // ---- unit table (wave 0): units of <= 32 pairs in (offset, position) order; table of the non-empty offsets
    if (wv == 0) {
        int running = 0, orun = 0;
        for (int k0 = 0; k0 < KR; k0 += 64) {
            const int kk = k0 + lane;
            const int c = kk < KR ? cnt[kk] : 0;
            const int nu = (c + 31) >> 5;
            int inc = nu;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int t = __shfl_up(inc, d);
                if (lane >= d) inc += t;
            }
            const int pos = running + inc - nu;
            for (int u = 0; u < nu; ++u) {
                const int rows = c - u * 32 < 32 ? c - u * 32 : 32;
                if (pos + u < p.ucap) tbl[pos + u] = int2v{kk * p.lcap + u * 32, (kk << 8) | rows};
            }
            const uint64_t ne = __ballot(nu > 0);
            if (nu > 0) otbl[orun + __popcll(ne & ((1ull << lane) - 1ull))] = int4v{kk, pos, pos + nu, 0};
            orun += __popcll(ne);
            running += __shfl(inc, 63);
        }
        if (lane == 0) { meta[0] = running < p.ucap ? running : p.ucap; meta[1] = orun; }
    }
    __syncthreads();
    const int TU = __builtin_amdgcn_readfirstlane(meta[0]), NO = __builtin_amdgcn_readfirstlane(meta[1]);

    // ---- gather roles: item s of this thread = float4 c4 of staged row `irow`
    int irow[NI], ic4[NI];
#pragma unroll
    for (int s = 0; s < NI; ++s) {
        const int it = tid + s * NT;
        irow[s] = it / C4;
        ic4[s] = it - irow[s] * C4;
    }
    constexpr bool ALL_ITEMS = (NI * NT == ITEMS);
    // Per-step request stream, TWO steps ahead of the multiply: in step t the rows of step t + 2 are requested (their list
    // entries arrived a step earlier) and the entries of step t + 3 are fetched; the rows are written to LDS late in step
    // t + 1 and multiplied in step t + 2 (two register stages, roles fixed by the step's parity).  Everything here is
    // UNCONDITIONAL, loop-carried values are only overwritten by loads: a conditional update or a select on a loaded value
    // makes hipcc wait for the load on the spot (and, vmcnt being in-order, for every request issued before it).  Past the
    // last unit the entries come from a block of (row 0, sink row) pairs behind the lists.
    auto load_idx = [&](int2v (&I)[NI], int step) {
        const int u = step / NCH;
        const int pos = u < TU ? tbl[u][0] : KR * p.lcap;      // uniform
#pragma unroll
        for (int s = 0; s < NI; ++s) I[s] = lst[pos + (irow[s] < 32 ? irow[s] : 31)];
    };
    int2v In[NI];
    auto issue = [&](f32x4_t (&G)[NI], int (&Y)[NI], int& Yu, const int2v (&I)[NI], int step) {
        const int gc = step % NCH;
#pragma unroll
        for (int s = 0; s < NI; ++s) {
            const int ch = gc * CK + ic4[s] * 4;
            const bool first = ch < p.C0;
            const float* base = first ? p.in0 : p.in1;
            const int ld = first ? p.ld0 : p.ld1;
            G[s] = *(const f32x4_t*)(base + (int64_t)I[s][0] * ld + (first ? ch : ch - p.C0));
            Y[s] = I[s][1];
        }
        Yu = step / NCH;
    };
    auto stage_store = [&](const f32x4_t (&G)[NI], const int (&Y)[NI], int Yu, int buf) {
#pragma unroll
        for (int s = 0; s < NI; ++s) {
            if (ALL_ITEMS || tid + s * NT < ITEMS) {
                *(f32x4_t*)(stage + (buf * 32 + irow[s]) * CKP + ic4[s] * 4) = G[s];
                if (ic4[s] == 0) ol[(Yu & 3) * 32 + irow[s]] = Y[s];
            }
        }
    };
    auto load_w = [&](f32x4_t (&Wd)[NCH * NQ], int kk) {      // this wave's 16 columns of W[kb + kk]: lane (n, g) holds W[n][16 q + 4 g ..]
        const float* wsrc = p.wt + ((int64_t)(kb + kk) * p.Cout + col0 + wv * 16 + n) * p.Cin + 4 * g;
#pragma unroll
        for (int q = 0; q < NCH * NQ; ++q) Wd[q] = *(const f32x4_t*)(wsrc + 16 * q);
    };

    f32x4_t W[NCH * NQ], Wn[WPF ? NCH * NQ : 1];
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    f32x4_t P0 = acc0, P1 = acc0;                              // finished product tile waiting to be added into the slab
    int4v po0 = {0, 0, 0, 0}, po1 = {0, 0, 0, 0};
    int pend = 0;                                              // 0 none, 1 = 16 rows, 2 = 32 rows
    f32x4_t G0[NI], G1[NI];
    int Y0[NI], Y1[NI], Yu0 = 0, Yu1 = 0;
    const int T = TU * NCH;
    if (T > 0) {
        int2v I0[NI], I1[NI];
        load_idx(I0, 0);
        load_idx(I1, 1);
        load_idx(In, 2);
        if constexpr (WPF) load_w(Wn, otbl[0][0]);
        issue(G0, Y0, Yu0, I0, 0);
        issue(G1, Y1, Yu1, I1, 1);                             // past the end: row 0, never multiplied
        stage_store(G0, Y0, Yu0, 0);
    }
    __syncthreads();
    auto rmw = [&]() {                                         // slab[out row][this wave's columns] += pending tile
        float* col = slab + wv * 16 + n;
        if (pend) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = col[po0[r] * SLD];
#pragma unroll
            for (int r = 0; r < 4; ++r) col[po0[r] * SLD] = v[r] + P0[r];
        }
        if (pend == 2) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = col[po1[r] * SLD];
#pragma unroll
            for (int r = 0; r < 4; ++r) col[po1[r] * SLD] = v[r] + P1[r];
        }
        pend = 0;
    };
    // one step; PAR = parity of the step (LDS buffer / register stage roles), C = chunk
    auto step = [&](auto par_c, auto c_c, int u) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr int C = decltype(c_c)::value;
        const int t = u * NCH + C;
        const int rows = __builtin_amdgcn_readfirstlane(tbl[u][1] & 255);
        {                                                      // request the rows of step t + 2, fetch the entries of step t + 3
            int2v I[NI];
#pragma unroll
            for (int s = 0; s < NI; ++s) I[s] = In[s];
            if (PAR == 0) issue(G0, Y0, Yu0, I, t + 2); else issue(G1, Y1, Yu1, I, t + 2);
            load_idx(In, t + 3);
        }
        rmw();                                                 // previous unit's tile (overlaps the multiplies below)
        const float* a_base = stage + (PAR * 32 + n) * CKP + 4 * g;
        if (rows > 16) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const f32x4_t a0 = *(const f32x4_t*)(a_base + 16 * q);
                const f32x4_t a1 = *(const f32x4_t*)(a_base + 16 * CKP + 16 * q);
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e4], W[C * NQ + q][e4], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e4], W[C * NQ + q][e4], acc1, 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const f32x4_t a0 = *(const f32x4_t*)(a_base + 16 * q);
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4)
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e4], W[C * NQ + q][e4], acc0, 0, 0, 0);
            }
        }
        // the rows of step t + 1 (requested a step ago) go to the other LDS buffer, last read in step t - 1
        if (t + 1 < T) {
            if (PAR == 0) stage_store(G1, Y1, Yu1, 1); else stage_store(G0, Y0, Yu0, 0);
        }
        if (C == NCH - 1) {                                    // unit complete: its tile is added during the next step
            const int* o = ol + (u & 3) * 32 + 4 * g;
            po0 = *(const int4v*)o;
            po1 = *(const int4v*)(o + 16);
            P0 = acc0; P1 = acc1;
            pend = rows > 16 ? 2 : 1;
            acc0 = f32x4_t{0.f, 0.f, 0.f, 0.f};
            acc1 = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
    };
    // all chunks of one unit; UPAR = u & 1 fixes the parity of its steps: (u * NCH + c) & 1 = (UPAR * NCH + c) & 1
    auto unit = [&](auto upar_c, int u) {
        constexpr int UP = decltype(upar_c)::value;
        step(std::integral_constant<int, (UP * NCH + 0) & 1>{}, std::integral_constant<int, 0>{}, u);
        if constexpr (NCH > 1) step(std::integral_constant<int, (UP * NCH + 1) & 1>{}, std::integral_constant<int, 1>{}, u);
        if constexpr (NCH > 2) step(std::integral_constant<int, (UP * NCH + 2) & 1>{}, std::integral_constant<int, 2>{}, u);
    };
    // Outer loop: the non-empty offsets.  This wave's columns of the NEXT offset are requested here, unconditionally, into
    // the register set the previous iteration's copy just freed (a load under a condition inside the unit loop would be
    // followed by a wait and a copy at the branch join).
    for (int o = 0; o < NO; ++o) {
        const int4v oe = otbl[o];
        const int kk = __builtin_amdgcn_readfirstlane(oe[0]);
        const int u0 = __builtin_amdgcn_readfirstlane(oe[1]), u1 = __builtin_amdgcn_readfirstlane(oe[2]);
        if constexpr (WPF) {
#pragma unroll
            for (int q = 0; q < NCH * NQ; ++q) W[q] = Wn[q];
            load_w(Wn, otbl[o + 1 < NO ? o + 1 : o][0]);
        } else {
            load_w(W, kk);
        }
        int u = u0;
        if (NCH != 2 && (u & 1)) { unit(std::integral_constant<int, 1>{}, u); ++u; }
        if constexpr (NCH == 2) {
            for (; u < u1; ++u) unit(std::integral_constant<int, 0>{}, u);
        } else {
            for (; u + 1 < u1; u += 2) {
                unit(std::integral_constant<int, 0>{}, u);
                unit(std::integral_constant<int, 1>{}, u + 1);
            }
            if (u < u1) unit(std::integral_constant<int, 0>{}, u);
        }
    }
    rmw();
    __syncthreads();

    // ---- epilogue: one coalesced write per output row
    const int c4n = SLD >> 2;
    if (p.ksplit > 1) {
        float* dst = p.part + ((int64_t)blockIdx.y * p.M + row0) * p.Cout + col0;
        for (int i = tid; i < nrows * c4n; i += NT) {
            const int r = i / c4n, q = (i - r * c4n) * 4;
            *(f32x4_t*)(dst + (int64_t)r * p.Cout + q) = *(const f32x4_t*)(slab + r * SLD + q);
        }
        return;
    }
    for (int i = tid; i < nrows * c4n; i += NT) {
        const int r = i / c4n, q = (i - r * c4n) * 4;
        const f32x4_t a = *(const f32x4_t*)(slab + r * SLD + q);
        float y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = col0 + q + j;
            float t = a[j] * (p.scale ? p.scale[col] : 1.f) + (p.shift ? p.shift[col] : 0.f);
            if (p.res) t += p.res[(row0 + r) * p.ld_res + col];
            y[j] = slab_act(t, p.act);
        }
        *(f32x4_t*)(p.out + (row0 + r) * p.ld_out + col0 + q) = f32x4_t{y[0], y[1], y[2], y[3]};
    }
}

#define SLAB_ENTRY(CK, NCH, NCB, WPF)                                                                         \
    __global__ __launch_bounds__(64 * NCB) void slab_conv_kernel_##CK##_##NCH##_##NCB(const SlabParams p) {  \
        extern __shared__ __attribute__((aligned(16))) float slab_smem[];                                     \
        slab_conv_body<CK, NCH, NCB, WPF>(p, slab_smem);                                                      \
    }
// (channels per step, steps per unit, waves, weights of the next offset prefetched into a second register set): every
// (Cin, columns per workgroup) the two U-Nets use
SLAB_ENTRY(32, 1, 2, true)
SLAB_ENTRY(32, 1, 4, true)
SLAB_ENTRY(64, 1, 4, true)
SLAB_ENTRY(64, 1, 8, true)
SLAB_ENTRY(96, 1, 6, true)
SLAB_ENTRY(128, 1, 4, true)
SLAB_ENTRY(128, 1, 6, true)
SLAB_ENTRY(128, 1, 8, true)
SLAB_ENTRY(96, 2, 8, true)
SLAB_ENTRY(128, 2, 8, false)
SLAB_ENTRY(128, 3, 8, false)
SLAB_ENTRY(128, 1, 16, false)
SLAB_ENTRY(128, 2, 16, false)

// ---- barrier-free variant -------------------------------------------------------------------------------------------------
// The staged kernel above keeps its waves in lock-step (one barrier per 32 pairs), so every wave's non-matrix work - the slab
// adds, the weight reloads, the wait for the gathers - lands in the same gap of the matrix pipe (measured: matrix phase 211 us
// of 391 us on the level-2 128->128 layer, the rest does not overlap).  Here a wave owns 32 output columns and NOTHING is
// shared between waves but the read-only rulebook: each wave gathers the pairs' rows itself, straight into the MFMA A layout
// (lane (n, g): rows n and 16 + n, channels 16 q + 4 g ..), holds its columns of W[k] in registers and adds into its columns of
// the slab.  The gathers are redundant across the workgroup's waves (L1 / L2 hits); there is no LDS staging and no barrier
// between prologue and epilogue, so with two or more waves per SIMD one wave's stalls are covered by another's multiplies.
template <int CK, int NCH, int NW>
__device__ __forceinline__ void slab_direct_body(const SlabParams& p, float* smem) {
    constexpr int NT = 64 * NW;
    constexpr int NQ = CK / 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int R = p.R, SLD = p.ncols;
    const int64_t row0 = (int64_t)blockIdx.x * R;
    const int nrows = (int)((p.M - row0) < R ? (p.M - row0) : R);
    const int kb = (int)((int64_t)blockIdx.y * p.K / gridDim.y), ke = (int)((int64_t)(blockIdx.y + 1) * p.K / gridDim.y);
    const int KR = ke - kb;
    float* slab = smem;                                        // [(R + 1)][SLD], row R = sink of the padding entries
    int* cnt = (int*)(slab + (size_t)(R + 1) * SLD);           // [KR]
    int* meta = cnt + 128;
    int4v* otbl = (int4v*)(meta + 4);                          // [KR] non-empty offsets: (k, first unit, end unit, 0)
    int2v* tbl = (int2v*)(otbl + 128);                         // [ucap] (list position, k << 8 | rows)
    int2v* lst = p.lists + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * ((int64_t)p.K * p.lcap + 32);

    for (int i = tid * 4; i < (R + 1) * SLD; i += NT * 4) *(f32x4_t*)(slab + i) = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int kk = wv; kk < KR; kk += NW) {
        const int32_t* src = p.nbr + (int64_t)(kb + kk) * p.M + row0;
        int2v* dst = lst + (int64_t)kk * p.lcap;
        int base = 0;
        for (int it0 = 0; it0 < R; it0 += 256) {
            int v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = it0 + u * 64 + lane;
                v[u] = r < nrows ? src[r] : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint64_t m = __ballot(v[u] >= 0);
                if (v[u] >= 0) dst[base + __popcll(m & ((1ull << lane) - 1ull))] = int2v{v[u], it0 + u * 64 + lane};
                base += __popcll(m);
            }
        }
        const int padded = (base + 31) & ~31;
        if (lane < padded - base) dst[base + lane] = int2v{0, R};
        if (lane == 0) cnt[kk] = base;
    }
    if (wv == 0 && lane < 32) lst[(int64_t)KR * p.lcap + lane] = int2v{0, R};
    __syncthreads();
    if (wv == 0) {
        int running = 0, orun = 0;
        for (int k0 = 0; k0 < KR; k0 += 64) {
            const int kk = k0 + lane;
            const int c = kk < KR ? cnt[kk] : 0;
            const int nu = (c + 31) >> 5;
            int inc = nu;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int t = __shfl_up(inc, d);
                if (lane >= d) inc += t;
            }
            const int pos = running + inc - nu;
            for (int u = 0; u < nu; ++u) {
                const int rows = c - u * 32 < 32 ? c - u * 32 : 32;
                if (pos + u < p.ucap) tbl[pos + u] = int2v{kk * p.lcap + u * 32, (kk << 8) | rows};
            }
            const uint64_t ne = __ballot(nu > 0);
            if (nu > 0) otbl[orun + __popcll(ne & ((1ull << lane) - 1ull))] = int4v{kk, pos, pos + nu, 0};
            orun += __popcll(ne);
            running += __shfl(inc, 63);
        }
        if (lane == 0) { meta[0] = running < p.ucap ? running : p.ucap; meta[1] = orun; }
    }
    __syncthreads();
    const int TU = __builtin_amdgcn_readfirstlane(meta[0]), NO = __builtin_amdgcn_readfirstlane(meta[1]);
    const int T = TU * NCH;

    // (in row, local out row) of rows n and 16 + n of a step's unit; past the end: the (row 0, sink) block behind the lists
    auto load_entries = [&](int2v (&E)[2], int step) {
        const int u = step / NCH;
        const int pos = u < TU ? tbl[u][0] : KR * p.lcap;      // uniform
        E[0] = lst[pos + n];
        E[1] = lst[pos + 16 + n];
    };
    auto issue = [&](f32x4_t (&A)[2][NQ], const int2v (&E)[2], int step) {
        const int gc = step % NCH;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int ch = gc * CK + 16 * q;                   // uniform; a 16-channel group never straddles the concat split
            const bool first = ch < p.C0;
            const float* base = (first ? p.in0 + ch : p.in1 + (ch - p.C0)) + 4 * g;
            const int ld = first ? p.ld0 : p.ld1;
#pragma unroll
            for (int h = 0; h < 2; ++h) A[h][q] = *(const f32x4_t*)(base + (int64_t)E[h][0] * ld);
        }
    };
    f32x4_t W[2][NCH * NQ];                                    // [column block][16-channel group]: lane (n, g) holds W[cb * 16 + n][16 q + 4 g ..]
    f32x4_t acc[2][2];                                         // [row half][column block]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[h][cb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    f32x4_t A0[2][NQ], A1[2][NQ];
    int2v E0[2], E1[2], In[2];
    if (T > 0) {
        load_entries(E0, 0);
        load_entries(In, 1);
        issue(A0, E0, 0);
    }
    float* mycol = slab + wv * 32 + n;
    auto step = [&](auto par_c, auto c_c, int u) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr int C = decltype(c_c)::value;
        const int t = u * NCH + C;
        const int rows = __builtin_amdgcn_readfirstlane(tbl[u][1] & 255);
        // request the rows of step t + 1 (entries fetched a step ago) and fetch the entries of step t + 2: unconditional
        if (PAR == 0) { E1[0] = In[0]; E1[1] = In[1]; issue(A1, E1, t + 1); } else { E0[0] = In[0]; E0[1] = In[1]; issue(A0, E0, t + 1); }
        load_entries(In, t + 2);
        if (rows > 16) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const f32x4_t a0 = PAR == 0 ? A0[0][q] : A1[0][q], a1 = PAR == 0 ? A0[1][q] : A1[1][q];
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e4], W[0][C * NQ + q][e4], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e4], W[1][C * NQ + q][e4], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e4], W[0][C * NQ + q][e4], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e4], W[1][C * NQ + q][e4], acc[1][1], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const f32x4_t a0 = PAR == 0 ? A0[0][q] : A1[0][q];
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e4], W[0][C * NQ + q][e4], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e4], W[1][C * NQ + q][e4], acc[0][1], 0, 0, 0);
                }
            }
        }
        if (C == NCH - 1) {                                    // unit complete: add its tile into this wave's slab columns
            const int oy0 = PAR == 0 ? E0[0][1] : E1[0][1], oy1 = PAR == 0 ? E0[1][1] : E1[1][1];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = __shfl(oy0, 4 * g + r);          // local out row of pair 4 g + r (held by lane n' = 4 g + r)
                mycol[o * SLD] += acc[0][0][r];
                mycol[o * SLD + 16] += acc[0][1][r];
            }
            if (rows > 16) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = __shfl(oy1, 4 * g + r);
                    mycol[o * SLD] += acc[1][0][r];
                    mycol[o * SLD + 16] += acc[1][1][r];
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc[h][cb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto unit = [&](auto upar_c, int u) {
        constexpr int UP = decltype(upar_c)::value;
        step(std::integral_constant<int, (UP * NCH + 0) & 1>{}, std::integral_constant<int, 0>{}, u);
        if constexpr (NCH > 1) step(std::integral_constant<int, (UP * NCH + 1) & 1>{}, std::integral_constant<int, 1>{}, u);
        if constexpr (NCH > 2) step(std::integral_constant<int, (UP * NCH + 2) & 1>{}, std::integral_constant<int, 2>{}, u);
        if constexpr (NCH > 3) step(std::integral_constant<int, (UP * NCH + 3) & 1>{}, std::integral_constant<int, 3>{}, u);
    };
    for (int o = 0; o < NO; ++o) {
        const int4v oe = otbl[o];
        const int kk = __builtin_amdgcn_readfirstlane(oe[0]);
        const int u0 = __builtin_amdgcn_readfirstlane(oe[1]), u1 = __builtin_amdgcn_readfirstlane(oe[2]);
        {
            const float* wsrc = p.wt + ((int64_t)(kb + kk) * p.Cout + p.col0 + wv * 32 + n) * p.Cin + 4 * g;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < NCH * NQ; ++q) W[cb][q] = *(const f32x4_t*)(wsrc + (int64_t)cb * 16 * p.Cin + 16 * q);
        }
        int u = u0;
        if ((NCH & 1) && (u & 1)) { unit(std::integral_constant<int, 1>{}, u); ++u; }
        if constexpr ((NCH & 1) == 0) {
            for (; u < u1; ++u) unit(std::integral_constant<int, 0>{}, u);
        } else {
            for (; u + 1 < u1; u += 2) {
                unit(std::integral_constant<int, 0>{}, u);
                unit(std::integral_constant<int, 1>{}, u + 1);
            }
            if (u < u1) unit(std::integral_constant<int, 0>{}, u);
        }
    }
    __syncthreads();
    const int c4n = SLD >> 2;
    if (p.ksplit > 1) {
        float* dst = p.part + ((int64_t)blockIdx.y * p.M + row0) * p.Cout + p.col0;
        for (int i = tid; i < nrows * c4n; i += NT) {
            const int r = i / c4n, q = (i - r * c4n) * 4;
            *(f32x4_t*)(dst + (int64_t)r * p.Cout + q) = *(const f32x4_t*)(slab + r * SLD + q);
        }
        return;
    }
    for (int i = tid; i < nrows * c4n; i += NT) {
        const int r = i / c4n, q = (i - r * c4n) * 4;
        const f32x4_t a = *(const f32x4_t*)(slab + r * SLD + q);
        float y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = p.col0 + q + j;
            float t = a[j] * (p.scale ? p.scale[col] : 1.f) + (p.shift ? p.shift[col] : 0.f);
            if (p.res) t += p.res[(row0 + r) * p.ld_res + col];
            y[j] = slab_act(t, p.act);
        }
        *(f32x4_t*)(p.out + (row0 + r) * p.ld_out + p.col0 + q) = f32x4_t{y[0], y[1], y[2], y[3]};
    }
}

#define SLABD_ENTRY(CK, NCH, NW, WPE)                                                                                  \
    __global__ __launch_bounds__(64 * NW, WPE) void slab_direct_kernel_##CK##_##NCH##_##NW(const SlabParams p) {       \
        extern __shared__ __attribute__((aligned(16))) float slab_smem[];                                              \
        slab_direct_body<CK, NCH, NW>(p, slab_smem);                                                                   \
    }
// (channels per step, steps per unit, waves = 32-column blocks, waves per SIMD the register budget is sized for)
SLABD_ENTRY(32, 1, 1, 2)
SLABD_ENTRY(32, 1, 2, 2)
SLABD_ENTRY(64, 1, 2, 2)
SLABD_ENTRY(64, 1, 4, 2)
SLABD_ENTRY(96, 1, 3, 2)
SLABD_ENTRY(64, 2, 3, 2)
SLABD_ENTRY(64, 2, 4, 2)
SLABD_ENTRY(96, 2, 4, 2)
SLABD_ENTRY(64, 4, 4, 2)
SLABD_ENTRY(64, 2, 8, 2)
SLABD_ENTRY(64, 4, 8, 2)

// out = act(scale * sum_z part[z] + shift + res), z ascending (= offsets ascending): the k-split's second pass
struct SlabReduceParams {
    const float* part; int ksplit; int64_t M; int Cout;
    const float* scale; const float* shift; const float* res; int ld_res; float* out; int ld_out; int act;
};
__global__ __launch_bounds__(256) void slab_reduce_kernel(const SlabReduceParams p) {
    const int c4 = p.Cout >> 2;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = e / c4;
    if (r >= p.M) return;
    const int q = (int)(e - r * c4) * 4;
    f32x4_t a = *(const f32x4_t*)(p.part + r * p.Cout + q);
    for (int z = 1; z < p.ksplit; ++z) a += *(const f32x4_t*)(p.part + ((int64_t)z * p.M + r) * p.Cout + q);
    float y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float t = a[j] * (p.scale ? p.scale[q + j] : 1.f) + (p.shift ? p.shift[q + j] : 0.f);
        if (p.res) t += p.res[r * p.ld_res + q + j];
        y[j] = slab_act(t, p.act);
    }
    *(f32x4_t*)(p.out + r * p.ld_out + q) = f32x4_t{y[0], y[1], y[2], y[3]};
}

// ---- launcher ---------------------------------------------------------------------------------
struct SlabPlan { int ck, nch, ncb, ncg, R, ksplit, lcap, ucap; size_t lds, ws_bytes; };

static int slab_lds_bytes(int R, int ncols, int ck, int K, int ucap) {
    return (int)(((size_t)(R + 1) * ncols + 2 * 32 * (ck + 4)) * 4 + (128 + 128 + 4) * 4 + 128 * 16 + (size_t)ucap * 8);
}

// Geometry of one convolution; 0 = this shape is not handled by the slab kernel (the caller keeps the pair-major path).
int slab_conv_plan(int K, int Cin, int Cout, int64_t M, int64_t n_pairs, int n_cu, SlabPlan* pl) {
    if (K > 128 || Cout % 16 || Cin % 32 || M <= 0) return 0;
    int ck = 0;
    for (int c = 128; c >= 32; c -= 32) if (Cin % c == 0 && Cin / c <= 3) { ck = c; break; }
    if (!ck) return 0;
    const int nch = Cin / ck;
    // waves per workgroup: 16 columns each; the weights of a wave live in Cin / 4 registers, so wide inputs cap the
    // workgroup at 8 waves (256 registers each) and wider outputs run as column groups
    int ncb = Cout / 16, ncg = 1;
    const int max_waves = (Cin > 256) ? 8 : 16;
    while (ncb > max_waves) { if (ncb % 2) return 0; ncb /= 2; ncg *= 2; }
    static const int have[][3] = {{32, 1, 2}, {32, 1, 4}, {64, 1, 4}, {64, 1, 8}, {96, 1, 6}, {128, 1, 6}, {128, 1, 8}, {96, 2, 8},
                                  {128, 2, 8}, {128, 3, 8}, {128, 1, 16}, {128, 2, 16}, {128, 1, 4}};
    { static int ncb_env = -1; if (ncb_env < 0) { const char* e = getenv("SD3D_SLAB_NCB"); ncb_env = e ? atoi(e) : 0; }
      if (ncb_env > 0) while (ncb > ncb_env && ncb % 2 == 0) { ncb /= 2; ncg *= 2; } }
    bool ok = false;
    for (auto& h : have) ok |= (h[0] == ck && h[1] == nch && h[2] == ncb);
    if (!ok) return 0;
    const int ncols = ncb * 16;
    // slab rows: as many as LDS allows for the targeted workgroups per CU (narrow layers keep several workgroups resident)
    const int wg_per_cu = ncb >= 6 ? 1 : (ncb == 4 ? 2 : 4);
    const int budget = (160 * 1024) / wg_per_cu - 1024;
    int R = 512;
    for (; R >= 32; R -= 16) {
        const int ucap = K * (R / 32 + 2);
        if (slab_lds_bytes(R, ncols, ck, K, ucap) <= budget) break;
    }
    if (R < 32) return 0;
    const double density = (double)n_pairs / ((double)K * (double)M);
    (void)density;
    // few rows: spread the slabs over the chip, then split the offsets
    int64_t slabs = cdiv(M, R);
    int ksplit = 1;
    const int64_t slots = (int64_t)n_cu * wg_per_cu;
    if (slabs * ncg < slots) {
        // shrink the slab until the grid covers the chip once, but keep >= 64 rows (unit fill); then split offsets
        int Rt = (int)cdiv(M * ncg, slots);
        Rt = (Rt + 15) / 16 * 16;
        if (Rt < 64) Rt = 64;
        if (Rt < R) R = Rt;
        slabs = cdiv(M, R);
        while (slabs * ncg * ksplit * 2 <= slots && ksplit * 2 <= 8 && K / (ksplit * 2) >= 3) ksplit *= 2;
    }
    pl->ck = ck; pl->nch = nch; pl->ncb = ncb; pl->ncg = ncg; pl->R = R; pl->ksplit = ksplit;
    pl->lcap = (R + 31) / 32 * 32;
    pl->ucap = K * (R / 32 + 2);
    pl->lds = (size_t)slab_lds_bytes(R, ncols, ck, K, pl->ucap);
    pl->ws_bytes = (size_t)slabs * ksplit * ncg * ((size_t)K * pl->lcap + 32) * sizeof(int2v) + (ksplit > 1 ? (size_t)ksplit * M * Cout * sizeof(float) : 0) + 256;
    return 1;
}

static int slab_direct_lds_bytes(int R, int ncols, int ucap) {
    return (int)((size_t)(R + 1) * ncols * 4 + (128 + 4) * 4 + 128 * 16 + (size_t)ucap * 8);
}

static int slab_mode() {
    static int m = -1;
    if (m < 0) { const char* e = getenv("SD3D_SLAB_MODE"); m = e ? atoi(e) : 0; }
    return m;
}

// geometry of the barrier-free variant; 0 = shape not handled
int slab_direct_plan(int K, int Cin, int Cout, int64_t M, int64_t n_pairs, int n_cu, SlabPlan* pl) {
    if (K > 128 || Cout % 32 || Cin % 32 || M <= 0) return 0;
    int ck = 0, nch = 0;
    if (Cin <= 96) { ck = Cin; nch = 1; }
    else if (Cin == 128) { ck = 64; nch = 2; }       // 128-channel steps would need 128 gather registers per stage: spills
    else if (Cin == 192) { ck = 96; nch = 2; }
    else if (Cin == 256) { ck = 64; nch = 4; }
    else return 0;
    int nw = Cout / 32, ncg = 1;
    while (nw > 8) { if (nw % 2) return 0; nw /= 2; ncg *= 2; }
    static const int have[][3] = {{32, 1, 1}, {32, 1, 2}, {64, 1, 2}, {64, 1, 4}, {96, 1, 3}, {64, 2, 3}, {64, 2, 4}, {96, 2, 4},
                                  {64, 4, 4}, {64, 2, 8}, {64, 4, 8}};
    bool ok = false;
    for (auto& h : have) ok |= (h[0] == ck && h[1] == nch && h[2] == nw);
    if (!ok) return 0;
    const int ncols = nw * 32;
    // two waves per SIMD: 8 waves per CU -> workgroups per CU = 8 / waves
    int wg_per_cu = 8 / nw; if (wg_per_cu < 1) wg_per_cu = 1; if (nw == 3) wg_per_cu = 2;
    const int budget = (160 * 1024) / wg_per_cu - 512;
    int R = 512;
    for (; R >= 32; R -= 16) if (slab_direct_lds_bytes(R, ncols, K * (R / 32 + 2)) <= budget) break;
    if (R < 32) return 0;
    int64_t slabs = cdiv(M, R);
    int ksplit = 1;
    const int64_t slots = (int64_t)n_cu * wg_per_cu;
    if (slabs * ncg < slots) {
        int Rt = (int)cdiv(M * ncg, slots);
        Rt = (Rt + 15) / 16 * 16;
        if (Rt < 64) Rt = 64;
        if (Rt < R) R = Rt;
        slabs = cdiv(M, R);
        while (slabs * ncg * ksplit * 2 <= slots && ksplit * 2 <= 8 && K / (ksplit * 2) >= 3) ksplit *= 2;
    }
    pl->ck = ck; pl->nch = nch; pl->ncb = nw; pl->ncg = ncg; pl->R = R; pl->ksplit = ksplit;
    pl->lcap = (R + 31) / 32 * 32;
    pl->ucap = K * (R / 32 + 2);
    pl->lds = (size_t)slab_direct_lds_bytes(R, ncols, pl->ucap);
    pl->ws_bytes = (size_t)slabs * ksplit * ((size_t)K * pl->lcap + 32) * sizeof(int2v) + (ksplit > 1 ? (size_t)ksplit * M * Cout * sizeof(float) : 0) + 256;
    return 1;
}

size_t slab_conv_ws_bytes(int K, int Cin, int Cout, int64_t M, int64_t n_pairs) {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) n_cu = 256;
        else n_cu = prop.multiProcessorCount;
    }
    SlabPlan pl;
    if (slab_mode() == 1) return slab_direct_plan(K, Cin, Cout, M, n_pairs, n_cu, &pl) ? pl.ws_bytes : 0;
    if (!slab_conv_plan(K, Cin, Cout, M, n_pairs, n_cu, &pl)) return 0;
    return pl.ws_bytes;
}

int launch_slab_conv(const float* in0, int ld0, int C0, const float* in1, int ld1, const int32_t* nbr, int64_t n_pairs, const float* wt,
                     int K, int Cin, int Cout, int64_t M, const float* scale, const float* shift, const float* res, int ld_res,
                     float* out, int ld_out, int act, void* ws, size_t ws_bytes, hipStream_t st) {
    if (M <= 0 || Cout <= 0) return SD3D_OK;
    if (in1 && ((C0 & 31) || C0 > Cin)) return sd3d_set_error(SD3D_ERR_ARG, "slab_conv: concat split must be a multiple of 32");
    if (!in1) C0 = Cin;
    if ((ld0 & 3) || (in1 && (ld1 & 3)) || (ld_out & 3)) return sd3d_set_error(SD3D_ERR_ARG, "slab_conv: row strides must be multiples of 4 floats");
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return sd3d_set_error(SD3D_ERR_LAUNCH, "slab_conv: no device");
        n_cu = prop.multiProcessorCount;
    }
    SlabPlan pl;
    const bool direct = slab_mode() == 1;
    if (!(direct ? slab_direct_plan(K, Cin, Cout, M, n_pairs, n_cu, &pl) : slab_conv_plan(K, Cin, Cout, M, n_pairs, n_cu, &pl)))
        return sd3d_set_error(SD3D_ERR_ARG, "slab_conv: shape not supported");
    { static int r_env = -1, ks_env = -1;                       // SD3D_SLAB_R / SD3D_SLAB_KSPLIT: tuning overrides
      if (r_env < 0) { const char* e = getenv("SD3D_SLAB_R"); r_env = e ? atoi(e) : 0; }
      if (ks_env < 0) { const char* e = getenv("SD3D_SLAB_KSPLIT"); ks_env = e ? atoi(e) : 0; }
      if (r_env > 0 || ks_env > 0) {
          if (r_env > 0) pl.R = r_env;
          if (ks_env > 0) pl.ksplit = ks_env;
          pl.lcap = (pl.R + 31) / 32 * 32; pl.ucap = K * (pl.R / 32 + 2);
          pl.lds = direct ? (size_t)slab_direct_lds_bytes(pl.R, pl.ncb * 32, pl.ucap) : (size_t)slab_lds_bytes(pl.R, pl.ncb * 16, pl.ck, K, pl.ucap);
          pl.ws_bytes = (size_t)cdiv(M, pl.R) * pl.ksplit * pl.ncg * ((size_t)K * pl.lcap + 32) * sizeof(int2v) + (pl.ksplit > 1 ? (size_t)pl.ksplit * M * Cout * sizeof(float) : 0) + 256;
      } }
    if (pl.lds > 160 * 1024) return sd3d_set_error(SD3D_ERR_ARG, "slab_conv: slab does not fit LDS");
    if (ws_bytes < pl.ws_bytes) return sd3d_set_error(SD3D_ERR_WS, "slab_conv: workspace too small (sd3d_slab_conv_ws_bytes)");
    const int64_t slabs = cdiv(M, pl.R);
    SlabParams p;
    p.in0 = in0; p.ld0 = ld0; p.C0 = C0; p.in1 = in1; p.ld1 = ld1; p.nbr = nbr; p.K = K; p.M = M; p.wt = wt; p.Cin = Cin; p.Cout = Cout;
    p.scale = scale; p.shift = shift; p.res = res; p.ld_res = ld_res; p.out = out; p.ld_out = ld_out; p.act = act;
    p.R = pl.R; p.lcap = pl.lcap; p.ucap = pl.ucap; p.ksplit = pl.ksplit;
    p.lists = (int2v*)ws;
    p.part = pl.ksplit > 1 ? (float*)((char*)ws + align_up((size_t)slabs * pl.ksplit * pl.ncg * ((size_t)K * pl.lcap + 32) * sizeof(int2v), 256)) : nullptr;
    p.ncols = pl.ncb * (direct ? 32 : 16);
    const dim3 grid((unsigned)slabs, (unsigned)pl.ksplit), block(64 * pl.ncb);
#define SLAB_CASE(CK, NCH, NCB)                                                                                          \
    if (pl.ck == CK && pl.nch == NCH && pl.ncb == NCB) {                                                                 \
        static bool attr = false;                                                                                        \
        if (!attr) { (void)hipFuncSetAttribute((const void*)slab_conv_kernel_##CK##_##NCH##_##NCB,                       \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }   \
        p.col0 = 0;                                                                                                      \
        hipLaunchKernelGGL(slab_conv_kernel_##CK##_##NCH##_##NCB, dim3(grid.x, grid.y, pl.ncg), block, pl.lds, st, p);   \
    } else
#define SLABD_CASE(CK, NCH, NW)                                                                                          \
    if (direct && pl.ck == CK && pl.nch == NCH && pl.ncb == NW) {                                                        \
        static bool attr = false;                                                                                        \
        if (!attr) { (void)hipFuncSetAttribute((const void*)slab_direct_kernel_##CK##_##NCH##_##NW,                      \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }   \
        for (int cg = 0; cg < pl.ncg; ++cg) {                                                                            \
            p.col0 = cg * p.ncols;                                                                                       \
            hipLaunchKernelGGL(slab_direct_kernel_##CK##_##NCH##_##NW, grid, block, pl.lds, st, p);                      \
        }                                                                                                                \
    } else
    SLABD_CASE(32, 1, 1) SLABD_CASE(32, 1, 2) SLABD_CASE(64, 1, 2) SLABD_CASE(64, 1, 4) SLABD_CASE(96, 1, 3) SLABD_CASE(64, 2, 3)
    SLABD_CASE(64, 2, 4) SLABD_CASE(96, 2, 4) SLABD_CASE(64, 4, 4) SLABD_CASE(64, 2, 8) SLABD_CASE(64, 4, 8)
    if (direct) { return sd3d_set_error(SD3D_ERR_ARG, "slab_conv: no kernel variant"); } else
    SLAB_CASE(32, 1, 2) SLAB_CASE(32, 1, 4) SLAB_CASE(64, 1, 4) SLAB_CASE(64, 1, 8) SLAB_CASE(96, 1, 6) SLAB_CASE(128, 1, 6)
    SLAB_CASE(128, 1, 8) SLAB_CASE(96, 2, 8) SLAB_CASE(128, 2, 8) SLAB_CASE(128, 3, 8) SLAB_CASE(128, 1, 16) SLAB_CASE(128, 2, 16) SLAB_CASE(128, 1, 4)
    { return sd3d_set_error(SD3D_ERR_ARG, "slab_conv: no kernel variant"); }
#undef SLAB_CASE
    if (pl.ksplit > 1) {
        SlabReduceParams r;
        r.part = p.part; r.ksplit = pl.ksplit; r.M = M; r.Cout = Cout; r.scale = scale; r.shift = shift; r.res = res; r.ld_res = ld_res;
        r.out = out; r.ld_out = ld_out; r.act = act;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)cdiv(M * (Cout / 4), 256)), dim3(256), 0, st, r);
    }
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
