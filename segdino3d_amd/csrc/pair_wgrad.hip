// Weight gradient of the pair-major sparse convolution (training step):
//     dW[k][co][ci] = sum over the pairs p of offset k of  dY[out_row(p)][co] * X[in_idx(p)][ci]
// i.e. one [Cout x P_k] x [P_k x Cin] GEMM per kernel offset whose contraction runs over the rulebook's pairs - the
// transpose of what MinkowskiEngine / spconv compute with their gather-GEMM-scatter backward (third-party, not in the
// reference tree; the reference reaches it through autograd, train_engine_3d.py:88-122).  The input gradient needs no
// kernel of its own: it is the forward pair_conv on the transposed rulebook (see ops.pair_conv_backward).
//
// Work split: the offset-major pair list (pair_gemm.hip) is cut into balanced contiguous tile ranges, one per
// blockIdx.x; a workgroup accumulates a [<=128 x <=128] block of dW in MFMA accumulators (v_mfma_f32_32x32x2_f32: exact
// fp32, the contraction index of the instruction IS the pair index) over the 32-pair steps of its range, staging the
// gathered dY / X rows through LDS with dwordx4 loads.  Whenever the offset changes inside a range - and at its end - the
// block goes to a partial slot; slots are numbered in list order (range index + offset changes before it), so the slots
// of one offset are contiguous and pass 2 adds them up in a fixed order: bit-reproducible, no atomics.
#include "common.h"
#include "../../include/segdino3d_hip.h"
#include <stdlib.h>

#define PT 128                       // pairs per tile of the pair lists
#define WG_STEP 64                   // pairs staged per step (32: same speed within 3 %)

struct WGParams {
    const float* dy; int ld_dy; const float* x; int ld_x;
    const int32_t* in_idx; const int32_t* out_idx; const int32_t* tile_k; int n_tiles;
    int Cin, Cout;
    float* wpart; int32_t* slot_k; int n_slots;
    int bf16;                        // round the staged operands to bf16 (nearest even): bf16-operand numerics on the fp32 matrix pipe
    int bias;                        // 1 (K = 1 only): a slot is Cout*Cin + Cout floats, the tail = column sums of dY over the slot's pairs
};

__device__ __forceinline__ float round_to_bf16(float v) {
    uint32_t u = __float_as_uint(v);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return __uint_as_float(u & 0xFFFF0000u);
}

__global__ __launch_bounds__(256) void pair_out_rows_kernel(const int32_t* __restrict__ pos, int K, int64_t M, int32_t* __restrict__ out_idx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)K * M) return;
    const int p = pos[i];
    if (p >= 0) out_idx[p] = (int32_t)(i % M);
}

template <int NCO, int NCI, int NW>
__global__ __launch_bounds__(NW * 64) void pair_wgrad_kernel(const WGParams p) {
    constexpr int NTHR = NW * 64;
    constexpr int TCO = NCO * 32, TCI = NCI * 32;
    constexpr int LDA = (TCO % 64) ? TCO : TCO + 32;           // row stride = 32 mod 64 floats: the two pair rows an MFMA reads sit in different bank halves
    constexpr int LDB = (TCI % 64) ? TCI : TCI + 32;
    constexpr int NT = NCO * NCI;                              // 32 x 32 output tiles of the block
    constexpr int TPW = (NT + NW - 1) / NW;                    // tiles per wave
    constexpr int A4 = WG_STEP * TCO / 4, B4 = WG_STEP * TCI / 4;      // float4 pieces per step
    constexpr int NA = (A4 + NTHR - 1) / NTHR, NB = (B4 + NTHR - 1) / NTHR;
    __shared__ __attribute__((aligned(16))) float As[WG_STEP * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[WG_STEP * LDB];
    __shared__ int red[NW], red2[NW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: tile offsets live in SGPRs, no exec-masked branches
    const int n_real = p.tile_k[p.n_tiles];
    const int t0 = (int)((int64_t)blockIdx.x * n_real / gridDim.x);
    const int t1 = (int)((int64_t)(blockIdx.x + 1) * n_real / gridDim.x);
    const bool owner = blockIdx.y == 0 && blockIdx.z == 0;     // the workgroup of the range that keeps the slot table
    if (t1 <= t0 && !owner) return;
    const int co0 = blockIdx.y * TCO, ci0 = blockIdx.z * TCI;
    // slot of the first run = range index + offset changes in tiles (0, t0]; `more` = changes in (t0, t1] (a tile index past the
    // last real tile is no change).  Range r owns the slots [slot_r, slot_{r+1}) with slot_{r+1} = r + 1 + changes in (0, t1]: the
    // ones its runs do not fill are marked empty by its owner workgroup, so the table needs no memset before the launch.
    int changes = 0, more = 0;
    for (int t = 1 + tid; t <= t1 && t < n_real; t += NTHR) {
        const int c = p.tile_k[t] != p.tile_k[t - 1];
        if (t <= t0) changes += c; else more += c;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { changes += __shfl_xor(changes, o, 64); more += __shfl_xor(more, o, 64); }
    if (lane == 0) { red[wv] = changes; red2[wv] = more; }
    __syncthreads();
    int slot = blockIdx.x, slot_end = blockIdx.x + 1;
#pragma unroll
    for (int i = 0; i < NW; ++i) { slot += red[i]; slot_end += red[i] + red2[i]; }
    if (blockIdx.x + 1 == gridDim.x) slot_end = p.n_slots;
    if (t1 <= t0) {                                            // empty range (fewer tiles than ranges): its slots are empty
        for (int s2 = slot + tid; s2 < slot_end; s2 += NTHR) p.slot_k[s2] = -1;
        return;
    }

    f32x16 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // bias gradient of a Linear (K = 1): the column sums of dY ride along - the dY rows are in LDS anyway; thread c of the
    // first input-channel block adds column c of every staged step in row order (fixed order: bit-reproducible)
    const bool do_bias = p.bias && blockIdx.z == 0 && tid < TCO;
    const int64_t slab = (int64_t)p.Cout * p.Cin + (p.bias ? p.Cout : 0);
    float bsum = 0.f;

    // float4 piece f of a step: pair row f / (T/4), column piece f % (T/4)
    f32x4 ra[NA], rb[NB];
    auto fetch = [&](int tile, int sub) {
        const int64_t pb = (int64_t)tile * PT + sub * WG_STEP;
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int f = u * NTHR + tid;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (f < A4) {
                const int row = f / (TCO / 4), c4 = f % (TCO / 4);
                const int r = p.out_idx[pb + row];
                if (r >= 0 && co0 + c4 * 4 < p.Cout) v = *(const f32x4*)(p.dy + (int64_t)r * p.ld_dy + co0 + c4 * 4);
            }
            ra[u] = v;
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int f = u * NTHR + tid;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (f < B4) {
                const int row = f / (TCI / 4), c4 = f % (TCI / 4);
                const int r = p.in_idx[pb + row];
                if (r >= 0 && ci0 + c4 * 4 < p.Cin) v = *(const f32x4*)(p.x + (int64_t)r * p.ld_x + ci0 + c4 * 4);
            }
            rb[u] = v;
        }
    };
    auto stage = [&]() {
        if (p.bf16) {                                          // uniform; here (not in fetch) so the loads stay in flight over the MFMA phase
#pragma unroll
            for (int u = 0; u < NA; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[u][e] = round_to_bf16(ra[u][e]);
#pragma unroll
            for (int u = 0; u < NB; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[u][e] = round_to_bf16(rb[u][e]);
        }
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int f = u * NTHR + tid;
            if (f < A4) *(f32x4*)(As + (f / (TCO / 4)) * LDA + (f % (TCO / 4)) * 4) = ra[u];
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int f = u * NTHR + tid;
            if (f < B4) *(f32x4*)(Bs + (f / (TCI / 4)) * LDB + (f % (TCI / 4)) * 4) = rb[u];
        }
    };
    auto flush = [&](int k) {
        float* dst = p.wpart + (int64_t)slot * slab;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int t = wv + NW * i;
            if (t < NT) {
                const int co = co0 + (t / NCI) * 32, ci = ci0 + (t % NCI) * 32 + (lane & 31);
                if (ci < p.Cin) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = co + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                        if (row < p.Cout) dst[(int64_t)row * p.Cin + ci] = acc[i][r];
                        acc[i][r] = 0.f;
                    }
                }
            }
        }
        if (do_bias) {
            if (co0 + tid < p.Cout) dst[(int64_t)p.Cout * p.Cin + co0 + tid] = bsum;
            bsum = 0.f;
        }
        if (tid == 0 && blockIdx.y == 0 && blockIdx.z == 0) p.slot_k[slot] = k;
        ++slot;
    };

    int a_off[TPW], b_off[TPW];
    bool t_ok[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = wv + NW * i;
        t_ok[i] = t < NT;
        const int tt = t < NT ? t : NT - 1;
        a_off[i] = (tt / NCI) * 32; b_off[i] = (tt % NCI) * 32;
    }
    int cur_k = p.tile_k[t0];
    fetch(t0, 0);
    for (int tile = t0; tile < t1; ++tile) {
        const int k = p.tile_k[tile];
        if (k != cur_k) { flush(cur_k); cur_k = k; }
        for (int sub = 0; sub < PT / WG_STEP; ++sub) {
            __syncthreads();                                   // the previous step's MFMAs have read LDS
            stage();
            __syncthreads();
            // next step's rows travel while this one multiplies
            if (sub + 1 < PT / WG_STEP) fetch(tile, sub + 1);
            else if (tile + 1 < t1) fetch(tile + 1, 0);
            // branch-free inner loop: a wave whose i-th tile does not exist (NT not a multiple of 4) multiplies zeros into an
            // accumulator it never flushes - it would wait at the barrier anyway, and a guard around the MFMA costs every wave an
            // exec-masked branch and a full LDS wait per instruction
#pragma unroll
            for (int s = 0; s < WG_STEP / 2; ++s) {
                const float* ar = As + (2 * s + (lane >> 5)) * LDA + (lane & 31);
                const float* br = Bs + (2 * s + (lane >> 5)) * LDB + (lane & 31);
                float av[TPW], bv[TPW];
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    const float a = ar[a_off[i]];
                    av[i] = (NT % NW == 0 || t_ok[i]) ? a : 0.f;
                    bv[i] = br[b_off[i]];
                }
#pragma unroll
                for (int i = 0; i < TPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc[i], 0, 0, 0);
            }
            if (do_bias) {
#pragma unroll 16
                for (int r = 0; r < WG_STEP; ++r) bsum += As[r * LDA + tid];
            }
        }
    }
    flush(cur_k);
    if (owner)
        for (int s2 = slot + tid; s2 < slot_end; s2 += NTHR) p.slot_k[s2] = -1;
}

// pass 2: dW[k] (+)= sum of the slots of offset k, in slot order
__global__ __launch_bounds__(256) void pair_wgrad_reduce_kernel(const float* __restrict__ wpart, const int32_t* __restrict__ slot_k, int n_slots,
                                                                int64_t elems, float* __restrict__ dw, int accumulate) {
    __shared__ int s_first, s_last;
    const int k = blockIdx.y;
    if (threadIdx.x == 0) { s_first = n_slots; s_last = -1; }
    __syncthreads();
    for (int s = threadIdx.x; s < n_slots; s += 256)
        if (slot_k[s] == k) { atomicMin(&s_first, s); atomicMax(&s_last, s); }
    __syncthreads();
    const int first = s_first, last = s_last;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < elems; e += (int64_t)gridDim.x * 256) {
        float a = 0.f;
        for (int s = first; s <= last; ++s)
            if (slot_k[s] == k) a += wpart[(int64_t)s * elems + e];
        float* d = dw + (int64_t)k * elems + e;
        *d = accumulate ? *d + a : a;
    }
}

// ---- weight (and bias) gradient of a Linear on a few thousand rows: ONE launch, no partial blocks -------------------------------------
// dW[co][ci] = sum_m g[m][co] x[m][ci] for the decoder's Linears (M = 2 441 queries / 3 000 superpoints per step, ~130 of them): through the
// pair-list kernel above such a product is two launches (20 one-tile ranges write 20 partial blocks, the reduce adds them up): 21 + 13 us
// for 0.3 GFLOP.  Here a workgroup owns one 32 x 32 block of dW for ALL rows: rounds of 128 rows are staged in LDS (one dwordx4 per thread and operand), each of its sixteen waves multiplies eight of them
// (four v_mfma_f32_32x32x2_f32), and the sixteen accumulators are added through
// LDS in a fixed order - bit-reproducible, no atomics.  db (column sums of g) rides along in the workgroups of the first ci block.
#define LW_WAVES 16
#define LW_ROWS 128              // rows staged per round: eight per wave = four MFMA steps
#define LW_STAGES 4              // rounds of requests in flight per thread
#define LW_LD 36                 // floats per staged row (32 + 4: rows stay 16-byte aligned, the two rows of an MFMA step fall into different banks)
__global__ __launch_bounds__(LW_WAVES * 64) void linear_wgrad_small_kernel(const float* __restrict__ g, int ld_g, const float* __restrict__ x, int ld_x,
                                                                         int64_t M, int Cin, int Cout, float* __restrict__ dw, float* __restrict__ db,
                                                                         int bf16, int accumulate) {
    // (a first version fed the MFMAs straight from memory, one dword per lane and operand: two wave-wide loads per MFMA through the CU's
    //  one address unit - 29 us per 256 x 256 x 2 441 product whatever the number of waves or loads in flight; here a round of 128 rows is
    //  fetched with one dwordx4 per thread and operand and multiplied out of LDS)
    __shared__ __attribute__((aligned(16))) float smem[2 * LW_ROWS * LW_LD > (LW_WAVES / 2) * 1024 ? 2 * LW_ROWS * LW_LD : (LW_WAVES / 2) * 1024];
    __shared__ float bred[LW_WAVES][32];
    float* const As = smem;                                    // [LW_ROWS][LW_LD]  g rows (this block's 32 output columns)
    float* const Bs = smem + LW_ROWS * LW_LD;                  // [LW_ROWS][LW_LD]  x rows (this block's 32 input columns)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, k = lane >> 5;
    const int co0 = blockIdx.x * 32, ci0 = blockIdx.y * 32;
    const int srow = tid >> 3, seg = (tid & 7) * 4;            // staging: thread = (row of the round, four columns)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bs = 0.f;
    // fetch = the two requests and nothing else (clamped addresses: unconditional, nothing looks at the values until the round is staged -
    // a select or a rounding right behind the load, or a load under a branch, makes hipcc wait for it on the spot: `global_load; s_waitcnt
    // vmcnt(0)` put one memory round trip into every round of the first versions); finish = masks (+ bf16 rounding) when the round is staged
    const bool a_in = co0 + seg < ((Cout + 3) & ~3), b_in = ci0 + seg < ((Cin + 3) & ~3);
    const float* const gp = g + (a_in ? co0 + seg : 0);
    const float* const xp = x + (b_in ? ci0 + seg : 0);
    auto fetch = [&](int64_t r0, f32x4& va, f32x4& vb) {
        const int64_t row = r0 + srow;
        const int64_t rc = row < M ? row : M - 1;
        va = *(const f32x4*)(gp + rc * ld_g);
        vb = *(const f32x4*)(xp + rc * ld_x);
    };
    auto finish = [&](int64_t r0, f32x4& va, f32x4& vb) {
        const bool row_ok = r0 + srow < M;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (!row_ok || co0 + seg + c >= Cout) va[c] = 0.f;
            if (!row_ok || ci0 + seg + c >= Cin) vb[c] = 0.f;
        }
        if (bf16) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { va[c] = round_to_bf16(va[c]); vb[c] = round_to_bf16(vb[c]); }
        }
    };
    // LW_STAGES rounds of requests in flight (registers): a round multiplies for ~0.3 us, a request takes 1 - 2 us - with one round in flight the
    // launch lasted 20 rounds x the memory latency, 25 us whatever the number of workgroups (profiles/EXPERIMENTS.md round 6)
    f32x4 va[LW_STAGES], vb[LW_STAGES];
    const int64_t nrounds = (M + LW_ROWS - 1) / LW_ROWS;
#pragma unroll
    for (int u = 0; u < LW_STAGES; ++u) fetch((int64_t)(u < nrounds ? u : nrounds - 1) * LW_ROWS, va[u], vb[u]);
    for (int64_t rd = 0; rd < nrounds; rd += LW_STAGES) {
#pragma unroll
        for (int u = 0; u < LW_STAGES; ++u) {
            if (rd + u < nrounds) {                             // uniform
                finish((rd + u) * LW_ROWS, va[u], vb[u]);
                __syncthreads();                                // the previous round's products have read LDS
                *(f32x4*)(As + srow * LW_LD + seg) = va[u];
                *(f32x4*)(Bs + srow * LW_LD + seg) = vb[u];
                __syncthreads();
                {   // UNCONDITIONAL (a round past the end reloads the last one: hipcc waits on the spot for loads it issues under a branch)
                    const int64_t nx = rd + u + LW_STAGES;
                    fetch((nx < nrounds ? nx : nrounds - 1) * LW_ROWS, va[u], vb[u]);
                }
                const float* ar = As + (8 * wv + k) * LW_LD + i;
                const float* br = Bs + (8 * wv + k) * LW_LD + i;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float a = ar[2 * q * LW_LD], b = br[2 * q * LW_LD];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                    bs += a;
                }
            }
        }
    }
    // D[row][col]: lane holds column (lane & 31), register r row 8 (r >> 2) + 4 (lane >> 5) + (r & 3).  Waves 8 .. 15 hand their block to
    // waves 0 .. 7 (acc_w + acc_{w + 8}), then the eight sums are added in wave order: one fixed order whatever the timing.
    float (*red)[1024] = (float (*)[1024])smem;
    bs += __shfl_xor(bs, 32);
    if (k == 0) bred[wv][i] = bs;
    __syncthreads();                                            // the staging area becomes the reduction area
    if (wv >= LW_WAVES / 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wv - LW_WAVES / 2][(8 * (r >> 2) + 4 * k + (r & 3)) * 32 + i] = acc[r];
    }
    __syncthreads();
    if (wv < LW_WAVES / 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float* slot = &red[wv][(8 * (r >> 2) + 4 * k + (r & 3)) * 32 + i];
            *slot = acc[r] + *slot;
        }
    }
    __syncthreads();
    for (int e = tid; e < 1024; e += LW_WAVES * 64) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < LW_WAVES / 2; ++w) a += red[w][e];
        const int row = co0 + (e >> 5), col = ci0 + (e & 31);
        if (row < Cout && col < Cin) {
            float* d = dw + (int64_t)row * Cin + col;
            *d = accumulate ? *d + a : a;
        }
    }
    if (db && blockIdx.y == 0 && tid < 32 && co0 + tid < Cout) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < LW_WAVES; ++w) a += bred[w][tid];
        db[co0 + tid] = accumulate ? db[co0 + tid] + a : a;
    }
}

// tile ranges (= workgroups per output block): enough to fill the chip, never more than there are tiles (a 200-row Linear of
// the decoder has two tiles: 768 ranges would leave 766 empty slots for pass 2 to scan)
static int wgrad_ranges(int K, int Cin, int Cout, int64_t p_cap) {
    const int blocks = (int)(cdiv(Cout, 128) * cdiv(Cin, 128));
    // workgroups in total: one per CU for the wide layers (every extra range is another Cout x Cin partial block to write and to
    // re-read in pass 2, and counts between whole multiples of the CU count run a half-empty last round: level-3 256 -> 256
    // 604 us with 768, 553 with 512, 533 with 256, 710 with 384), three per CU for the narrow ones (64 -> 64: 124 / 140 / 188 us
    // with 768 / 512 / 256).  SD3D_WGRAD_WGS overrides (tuning).
    static int total_env = -1;
    if (total_env < 0) { const char* e = getenv("SD3D_WGRAD_WGS"); total_env = e ? atoi(e) : 0; }
    // (the 5^3 stem, 288 -> 32 with 125 offsets: 868 / 517 / 659 us with 256 / 512 / 768)
    const int total = total_env > 0 ? total_env : ((int64_t)Cin * Cout > 96 * 96 || (Cin >= 96 && Cout >= 96) ? 256 : (K > 27 ? 512 : 768));
    int r = total / blocks;
    r = r < 32 ? 32 : r;
    const int64_t tiles = p_cap / PT;
    return (int)(tiles < r ? (tiles > 0 ? tiles : 1) : r);
}

#define ST ((hipStream_t)stream)
extern "C" {

int sd3d_pair_out_rows(const int32_t* pos, int K, int64_t M, int64_t p_cap, int32_t* out_idx, void* stream) {
    if (K <= 0 || M <= 0) return SD3D_OK;
    if (hipMemsetAsync(out_idx, 0xFF, (size_t)p_cap * sizeof(int32_t), ST) != hipSuccess) return sd3d_set_error(SD3D_ERR_LAUNCH, "pair_out_rows: memset failed");
    pair_out_rows_kernel<<<(unsigned)cdiv((int64_t)K * M, 256), 256, 0, ST>>>(pos, K, M, out_idx);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

size_t sd3d_pair_wgrad_ws_bytes(int K, int Cin, int Cout) {
    const size_t slots = (size_t)wgrad_ranges(K, Cin, Cout, (int64_t)1 << 40) + K;          // the most any p_cap can ask for
    return align_up(slots * sizeof(int32_t), 256) + slots * ((size_t)Cin * Cout + Cout) * sizeof(float);   // + Cout: SD3D_WGRAD_BIAS
}

int sd3d_pair_wgrad(const float* dy, int ld_dy, const float* x, int ld_x, const int32_t* in_idx, const int32_t* out_idx,
                    const int32_t* tile_k, int64_t p_cap, int K, int Cin, int Cout, float* dw, int flags, void* ws, size_t ws_bytes,
                    void* stream) {
    if (K <= 0 || Cin <= 0 || Cout <= 0) return SD3D_OK;
    if ((Cin & 3) || (Cout & 3) || (ld_dy & 3) || (ld_x & 3)) return sd3d_set_error(SD3D_ERR_ARG, "pair_wgrad: channel counts and row strides must be multiples of 4");
    if (p_cap <= 0 || (p_cap % PT)) return sd3d_set_error(SD3D_ERR_ARG, "pair_wgrad: p_cap must be a positive multiple of 128");
    if (ws_bytes < sd3d_pair_wgrad_ws_bytes(K, Cin, Cout)) return sd3d_set_error(SD3D_ERR_WS, "pair_wgrad: workspace too small");
    const int ranges = wgrad_ranges(K, Cin, Cout, p_cap);
    WGParams p;
    p.dy = dy; p.ld_dy = ld_dy; p.x = x; p.ld_x = ld_x; p.in_idx = in_idx; p.out_idx = out_idx; p.tile_k = tile_k;
    p.n_tiles = (int)(p_cap / PT); p.Cin = Cin; p.Cout = Cout;
    p.n_slots = ranges + K;
    p.bf16 = (flags & SD3D_WGRAD_BF16_OPERANDS) ? 1 : 0;
    p.bias = (flags & SD3D_WGRAD_BIAS) ? 1 : 0;
    if (p.bias && K != 1) return sd3d_set_error(SD3D_ERR_ARG, "pair_wgrad: SD3D_WGRAD_BIAS is for K = 1 (a Linear's identity pair list)");
    const int accumulate = flags & SD3D_WGRAD_ACCUMULATE;
    p.slot_k = (int32_t*)ws;
    p.wpart = (float*)((char*)ws + align_up((size_t)p.n_slots * sizeof(int32_t), 256));
    // block = the channel count split evenly over its ceil(C / 128) blocks (192 channels: 2 x 96, not 128 + 64)
    auto sub = [](int C) { const int n32 = (C + 31) / 32, nb = (n32 + 3) / 4; return (n32 + nb - 1) / nb; };
    const int nco = sub(Cout), nci = sub(Cin);
    const dim3 grid(ranges, (unsigned)cdiv(Cout, nco * 32), (unsigned)cdiv(Cin, nci * 32));
    // waves per workgroup: one 32 x 32 output tile per wave where the block has >= 8 tiles (16 waves share one staged step of the
    // 128 x 128 block: level-3 256 -> 256 530 -> 354 us, level-2 128 -> 128 415 -> 277 us against 4 waves x 4 tiles), 4 waves otherwise.
    // SD3D_WGRAD_NW=4 forces the 4-wave kernel (cross-check).
    static int nw_env = -1;
    if (nw_env < 0) { const char* e = getenv("SD3D_WGRAD_NW"); nw_env = e ? atoi(e) : 0; }
#define WG_LAUNCH(a, b, nw) pair_wgrad_kernel<a, b, nw><<<grid, (nw) * 64, 0, ST>>>(p)
#define WG_CASE(a, b) if (nco == a && nci == b) {                                                   \
        if (nw_env == 4 || a * b < 8) WG_LAUNCH(a, b, 4);                                           \
        else if (a * b == 16) WG_LAUNCH(a, b, 16);                                                  \
        else if (a * b == 12) WG_LAUNCH(a, b, 12);                                                  \
        else if (a * b == 9) WG_LAUNCH(a, b, 9);                                                    \
        else WG_LAUNCH(a, b, 8); }
    WG_CASE(1, 1) WG_CASE(1, 2) WG_CASE(1, 3) WG_CASE(1, 4)
    WG_CASE(2, 1) WG_CASE(2, 2) WG_CASE(2, 3) WG_CASE(2, 4)
    WG_CASE(3, 1) WG_CASE(3, 2) WG_CASE(3, 3) WG_CASE(3, 4)
    WG_CASE(4, 1) WG_CASE(4, 2) WG_CASE(4, 3) WG_CASE(4, 4)
#undef WG_CASE
#undef WG_LAUNCH
    const int64_t elems = (int64_t)Cin * Cout + (p.bias ? Cout : 0);
    int64_t gx = cdiv(elems, 256);                           // ~2048 workgroups over all offsets: a K = 1 Linear gets as many as a 27-offset convolution
    const int64_t cap = 2048 / K > 64 ? 2048 / K : 64;
    gx = gx < cap ? gx : cap;
    pair_wgrad_reduce_kernel<<<dim3((unsigned)gx, K), 256, 0, ST>>>(p.wpart, p.slot_k, p.n_slots, elems, dw, accumulate);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

int sd3d_linear_wgrad(const float* g, int ld_g, const float* x, int ld_x, int64_t M, int Cin, int Cout, float* dw, float* db, int flags,
                      void* stream) {
    if (M <= 0 || Cin <= 0 || Cout <= 0) return SD3D_OK;
    if (!g || !x || !dw) return sd3d_set_error(SD3D_ERR_ARG, "linear_wgrad: null pointer");
    if ((ld_g & 3) || (ld_x & 3) || ld_g < ((Cout + 3) & ~3) || ld_x < ((Cin + 3) & ~3))
        return sd3d_set_error(SD3D_ERR_ARG, "linear_wgrad: row strides must be multiples of 4 floats and cover the columns rounded up to 4");
    const dim3 grid((unsigned)cdiv(Cout, 32), (unsigned)cdiv(Cin, 32));
    linear_wgrad_small_kernel<<<grid, LW_WAVES * 64, 0, ST>>>(g, ld_g, x, ld_x, M, Cin, Cout, dw, db, (flags & SD3D_WGRAD_BF16_OPERANDS) ? 1 : 0,
                                                             (flags & SD3D_WGRAD_ACCUMULATE) ? 1 : 0);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

}  // extern "C"
