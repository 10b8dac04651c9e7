// Pair-major sparse convolution (exact fp32 MFMA).
//
// The output-stationary kernels of gather_gemm.hip walk all K offsets for every 32-row output tile, so
// at rulebook density d they spend 1/d of their MFMA time on zeros (d = 0.06 .. 0.6 on a ScanNet
// scene), and the pair-compacted variants there are limited by LDS capacity (small row tiles -> the
// 32-pair MFMA chunks are half empty and every small tile re-reads all K weight matrices).
//
// Here the rulebook is laid out pair-major ONCE per neighbour table (it is shared by every
// convolution of a U-Net level):
//     for k in 0..K-1:  the pairs (in = nbr[k][r], out = r) with in >= 0, in row order,
//     each k segment padded to a multiple of 128 pairs (pad entries have in = -1)
// so that a 128-pair tile has ONE weight matrix W[k] and every MFMA row is a real pair:
//     pass 1  part[p][:] = X[in_idx[p]][:] . W[k(p)]^T        dense 128 x Cout x Cin tiles, lock-step
//                                                              LDS-shared weights, fp32 MFMA
//     pass 2  out[r][:]  = act(scale * sum_k part[pos[k][r]][:] + shift + res[r][:])   k ascending
// pos[k][r] = position of pair (k, r) in the list or -1.  Pass 2 is pure streaming (HBM-bound);
// the partial products cost 2 x 4 x P x Cout bytes of extra traffic, cheap next to the 157 TFLOP/s
// fp32 matrix rate (>= 16 flop per byte moved at Cin = 96).  The sum order is fixed (full Cin dot
// product per pair, then k ascending) so results are deterministic.
#include "gg_common.h"
#include "../../include/segdino3d_hip.h"
#include <stdlib.h>
#include <atomic>
#include <vector>

static std::atomic<int> g_scenes_in_flight{1};
extern "C" int sd3d_set_scenes_in_flight(int n) { return g_scenes_in_flight.exchange(n < 1 ? 1 : n); }

// Partial products are written once by pass 1 and read once by pass 2, hundreds of megabytes per convolution.  Measured (same box,
// tools/ab_lib.sh; conv ms per scene in the instrumented replay / single-scene latency): non-temporal READS in pass 2 7.49 -> 7.34 ms /
// 12.10 -> 12.00 ms (they stop displacing the activation rows the next pass 1 gathers); non-temporal STORES in pass 1 7.49 -> 7.68 ms /
// 12.1 -> 12.7 ms (one scene in flight: 7.92 -> 8.43 ms) - the plain stores are combined in L2 and part of them is still on chip when
// pass 2 asks.  So: plain stores, non-temporal loads.
// (with SEVERAL scenes in flight the trade flips: non-temporal stores 114.5 -> 116.8 scenes/s - the other scenes' kernels, not this
// convolution's pass 2, are what the written-back lines would displace - so pass 1 takes the store flavour from `p.nt_part`,
// set by the launcher from the scenes-in-flight hint.)
#define PART_STORE4(PTR, V) do { if (p.nt_part) __builtin_nontemporal_store((V), (f32x4*)(PTR)); else *(f32x4*)(PTR) = (V); } while (0)
#define PART_STORE1(PTR, V) do { if (p.nt_part) __builtin_nontemporal_store((V), (float*)(PTR)); else *(float*)(PTR) = (V); } while (0)
#ifndef SD3D_NT_PART_LOAD
#define SD3D_NT_PART_LOAD 1
#endif
#if SD3D_NT_PART_LOAD
#define PART_LOAD4(PTR) __builtin_nontemporal_load((const f32x4*)(PTR))
#else
#define PART_LOAD4(PTR) (*(const f32x4*)(PTR))
#endif
#define PL_ROWS 2048            // rows per workgroup of the list-building kernels
#define PT 128                  // pairs per tile / segment padding
#define PBS_LD 36

__device__ static inline int block_excl_scan_256p(int v, int* total, int* smem4) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    if (lane == 63) smem4[w] = inc;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < w; ++i) base += smem4[i];
    *total = smem4[0] + smem4[1] + smem4[2] + smem4[3];
    __syncthreads();
    return base + inc - v;
}

// ---- list building ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pair_count_kernel(const int32_t* __restrict__ nbr, int64_t M, int nblk,
                                                         int32_t* __restrict__ blk_cnt) {
    __shared__ int sm[4];
    const int k = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    int c = 0;
#pragma unroll
    for (int i = 0; i < PL_ROWS / 256; ++i) {
        const int64_t row = (int64_t)blk * PL_ROWS + i * 256 + tid;
        const bool v = row < M && nbr[(int64_t)k * M + row] >= 0;
        c += __popcll(__ballot(v));
    }
    if ((tid & 63) == 0) sm[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) blk_cnt[(int64_t)k * nblk + blk] = sm[0] + sm[1] + sm[2] + sm[3];
}

// one workgroup per offset k: exclusive scan of its block counts (in place) and its total
__global__ __launch_bounds__(256) void pair_scan_kernel(int32_t* __restrict__ blk_cnt, int nblk, int32_t* __restrict__ totals) {
    __shared__ int sm[4];
    const int k = blockIdx.x, tid = threadIdx.x;
    int running = 0;
    for (int base = 0; base < nblk; base += 256) {
        const int i = base + tid;
        const int v = i < nblk ? blk_cnt[(int64_t)k * nblk + i] : 0;
        int total;
        const int ex = block_excl_scan_256p(v, &total, sm);
        if (i < nblk) blk_cnt[(int64_t)k * nblk + i] = running + ex;
        running += total;
    }
    if (tid == 0) totals[k] = running;
}

__global__ __launch_bounds__(256) void pair_fill_kernel(const int32_t* __restrict__ nbr, int K, int64_t M, int nblk,
                                                        const int32_t* __restrict__ blk_off, const int32_t* __restrict__ totals,
                                                        int64_t p_cap, int32_t* __restrict__ pos, int32_t* __restrict__ in_idx,
                                                        int32_t* __restrict__ tile_k) {
    __shared__ int sm[4];
    __shared__ int wcnt[4];
    const int k = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // segment start of offset k = sum of the padded totals before it
    int s = 0;
    for (int kk = tid; kk < k; kk += 256) s += (totals[kk] + PT - 1) / PT * PT;
    int seg;
    block_excl_scan_256p(s, &seg, sm);
    const int seg_len = (totals[k] + PT - 1) / PT * PT;
    if (blk == 0) {
        for (int t = tid; t < seg_len / PT; t += 256)
            if ((int64_t)(seg / PT + t) * PT < p_cap) tile_k[seg / PT + t] = k;
        if (k == K - 1 && tid == 0) {                          // number of real tiles, after the last slot
            const int64_t end = (int64_t)seg + seg_len;
            tile_k[p_cap / PT] = (int)((end < p_cap ? end : p_cap) / PT);
        }
    }
    int base = seg + blk_off[(int64_t)k * nblk + blk];
    const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll 1
    for (int i = 0; i < PL_ROWS / 256; ++i) {
        const int64_t row = (int64_t)blk * PL_ROWS + i * 256 + tid;
        const int id = row < M ? nbr[(int64_t)k * M + row] : -1;
        const uint64_t bal = __ballot(id >= 0);
        if (lane == 0) wcnt[wv] = __popcll(bal);
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wv; ++w) before += wcnt[w];
        const int all = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
        const int p = base + before + __popcll(bal & lt);
        if (row < M) pos[(int64_t)k * M + row] = (id >= 0 && p < p_cap) ? p : -1;
        if (id >= 0 && p < p_cap) in_idx[p] = id;
        base += all;
    }
}

// ---- the same three steps for SEVERAL tables in one launch set ---------------------------------------------------------------
// A U-Net forward needs the lists of ~14 tables (one 5^3, five 3^3, four down, four up); built one by one that was 14 x (2
// memsets + count + scan + fill) = 70 launches of a few microseconds each on the critical path of every scene.  Here a launch
// covers all tables: a workgroup finds its (table, offset, row block) from the tables' cumulative block counts, and the fill
// kernel writes the -1 padding itself (segment tails, the unused end of in_idx / tile_k) instead of two memsets per table.
#define PL_MAX_TABLES 16
#define RL_ROWS 256             // rows per workgroup of the row-list kernel
struct PLTable {
    const int32_t* nbr; int32_t* pos; int32_t* in_idx; int32_t* tile_k; int32_t* blk_cnt; int32_t* totals;
    int32_t* rlist;                // [M][rl_stride]: per output row {count, list positions of its pairs in offset order} (NULL: not built)
    int32_t* out_idx;              // [p_cap]: output row of every list entry, -1 on padding (NULL: not built)
    int64_t M, p_cap;
    int K, nblk, wg0, k0;          // wg0: first workgroup of this table in the (K * nblk)-flattened grid; k0: first of the K-flattened grid
    int rl_stride, rb0, meta;      // rb0: first workgroup of the table in the row-list grid; meta: tile_k carries two reserved (zero) slots behind the tile count
};
struct PLBatch { int n; PLTable t[PL_MAX_TABLES]; };

__device__ __forceinline__ int pl_find_table(const PLBatch& b, int wg, int by) {           // by: 0 = wg0, 1 = k0, 2 = rb0
    int ti = 0;
    for (int i = 1; i < b.n; ++i) if (wg >= (by == 1 ? b.t[i].k0 : (by == 2 ? b.t[i].rb0 : b.t[i].wg0))) ti = i;
    return ti;
}

__global__ __launch_bounds__(256) void pair_count_batch_kernel(const PLBatch b) {
    __shared__ int sm[4];
    const int ti = pl_find_table(b, blockIdx.x, 0);
    const PLTable& T = b.t[ti];
    const int local = blockIdx.x - T.wg0, k = local / T.nblk, blk = local - k * T.nblk, tid = threadIdx.x;
    int c = 0;
#pragma unroll
    for (int i = 0; i < PL_ROWS / 256; ++i) {
        const int64_t row = (int64_t)blk * PL_ROWS + i * 256 + tid;
        const bool v = row < T.M && T.nbr[(int64_t)k * T.M + row] >= 0;
        c += __popcll(__ballot(v));
    }
    if ((tid & 63) == 0) sm[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) T.blk_cnt[(int64_t)k * T.nblk + blk] = sm[0] + sm[1] + sm[2] + sm[3];
}

__device__ __forceinline__ void pair_scan_batch_body(const PLBatch& b, const int bx) {
    __shared__ int sm[4];
    const int ti = pl_find_table(b, bx, 1);
    const PLTable& T = b.t[ti];
    const int k = bx - T.k0, tid = threadIdx.x;
    int running = 0;
    for (int base = 0; base < T.nblk; base += 256) {
        const int i = base + tid;
        const int v = i < T.nblk ? T.blk_cnt[(int64_t)k * T.nblk + i] : 0;
        int total;
        const int ex = block_excl_scan_256p(v, &total, sm);
        if (i < T.nblk) T.blk_cnt[(int64_t)k * T.nblk + i] = running + ex;
        running += total;
    }
    if (tid == 0) T.totals[k] = running;
}
__global__ __launch_bounds__(256) void pair_scan_batch_kernel(const PLBatch b) { pair_scan_batch_body(b, (int)blockIdx.x); }

__global__ __launch_bounds__(256) void pair_fill_batch_kernel(const PLBatch b) {
    __shared__ int sm[4];
    __shared__ int wcnt[4];
    const int ti = pl_find_table(b, blockIdx.x, 0);
    const PLTable& T = b.t[ti];
    const int local = blockIdx.x - T.wg0, k = local / T.nblk, blk = local - k * T.nblk;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int s = 0;
    for (int kk = tid; kk < k; kk += 256) s += (T.totals[kk] + PT - 1) / PT * PT;
    int seg;
    block_excl_scan_256p(s, &seg, sm);
    const int tot_k = T.totals[k];
    const int seg_len = (tot_k + PT - 1) / PT * PT;
    if (blk == 0) {
        for (int t = tid; t < seg_len / PT; t += 256)
            if ((int64_t)(seg / PT + t) * PT < T.p_cap) T.tile_k[seg / PT + t] = k;
        for (int e = tot_k + tid; e < seg_len; e += 256)       // the segment's padding
            if ((int64_t)seg + e < T.p_cap) {
                T.in_idx[seg + e] = -1;
                if (T.out_idx) T.out_idx[seg + e] = -1;
            }
        if (k == T.K - 1 && tid == 0) {                        // number of real tiles, after the last slot
            const int64_t end = (int64_t)seg + seg_len < T.p_cap ? (int64_t)seg + seg_len : T.p_cap;
            T.tile_k[T.p_cap / PT] = (int)(end / PT);
            if (T.meta & 1) { T.tile_k[T.p_cap / PT + 1] = 0; T.tile_k[T.p_cap / PT + 2] = 0; }
        }
    }
    if (k == T.K - 1 && !(T.meta & 2)) {
        // past the last segment: unused capacity reads as "no pair".  Every row block of the last offset fills its slice
        // (with worst-case sized lists, SD3D_EXACT_PAIRS=0, the tail is tens of MB: one workgroup would sit on the critical path;
        //  meta & 2: the caller's consumers walk tile_k[p_cap / 128] tiles and nothing else - the tail stays unwritten)
        const int64_t end = (int64_t)seg + seg_len < T.p_cap ? (int64_t)seg + seg_len : T.p_cap;
        for (int64_t e = end + (int64_t)blk * 256 + tid; e < T.p_cap; e += (int64_t)T.nblk * 256) {
            T.in_idx[e] = -1;
            if (T.out_idx) T.out_idx[e] = -1;
        }
        for (int64_t t = end / PT + (int64_t)blk * 256 + tid; t < T.p_cap / PT; t += (int64_t)T.nblk * 256) T.tile_k[t] = -1;
    }
    // Round 5: a wave owns 512 consecutive rows of the block and requests all eight of its 64-row slices at once; the waves meet ONCE
    // (their totals through LDS).  Before: eight rounds of {load, ballot, two barriers} per workgroup - 84 us for the stem's 5^3 table,
    // whose 12.5 M slots it walks at 1.2 TB/s.  Same positions (rows ascending within an offset).
    int base = seg + T.blk_cnt[(int64_t)k * T.nblk + blk];
    const uint64_t lt = (1ull << lane) - 1ull;
    constexpr int NS = PL_ROWS / 256;
    const int64_t row0 = (int64_t)blk * PL_ROWS + (int64_t)wv * (PL_ROWS / 4) + lane;
    int id[NS];
    uint64_t bal[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int64_t row = row0 + i * 64;
        id[i] = T.nbr[(int64_t)k * T.M + (row < T.M ? row : 0)];
        if (row >= T.M) id[i] = -1;
    }
    int wtot = 0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        bal[i] = __ballot(id[i] >= 0);
        wtot += __popcll(bal[i]);
    }
    if (lane == 0) wcnt[wv] = wtot;
    __syncthreads();
    for (int w = 0; w < wv; ++w) base += wcnt[w];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int64_t row = row0 + i * 64;
        const int p = base + __popcll(bal[i] & lt);
        if (T.pos && row < T.M) T.pos[(int64_t)k * T.M + row] = (id[i] >= 0 && p < T.p_cap) ? p : -1;
        if (id[i] >= 0 && p < T.p_cap) {
            T.in_idx[p] = id[i];
            if (T.out_idx) T.out_idx[p] = (int32_t)row;
        }
        base += __popcll(bal[i]);
    }
}

// Per output row: how many partial products pass 2 has to add up and where they are - {count, positions in offset order} in
// rl_stride ints per row (the count and the first three positions arrive with ONE 16-byte load; a level-0 row has ~2 partners, so
// walking all K slots of pos[k][r] was 27 loads for 3 hits).  One thread per row; pos is read offset-major (coalesced over rows).
__global__ __launch_bounds__(RL_ROWS) void pair_rowlist_batch_kernel(const PLBatch b) {
    const int ti = pl_find_table(b, blockIdx.x, 2);
    const PLTable& T = b.t[ti];
    if (!T.rlist) return;
    const int64_t row = (int64_t)(blockIdx.x - T.rb0) * RL_ROWS + threadIdx.x;
    if (row >= T.M) return;
    int32_t* rl = T.rlist + row * T.rl_stride;
    int cnt = 0;
    for (int k0 = 0; k0 < T.K; k0 += 16) {                    // sixteen columns requested together (the stem's table has 125)
        int v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            v[u] = T.pos[(int64_t)(k0 + u < T.K ? k0 + u : T.K - 1) * T.M + row];
            if (k0 + u >= T.K) v[u] = -1;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (v[u] >= 0) rl[1 + cnt++] = v[u];
    }
    rl[0] = cnt;
}

// ---- plain lists without a position table (round 5) -------------------------------------------------------------------------------
// An evaluation forward never reads pos [K, M]: pass 2 walks the per-row lists.  For the stem's 5^3 table pos is 69 MB written by the
// fill launch and read back by the row-list launch (and the lists' unused capacity, sized for the worst case when one scene is in
// flight, another 55 MB of -1).  When the caller asks for row lists but no pos (pos == NULL), the builder takes the row-block form
// of the chained builder: a workgroup owns 256 rows and walks all K offsets of them, the fill launch writes the rows' lists itself.
// Same in_idx / tile_k / rlist as the (offset, row block) form, entry for entry.
#define PR_MAX_K 128
template <class F>
__device__ __forceinline__ void pr_walk(const PLTable& T, int64_t row, bool live, F&& f) {
    const int64_t rc = live ? row : 0;
    for (int k0 = 0; k0 < T.K; k0 += 8) {
        int id[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) id[u] = T.nbr[(int64_t)(k0 + u < T.K ? k0 + u : T.K - 1) * T.M + rc];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + u < T.K) f(k0 + u, live ? id[u] : -1);
    }
}
__device__ __forceinline__ void pair_count_rows_body(const PLBatch& b, const int bx) {
    __shared__ int wc[4][PR_MAX_K];
    const int ti = pl_find_table(b, bx, 0);
    const PLTable& T = b.t[ti];
    const int blk = bx - T.wg0, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t row = (int64_t)blk * 256 + tid;
    pr_walk(T, row, row < T.M, [&](int k, int id) {
        const int c = __popcll(__ballot(id >= 0));
        if (lane == 0) wc[wv][k] = c;
    });
    __syncthreads();
    if (tid < T.K) T.blk_cnt[(int64_t)tid * T.nblk + blk] = wc[0][tid] + wc[1][tid] + wc[2][tid] + wc[3][tid];
}
__global__ __launch_bounds__(256) void pair_count_rows_kernel(const PLBatch b) { pair_count_rows_body(b, (int)blockIdx.x); }
__device__ __forceinline__ void pair_fill_rows_body(const PLBatch& b, const int bx) {
    __shared__ int sm[4];
    __shared__ int wb[4][PR_MAX_K];
    __shared__ int seg[PR_MAX_K + 1], tot[PR_MAX_K];
    const int ti = pl_find_table(b, bx, 0);
    const PLTable& T = b.t[ti];
    const int blk = bx - T.wg0, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t row = (int64_t)blk * 256 + tid;
    const bool live = row < T.M;
    {   // first list position of every offset's segment = the padded totals before it
        const int t = tid < T.K ? T.totals[tid] : 0;
        int end;
        const int ex = block_excl_scan_256p((t + PT - 1) / PT * PT, &end, sm);
        if (tid < T.K) { seg[tid] = ex; tot[tid] = t; }
        if (tid == 0) seg[T.K] = end;
    }
    pr_walk(T, row, live, [&](int k, int id) {
        const int c = __popcll(__ballot(id >= 0));
        if (lane == 0) wb[wv][k] = c;
    });
    __syncthreads();
    if (tid < T.K) {
        int run = T.blk_cnt[(int64_t)tid * T.nblk + blk];
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int c = wb[w][tid]; wb[w][tid] = run; run += c; }
    }
    __syncthreads();
    for (int k = blk; k < T.K; k += T.nblk) {                      // tile headers and the -1 padding of the segments, dealt out to the row blocks
        const int seg_len = (tot[k] + PT - 1) / PT * PT;
        for (int t = tid; t < seg_len / PT; t += 256)
            if ((int64_t)(seg[k] / PT + t) * PT < T.p_cap) T.tile_k[seg[k] / PT + t] = k;
        for (int e = tot[k] + tid; e < seg_len; e += 256)
            if ((int64_t)seg[k] + e < T.p_cap) {
                T.in_idx[seg[k] + e] = -1;
                if (T.out_idx) T.out_idx[seg[k] + e] = -1;
            }
    }
    {
        const int64_t end = (int64_t)seg[T.K] < T.p_cap ? (int64_t)seg[T.K] : T.p_cap;
        if (blk == 0 && tid == 0) {
            T.tile_k[T.p_cap / PT] = (int)(end / PT);
            if (T.meta & 1) { T.tile_k[T.p_cap / PT + 1] = 0; T.tile_k[T.p_cap / PT + 2] = 0; }
        }
        if (!(T.meta & 2)) {
            for (int64_t e = end + (int64_t)blk * 256 + tid; e < T.p_cap; e += (int64_t)T.nblk * 256) {
                T.in_idx[e] = -1;
                if (T.out_idx) T.out_idx[e] = -1;
            }
            for (int64_t t = end / PT + (int64_t)blk * 256 + tid; t < T.p_cap / PT; t += (int64_t)T.nblk * 256) T.tile_k[t] = -1;
        }
    }
    const uint64_t lt = (1ull << lane) - 1ull;
    int32_t* rl = T.rlist ? T.rlist + (live ? row : 0) * T.rl_stride : nullptr;
    int cnt = 0;
    pr_walk(T, row, live, [&](int k, int id) {
        const uint64_t bal = __ballot(id >= 0);
        if (id >= 0) {
            const int p = seg[k] + wb[wv][k] + __popcll(bal & lt);
            if (p < T.p_cap) {
                T.in_idx[p] = id;
                if (T.out_idx) T.out_idx[p] = (int32_t)row;
                if (rl) rl[1 + cnt] = p;
            }
            ++cnt;
        }
    });
    if (live && rl) rl[0] = cnt;
}
__global__ __launch_bounds__(256) void pair_fill_rows_kernel(const PLBatch b) { pair_fill_rows_body(b, (int)blockIdx.x); }

// ---- chained lists: mirror offsets and the centre share ONE partial product ----------------------------------------------------
// A stride-1 table of a voxel set onto itself with an odd kernel (3^3) enumerates its offsets symmetrically: off[K-1-k] == -off[k],
// centre = K / 2.  On a surface a row that has a neighbour at +d mostly has one at -d too (measured on the benchmark scene: 48 % /
// 76 % / 75 % of the non-centre entries of levels 1 / 2 / 3 come in such mirror pairs, tools/mirror_pairs.py), and EVERY row has
// its centre.  The pair-major convolution pays 2 x 4 x Cout bytes of HBM traffic per list entry for the partial product (written
// by pass 1, read by pass 2).  Here the entries of one output row that belong to the same mirror group g = {k = g, K-1-g}, plus
// the centre in the row's first non-empty group, share ONE partial product: pass 1 keeps accumulating across up to three
// consecutive "sub-tiles" (same 128 rows, sources centre -> +d -> -d, each with its own gather indices and its own W[k]) and
// stores once (tile_k carries PG_CHAIN on all but the last sub-tile).  Rows are sorted into SEGMENTS by (group, pattern) so that
// every sub-tile row is a real product: no MFMA row is wasted, the flops are exactly the rulebook's.  Partial rows at 150 k points:
// level 0 -27 %, level 1 -35 %, levels 2-4 -40...-44 %.  Pass 2 is unchanged (per-row lists of partial positions, ascending).
//   pattern: 0 {a} 1 {b} 2 {a,b} 3 {c,a} 4 {c,b} 5 {c,a,b}   (a = nbr[g][r], b = nbr[K-1-g][r], c = the centre: idx r)
//   segment g * 6 + pattern for g < G = K / 2;  segment 6 G = rows without any neighbour (centre alone)
//   entry i of a segment with n sources: sub-tile (i / 128) * n + s of the segment, slot i % 128
#define PG_CHAIN 0x40000000
#define PG_KMASK 0x3FFFFFFF
#define CH_NPAT 6
#define CH_ROWS 256              // rows per workgroup of the chained builder (one row per thread)
#define CH_GB 7                  // mirror groups whose two columns are requested together
#define CH_MAX_G 62              // K <= 125
#define CH_MAX_SEG (CH_MAX_G * CH_NPAT + 1)
// Round 5 form of the builder: a workgroup owns 256 ROWS and walks all mirror groups of them (before: one workgroup per (group, 2048
// rows), five launches - first group per row, count, scan, fill, per-row lists - with the group positions [G + 1][M] and the first
// groups [M] going through memory in between: 0.49 ms of kernel time per scene).  A row's first non-empty group is known on the fly
// when the groups are walked in ascending order, so the count launch needs no first-group pass, and the fill launch knows ALL partial
// positions of its rows: it writes the per-row lists itself.  Three launches (count, scan, fill); the table is read twice instead of
// ~4.5 times and nothing but the per-(segment, row block) counters sits between them.  The lists are the same, entry for entry.
struct CHTable {
    const int32_t* nbr; int32_t* in_idx; int32_t* tile_k; int32_t* blk_cnt; int32_t* totals; int32_t* rlist;
    int64_t M, p_cap;
    int K, G, nblk, rl_stride;
    int wg0, sg0;                  // first workgroup of the table in the row-block grid / the segment grid
    int lean;                      // 1: the unused capacity behind the last segment stays unwritten (sd3d_pair_table_desc.meta & 2)
};
struct CHBatch { int n; CHTable t[PL_MAX_TABLES]; };
__device__ __forceinline__ int ch_find(const CHBatch& b, int wg, int by) {
    int ti = 0;
    for (int i = 1; i < b.n; ++i) if (wg >= (by == 0 ? b.t[i].wg0 : b.t[i].sg0)) ti = i;
    return ti;
}
__device__ __forceinline__ int ch_nsrc(int pat) { return pat < 2 ? 1 : (pat < 5 ? 2 : 3); }

// The mirror groups of one row in ascending order: f(g, pattern, ia, ib) for EVERY group (pattern -1: no neighbour in it; the calls are
// wave-uniform, f may ballot).  Returns whether the row has any neighbour at all.  The two columns of CH_GB groups are requested
// together (coalesced over the rows, independent).
template <class F>
__device__ __forceinline__ bool ch_walk(const CHTable& T, int64_t row, bool live, F&& f) {
    bool seen = false;
    const int64_t rc = live ? row : 0;
    for (int g0 = 0; g0 < T.G; g0 += CH_GB) {
        int a[CH_GB], b[CH_GB];
#pragma unroll
        for (int u = 0; u < CH_GB; ++u) {
            const int g = g0 + u < T.G ? g0 + u : T.G - 1;
            a[u] = T.nbr[(int64_t)g * T.M + rc];
            b[u] = T.nbr[(int64_t)(T.K - 1 - g) * T.M + rc];
        }
#pragma unroll
        for (int u = 0; u < CH_GB; ++u) {
            if (g0 + u < T.G) {
                const int ia = live ? a[u] : -1, ib = live ? b[u] : -1;
                int pat = -1;
                if (ia >= 0 || ib >= 0) {
                    pat = (ia >= 0 ? (ib >= 0 ? 2 : 0) : 1) + (seen ? 0 : 3);     // the centre rides in the row's first non-empty group
                    seen = true;
                }
                f(g0 + u, pat, ia, ib);
            }
        }
    }
    return seen;
}

__device__ __forceinline__ void chain_count_body(const CHBatch& b, const int bx) {
    __shared__ int wc[4][CH_MAX_SEG];
    const int ti = ch_find(b, bx, 0);
    const CHTable& T = b.t[ti];
    const int blk = bx - T.wg0, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t row = (int64_t)blk * CH_ROWS + tid;
    const bool live = row < T.M;
    const bool seen = ch_walk(T, row, live, [&](int g, int pat, int, int) {
#pragma unroll
        for (int q = 0; q < CH_NPAT; ++q) {
            const int c = __popcll(__ballot(pat == q));
            if (lane == 0) wc[wv][g * CH_NPAT + q] = c;
        }
    });
    {
        const int c = __popcll(__ballot(live && !seen));        // the centre-only segment
        if (lane == 0) wc[wv][T.G * CH_NPAT] = c;
    }
    __syncthreads();
    const int nseg = T.G * CH_NPAT + 1;
    for (int sg = tid; sg < nseg; sg += 256) T.blk_cnt[(int64_t)sg * T.nblk + blk] = wc[0][sg] + wc[1][sg] + wc[2][sg] + wc[3][sg];
}
__global__ __launch_bounds__(256) void chain_count_kernel(const CHBatch b) { chain_count_body(b, (int)blockIdx.x); }
__device__ __forceinline__ void chain_scan_body(const CHBatch& b, const int bx) {
    __shared__ int sm[4];
    const int ti = ch_find(b, bx, 1);
    const CHTable& T = b.t[ti];
    const int seg = bx - T.sg0, tid = threadIdx.x;
    int running = 0;
    for (int base = 0; base < T.nblk; base += 256) {
        const int i = base + tid;
        const int v = i < T.nblk ? T.blk_cnt[(int64_t)seg * T.nblk + i] : 0;
        int total;
        const int ex = block_excl_scan_256p(v, &total, sm);
        if (i < T.nblk) T.blk_cnt[(int64_t)seg * T.nblk + i] = running + ex;
        running += total;
    }
    if (tid == 0) T.totals[seg] = running;
}
__global__ __launch_bounds__(256) void chain_scan_kernel(const CHBatch b) { chain_scan_body(b, (int)blockIdx.x); }
__device__ __forceinline__ void chain_fill_body(const CHBatch& b, const int bx) {
    __shared__ int sm[4];
    __shared__ int wb[4][CH_MAX_SEG];          // per wave: entries of the segment in this wave, then its first position in the segment
    __shared__ int seg_tile[CH_MAX_SEG], seg_tot[CH_MAX_SEG];
    const int ti = ch_find(b, bx, 0);
    const CHTable& T = b.t[ti];
    const int blk = bx - T.wg0, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nseg = T.G * CH_NPAT + 1, centre = T.K / 2;
    const int64_t row = (int64_t)blk * CH_ROWS + tid;
    const bool live = row < T.M;
    // first tile of every segment = sum over the segments before it of ceil(total / 128) * sources
    int end_tile = 0;
    for (int base = 0; base < nseg; base += 256) {
        const int sg = base + tid;
        const int tot = sg < nseg ? T.totals[sg] : 0;
        const int v = (tot + PT - 1) / PT * (sg == nseg - 1 ? 1 : ch_nsrc(sg % CH_NPAT));
        int total;
        const int ex = block_excl_scan_256p(v, &total, sm);
        if (sg < nseg) { seg_tile[sg] = end_tile + ex; seg_tot[sg] = tot; }
        end_tile += total;
    }
    // this wave's entries per segment, then their first position: the block's start in the segment + the waves before this one
    const bool any = ch_walk(T, row, live, [&](int g, int pat, int, int) {
#pragma unroll
        for (int q = 0; q < CH_NPAT; ++q) {
            const int c = __popcll(__ballot(pat == q));
            if (lane == 0) wb[wv][g * CH_NPAT + q] = c;
        }
    });
    {
        const int c = __popcll(__ballot(live && !any));
        if (lane == 0) wb[wv][T.G * CH_NPAT] = c;
    }
    __syncthreads();
    for (int sg = tid; sg < nseg; sg += 256) {
        int run = T.blk_cnt[(int64_t)sg * T.nblk + blk];
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int c = wb[w][sg]; wb[w][sg] = run; run += c; }
    }
    __syncthreads();
    // tile headers and the -1 padding of the segments, dealt out to the row blocks
    for (int sg = blk; sg < nseg; sg += T.nblk) {
        const int g = sg / CH_NPAT, q = sg - g * CH_NPAT;
        const int ns = g == T.G ? 1 : ch_nsrc(q), nt = (seg_tot[sg] + PT - 1) / PT;
        for (int e = tid; e < nt * ns; e += 256) {
            const int sub = e % ns;
            int k;                                      // source order: centre, +d (k = g), -d (k = K - 1 - g)
            if (g == T.G) k = centre;
            else if (q == 0) k = g;
            else if (q == 1) k = T.K - 1 - g;
            else if (q == 2) k = sub == 0 ? g : T.K - 1 - g;
            else if (q == 3) k = sub == 0 ? centre : g;
            else if (q == 4) k = sub == 0 ? centre : T.K - 1 - g;
            else k = sub == 0 ? centre : (sub == 1 ? g : T.K - 1 - g);
            if ((int64_t)(seg_tile[sg] + e) * PT < T.p_cap) T.tile_k[seg_tile[sg] + e] = k | (sub < ns - 1 ? PG_CHAIN : 0);
        }
        if (nt > 0) {
            const int fill0 = seg_tot[sg] - (nt - 1) * PT;       // real entries of the last tile row-block
            for (int e = tid; e < (PT - fill0) * ns; e += 256) {
                const int sub = e / (PT - fill0), off = fill0 + e % (PT - fill0);
                const int64_t pp = (int64_t)(seg_tile[sg] + (nt - 1) * ns + sub) * PT + off;
                if (pp < T.p_cap) T.in_idx[pp] = -1;
            }
        }
    }
    {
        // past the last segment: unused capacity reads as "no pair"; the number of real tiles after the last slot
        const int64_t cap_tiles = T.p_cap / PT;
        if (blk == 0 && tid == 0) { T.tile_k[cap_tiles] = (int)(end_tile < cap_tiles ? end_tile : cap_tiles); T.tile_k[cap_tiles + 1] = 0; T.tile_k[cap_tiles + 2] = 0; }
        if (!T.lean) {
            for (int64_t e = (int64_t)end_tile * PT + (int64_t)blk * 256 + tid; e < T.p_cap; e += (int64_t)T.nblk * 256) T.in_idx[e] = -1;
            for (int64_t tt = end_tile + (int64_t)blk * 256 + tid; tt < cap_tiles; tt += (int64_t)T.nblk * 256) T.tile_k[tt] = -1;
        }
    }
    // the entries of this row, group by group, and its list of partial positions (the last sub-tile of each of its chains, ascending)
    const uint64_t lt = (1ull << lane) - 1ull;
    int32_t* rl = T.rlist + (live ? row : 0) * T.rl_stride;
    int cnt = 0;
    ch_walk(T, row, live, [&](int g, int pat, int ia, int ib) {
        uint64_t mine = 0;
#pragma unroll
        for (int q = 0; q < CH_NPAT; ++q) {
            const uint64_t bal = __ballot(pat == q);
            if (pat == q) mine = bal;
        }
        if (pat >= 0) {
            const int sg = g * CH_NPAT + pat, ns = ch_nsrc(pat);
            const int my = wb[wv][sg] + __popcll(mine & lt);
            const int64_t tile = seg_tile[sg] + (int64_t)(my >> 7) * ns;
            const int off = my & (PT - 1);
            if ((tile + ns) * PT <= T.p_cap) {
                int64_t pp = tile * PT + off;
                if (pat >= 3) { T.in_idx[pp] = (int)row; pp += PT; }      // the centre: in = out = r
                if (ia >= 0) { T.in_idx[pp] = ia; pp += PT; }
                if (ib >= 0) { T.in_idx[pp] = ib; pp += PT; }
                rl[1 + cnt++] = (int32_t)(pp - PT);
            }
        }
    });
    {
        const bool lone = live && !any;
        const uint64_t bal = __ballot(lone);
        if (lone) {
            const int sg = T.G * CH_NPAT;
            const int my = wb[wv][sg] + __popcll(bal & lt);
            const int64_t pp = (int64_t)(seg_tile[sg] + (my >> 7)) * PT + (my & (PT - 1));
            if (pp < T.p_cap) { T.in_idx[pp] = (int)row; rl[1 + cnt++] = (int32_t)pp; }
        }
    }
    if (live) rl[0] = cnt;
}
__global__ __launch_bounds__(256) void chain_fill_kernel(const CHBatch b) { chain_fill_body(b, (int)blockIdx.x); }
// The chained tables and the position-free plain tables of a scene in the SAME three launches (count, scan, fill): two independent
// chains of three dependent launches on one stream were six launch latencies in front of the first convolution.  A workgroup below
// `n_chain` runs the chained builder's body, the others the row-block builder's - the same code on the same data.
__global__ __launch_bounds__(256) void lists_count_kernel(const CHBatch cb, const PLBatch rb, const int n_chain) {
    if ((int)blockIdx.x < n_chain) chain_count_body(cb, (int)blockIdx.x);
    else pair_count_rows_body(rb, (int)blockIdx.x - n_chain);
}
__global__ __launch_bounds__(256) void lists_scan_kernel(const CHBatch cb, const PLBatch rb, const int n_chain) {
    if ((int)blockIdx.x < n_chain) chain_scan_body(cb, (int)blockIdx.x);
    else pair_scan_batch_body(rb, (int)blockIdx.x - n_chain);
}
__global__ __launch_bounds__(256) void lists_fill_kernel(const CHBatch cb, const PLBatch rb, const int n_chain) {
    if ((int)blockIdx.x < n_chain) chain_fill_body(cb, (int)blockIdx.x);
    else pair_fill_rows_body(rb, (int)blockIdx.x - n_chain);
}
size_t chain_lists_ws_bytes(int K, int64_t M) {
    const int64_t nblk = cdiv(M, CH_ROWS);
    const int64_t nseg = (int64_t)(K / 2) * CH_NPAT + 1;
    return (size_t)(nseg * nblk + nseg) * sizeof(int32_t) + 256;
}

// ---- pass 1: dense tiles over the pair list --------------------------------------------------
struct PGParams {
    const float* in0; int ld0; int C0;
    const float* in1; int ld1;
    const int32_t* in_idx;                    // [n_tiles * 128]
    const int32_t* tile_k;                    // [n_tiles], -1 = past the end
    const float* wt;                          // [K][Cout][Cin]
    int Cin, Cout;
    float* part;                              // [n_tiles * 128][Cout]
    int n_tiles;                              // capacity; the real count is tile_k[n_tiles]
    // direct epilogue (out_idx != NULL): every output row has exactly ONE pair (transposed k2s2 convolution), so the tile's
    // products ARE the output rows: out[out_idx[p]] = act(scale * acc + shift + res) straight from pass 1, no partial products
    const int32_t* out_idx;
    const float* scale; const float* shift; const float* res; int ld_res;
    float* out; int ld_out; int act;
    int nt_part;                              // 1: non-temporal partial-product stores (several scenes in flight; see PART_STORE4)
    int chained;                              // 1: tile_k carries PG_CHAIN flags (chained lists: a tile's products add onto the next tile's)
    int64_t dense_rows;                       // > 0: no lists at all - pair p is (in = out = row p) of a dense [rows, Cin] x W[0]^T product (direct epilogue)
    unsigned long long* pool_ctr;             // lock-step kernel: one {epoch, next unit} word per column group of this launch's slot, or NULL (static partition only)
    unsigned int pool_epoch;                  // this launch's epoch (never 0)
    int k_flip;                               // >= 0: the weight matrix of offset k is W[k_flip - k] (mirrored offset order: SD3D_PAIR_MIRROR_W), -1: W[k]
#ifdef PG_ABLATE
    int dbg;                                  // timing-only switches of the lab build (tools/ablate.sh): 1 no epilogue stores, 2 gathers from row 0, 4 weights from chunk 0, 8 no step barrier
#endif
};
#ifdef PG_ABLATE
#define PG_DBG(p, bit) ((p).dbg & (bit))
static int pg_dbg_env() { static const int v = [] { const char* e = getenv("SD3D_PG_ABLATE"); return e ? atoi(e) : 0; }(); return v; }
#else
#define PG_DBG(p, bit) 0
#endif

// The shared tail of a lock-step launch.  A static partition gives every workgroup the same number of tiles, not the same time: a chained
// table's ranges are snapped to chain boundaries (+-2 tiles of the 6 - 10 a workgroup has at levels 0 - 2), the workgroups of a CU are
// served oldest first, the tiles' gathers hit or miss L2 - the counters of profiles/r05_pmc_pair_gemm.md put the MEAN wave lifetime at
// 68 - 80 % of a launch while the pipe is 73 - 82 % busy during a wave's lifetime: a fifth of every launch is waiting for the last
// workgroups.  So only the first 13/16 of the tile list is dealt out statically; the rest is a pool of small units (>= 6 steps each) that
// the workgroups draw from a counter as they run dry.  Which workgroup multiplies a tile never changes what is stored for it: the
// results are the same bits.
// Round 6: the counters carry the EPOCH of the launch that owns them.  A counter is one 64-bit word {epoch : 32 | next unit : 32} per
// (launch slot, column group); the host hands every pooled launch a ticket t -> slot t % PG_POOL_SLOTS, epoch t / PG_POOL_SLOTS + 1 (never
// 0: zero-initialised memory is "no launch yet").  A workgroup that draws a unit and finds another epoch in the word starts the word over
// at {its epoch, 1} and takes unit 0 - so a launch assumes NOTHING about what the slot held before it: a launch that died half-way (device
// fault, a killed debugger, hipDeviceReset in a long-lived server) cannot make a later launch skip units, no launch has to leave the slot
// clean, and no "workgroups done" counter is needed (rounds 5's form zeroed the slot from the launch's last workgroup and trusted it to be
// zero on entry: VERDICT r5 "the shared tail has no in-suite guard").  What is still assumed, and checked by sd3d_pair_pool_check(): the ring
// is long enough that the previous user of a slot (16384 pooled launches ago, ~150 forwards) has finished, and nobody else writes the ring.
// A launch that is being captured into a HIP graph gets NO pool (a replay would meet its own exhausted epoch): static partition.
#define PG_POOL_SLOTS 16384
#define PG_POOL_WORDS 4                                         // per slot: one {epoch, next} word per column group (<= 4)
__device__ unsigned long long g_pool_ctr[PG_POOL_SLOTS * PG_POOL_WORDS];
__device__ unsigned int g_pool_bad;                             // sd3d_pair_pool_check's result
__device__ unsigned int g_pool_bad_seen;                        // workgroups that met a poisoned word since the last check (they fell back to a static split)

// A workgroup CLAIMS its launch's word when it starts - an atomic max with {epoch, 0}, no return value waited for: epochs of a slot only
// grow, so whatever an older launch left (finished or dead) is replaced by the first claim and every later claim is a no-op - and DRAWS
// with one atomic add when it runs dry: its own claim precedes its draw in program order on the same address, so a draw sees this launch's
// epoch or - only after a foreign write - a newer one.  (Rejected forms: a compare-and-swap loop per draw cost 1 - 2 ms per launch, the
// ~500 workgroups of a launch run dry together and retry against each other; claim-on-first-draw tripled the atomics at that moment:
// +25 us on the 400 us layers.)  A newer epoch cannot come from any launch: POOL_POISONED tells the caller to split the pool statically,
// and the event is counted in g_pool_bad_seen for sd3d_pair_pool_check.
#define POOL_POISONED 0xFFFFFFFFu
__device__ __forceinline__ void pool_claim(unsigned long long* ctr, unsigned int epoch) {
    (void)__hip_atomic_fetch_max(ctr, (unsigned long long)epoch << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned int pool_draw(unsigned long long* ctr, unsigned int epoch) {
    const unsigned long long old = __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (unsigned int)(old >> 32) == epoch ? (unsigned int)old : POOL_POISONED;
}
// slots in a state no sequence of finished launches can leave behind: an epoch the host has not handed out yet for that slot
// (`tickets` pooled launches so far on this device's ring: slot s was last owned by epoch ceil((tickets - s) / PG_POOL_SLOTS))
__global__ __launch_bounds__(256) void pool_check_kernel(unsigned long long tickets) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= PG_POOL_SLOTS) return;
    const unsigned long long last = tickets > (unsigned long long)s ? (tickets - 1 - s) / PG_POOL_SLOTS + 1 : 0;
    int bad = 0;
#pragma unroll
    for (int w = 0; w < PG_POOL_WORDS; ++w) {
        const unsigned long long v = g_pool_ctr[(size_t)s * PG_POOL_WORDS + w];
        if ((v >> 32) > last || ((v >> 32) == 0 && v != 0)) ++bad;    // (epoch 0 = never used: the whole word is zero)
    }
    if (bad) atomicAdd(&g_pool_bad, (unsigned int)bad);
}


// scale * x + shift as ONE fused multiply-add in every epilogue of this file, so that the paths agree bit for bit
__device__ __forceinline__ float pg_affine(float x, const float* scale, const float* shift, int n) {
    return __builtin_fmaf(x, scale ? scale[n] : 1.f, shift ? shift[n] : 0.f);
}
// (GELU / sigmoid never follow a sparse convolution in the shipped networks: kept out of line so that the 64 epilogue sites of a
//  kernel do not each carry an inlined erff / expf)
__device__ __attribute__((noinline)) float pg_act_slow(float t, int act) {
    return act == 2 ? 0.5f * t * (1.f + erff(t * 0.70710678118654752440f)) : 1.f / (1.f + expf(-t));
}
__device__ __forceinline__ float pg_act(float t, int act) {
    if (act == 1) t = fmaxf(t, 0.f);
    else if (act >= 2) t = pg_act_slow(t, act);
    return t;
}

// ---- pass 1, lock-step variant ------------------------------------------------------------------------------------------
// A workgroup walks its range of consecutive 128-pair tiles as ONE flat stream of (tile, 32-channel chunk) steps: the weight chunk
// of step s + 1 is staged global -> registers -> LDS while step s runs on the matrix cores, the gathered activation rows are
// requested TWO steps ahead (three register stages, renamed by unrolling), across tile boundaries.  The four waves share the weight
// chunk (lock-step, one barrier per step); each owns 32 pairs x (32 NT) output columns.
// Round 5 form.  The ISA of the round 1-4 form (hipcc 7.2, -save-temps) showed two full memory round trips outside the matrix work:
// every tile switch loaded the next gather indices and the next tile's offset from global memory and waited for them on the spot
// (`s_waitcnt vmcnt(0)`: the offset must become a scalar, the index register is copied) with all four waves parked behind the
// barrier - every third step of a 96-channel layer, EVERY step of a 32-channel one - and the step at the loop header waited
// `vmcnt(0)` right after it had ISSUED the next weight chunk's loads.  Here the range's metadata (gather rows, offsets) sits in LDS,
// a piece of PG_PIECE tiles at a time, so the steady state has no global load whose value is needed at once, and a step issues its
// requests (next weight chunk, the rows of step s + 2) BEHIND its first group of MFMAs, so whatever the compiler waits for at the top
// of a step was requested a whole step earlier.  Same tiles, same MFMA order per accumulator, same stores: the partial products are
// bit-identical to the old form's.  Measured (tools/r05_pair_ab.sh, profiles/EXPERIMENTS.md): -4 % on the 96-column layers of
// levels 0-1, -1...-3 % elsewhere - the other workgroup of the CU was already covering most of those waits.
#define PG_PIECE 24
// The step is written for few vector instructions (376 instead of ~2400 per 204 MFMAs at NT = 4; it bought ~1 %, which is how the round
// learned that instruction issue is not what bounds this kernel - profiles/EXPERIMENTS.md): the MFMA is issued transposed (A = weights, B = gathered rows: a lane owns one pair row and its accumulator groups are
// four consecutive columns - 4 NT `dwordx4` stores from ONE 64-bit address per tile instead of 16 NT dword stores with an index product
// each), the weight requests are a wave-uniform 64-bit base (scalar unit) + a per-thread 32-bit offset fixed for the launch, a gather
// request is one 64-bit multiply-add per step.
// MODE 0: partial products (pass 2 follows); 1: direct epilogue over out_idx (transposed convolutions); 2: dense rows (the identity as the
// rulebook: the U-Net's 1x1 convolutions, launch_pair_dense) - its own kernel symbols (`pair_dense_kernel_*`), so that a kernel trace lists
// the sparse convolutions the roofline counts apart from the 1x1s
template <int NT, int MODE>
__device__ __forceinline__ void pair_gemm_body(const PGParams& p, float (*Bs)[NT * 32 * PBS_LD], int* Ix, int* Kx) {
    constexpr bool DIRECT = MODE >= 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    constexpr bool dense = MODE == 2;
    const int n_real = dense ? (int)((p.dense_rows + PT - 1) / PT) : p.tile_k[p.n_tiles];
    const int nchunks = p.Cin >> 5;
    // static part [0, n_static) in equal ranges, then the pool [n_static, n_real) in units of `unit` tiles (see g_pool_ctr above)
    // (not for the 32- / 64-channel layers: a unit of six one- or two-step tiles is shorter than the pipeline restart it costs - measured
    //  +12 % / +4 % on the level-1 32 -> 32 and level-2 64 -> 64 layers, -3 ... -5 % on the 96- ... 192-channel ones)
    const bool pool = p.pool_ctr != nullptr && nchunks >= 3 && n_real >= 6 * (int)gridDim.x;
    int n_static = n_real;
    // (13/16 static and units of six steps: the best of a sweep over 11 - 14 sixteenths x 4 - 12 steps, profiles/r05_pool_sweep.txt)
    const int unit = (6 + nchunks - 1) / nchunks;
    if (pool) {
        if (tid == 0) pool_claim(p.pool_ctr + blockIdx.y, p.pool_epoch);
        n_static = (int)((int64_t)n_real * 13 / 16);
        if (p.chained) while (n_static < n_real && (p.tile_k[n_static - 1] & PG_CHAIN)) ++n_static;
    }
    int range0 = (int)((int64_t)blockIdx.x * n_static / gridDim.x);
    int range1 = (int)((int64_t)(blockIdx.x + 1) * n_static / gridDim.x);
    if (p.chained) {                                           // a chain of sub-tiles (<= 3) is never split between workgroups
        while (range0 > 0 && range0 < n_static && (p.tile_k[range0 - 1] & PG_CHAIN)) ++range0;
        while (range1 > 0 && range1 < n_static && (p.tile_k[range1 - 1] & PG_CHAIN)) ++range1;
    }
    if (range1 <= range0 && !pool) return;                     // uniform over the workgroup
    const int ncol0 = blockIdx.y * NT * 32;
    __shared__ int pool_unit;
    bool poisoned = false;                                     // (uniform) this workgroup met a counter word no launch can have written
    int fallback_round = 0;
    const uint64_t wstride_b = (uint64_t)p.Cout * (uint64_t)p.Cin * 4ull;

    f32x16 acc[NT];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool fresh = true;                                         // the next product starts a new tile (chain): its MFMA takes C = 0 instead of 16 NT register clears per tile

    // weight staging: thread f = tid + 256 i moves 16 bytes of row (f >> 3) of the chunk; its byte offset inside W[k] and its LDS slot
    // are fixed for the launch, the (offset, chunk) part of the address is wave-uniform
    f32x4 bst[NT];
    uint32_t woff[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int f = tid + i * 256;
        int n = ncol0 + (f >> 3);
        n = n < p.Cout ? n : p.Cout - 1;
        woff[i] = (uint32_t)(n * p.Cin + (f & 7) * 4) * 4u;
    }
    float* const sts = &Bs[0][(tid >> 3) * PBS_LD + (tid & 7) * 4];         // + i * 32 rows, + buf * buffer
    auto stage_load = [&](int kf, int chunk) {
        if (PG_DBG(p, 4)) { kf = 0; chunk = 0; }
        const int kw = p.k_flip >= 0 ? p.k_flip - (kf & PG_KMASK) : (kf & PG_KMASK);
        const char* Wk = (const char*)p.wt + (uint64_t)(uint32_t)kw * wstride_b + (uint32_t)chunk * 128u;    // scalar
#pragma unroll
        for (int i = 0; i < NT; ++i) bst[i] = *(const f32x4*)(Wk + woff[i]);
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NT; ++i) *(f32x4*)(sts + buf * (NT * 32 * PBS_LD) + i * 32 * PBS_LD) = bst[i];
    };
    const uint64_t lane_b = (uint64_t)(h * 64);                 // this lane's 16 channels of a 32-channel chunk
    auto load_a = [&](f32x4 (&a)[4], int row, int chunk) {      // branch-free: row >= 0 always (padding reads row 0)
        if (PG_DBG(p, 2)) row = 0;
        const int cc = chunk * 32;                              // wave-uniform from here ...
        const bool first = cc < p.C0;
        const char* base = (const char*)(first ? p.in0 : p.in1) + (uint32_t)(first ? cc : cc - p.C0) * 4u;
        const uint32_t ld4 = (uint32_t)(first ? p.ld0 : p.ld1) * 4u;
        const char* src = base + ((uint64_t)(uint32_t)row * ld4 + lane_b);                 // ... one 64-bit multiply-add per lane
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = *(const f32x4*)(src + q * 16);
    };
    const bool full_cols = ncol0 + NT * 32 <= p.Cout;           // every layer of the shipped networks (Cout a multiple of 32 NT)

  for (;;) {                                                     // the static range, then pool units until the pool is empty
    for (int piece0 = range0; piece0 < range1; piece0 += PG_PIECE) {
        const int ntl = range1 - piece0 < PG_PIECE ? range1 - piece0 : PG_PIECE;
        __syncthreads();                                       // nobody still reads the previous piece's metadata / weight buffers
        if (dense) {
            for (int f = tid; f < ntl * PT; f += 256) {
                const int64_t r = (int64_t)piece0 * PT + f;
                Ix[f] = (int)(r < p.dense_rows ? r : p.dense_rows - 1);
            }
            if (tid < PG_PIECE + 2) Kx[tid] = 0;
        } else {
        for (int f = tid; f < ntl * PT; f += 256) {
            const int v = p.in_idx[(int64_t)piece0 * PT + f];
            Ix[f] = v < 0 ? 0 : v;                             // padding gathers row 0: its products are never read back
        }
        if (tid < PG_PIECE + 2) Kx[tid] = tid < ntl ? p.tile_k[piece0 + tid] : 0;
        }
        __syncthreads();
        const int nsteps = ntl * nchunks;
        const int* Iw = Ix + wv * 32 + j;                       // this lane's gather row of tile t: Iw[t * PT]
        int k_cur = __builtin_amdgcn_readfirstlane(Kx[0]), k_nxt = __builtin_amdgcn_readfirstlane(Kx[1]);
        f32x4 a0[4], a1[4], a2[4];
        int pf_t = 0, pf_c = 0;                                 // prefetch pointer = step s + 2
#define PF_ADVANCE() do { if (++pf_c == nchunks) { pf_c = 0; ++pf_t; } } while (0)
#define PF_ROW() Iw[(pf_t < ntl ? pf_t : ntl - 1) * PT]        /* past the end: a harmless reload of the last tile's row */
        load_a(a0, PF_ROW(), pf_c); PF_ADVANCE();
        load_a(a1, PF_ROW(), pf_c); PF_ADVANCE();
        stage_load(k_cur, 0);
        stage_store(0);
        __syncthreads();
        int buf = 0, cur_lt = 0, cur_c = 0, s = 0;
#define PAIR_STEP(CUR, PF)                                                                                            \
    {                                                                                                                 \
        const bool has_next = s + 1 < nsteps;                                                                         \
        const bool last_chunk = cur_c + 1 == nchunks;                                                                 \
        const int pf_row = PF_ROW();                /* LDS: asked for before the matrix work, used behind its first group */ \
        const int k_after = Kx[cur_lt + 2];                                                                           \
        {                                                                                                             \
            const float* bb = &Bs[buf][j * PBS_LD + h * 16];                                                          \
            f32x4 bq[2][NT];                                                                                          \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) bq[0][t] = *(const f32x4*)(bb + t * 32 * PBS_LD);          \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                           \
                if (q < 3) {                                                                                          \
                    _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                    \
                        bq[(q + 1) & 1][t] = *(const f32x4*)(bb + t * 32 * PBS_LD + (q + 1) * 4);                     \
                }                                                                                                     \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                       \
                    if (q == 0 && e == 0) {                                                                           \
                        if (fresh) {                                                                                  \
                            _Pragma("unroll") for (int t = 0; t < NT; ++t)                                            \
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[0][t][0], CUR[0][0], zero16, 0, 0, 0); \
                        } else {                                                                                      \
                            _Pragma("unroll") for (int t = 0; t < NT; ++t)                                            \
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[0][t][0], CUR[0][0], acc[t], 0, 0, 0); \
                        }                                                                                             \
                    } else {                                                                                          \
                        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                \
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[q & 1][t][e], CUR[q][e], acc[t], 0, 0, 0); \
                    }                                                                                                 \
                }                                                                                                     \
                __builtin_amdgcn_sched_barrier(0);                                                                    \
                if (q == 0) {   /* this step's requests, behind 4 NT MFMAs: next weight chunk first (consumed first) */ \
                    if (has_next) stage_load(last_chunk ? k_nxt : k_cur, last_chunk ? 0 : cur_c + 1);                 \
                    load_a(PF, pf_row, pf_c); PF_ADVANCE();                                                           \
                    __builtin_amdgcn_sched_barrier(0);                                                                \
                }                                                                                                     \
            }                                                                                                         \
        }                                                                                                             \
        if (last_chunk && !(k_cur & PG_CHAIN)) { /* tile (chain) complete: lane = pair row, register group g = columns 8g + 4h .. +3 */ \
            int64_t prow = (int64_t)(piece0 + cur_lt) * PT + wv * 32 + j;                                             \
            if (PG_DBG(p, 16)) prow = (prow & 127) + (int64_t)(blockIdx.x & 63) * 128;   /* every store lands in an L2-resident window */ \
            if (PG_DBG(p, 1)) {                                                                                       \
                if (p.Cin < 0) { _Pragma("unroll") for (int t = 0; t < NT; ++t) p.part[prow + t] = acc[t][0] + acc[t][5] + acc[t][10] + acc[t][15]; } \
            } else if (DIRECT) {                               /* one pair per output row: write the row itself */          \
                const int64_t o = dense ? (prow < p.dense_rows ? prow : -1) : (int64_t)p.out_idx[prow];               \
                if (o >= 0) {                                                                                         \
                    _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                    \
                        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                               \
                            const int c = ncol0 + 4 * h + t * 32 + 8 * g;                                             \
                            if (full_cols || c < p.Cout) {                                                            \
                                f32x4 y;                                                                              \
                                _Pragma("unroll") for (int i = 0; i < 4; ++i) y[i] = pg_affine(acc[t][4 * g + i], p.scale, p.shift, c + i); \
                                if (p.res) y += *(const f32x4*)(p.res + o * p.ld_res + c);                            \
                                *(f32x4*)(p.out + o * p.ld_out + c) = f32x4{pg_act(y[0], p.act), pg_act(y[1], p.act), pg_act(y[2], p.act), pg_act(y[3], p.act)}; \
                            }                                                                                         \
                        }                                                                                             \
                }                                                                                                     \
            } else {                                                                                                  \
                float* dst = p.part + prow * p.Cout + ncol0 + 4 * h;                                                  \
                _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                        \
                    _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                     \
                        if (full_cols || ncol0 + 4 * h + t * 32 + 8 * g < p.Cout)                                     \
                            PART_STORE4(dst + t * 32 + 8 * g, (f32x4{acc[t][4 * g], acc[t][4 * g + 1],                \
                                                                     acc[t][4 * g + 2], acc[t][4 * g + 3]}));         \
            }                                                                                                         \
            fresh = true;                                                                                             \
        } else {                                                                                                      \
            fresh = false;                                                                                            \
        }                                                                                                             \
        if (!has_next) break;                                                                                         \
        stage_store(buf ^ 1);                                                                                         \
        if (!PG_DBG(p, 8)) __syncthreads();                                                                           \
        buf ^= 1;                                                                                                     \
        ++s;                                                                                                          \
        if (last_chunk) {                                                                                             \
            cur_c = 0; ++cur_lt;                                                                                      \
            k_cur = k_nxt; k_nxt = __builtin_amdgcn_readfirstlane(k_after);                                           \
        } else {                                                                                                      \
            ++cur_c;                                                                                                  \
        }                                                                                                             \
    }
        for (;;) {
            PAIR_STEP(a0, a2)
            PAIR_STEP(a1, a0)
            PAIR_STEP(a2, a1)
        }
#undef PAIR_STEP
#undef PF_ADVANCE
#undef PF_ROW
    }
    if (!pool) break;
    __syncthreads();
    if (tid == 0) {
        unsigned int d = poisoned ? POOL_POISONED : pool_draw(p.pool_ctr + blockIdx.y, p.pool_epoch);
        if (d == POOL_POISONED) {                              // a foreign write in the slot: the pool is split statically, unit b, b + grid, ...
            if (!poisoned) atomicAdd(&g_pool_bad_seen, 1u);
            d = blockIdx.x + (unsigned)fallback_round * gridDim.x;
            pool_unit = -1 - (int)d;
        } else {
            pool_unit = (int)d;
        }
    }
    __syncthreads();
    int u = __builtin_amdgcn_readfirstlane(pool_unit);
    if (u < 0) { poisoned = true; ++fallback_round; u = -1 - u; }
    range0 = n_static + u * unit;
    if (range0 >= n_real) break;
    range1 = range0 + unit < n_real ? range0 + unit : n_real;
    if (p.chained) {                                           // the same snapping rule at both ends: the units tile the pool without gaps or overlaps
        while (range0 < n_real && (p.tile_k[range0 - 1] & PG_CHAIN)) ++range0;
        while (range1 < n_real && (p.tile_k[range1 - 1] & PG_CHAIN)) ++range1;
    }
  }
}

#define PAIR_GEMM_ENTRY(NAME, NT, WAVES, MODE)                                                                  \
    __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void NAME(const PGParams p) { \
        __shared__ __attribute__((aligned(16))) float Bs[2][NT * 32 * PBS_LD];                                  \
        __shared__ int Ix[PG_PIECE * PT];                                                                       \
        __shared__ int Kx[PG_PIECE + 2];                                                                        \
        pair_gemm_body<NT, MODE>(p, Bs, Ix, Kx);                                                               \
    }
PAIR_GEMM_ENTRY(pair_gemm_kernel_1, 1, 3, 0)
PAIR_GEMM_ENTRY(pair_gemm_kernel_2, 2, 2, 0)
PAIR_GEMM_ENTRY(pair_gemm_kernel_3, 3, 2, 0)
PAIR_GEMM_ENTRY(pair_gemm_kernel_4, 4, 2, 0)
PAIR_GEMM_ENTRY(pair_gemm_direct_kernel_1, 1, 3, 1)
PAIR_GEMM_ENTRY(pair_gemm_direct_kernel_2, 2, 2, 1)
PAIR_GEMM_ENTRY(pair_gemm_direct_kernel_3, 3, 2, 1)
PAIR_GEMM_ENTRY(pair_gemm_direct_kernel_4, 4, 2, 1)
PAIR_GEMM_ENTRY(pair_dense_kernel_1, 1, 3, 2)
PAIR_GEMM_ENTRY(pair_dense_kernel_2, 2, 2, 2)
PAIR_GEMM_ENTRY(pair_dense_kernel_3, 3, 2, 2)
PAIR_GEMM_ENTRY(pair_dense_kernel_4, 4, 2, 2)

#ifdef SD3D_WITH_PC
// ---- pass 1, producer / consumer variant (round 6; LAB BUILD ONLY: -DSD3D_WITH_PC, tools/ablate.sh) ---------------------------------
// Built, bit-identical to the lock-step kernel on every table of the benchmark scene, measured 9 - 70 % SLOWER (profiles/r06_pc_check.md,
// profiles/r06_ablate_pc.md; the reading is in profiles/EXPERIMENTS.md round 6): not in the product library.
// What the round-6 ablations of the lock-step kernel say (profiles/r06_ablate.md, profiles/r06_dense_ceiling.md): with the IDENTITY as
// rulebook (no gather randomness, no partial products) the kernel reaches the same 0.43 - 0.58 of the fp32 matrix peak it reaches on the
// real rulebooks - the lists and the chains are not what bounds pass 1; removing the per-step barrier or the weight requests changes
// nothing; removing the epilogue STORES (even when they land in an L2-resident window) and serving the gathers from cache lifts it to
// 0.64 - 0.75.  A wave that multiplies also issues 4 gathers + NT weight requests per step and 16 NT scattered 16-byte stores per tile
// through the CU's one address unit, in order with its own `s_waitcnt vmcnt`: what the matrix pipe waits for is its own wave's memory
// instructions.  Here the roles are split by WAVE: four consumer waves (one per SIMD) only read LDS and issue MFMAs; four producer waves
// stage the next step's gathered rows and weight chunk global -> registers -> LDS (row-coalesced: eight lanes per 128-byte row), and
// carry a finished tile from its LDS staging area to memory with full-line stores.  One `s_barrier` per step is the only
// synchronisation (no flags, no polling: round 2's ring-and-counter version lost to its own flag traffic).  The MFMA order per
// accumulator is the lock-step kernel's (chunk, then q, e; offsets of a chain in list order): the partial products are the same bits.
#define PC_LD 36                 // floats per LDS row of a 32-channel chunk (32 + 4: conflict-free b128 reads of a lane's 16 channels)
template <int NT, int MODE>
__device__ __forceinline__ void pair_gemm_pc_body(const PGParams& p, float* smem, const int n_os) {
    constexpr bool DIRECT = MODE >= 1;
    constexpr bool dense = MODE == 2;
    constexpr int WSZ = NT * 32 * PC_LD, ASZ = PT * PC_LD, LDO = NT * 32 + 4, OSZ = PT * LDO;
    float* const Ws = smem;                                   // [2][NT * 32][PC_LD]  weight chunk of a step
    float* const As = Ws + 2 * WSZ;                           // [2][128][PC_LD]      gathered rows of a step
    float* const Os = As + 2 * ASZ;                           // [n_os][128][LDO]     a finished tile on its way out
    int* const Ix = (int*)(Os + n_os * OSZ);                  // gather rows of the piece
    int* const Kx = Ix + PG_PIECE * PT;                       // offsets (+ chain flags) of the piece's tiles
    __shared__ int pool_unit;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool mma = wave < 4;                                // consumer (MFMA) wave / producer (memory) wave
    const int wv = wave & 3;                                  // consumer: its 32 rows of the tile; producer: its quarter of the staging work
    const int j = lane & 31, h = lane >> 5;
    const int r8 = lane >> 3, s8 = lane & 7;                  // producer: row within a group of eight, 16-byte segment of the 128-byte row
    const int n_real = dense ? (int)((p.dense_rows + PT - 1) / PT) : p.tile_k[p.n_tiles];
    const int nchunks = p.Cin >> 5;
    const bool pool = p.pool_ctr != nullptr && nchunks >= 3 && n_real >= 6 * (int)gridDim.x;
    int n_static = n_real;
    const int unit = (6 + nchunks - 1) / nchunks;
    if (pool) {
        if (tid == 0) pool_claim(p.pool_ctr + blockIdx.y, p.pool_epoch);
        n_static = (int)((int64_t)n_real * 13 / 16);
        if (p.chained) while (n_static < n_real && (p.tile_k[n_static - 1] & PG_CHAIN)) ++n_static;
    }
    int range0 = (int)((int64_t)blockIdx.x * n_static / gridDim.x);
    int range1 = (int)((int64_t)(blockIdx.x + 1) * n_static / gridDim.x);
    if (p.chained) {
        while (range0 > 0 && range0 < n_static && (p.tile_k[range0 - 1] & PG_CHAIN)) ++range0;
        while (range1 > 0 && range1 < n_static && (p.tile_k[range1 - 1] & PG_CHAIN)) ++range1;
    }
    if (range1 <= range0 && !pool) return;
    const int ncol0 = blockIdx.y * NT * 32;
    const uint64_t wstride_b = (uint64_t)p.Cout * (uint64_t)p.Cin * 4ull;
    const bool full_cols = ncol0 + NT * 32 <= p.Cout;
    bool poisoned = false;
    int fallback_round = 0;

    // consumer state
    f32x16 acc[NT];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool fresh = true;
    // producer state: two register sets (steps of even / odd parity), byte offsets of its weight rows fixed for the launch
    f32x4 ga0[4], ga1[4], gw0[NT], gw1[NT];
    uint32_t woff[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        int n = ncol0 + wv * (NT * 8) + i * 8 + r8;
        n = n < p.Cout ? n : p.Cout - 1;
        woff[i] = (uint32_t)(n * p.Cin + s8 * 4) * 4u;
    }
    auto issue = [&](f32x4 (&ga)[4], f32x4 (&gw)[NT], int lt, int c, int kf) {          // the requests of step (tile lt of the piece, chunk c)
        const int cc = c * 32;
        const bool first = cc < p.C0;
        const char* base = (const char*)(first ? p.in0 : p.in1) + (uint32_t)((first ? cc : cc - p.C0) + s8 * 4) * 4u;
        const uint32_t ld4 = (uint32_t)(first ? p.ld0 : p.ld1) * 4u;
        const int* ix = Ix + lt * PT + wv * 32 + r8;
#pragma unroll
        for (int i = 0; i < 4; ++i) ga[i] = *(const f32x4*)(base + (uint64_t)(uint32_t)(PG_DBG(p, 2) ? 0 : ix[i * 8]) * ld4);
        if (PG_DBG(p, 4)) { kf = 0; c = 0; }
        const int kw = p.k_flip >= 0 ? p.k_flip - (kf & PG_KMASK) : (kf & PG_KMASK);
        const char* Wk = (const char*)p.wt + (uint64_t)(uint32_t)kw * wstride_b + (uint32_t)c * 128u;
#pragma unroll
        for (int i = 0; i < NT; ++i) gw[i] = *(const f32x4*)(Wk + woff[i]);
    };
    auto put = [&](const f32x4 (&ga)[4], const f32x4 (&gw)[NT], int st) {
        float* a = As + st * ASZ + (wv * 32 + r8) * PC_LD + s8 * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)(a + i * 8 * PC_LD) = ga[i];
        float* w = Ws + st * WSZ + (wv * (NT * 8) + r8) * PC_LD + s8 * 4;
#pragma unroll
        for (int i = 0; i < NT; ++i) *(f32x4*)(w + i * 8 * PC_LD) = gw[i];
    };
    // a finished tile: Os -> memory.  A producer wave owns 32 rows; an instruction moves 64 / (8 NT) whole rows of 32 NT floats
    constexpr int LPR = NT * 8, RPI = 64 / LPR, NOUT = (32 + RPI - 1) / RPI;
    auto copy_out = [&](int tile, int par) {
        if (PG_DBG(p, 1)) return;
        const bool act_lane = lane < RPI * LPR;
        const int rr = lane / LPR, cs = (lane - rr * LPR) * 4;
        const float* o = Os + par * OSZ + (wv * 32 + rr) * LDO + cs;
        const int c = ncol0 + cs;
        constexpr int HB = DIRECT ? (NOUT + 3) / 4 : (NOUT + 1) / 2;    // batches of requests (registers: the accumulators of the consumer role are allocated in every wave)
#pragma unroll
        for (int b0 = 0; b0 < NOUT; b0 += HB) {
            f32x4 v[HB];
#pragma unroll
            for (int i = 0; i < HB; ++i) v[i] = *(const f32x4*)(o + ((b0 + i) * RPI < 32 - rr ? (b0 + i) * RPI : 0) * LDO);
#pragma unroll
            for (int i = 0; i < HB; ++i) {
                const int ri = (b0 + i) * RPI + rr;
                if (b0 + i >= NOUT || !act_lane || ri >= 32 || !(full_cols || c < p.Cout)) continue;
                const int64_t prow = (int64_t)tile * PT + wv * 32 + ri;
                if (DIRECT) {
                    const int64_t orow = dense ? (prow < p.dense_rows ? prow : -1) : (int64_t)p.out_idx[prow];
                    if (orow >= 0) {
                        f32x4 y;
#pragma unroll
                        for (int u = 0; u < 4; ++u) y[u] = pg_affine(v[i][u], p.scale, p.shift, c + u);
                        if (p.res) y += *(const f32x4*)(p.res + orow * p.ld_res + c);
                        *(f32x4*)(p.out + orow * p.ld_out + c) = f32x4{pg_act(y[0], p.act), pg_act(y[1], p.act), pg_act(y[2], p.act), pg_act(y[3], p.act)};
                    }
                } else {
                    PART_STORE4(p.part + prow * p.Cout + c, v[i]);
                }
            }
        }
    };

  for (;;) {
    for (int piece0 = range0; piece0 < range1; piece0 += PG_PIECE) {
        const int ntl = range1 - piece0 < PG_PIECE ? range1 - piece0 : PG_PIECE;
        __syncthreads();
        if (dense) {
            for (int f = tid; f < ntl * PT; f += 512) {
                const int64_t r = (int64_t)piece0 * PT + f;
                Ix[f] = (int)(r < p.dense_rows ? r : p.dense_rows - 1);
            }
            if (tid < PG_PIECE + 2) Kx[tid] = 0;
        } else {
            for (int f = tid; f < ntl * PT; f += 512) {
                const int v = p.in_idx[(int64_t)piece0 * PT + f];
                Ix[f] = v < 0 ? 0 : v;
            }
            if (tid < PG_PIECE + 2) Kx[tid] = tid < ntl ? p.tile_k[piece0 + tid] : 0;
        }
        __syncthreads();
        const int nsteps = ntl * nchunks;
        int k_cur = __builtin_amdgcn_readfirstlane(Kx[0]), k_nxt = __builtin_amdgcn_readfirstlane(Kx[1]);
        int pf_t = 0, pf_c = 0;                                 // the producers' request pointer = step s + 2
#define PC_ADVANCE() do { if (++pf_c == nchunks) { pf_c = 0; ++pf_t; } } while (0)
#define PC_TILE() (pf_t < ntl ? pf_t : ntl - 1)
        if (!mma) {
            { const int t_ = PC_TILE(); issue(ga0, gw0, t_, pf_c, Kx[t_]); } PC_ADVANCE();
            { const int t_ = PC_TILE(); issue(ga1, gw1, t_, pf_c, Kx[t_]); } PC_ADVANCE();
            put(ga0, gw0, 0);
        } else {
            PC_ADVANCE(); PC_ADVANCE();
        }
        __syncthreads();
        int buf = 0, cur_lt = 0, cur_c = 0, s = 0;
        int out_tile = -1, out_par = 0, os_par = 0;             // a tile waiting in Os[out_par]; the Os slot the next finished tile takes
#define PC_STEP(PUT_A, PUT_W, LD_A, LD_W)                                                                            \
    {                                                                                                                 \
        const bool has_next = s + 1 < nsteps;                                                                         \
        const bool last_chunk = cur_c + 1 == nchunks;                                                                 \
        const bool tile_done = last_chunk && !(k_cur & PG_CHAIN);                                                     \
        const int k_after = Kx[cur_lt + 2];                                                                           \
        if (mma) {                                                                                                    \
            const float* wb = Ws + buf * WSZ + j * PC_LD + h * 16;                                                    \
            const float* ab = As + buf * ASZ + (wv * 32 + j) * PC_LD + h * 16;                                        \
            f32x4 a[4], wq[2][NT];                                                                                    \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) a[q] = *(const f32x4*)(ab + q * 4);                         \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) wq[0][t] = *(const f32x4*)(wb + t * 32 * PC_LD);           \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                           \
                if (q < 3) {    /* the next group's weight fragments are requested BEFORE this group's MFMAs (pinned: hipcc sinks them behind the MFMAs otherwise) */ \
                    _Pragma("unroll") for (int t = 0; t < NT; ++t) wq[(q + 1) & 1][t] = *(const f32x4*)(wb + t * 32 * PC_LD + (q + 1) * 4); \
                }                                                                                                     \
                __builtin_amdgcn_sched_barrier(0);                                                                    \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                       \
                    if (q == 0 && e == 0) {                                                                           \
                        if (fresh) {                                                                                  \
                            _Pragma("unroll") for (int t = 0; t < NT; ++t)                                            \
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[0][t][0], a[0][0], zero16, 0, 0, 0); \
                        } else {                                                                                      \
                            _Pragma("unroll") for (int t = 0; t < NT; ++t)                                            \
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[0][t][0], a[0][0], acc[t], 0, 0, 0); \
                        }                                                                                             \
                    } else {                                                                                          \
                        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                \
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[q & 1][t][e], a[q][e], acc[t], 0, 0, 0); \
                    }                                                                                                 \
                }                                                                                                     \
                __builtin_amdgcn_sched_barrier(0);                                                                    \
            }                                                                                                         \
            if (tile_done && !(PG_DBG(p, 32) && p.Cin > 0)) {     /* lane = pair row, register group g = columns 8g + 4h .. + 3 of column block t */   \
                float* o = Os + os_par * OSZ + (wv * 32 + j) * LDO + 4 * h;                                           \
                _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                        \
                    _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                     \
                        *(f32x4*)(o + t * 32 + 8 * g) = f32x4{acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]}; \
            }                                                                                                         \
            fresh = tile_done;                                                                                        \
        } else {                                                                                                      \
            if (out_tile >= 0) copy_out(out_tile, out_par);     /* the tile finished one step ago */                  \
            if (has_next) put(PUT_A, PUT_W, buf ^ 1);                                                                 \
            if (s + 2 < nsteps) { const int t_ = PC_TILE(); issue(LD_A, LD_W, t_, pf_c, Kx[t_]); }                    \
        }                                                                                                             \
        PC_ADVANCE();                                                                                                 \
        out_tile = tile_done ? piece0 + cur_lt : -1;                                                                  \
        out_par = os_par;                                                                                             \
        if (tile_done && n_os > 1) os_par ^= 1;                                                                       \
        __syncthreads();                                                                                              \
        if (!has_next) break;                                                                                         \
        buf ^= 1;                                                                                                     \
        ++s;                                                                                                          \
        if (last_chunk) {                                                                                             \
            cur_c = 0; ++cur_lt;                                                                                      \
            k_cur = k_nxt; k_nxt = __builtin_amdgcn_readfirstlane(k_after);                                           \
        } else {                                                                                                      \
            ++cur_c;                                                                                                  \
        }                                                                                                             \
    }
        for (;;) {
            PC_STEP(ga1, gw1, ga0, gw0)
            PC_STEP(ga0, gw0, ga1, gw1)
        }
#undef PC_STEP
#undef PC_ADVANCE
#undef PC_TILE
        if (!mma && out_tile >= 0) copy_out(out_tile, out_par);  // the piece's last tile (the barrier at the top of the next piece / of the pool draw frees Os)
    }
    if (!pool) break;
    __syncthreads();
    if (tid == 0) {
        unsigned int d = poisoned ? POOL_POISONED : pool_draw(p.pool_ctr + blockIdx.y, p.pool_epoch);
        if (d == POOL_POISONED) {
            if (!poisoned) atomicAdd(&g_pool_bad_seen, 1u);
            d = blockIdx.x + (unsigned)fallback_round * gridDim.x;
            pool_unit = -1 - (int)d;
        } else {
            pool_unit = (int)d;
        }
    }
    __syncthreads();
    int u = __builtin_amdgcn_readfirstlane(pool_unit);
    if (u < 0) { poisoned = true; ++fallback_round; u = -1 - u; }
    range0 = n_static + u * unit;
    if (range0 >= n_real) break;
    range1 = range0 + unit < n_real ? range0 + unit : n_real;
    if (p.chained) {
        while (range0 < n_real && (p.tile_k[range0 - 1] & PG_CHAIN)) ++range0;
        while (range1 < n_real && (p.tile_k[range1 - 1] & PG_CHAIN)) ++range1;
    }
  }
}

static size_t pair_pc_lds_bytes(int nt, int n_os) {
    return (size_t)(2 * nt * 32 * PC_LD + 2 * PT * PC_LD + n_os * PT * (nt * 32 + 4)) * sizeof(float) + (size_t)(PG_PIECE * PT + PG_PIECE + 2) * sizeof(int);
}
#define PAIR_GEMM_PC_ENTRY(NAME, NT, MODE)                                                                           \
    __global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void NAME(const PGParams p, const int n_os) { \
        extern __shared__ __attribute__((aligned(16))) float pc_smem[];                                              \
        pair_gemm_pc_body<NT, MODE>(p, pc_smem, n_os);                                                               \
    }
PAIR_GEMM_PC_ENTRY(pair_gemm_pc_kernel_1, 1, 0)
PAIR_GEMM_PC_ENTRY(pair_gemm_pc_kernel_2, 2, 0)
PAIR_GEMM_PC_ENTRY(pair_gemm_pc_kernel_3, 3, 0)
PAIR_GEMM_PC_ENTRY(pair_gemm_pc_kernel_4, 4, 0)
PAIR_GEMM_PC_ENTRY(pair_gemm_pc_direct_kernel_1, 1, 1)
PAIR_GEMM_PC_ENTRY(pair_gemm_pc_direct_kernel_2, 2, 1)
PAIR_GEMM_PC_ENTRY(pair_gemm_pc_direct_kernel_3, 3, 1)
PAIR_GEMM_PC_ENTRY(pair_gemm_pc_direct_kernel_4, 4, 1)
PAIR_GEMM_PC_ENTRY(pair_dense_pc_kernel_1, 1, 2)
PAIR_GEMM_PC_ENTRY(pair_dense_pc_kernel_2, 2, 2)
PAIR_GEMM_PC_ENTRY(pair_dense_pc_kernel_3, 3, 2)
PAIR_GEMM_PC_ENTRY(pair_dense_pc_kernel_4, 4, 2)

#endif  // SD3D_WITH_PC

// ---- pass 1, weight-stationary variant ----------------------------------------------------------
// For layers whose whole W[k] (Cout x Cin fp32, Cout = 32*NT <= 128) fits in LDS next to a second
// workgroup: the workgroup stages W[k] ONCE per run of same-offset tiles and its four waves then walk
// their own 32-pair sub-tiles independently - no per-step barrier, no re-staging of weight chunks.
// The MFMA is issued transposed (A = weights, B = gathered rows) so a lane owns one pair row and its
// accumulator registers are 4-column groups: the partial products leave as dwordx4 stores.
#define WS_RANGE_TILES 24       // most tiles one workgroup may be given: its gather indices live in LDS (12 KB)
#define WS_STAGE_BATCH 16       // 16-byte weight requests per thread in flight while W[k] is staged

// (An inline-asm variant of the gather with hand-counted s_waitcnt vmcnt(N) was tried to keep two steps
// of loads in flight past hipcc's conservative waits: it was not faster - latency is not what limits this
// kernel - and the register copies the compiler places around asm outputs at the loop back-edge read
// registers with loads still in flight.  Plain loads it is.)
#define LOAD_A4(A, PTR)                                                                                            \
    do { _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) A[q_] = *(const f32x4*)((PTR) + q_ * 4); } while (0)

// RT = 32-pair row tiles one wave multiplies per weight fragment.  RT = 4 / 2 for the narrow layers
// (Cout = 32 / 64; 4 independent accumulators per fragment read, epilogue amortised over 64 MFMAs per step)
// measured SLOWER than RT = 1 (stem 296 -> 333 us, 32->32 62 -> 72, 64->64 108 -> 121): those layers are
// bound by the gathers and partial stores per flop, not by the matrix pipe, and RT costs them half their
// resident waves.  All variants therefore run RT = 1.
template <int NT, int RT, int ST, bool DIRECT>
__device__ __forceinline__ void pair_gemm_ws_body(const PGParams& p, float* Ws) {   // no __restrict__: LDS shared across waves
    constexpr int UPT = 4 / RT;                                // wave units (RT x 32 pairs) per 128-pair tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform values must live in SGPRs (scalar branches)
    const int j = lane & 31, h = lane >> 5;
    const int n_real = p.tile_k[p.n_tiles];
    const int range0 = (int)((int64_t)blockIdx.x * n_real / gridDim.x);
    const int range1 = (int)((int64_t)(blockIdx.x + 1) * n_real / gridDim.x);
    if (range1 <= range0) return;
    const int nchunks = p.Cin >> 5;
    const int ldw = p.Cin + 4;                                 // (Cin + 4) mod 64 is 4 or 36: conflict-free b128 rows
    const int c4 = p.Cin >> 2;
    const int npieces = 32 * NT * c4;
    const int ncol0 = blockIdx.y * NT * 32;                    // column group (256-column layers run as two groups of 128)
    int* Ix = (int*)(Ws + 32 * NT * ldw);                      // gather rows of this workgroup's pairs; padding (-1) -> row 0

    f32x16 acc[RT][NT];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool fresh = true;                                         // the next product starts a new unit: its MFMAs take C = 0 (no register clears: VALU time is matrix time here)

  // the range is walked in pieces whose gather indices fit the LDS buffer (the capacity-sized lists of a scene that
  // skipped the rulebook-size read-back can make a range longer than that)
  for (int tile0 = range0; tile0 < range1; tile0 += WS_RANGE_TILES) {
    const int ntl = range1 - tile0 < WS_RANGE_TILES ? range1 - tile0 : WS_RANGE_TILES;
    __syncthreads();                                           // nobody still reads the previous piece's indices
    for (int f = tid; f < ntl * PT; f += 256) {
        const int v = p.in_idx[(int64_t)(tile0 + (f >> 7)) * PT + (f & (PT - 1))];
        Ix[f] = v < 0 ? 0 : v;                                 // their partial products are never read back
    }
    int run_start = 0;
    while (run_start < ntl) {                                  // runs of tiles with the same offset (uniform)
        const int k = p.tile_k[(tile0 + run_start)];
        int run_end = run_start + 1;
        while (run_end < ntl && p.tile_k[(tile0 + run_end)] == k) ++run_end;
        __syncthreads();                                       // everyone is done reading the previous W (and Ix is written)
        {
            const float* __restrict__ W = p.wt + ((int64_t)(p.k_flip >= 0 ? p.k_flip - k : k) * p.Cout + ncol0) * p.Cin;
            // sixteen requests per thread in flight (four before round 5): staging W[k] was 4 - 8 dependent L2 round trips of ~1.5 us before
            // a workgroup's first MFMA - a fifth of a level-3 128 -> 128 launch, where a workgroup has 3.5 tiles.  Unconditional loads
            // (clamped), conditional LDS writes: no exec-masked load for the compiler to wait on.
            constexpr int SB = (NT == 1 || NT == 3) ? WS_STAGE_BATCH / 2 : WS_STAGE_BATCH;    // (the 3- and 4-waves-per-SIMD variants have 128 / 168 registers)
            for (int f0 = 0; f0 < npieces; f0 += 256 * SB) {
                f32x4 v[SB];
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int f = f0 + u * 256 + tid;
                    v[u] = *(const f32x4*)(W + (int64_t)(f < npieces ? f : npieces - 1) * 4);
                }
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int f = f0 + u * 256 + tid;
                    if (f < npieces) {
                        const int row = f / c4, col = f - row * c4;
                        *(f32x4*)(Ws + row * ldw + col * 4) = v[u];
                    }
                }
            }
        }
        __syncthreads();
        // this wave's units of the run: local unit numbers u0, u0 + 4, ...
        const int u0 = run_start * UPT + wv;
        const int ns = (run_end * UPT - u0 + 3) / 4;
        const int nsteps = ns * nchunks;
        if (nsteps > 0) {
            // fragments of step (unit i, chunk c); past the end -> unit 0 (harmless reload)
            auto load_unit = [&](f32x4 (&A)[RT][4], int i, int c) {
                const int ii = i < ns ? i : 0;
                const int cc = c * 32;                          // wave-uniform
                const bool first = cc < p.C0;
                const float* base = first ? p.in0 : p.in1;
                const int ld = first ? p.ld0 : p.ld1;
                const int coff = (first ? cc : cc - p.C0) + h * 16;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int row = Ix[((u0 + 4 * ii) * RT + rt) * 32 + j];
                    const float* q = base + (int64_t)row * ld + coff;
                    LOAD_A4(A[rt], q);
                }
            };
            // ST register stages: the rows of step s + ST - 1 are requested while step s multiplies.  The ring is
            // rotated by unrolling the loop ST times with the roles renamed (copying a stage would make the
            // compiler wait for the loads still in flight into it).  Narrow layers are latency-bound and want
            // ST = 3-4; the 96/128-column variants are better off with the registers (ST = 2).
            f32x4 a0[RT][4], a1[RT][4], a2[RT][4], a3[RT][4];
            int pf_i = 0, pf_c = 0;
#define WS_PF_ADVANCE() do { if (++pf_c == nchunks) { pf_c = 0; ++pf_i; } } while (0)
            load_unit(a0, pf_i, pf_c); WS_PF_ADVANCE();
            if (ST >= 3) { load_unit(a1, pf_i, pf_c); WS_PF_ADVANCE(); }
            if (ST >= 4) { load_unit(a2, pf_i, pf_c); WS_PF_ADVANCE(); }
            int cur_i = 0, cur_c = 0, s = 0;
#define WS_STEP(CUR, PF)                                                                                              \
    {                                                                                                                 \
        const bool last_chunk = cur_c + 1 == nchunks;                                                                 \
        load_unit(PF, pf_i, pf_c); WS_PF_ADVANCE();             /* next step's rows while this one multiplies */      \
        {                                                                                                             \
            const float* wb = Ws + j * ldw + cur_c * 32 + h * 16;                                                     \
            f32x4 wq[2][NT];                                                                                          \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) wq[0][t] = *(const f32x4*)(wb + t * 32 * ldw);             \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                           \
                if (q < 3) {                                                                                          \
                    _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                    \
                        wq[(q + 1) & 1][t] = *(const f32x4*)(wb + t * 32 * ldw + (q + 1) * 4);                        \
                }                                                                                                     \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                       \
                    if (q == 0 && e == 0 && fresh) {                                                                  \
                        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                \
                            _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                         \
                                acc[rt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[0][t][0], CUR[rt][0][0], zero16, 0, 0, 0); \
                    } else {                                                                                          \
                        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                \
                            _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                         \
                                acc[rt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[q & 1][t][e], CUR[rt][q][e],     \
                                                                                  acc[rt][t], 0, 0, 0);               \
                    }                                                                                                 \
                }                                                                                                     \
                __builtin_amdgcn_sched_barrier(0);                                                                    \
            }                                                                                                         \
        }                                                                                                             \
        if (last_chunk) { /* unit complete: lane = pair row, register group g = columns 8g + 4h .. +3 */              \
            _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) {                                                       \
                const int ue = (u0 + 4 * cur_i) * RT + rt;          /* 32-pair unit of the piece: tile ue / 4, quarter ue % 4 */ \
                const int64_t prow = (int64_t)(tile0 + (ue >> 2)) * PT + (ue & 3) * 32 + j;                       \
                if (DIRECT) {                                       /* one pair per output row: write the row itself */ \
                    const int64_t o = p.out_idx[prow];                                                                \
                    if (o >= 0) {                                                                                     \
                        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                \
                            _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                           \
                                const int c = ncol0 + 4 * h + t * 32 + 8 * g;                                         \
                                f32x4 y;                                                                              \
                                _Pragma("unroll") for (int i = 0; i < 4; ++i) y[i] = pg_affine(acc[rt][t][4 * g + i], p.scale, p.shift, c + i); \
                                if (p.res) y += *(const f32x4*)(p.res + o * p.ld_res + c);                            \
                                *(f32x4*)(p.out + o * p.ld_out + c) = f32x4{pg_act(y[0], p.act), pg_act(y[1], p.act), pg_act(y[2], p.act), pg_act(y[3], p.act)}; \
                            }                                                                                         \
                    }                                                                                                 \
                } else {                                                                                              \
                float* dst = p.part + prow * p.Cout + ncol0 + 4 * h;                                                  \
                _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                        \
                    _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                     \
                        PART_STORE4(dst + t * 32 + 8 * g, (f32x4{acc[rt][t][4 * g], acc[rt][t][4 * g + 1],            \
                                                                 acc[rt][t][4 * g + 2], acc[rt][t][4 * g + 3]}));     \
                }                                                                                                     \
            }                                                                                                         \
        }                                                                                                             \
        fresh = last_chunk;                                                                                           \
        if (s + 1 >= nsteps) break;                                                                                   \
        ++s;                                                                                                          \
        if (last_chunk) { cur_c = 0; ++cur_i; } else { ++cur_c; }                                                     \
    }
            if (ST == 2) {
                for (;;) {
                    WS_STEP(a0, a1)
                    WS_STEP(a1, a0)
                }
            } else if (ST == 3) {
                for (;;) {
                    WS_STEP(a0, a2)
                    WS_STEP(a1, a0)
                    WS_STEP(a2, a1)
                }
            } else {
                for (;;) {
                    WS_STEP(a0, a3)
                    WS_STEP(a1, a0)
                    WS_STEP(a2, a1)
                    WS_STEP(a3, a2)
                }
            }
#undef WS_STEP
#undef WS_PF_ADVANCE
        }
        run_start = run_end;
    }
  }
}

#define PAIR_GEMM_WS_ENTRY(NAME, NT, RT, ST, WAVES, DIRECT)                                                        \
    __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void NAME(const PGParams p) { \
        extern __shared__ __attribute__((aligned(16))) float ws_smem[];                                            \
        pair_gemm_ws_body<NT, RT, ST, DIRECT>(p, ws_smem);                                                         \
    }
PAIR_GEMM_WS_ENTRY(pair_gemm_ws_kernel_1, 1, 1, 3, 4, false)
PAIR_GEMM_WS_ENTRY(pair_gemm_ws_kernel_2, 2, 1, 3, 2, false)
PAIR_GEMM_WS_ENTRY(pair_gemm_ws_kernel_3, 3, 1, 2, 3, false)
PAIR_GEMM_WS_ENTRY(pair_gemm_ws_kernel_4, 4, 1, 2, 2, false)
PAIR_GEMM_WS_ENTRY(pair_gemm_ws_direct_kernel_1, 1, 1, 3, 4, true)
PAIR_GEMM_WS_ENTRY(pair_gemm_ws_direct_kernel_2, 2, 1, 3, 2, true)
PAIR_GEMM_WS_ENTRY(pair_gemm_ws_direct_kernel_3, 3, 1, 2, 3, true)
PAIR_GEMM_WS_ENTRY(pair_gemm_ws_direct_kernel_4, 4, 1, 2, 2, true)

// ---- pass 2: fixed-order reduction over the offsets + epilogue --------------------------------
struct PRParams {
    const int32_t* pos; int K; int64_t M;
    const float* part; int Cout;
    const float* scale; const float* shift;
    const float* res; int ld_res;
    float* out; int ld_out; int act;
};

__global__ __launch_bounds__(256) void pair_reduce_kernel(const PRParams p) {
    const int c4 = p.Cout >> 2;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = e / c4;
    if (r >= p.M) return;
    const int q = (int)(e - r * c4) * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < p.K; k0 += 8) {
        int ids[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) ids[u] = (k0 + u < p.K) ? p.pos[(int64_t)(k0 + u) * p.M + r] : -1;
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            v[u] = ids[u] >= 0 ? *(const f32x4*)(p.part + (int64_t)ids[u] * p.Cout + q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (ids[u] >= 0) a += v[u];
    }
    float y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = q + i;
        float t = pg_affine(a[i], p.scale, p.shift, n);
        if (p.res) t += p.res[r * p.ld_res + n];
        y[i] = pg_act(t, p.act);
    }
    *(f32x4*)(p.out + r * p.ld_out + q) = f32x4{y[0], y[1], y[2], y[3]};
}

// ---- pass 2 over the per-row lists ---------------------------------------------------------------
// The same fixed-order sum, but a row walks ITS OWN partial products {count, positions...} (pair_rowlist_batch_kernel) instead of
// all K slots of pos[k][r]: one 16-byte load brings the count and the first three positions.
struct PRLParams {
    const int32_t* rlist; int rl_stride; int64_t M;
    const float* part; int Cout;
    const float* scale; const float* shift;
    const float* res; int ld_res;
    float* out; int ld_out; int act;
};

__global__ __launch_bounds__(256) void pair_reduce_rl_kernel(const PRLParams p) {
    const int c4 = p.Cout >> 2;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = e / c4;
    if (r >= p.M) return;
    const int q = (int)(e - r * c4) * 4;
    const int32_t* rl = p.rlist + r * p.rl_stride;
    // the row's whole list in registers first (7 x 16 bytes cover count + 27 positions: every 3^3 table), then the partial
    // products eight at a time: two short dependent steps instead of a position -> product chain per group of four
    int4 hv[7];
    hv[0] = *(const int4*)rl;
    const int cnt = hv[0].x;
#pragma unroll
    for (int g = 1; g < 7; ++g) hv[g] = (4 * g <= cnt) ? *(const int4*)(rl + 4 * g) : int4{0, 0, 0, 0};
    auto id_at = [&](int i) -> int {                            // list entry i (< 27) = int 1 + i of the row; compile-time i after unrolling
        const int4& v = hv[(1 + i) >> 2];
        const int c = (1 + i) & 3;
        return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w));
    };
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i0 = 0; i0 < 27; i0 += 9) {
        if (i0 < cnt) {
            f32x4 v[9];
#pragma unroll
            for (int u = 0; u < 9; ++u)
                v[u] = i0 + u < cnt ? PART_LOAD4(p.part + (int64_t)id_at(i0 + u) * p.Cout + q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 9; ++u) if (i0 + u < cnt) a += v[u];
        }
    }
    for (int i0 = 27; i0 < cnt; i0 += 4) {                      // longer lists (the 5^3 stem): four at a time from memory (int 1 + 27 = 28: 16-byte aligned)
        const int4 ids = *(const int4*)(rl + 1 + i0);
        f32x4 v[4];
        const int idv[4] = {ids.x, ids.y, ids.z, ids.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i0 + u < cnt ? PART_LOAD4(p.part + (int64_t)idv[u] * p.Cout + q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i0 + u < cnt) a += v[u];
    }
    float y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = q + i;
        float t = pg_affine(a[i], p.scale, p.shift, n);
        if (p.res) t += p.res[r * p.ld_res + n];
        y[i] = pg_act(t, p.act);
    }
    *(f32x4*)(p.out + r * p.ld_out + q) = f32x4{y[0], y[1], y[2], y[3]};
}

// ---- the shared tail's host side ------------------------------------------------------------------
#define PG_MAX_DEVICES 64
#ifndef SD3D_PAIR_PC_DEFAULT
#define SD3D_PAIR_PC_DEFAULT 0
#endif
static std::atomic<unsigned long long*> g_pool_bases[PG_MAX_DEVICES];
static std::atomic<unsigned long long> g_pool_tickets[PG_MAX_DEVICES];        // pooled launches handed out so far, per device
static std::atomic<int> g_pool_on{-1};                                         // -1: not decided yet (SD3D_PAIR_POOL, default 1)
static bool pool_enabled() {
    int v = g_pool_on.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("SD3D_PAIR_POOL");
        v = e ? (atoi(e) != 0) : 1;
        int expected = -1;
        g_pool_on.compare_exchange_strong(expected, v);
        v = g_pool_on.load(std::memory_order_relaxed);
    }
    return v != 0;
}
#ifdef SD3D_WITH_PC
// pass-1 variant: 0 = lock-step / weight-stationary by the launcher's rules (rounds 1-5), 1 = the producer / consumer kernel wherever the
// lock-step kernel would run, 2 = everywhere (SD3D_PAIR_PC; sd3d_set_pair_pc at run time: tests compare the variants bit for bit in one process)
static std::atomic<int> g_pc_mode{-1};
static int pair_pc_mode() {
    int v = g_pc_mode.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("SD3D_PAIR_PC");
        v = e ? atoi(e) : SD3D_PAIR_PC_DEFAULT;
        int expected = -1;
        g_pc_mode.compare_exchange_strong(expected, v);
        v = g_pc_mode.load(std::memory_order_relaxed);
    }
    return v;
}
extern "C" int sd3d_set_pair_pc(int mode) {
    const int prev = pair_pc_mode();
    g_pc_mode.store(mode < 0 ? 0 : (mode > 2 ? 2 : mode), std::memory_order_relaxed);
    return prev;
}
template <class K>
static void pc_launch(K kern, dim3 grid, size_t lds, hipStream_t st, const PGParams& g, int n_os) {
    static thread_local const void* done[16];                  // (attribute set once per kernel and thread; the call is cheap and idempotent)
    bool seen = false;
    for (const void* d : done) if (d == (const void*)kern) seen = true;
    if (!seen) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);   // (the kernel also has a few static bytes)
        for (auto& d : done) if (!d) { d = (const void*)kern; break; }
    }
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, g, n_os);
}
// -> true when the producer / consumer kernel took the launch
static bool launch_pair_pc(PGParams& g, int nt, int cgs, int mode_kernel, int n_cu, int64_t n_tiles, hipStream_t st) {
    const int n_os = (g.Cin >> 5) == 1 ? 2 : 1;                // one-step tiles finish back to back: two staging slots
    const size_t lds = pair_pc_lds_bytes(nt, n_os);
    if (lds > 160 * 1024 - 256) return false;
    int gx = n_cu / cgs;
    gx = gx < 1 ? 1 : (gx < n_tiles ? gx : (int)n_tiles);
    { static int gx_env = -1; if (gx_env < 0) { const char* e = getenv("SD3D_PAIR_GX"); gx_env = e ? atoi(e) : 0; } if (gx_env > 0) gx = gx_env; }
    const dim3 grid((unsigned)gx, (unsigned)cgs);
    if (mode_kernel == 0) {
        switch (nt) {
            case 1: pc_launch(pair_gemm_pc_kernel_1, grid, lds, st, g, n_os); break;
            case 2: pc_launch(pair_gemm_pc_kernel_2, grid, lds, st, g, n_os); break;
            case 3: pc_launch(pair_gemm_pc_kernel_3, grid, lds, st, g, n_os); break;
            default: pc_launch(pair_gemm_pc_kernel_4, grid, lds, st, g, n_os); break;
        }
    } else if (mode_kernel == 1) {
        switch (nt) {
            case 1: pc_launch(pair_gemm_pc_direct_kernel_1, grid, lds, st, g, n_os); break;
            case 2: pc_launch(pair_gemm_pc_direct_kernel_2, grid, lds, st, g, n_os); break;
            case 3: pc_launch(pair_gemm_pc_direct_kernel_3, grid, lds, st, g, n_os); break;
            default: pc_launch(pair_gemm_pc_direct_kernel_4, grid, lds, st, g, n_os); break;
        }
    } else {
        switch (nt) {
            case 1: pc_launch(pair_dense_pc_kernel_1, grid, lds, st, g, n_os); break;
            case 2: pc_launch(pair_dense_pc_kernel_2, grid, lds, st, g, n_os); break;
            case 3: pc_launch(pair_dense_pc_kernel_3, grid, lds, st, g, n_os); break;
            default: pc_launch(pair_dense_pc_kernel_4, grid, lds, st, g, n_os); break;
        }
    }
    return true;
}
#endif  // SD3D_WITH_PC
extern "C" int sd3d_set_pair_pool(int on) {
    const int prev = pool_enabled() ? 1 : 0;
    g_pool_on.store(on ? 1 : 0, std::memory_order_relaxed);
    return prev;
}
// Number of counter words of the current device's ring in a state no sequence of finished launches leaves behind (an epoch the host has
// not handed out for that slot, or anything in a slot no launch has used) + the workgroups that met such a word since the last call (they
// fell back to a static split of their pool: the results are unaffected, the event is reported here and through sd3d_last_error);
// 0 is the healthy answer.  Synchronises the device.  < 0: error.
extern "C" int sd3d_pair_pool_check(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PG_MAX_DEVICES) { sd3d_set_error(SD3D_ERR_LAUNCH, "pair_pool_check: no device"); return -1; }
    if (hipDeviceSynchronize() != hipSuccess) { sd3d_set_error(SD3D_ERR_LAUNCH, "pair_pool_check: device synchronisation failed"); return -1; }
    unsigned int zero = 0, bad = 0, seen = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_pool_bad), &zero, sizeof(zero)) != hipSuccess) { sd3d_set_error(SD3D_ERR_LAUNCH, "pair_pool_check: symbol write failed"); return -1; }
    hipLaunchKernelGGL(pool_check_kernel, dim3(PG_POOL_SLOTS / 256), dim3(256), 0, nullptr, g_pool_tickets[dev].load(std::memory_order_relaxed));
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
        hipMemcpyFromSymbol(&bad, HIP_SYMBOL(g_pool_bad), sizeof(bad)) != hipSuccess ||
        hipMemcpyFromSymbol(&seen, HIP_SYMBOL(g_pool_bad_seen), sizeof(seen)) != hipSuccess ||
        hipMemcpyToSymbol(HIP_SYMBOL(g_pool_bad_seen), &zero, sizeof(zero)) != hipSuccess) { sd3d_set_error(SD3D_ERR_LAUNCH, "pair_pool_check: check kernel failed"); return -1; }
    if (seen) sd3d_set_error(SD3D_ERR_LAUNCH, "pair_pool_check: launches met counter words no launch can have written (they fell back to a static split of their shared tail: results unaffected)");
    bad += seen;                                               // (workgroups that fell back since the last check)
    return (int)bad;
}
// pooled launches so far on the current device (tests: did the pool really run?)
extern "C" int sd3d_pair_pool_launches(int64_t* launches_out) {
    int dev = 0;
    if (!launches_out || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PG_MAX_DEVICES) return sd3d_set_error(SD3D_ERR_ARG, "pair_pool_launches: no device / no output");
    *launches_out = (int64_t)g_pool_tickets[dev].load(std::memory_order_relaxed);
    return SD3D_OK;
}
// Test hook: overwrite the whole ring of the current device with `word` (a launch that died half-way, a foreign write) and forget
// nothing else - the next pooled launches must still produce the same bits (tests/test_gpu_pair_paths.py).
extern "C" int sd3d_pair_pool_poison(int64_t word) {
    unsigned long long* base = nullptr;
    if (hipGetSymbolAddress((void**)&base, HIP_SYMBOL(g_pool_ctr)) != hipSuccess) return sd3d_set_error(SD3D_ERR_LAUNCH, "pair_pool_poison: no symbol");
    std::vector<unsigned long long> h((size_t)PG_POOL_SLOTS * PG_POOL_WORDS, (unsigned long long)word);
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(base, h.data(), h.size() * sizeof(unsigned long long), hipMemcpyHostToDevice) != hipSuccess)
        return sd3d_set_error(SD3D_ERR_LAUNCH, "pair_pool_poison: copy failed");
    return SD3D_OK;
}

// ---- launchers --------------------------------------------------------------------------------
size_t pair_lists_ws_bytes(int K, int64_t M) {
    const int64_t nblk = cdiv(M, PL_ROWS);
    const size_t plain = (size_t)((int64_t)K * nblk + K) * sizeof(int32_t) + 256;
    const size_t chained = (K & 1) ? chain_lists_ws_bytes(K, M) : 0;      // (a table may be built either way: size for all)
    const size_t rows = (size_t)((int64_t)K * cdiv(M, 256) + K) * sizeof(int32_t) + 256;
    const size_t m = plain > chained ? plain : chained;
    return m > rows ? m : rows;
}

// p_cap: capacity of in_idx in pairs (multiple of 128, >= pairs + K * 127); tile_k has p_cap / 128 + 1 entries
// (the last one receives the number of real tiles).
int launch_pair_lists(const int32_t* nbr, int K, int64_t M, int64_t p_cap, int32_t* pos, int32_t* in_idx, int32_t* tile_k,
                      void* ws, size_t ws_bytes, hipStream_t st) {
    if (K <= 0 || M <= 0) return SD3D_OK;
    if (p_cap <= 0 || (p_cap % PT)) return sd3d_set_error(SD3D_ERR_ARG, "pair_lists: p_cap must be a positive multiple of 128");
    if (ws_bytes < pair_lists_ws_bytes(K, M)) return sd3d_set_error(SD3D_ERR_ARG, "pair_lists: workspace too small");
    const int nblk = (int)cdiv(M, PL_ROWS);
    int32_t* blk_cnt = (int32_t*)ws;
    int32_t* totals = blk_cnt + (int64_t)K * nblk;
    if (hipMemsetAsync(in_idx, 0xFF, (size_t)p_cap * sizeof(int32_t), st) != hipSuccess ||
        hipMemsetAsync(tile_k, 0xFF, (size_t)(p_cap / PT + 1) * sizeof(int32_t), st) != hipSuccess)
        return sd3d_set_error(SD3D_ERR_LAUNCH, "pair_lists: memset failed");
    hipLaunchKernelGGL(pair_count_kernel, dim3(nblk, K), dim3(256), 0, st, nbr, M, nblk, blk_cnt);
    hipLaunchKernelGGL(pair_scan_kernel, dim3(K), dim3(256), 0, st, blk_cnt, nblk, totals);
    hipLaunchKernelGGL(pair_fill_kernel, dim3(nblk, K), dim3(256), 0, st, nbr, K, M, nblk, blk_cnt, totals, p_cap, pos, in_idx, tile_k);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// n tables at once (n <= PL_MAX_TABLES); the tables' scratch sits back to back in ws (pair_lists_ws_bytes(K_i, M_i) bytes each,
// rounded up to 256).  rlist / out_idx / the two centre slots of tile_k are optional products (see sd3d_pair_table_desc).
int launch_pair_lists_desc(int n, const sd3d_pair_table_desc* d, void* ws, size_t ws_bytes, hipStream_t st) {
    if (n <= 0) return SD3D_OK;
    if (n > PL_MAX_TABLES) return sd3d_set_error(SD3D_ERR_ARG, "pair_lists_batch: at most 16 tables per call");
    PLBatch b, rbt;                                            // (offset, row block) form / row-block form (no pos table)
    b.n = 0;
    rbt.n = 0;
    int rwg = 0, rkk = 0;
    CHBatch cb;
    cb.n = 0;
    size_t off = 0;
    int wg = 0, kk = 0, rb = 0;
    int cwg = 0, csg = 0;
    for (int i = 0; i < n; ++i) {
        const int K = d[i].K;
        const int64_t M = d[i].M, p_cap = d[i].p_cap;
        if (d[i].center == SD3D_PAIR_CHAINED && K > 0 && M > 0) {
            // chained lists (mirror groups + centre share a partial product): their own builder
            if (!(K & 1) || K < 3) return sd3d_set_error(SD3D_ERR_ARG, "pair_lists_batch: chained lists need an odd kernel (symmetric offsets)");
            if (K / 2 > CH_MAX_G) return sd3d_set_error(SD3D_ERR_ARG, "pair_lists_batch: chained lists take kernels up to 5^3");
            if (p_cap <= 0 || (p_cap % PT) || !d[i].rlist || d[i].rl_stride < K / 2 + 2 || (d[i].rl_stride & 3))
                return sd3d_set_error(SD3D_ERR_ARG, "pair_lists_batch: chained lists need rlist and rl_stride >= K / 2 + 2");
            CHTable& T = cb.t[cb.n++];
            T.nbr = d[i].nbr; T.in_idx = d[i].in_idx; T.tile_k = d[i].tile_k; T.rlist = d[i].rlist; T.M = M; T.p_cap = p_cap;
            T.K = K; T.G = K / 2; T.nblk = (int)cdiv(M, CH_ROWS); T.rl_stride = d[i].rl_stride;
            const int nseg = T.G * CH_NPAT + 1;
            T.blk_cnt = (int32_t*)((char*)ws + off);
            T.totals = T.blk_cnt + (int64_t)nseg * T.nblk;
            off += align_up(chain_lists_ws_bytes(K, M), 256);
            T.wg0 = cwg; T.sg0 = csg; T.lean = (d[i].meta & 2) ? 1 : 0;
            cwg += T.nblk; csg += nseg;
            continue;
        }
        if (K <= 0 || M <= 0) {                                // no rows: a later pair_conv on this table must see "0 real tiles"
            if (d[i].tile_k && p_cap > 0 && hipMemsetAsync(d[i].tile_k + p_cap / PT, 0, ((d[i].meta & 1) ? 3 : 1) * sizeof(int32_t), st) != hipSuccess)
                return sd3d_set_error(SD3D_ERR_LAUNCH, "pair_lists_batch: memset failed");
            continue;
        }
        if (p_cap <= 0 || (p_cap % PT)) return sd3d_set_error(SD3D_ERR_ARG, "pair_lists_batch: p_cap must be a positive multiple of 128");
        if (d[i].rlist && (d[i].rl_stride < K + 4 || (d[i].rl_stride & 3))) return sd3d_set_error(SD3D_ERR_ARG, "pair_lists_batch: rl_stride must be a multiple of 4, >= K + 4");
        if (d[i].center >= 0) return sd3d_set_error(SD3D_ERR_ARG, "pair_lists_batch: center must be -1 or SD3D_PAIR_CHAINED");
        if (!d[i].pos && !d[i].rlist && !d[i].out_idx)
            return sd3d_set_error(SD3D_ERR_ARG, "pair_lists_batch: a table without pos needs rlist or out_idx (nothing could run pass 2 on it)");
        if (!d[i].pos && K <= PR_MAX_K) {                      // no position table wanted: the row-block form (three launches, the rows' lists from the fill)
            PLTable& T = rbt.t[rbt.n++];
            T.nbr = d[i].nbr; T.pos = nullptr; T.in_idx = d[i].in_idx; T.tile_k = d[i].tile_k; T.M = M; T.p_cap = p_cap; T.K = K;
            T.rlist = d[i].rlist; T.out_idx = d[i].out_idx; T.rl_stride = d[i].rl_stride; T.meta = d[i].meta;
            T.nblk = (int)cdiv(M, 256);
            T.blk_cnt = (int32_t*)((char*)ws + off);
            T.totals = T.blk_cnt + (int64_t)T.K * T.nblk;
            off += align_up(pair_lists_ws_bytes(K, M), 256);
            T.wg0 = rwg; T.k0 = rkk; T.rb0 = 0;
            rwg += T.nblk; rkk += T.K;
            continue;
        }
        if (!d[i].pos && d[i].rlist) return sd3d_set_error(SD3D_ERR_ARG, "pair_lists_batch: row lists without pos take kernels up to 128 offsets");
        PLTable& T = b.t[b.n++];
        T.nbr = d[i].nbr; T.pos = d[i].pos; T.in_idx = d[i].in_idx; T.tile_k = d[i].tile_k; T.M = M; T.p_cap = p_cap; T.K = K;
        T.rlist = d[i].rlist; T.out_idx = d[i].out_idx; T.rl_stride = d[i].rl_stride; T.meta = d[i].meta;
        T.nblk = (int)cdiv(M, PL_ROWS);
        T.blk_cnt = (int32_t*)((char*)ws + off);
        T.totals = T.blk_cnt + (int64_t)T.K * T.nblk;
        off += align_up(pair_lists_ws_bytes(K, M), 256);
        T.wg0 = wg; T.k0 = kk; T.rb0 = rb;
        wg += T.K * T.nblk; kk += T.K;
        rb += T.rlist ? (int)cdiv(M, RL_ROWS) : 0;
    }
    if (off > ws_bytes) return sd3d_set_error(SD3D_ERR_ARG, "pair_lists_batch: workspace too small");
    if (cb.n > 0 && rbt.n > 0) {
        hipLaunchKernelGGL(lists_count_kernel, dim3(cwg + rwg), dim3(256), 0, st, cb, rbt, cwg);
        hipLaunchKernelGGL(lists_scan_kernel, dim3(csg + rkk), dim3(256), 0, st, cb, rbt, csg);
        hipLaunchKernelGGL(lists_fill_kernel, dim3(cwg + rwg), dim3(256), 0, st, cb, rbt, cwg);
        SD3D_CHECK_LAUNCH();
    } else if (cb.n > 0) {
        hipLaunchKernelGGL(chain_count_kernel, dim3(cwg), dim3(256), 0, st, cb);
        hipLaunchKernelGGL(chain_scan_kernel, dim3(csg), dim3(256), 0, st, cb);
        hipLaunchKernelGGL(chain_fill_kernel, dim3(cwg), dim3(256), 0, st, cb);
        SD3D_CHECK_LAUNCH();
    } else if (rbt.n > 0) {
        hipLaunchKernelGGL(pair_count_rows_kernel, dim3(rwg), dim3(256), 0, st, rbt);
        hipLaunchKernelGGL(pair_scan_batch_kernel, dim3(rkk), dim3(256), 0, st, rbt);
        hipLaunchKernelGGL(pair_fill_rows_kernel, dim3(rwg), dim3(256), 0, st, rbt);
        SD3D_CHECK_LAUNCH();
    }
    if (b.n == 0) return SD3D_OK;
    hipLaunchKernelGGL(pair_count_batch_kernel, dim3(wg), dim3(256), 0, st, b);
    hipLaunchKernelGGL(pair_scan_batch_kernel, dim3(kk), dim3(256), 0, st, b);
    hipLaunchKernelGGL(pair_fill_batch_kernel, dim3(wg), dim3(256), 0, st, b);
    if (rb > 0) hipLaunchKernelGGL(pair_rowlist_batch_kernel, dim3(rb), dim3(RL_ROWS), 0, st, b);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

int launch_pair_lists_batch(int n, const int32_t* const* nbr, const int* K, const int64_t* M, const int64_t* p_cap, int32_t* const* pos,
                            int32_t* const* in_idx, int32_t* const* tile_k, void* ws, size_t ws_bytes, hipStream_t st) {
    if (n > PL_MAX_TABLES) return sd3d_set_error(SD3D_ERR_ARG, "pair_lists_batch: at most 16 tables per call");
    sd3d_pair_table_desc d[PL_MAX_TABLES];
    for (int i = 0; i < n; ++i) {
        d[i].nbr = nbr[i]; d[i].pos = pos[i]; d[i].in_idx = in_idx[i]; d[i].tile_k = tile_k[i]; d[i].rlist = nullptr; d[i].out_idx = nullptr;
        d[i].M = M[i]; d[i].p_cap = p_cap[i]; d[i].K = K[i]; d[i].center = -1; d[i].rl_stride = 0; d[i].meta = 0;
    }
    return launch_pair_lists_desc(n, d, ws, ws_bytes, st);
}

static int env_flag(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

// rlist / rl_stride / center / out_idx: optional products of launch_pair_lists_desc (NULL / -1: the round-1 path, pass 1 over all
// offsets + pair_reduce_kernel over pos).  out_idx != NULL promises ONE pair per output row (transposed k2s2 convolution).
int launch_pair_conv(const float* in0, int ld0, int C0, const float* in1, int ld1, const int32_t* in_idx, const int32_t* tile_k,
                     int64_t p_cap, const int32_t* pos, const int32_t* rlist, int rl_stride, int center, const int32_t* out_idx,
                     const float* wt, int K, int Cin, int Cout, int64_t M, const float* scale,
                     const float* shift, const float* res, int ld_res, float* out, int ld_out, int act, float* part,
                     size_t part_bytes, hipStream_t st) {
    if (M <= 0 || Cout <= 0) return SD3D_OK;
    if (Cin <= 0 || (Cin & 31)) return sd3d_set_error(SD3D_ERR_ARG, "pair_conv: Cin must be a positive multiple of 32");
    if (Cout & 3) return sd3d_set_error(SD3D_ERR_ARG, "pair_conv: Cout must be a multiple of 4");
    if (in1 && ((C0 & 31) || C0 > Cin)) return sd3d_set_error(SD3D_ERR_ARG, "pair_conv: concat split must be a multiple of 32");
    if (!in1) C0 = Cin;
    if ((ld0 & 3) || (in1 && (ld1 & 3)) || (ld_out & 3)) return sd3d_set_error(SD3D_ERR_ARG, "pair_conv: row strides must be multiples of 4 floats");
    if (p_cap <= 0 || (p_cap % PT)) return sd3d_set_error(SD3D_ERR_ARG, "pair_conv: p_cap must be a positive multiple of 128");
    if (part_bytes < (size_t)p_cap * Cout * sizeof(float)) return sd3d_set_error(SD3D_ERR_ARG, "pair_conv: partial-product buffer too small");
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return sd3d_set_error(SD3D_ERR_LAUNCH, "pair_conv: no device");
        n_cu = prop.multiProcessorCount;
    }
    // SD3D_PAIR_DIRECT / SD3D_PAIR_RL = 0 switch the round-3 paths off one by one (A/B, cross-checks in the tests)
    static const int direct_env = env_flag("SD3D_PAIR_DIRECT", 1), rl_env = env_flag("SD3D_PAIR_RL", 1);
    if ((ld_res & 3) && res) return sd3d_set_error(SD3D_ERR_ARG, "pair_conv: residual row stride must be a multiple of 4 floats");
    const bool mirror_w = center == SD3D_PAIR_MIRROR_W(-1) || center == SD3D_PAIR_MIRROR_W(SD3D_PAIR_CHAINED);
    if (mirror_w) center += 2;
    if (center >= 0 || center < SD3D_PAIR_CHAINED) return sd3d_set_error(SD3D_ERR_ARG, "pair_conv: center must be -1 or SD3D_PAIR_CHAINED, or SD3D_PAIR_MIRROR_W of them (the dense centre kernel of round 3 left the library: profiles/EXPERIMENTS.md)");
    const bool direct = out_idx != nullptr && direct_env;
    // (lean evaluation tables carry no position table, 'up' tables no row lists: with SD3D_PAIR_RL=0 / SD3D_PAIR_DIRECT=0, or a C caller that
    //  passes neither, pass 2 would walk pos == NULL - refuse before anything is launched.  ADVICE r5)
    if (!direct && !(rlist && rl_env) && !pos)
        return sd3d_set_error(SD3D_ERR_ARG, "pair_conv: pass 2 needs rlist (SD3D_PAIR_RL=1), or pos, or out_idx with the direct epilogue (SD3D_PAIR_DIRECT=1); "
                                            "this table was built without a position table (lean evaluation lists)");
    PGParams g;
    g.in0 = in0; g.ld0 = ld0; g.C0 = C0; g.in1 = in1; g.ld1 = ld1; g.in_idx = in_idx; g.tile_k = tile_k; g.wt = wt;
    g.Cin = Cin; g.Cout = Cout; g.part = part;
    g.chained = center == SD3D_PAIR_CHAINED ? 1 : 0;
    {
        static const int nt_env = env_flag("SD3D_PAIR_NT_STORE", -1);          // -1: by the scenes-in-flight hint
        g.nt_part = nt_env >= 0 ? (nt_env != 0) : (g_scenes_in_flight.load(std::memory_order_relaxed) > 1 ? 1 : 0);
    }
    g.dense_rows = 0;
    g.pool_ctr = nullptr;
    g.pool_epoch = 0;
    g.k_flip = mirror_w ? K - 1 : -1;
#ifdef PG_ABLATE
    g.dbg = pg_dbg_env();
#endif
    g.out_idx = direct ? out_idx : nullptr;
    g.scale = scale; g.shift = shift; g.res = res; g.ld_res = ld_res; g.out = out; g.ld_out = ld_out; g.act = act;
    const int sub = (Cout + 31) / 32;
    int nt = sub >= 4 ? 4 : sub;
    if (sub > 4 && sub % 4) { for (int c = 4; c >= 1; --c) if (sub % c == 0) { nt = c; break; } }
    const int cgs = (int)cdiv(sub, nt);
    g.n_tiles = (int)(p_cap / PT);
    static int slots_env = -1;                                 // SD3D_PAIR_SLOTS: workgroups per CU override (tuning)
    if (slots_env < 0) { const char* e = getenv("SD3D_PAIR_SLOTS"); slots_env = e ? atoi(e) : 0; }
    // weight-stationary variant: Cout = 32 * nt <= 128 and W[k] (padded rows) <= 68 KB of LDS
    static int ws_env = -1;
    if (ws_env < 0) { const char* e = getenv("SD3D_PAIR_WS"); ws_env = e ? atoi(e) : 1; }
    const size_t w_lds = (size_t)Cout * (Cin + 4) * sizeof(float);
    const size_t w_lds_cg = (size_t)(32 * nt) * (Cin + 4) * sizeof(float);       // one column group's share of W[k]
    static int ws_mask = -1;                                   // SD3D_PAIR_WS_MASK: bit (nt - 1) allows the weight-stationary variant for nt column tiles
    if (ws_mask < 0) { const char* e = getenv("SD3D_PAIR_WS_MASK"); ws_mask = e ? atoi(e) : 15; }
    // several scenes in flight: the 96 / 128-column weight-stationary variants (47-63 KB of LDS per workgroup, 2-3 per CU) leave
    // no room for the other scenes' kernels; measured with 4 scenes in flight 107 -> 112 scenes/s without them, while alone on the
    // GPU they are 0.3 ms per forward faster.  SD3D_PAIR_CROWD=0/1 overrides the hint.
    static int crowd_env = -2;
    if (crowd_env == -2) { const char* e = getenv("SD3D_PAIR_CROWD"); crowd_env = e ? atoi(e) : -1; }
    const bool crowded = crowd_env >= 0 ? crowd_env != 0 : g_scenes_in_flight.load(std::memory_order_relaxed) > 1;
    const bool ws_one = ws_env && !g.chained && ((ws_mask >> (nt - 1)) & 1) && !(crowded && nt >= 3) && cgs == 1 && Cout == 32 * nt && w_lds <= 68 * 1024;
    // wide layers (256 columns = two groups of 128): the group's half of W[k] (133 KB at Cin = 256) still fits LDS with ONE
    // workgroup per CU - no per-step barrier, no re-staging of weight chunks, like the narrow layers
    static int ws2_env = -1;
    if (ws2_env < 0) { const char* e = getenv("SD3D_PAIR_WS2"); ws2_env = e ? atoi(e) : 1; }
    // (measured: level-3 256->256, 1808 tiles: 358 -> 347 us; level-4, 472 tiles: 104 -> 113 us - too few tiles for half the workgroups)
    const bool ws_two = ws_env && !g.chained && ws2_env && !crowded && cgs == 2 && nt == 4 && Cout == 256 && g.n_tiles >= 1024 && w_lds_cg + WS_RANGE_TILES * PT * sizeof(int32_t) + 256 <= 160 * 1024;
    auto setup_pool = [&]() {
    if (pool_enabled() && cgs <= PG_POOL_WORDS) {           // the shared tail (sd3d_set_pair_pool / SD3D_PAIR_POOL=0: every tile dealt out statically, rounds 1-4)
        // (the counters are a per-device symbol: a process that drives several GPUs gets each device's own address and its own tickets)
        int dev = 0;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < PG_MAX_DEVICES &&
            (st == nullptr || (hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone))) {
            unsigned long long* base = g_pool_bases[dev].load(std::memory_order_acquire);
            if (!base && hipGetSymbolAddress((void**)&base, HIP_SYMBOL(g_pool_ctr)) == hipSuccess)
                g_pool_bases[dev].store(base, std::memory_order_release);
            if (base) {
                const unsigned long long t = g_pool_tickets[dev].fetch_add(1, std::memory_order_relaxed);
                g.pool_ctr = base + (size_t)(t % PG_POOL_SLOTS) * PG_POOL_WORDS;
                g.pool_epoch = (unsigned int)(t / PG_POOL_SLOTS + 1);
            }
        }
    }
    };
    bool pc_done = false;
#ifdef SD3D_WITH_PC
    const int pc_mode = pair_pc_mode();
    if (pc_mode == 2 || (pc_mode == 1 && !(ws_one || ws_two))) {
        setup_pool();
        pc_done = launch_pair_pc(g, nt, cgs, direct ? 1 : 0, n_cu, g.n_tiles, st);
        if (!pc_done) { g.pool_ctr = nullptr; g.pool_epoch = 0; }
    }
#endif
    if (pc_done) {
    } else if (ws_one || ws_two) {
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute((const void*)pair_gemm_ws_kernel_1, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            (void)hipFuncSetAttribute((const void*)pair_gemm_ws_kernel_2, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            (void)hipFuncSetAttribute((const void*)pair_gemm_ws_kernel_3, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            (void)hipFuncSetAttribute((const void*)pair_gemm_ws_kernel_4, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)pair_gemm_ws_direct_kernel_1, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            (void)hipFuncSetAttribute((const void*)pair_gemm_ws_direct_kernel_2, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            (void)hipFuncSetAttribute((const void*)pair_gemm_ws_direct_kernel_3, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            (void)hipFuncSetAttribute((const void*)pair_gemm_ws_direct_kernel_4, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_done = true;
        }
        // resident workgroups per CU: the pinned register budgets allow 4 / 3 / 3 / 2; LDS (160 KB) may allow fewer
        int per_cu = nt == 1 ? 4 : (nt == 3 ? 3 : 2);
        const size_t w_need = ws_two ? w_lds_cg : w_lds;
        const int by_lds = (int)((160 * 1024) / (w_need + WS_RANGE_TILES * PT * sizeof(int32_t) + 256));
        per_cu = per_cu < by_lds ? per_cu : by_lds;
        if (slots_env > 0) per_cu = slots_env;
        int gx = n_cu * per_cu / (ws_two ? 2 : 1);
        gx = gx < 1 ? 1 : (gx < g.n_tiles ? gx : g.n_tiles);
        { static int gx_env = -1; if (gx_env < 0) { const char* e = getenv("SD3D_PAIR_GX"); gx_env = e ? atoi(e) : 0; } if (gx_env > 0) gx = gx_env; }
        const dim3 wgrid((unsigned)gx, ws_two ? 2u : 1u);
        const size_t lds = w_need + (size_t)WS_RANGE_TILES * PT * sizeof(int32_t);
        if (direct) {
            switch (nt) {
                case 1: hipLaunchKernelGGL(pair_gemm_ws_direct_kernel_1, wgrid, dim3(256), lds, st, g); break;
                case 2: hipLaunchKernelGGL(pair_gemm_ws_direct_kernel_2, wgrid, dim3(256), lds, st, g); break;
                case 3: hipLaunchKernelGGL(pair_gemm_ws_direct_kernel_3, wgrid, dim3(256), lds, st, g); break;
                default: hipLaunchKernelGGL(pair_gemm_ws_direct_kernel_4, wgrid, dim3(256), lds, st, g); break;
            }
        } else {
            switch (nt) {
                case 1: hipLaunchKernelGGL(pair_gemm_ws_kernel_1, wgrid, dim3(256), lds, st, g); break;
                case 2: hipLaunchKernelGGL(pair_gemm_ws_kernel_2, wgrid, dim3(256), lds, st, g); break;
                case 3: hipLaunchKernelGGL(pair_gemm_ws_kernel_3, wgrid, dim3(256), lds, st, g); break;
                default: hipLaunchKernelGGL(pair_gemm_ws_kernel_4, wgrid, dim3(256), lds, st, g); break;
            }
        }
    } else {
        int per_cu = nt == 1 ? 3 : 2;
        if (slots_env > 0) per_cu = slots_env;
        int gx = n_cu * per_cu / cgs;
        gx = gx < 1 ? 1 : (gx < g.n_tiles ? gx : g.n_tiles);
        const dim3 grid((unsigned)gx, (unsigned)cgs);
        setup_pool();
        if (direct) {
            switch (nt) {
                case 1: hipLaunchKernelGGL(pair_gemm_direct_kernel_1, grid, dim3(256), 0, st, g); break;
                case 2: hipLaunchKernelGGL(pair_gemm_direct_kernel_2, grid, dim3(256), 0, st, g); break;
                case 3: hipLaunchKernelGGL(pair_gemm_direct_kernel_3, grid, dim3(256), 0, st, g); break;
                default: hipLaunchKernelGGL(pair_gemm_direct_kernel_4, grid, dim3(256), 0, st, g); break;
            }
        } else {
            switch (nt) {
                case 1: hipLaunchKernelGGL(pair_gemm_kernel_1, grid, dim3(256), 0, st, g); break;
                case 2: hipLaunchKernelGGL(pair_gemm_kernel_2, grid, dim3(256), 0, st, g); break;
                case 3: hipLaunchKernelGGL(pair_gemm_kernel_3, grid, dim3(256), 0, st, g); break;
                default: hipLaunchKernelGGL(pair_gemm_kernel_4, grid, dim3(256), 0, st, g); break;
            }
        }
    }
    if (direct) {                                              // pass 1 wrote the output rows
    } else if (rlist && rl_env) {               
        PRLParams r;
        r.rlist = rlist; r.rl_stride = rl_stride; r.M = M; r.part = part; r.Cout = Cout; r.scale = scale; r.shift = shift; r.res = res;
        r.ld_res = ld_res; r.out = out; r.ld_out = ld_out; r.act = act;
        hipLaunchKernelGGL(pair_reduce_rl_kernel, dim3((unsigned)cdiv(M * (Cout / 4), 256)), dim3(256), 0, st, r);
    } else {
        PRParams r;
        r.pos = pos; r.K = K; r.M = M; r.part = part; r.Cout = Cout; r.scale = scale; r.shift = shift; r.res = res; r.ld_res = ld_res;
        r.out = out; r.ld_out = ld_out; r.act = act;
        hipLaunchKernelGGL(pair_reduce_kernel, dim3((unsigned)cdiv(M * (Cout / 4), 256)), dim3(256), 0, st, r);
    }
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// A dense [M, Cin] x W^T product (+ scale / shift / residual / activation) on the pass-1 kernel: the 1x1 convolutions of the U-Net
// (BasicBlockBase.downsample, minkunet.py:314-328) on tens of thousands of rows.  The one-tile-per-workgroup kernel of gather_gemm.hip
// spends most of such a launch on its prologue and epilogue (Cin / 32 = 4 - 12 steps per workgroup: 49 TFLOP/s at level 0); here a
// persistent workgroup walks its share of the 128-row tiles through the same software pipeline as the sparse convolutions, the
// "rulebook" being the identity (no lists are read) and the epilogue writing the output rows themselves.
int launch_pair_dense(const GGParams& q, hipStream_t st) {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return sd3d_set_error(SD3D_ERR_LAUNCH, "pair_dense: no device");
        n_cu = prop.multiProcessorCount;
    }
    PGParams g;
    g.in0 = q.in0; g.ld0 = q.ld0; g.C0 = q.in1 ? q.C0 : q.Cin; g.in1 = q.in1; g.ld1 = q.ld1; g.in_idx = nullptr; g.tile_k = nullptr; g.wt = q.wt;
    g.Cin = q.Cin; g.Cout = q.Cout; g.part = nullptr; g.n_tiles = 0; g.out_idx = nullptr; g.scale = q.scale; g.shift = q.shift; g.res = q.res;
    g.ld_res = q.ld_res; g.out = q.out; g.ld_out = q.ld_out; g.act = q.act; g.nt_part = 0; g.chained = 0; g.dense_rows = q.M; g.pool_ctr = nullptr; g.pool_epoch = 0; g.k_flip = -1;
#ifdef PG_ABLATE
    g.dbg = pg_dbg_env();
#endif
    const int sub = (q.Cout + 31) / 32;
    int nt = sub >= 4 ? 4 : sub;
    if (sub > 4 && sub % 4) { for (int c = 4; c >= 1; --c) if (sub % c == 0) { nt = c; break; } }
    const int cgs = (int)cdiv(sub, nt);
    const int64_t tiles = cdiv(q.M, PT);
#ifdef SD3D_WITH_PC
    if (pair_pc_mode() >= 1 && launch_pair_pc(g, nt, cgs, 2, n_cu, tiles, st)) {
        SD3D_CHECK_LAUNCH();
        return SD3D_OK;
    }
#endif
    int gx = n_cu * (nt == 1 ? 3 : 2) / cgs;
    gx = gx < 1 ? 1 : (gx < tiles ? gx : (int)tiles);
    const dim3 grid((unsigned)gx, (unsigned)cgs);
    switch (nt) {
        case 1: hipLaunchKernelGGL(pair_dense_kernel_1, grid, dim3(256), 0, st, g); break;
        case 2: hipLaunchKernelGGL(pair_dense_kernel_2, grid, dim3(256), 0, st, g); break;
        case 3: hipLaunchKernelGGL(pair_dense_kernel_3, grid, dim3(256), 0, st, g); break;
        default: hipLaunchKernelGGL(pair_dense_kernel_4, grid, dim3(256), 0, st, g); break;
    }
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
