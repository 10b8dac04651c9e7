// Stable LSD radix sort of (u64 key, u32 value) pairs + exclusive scan, hand-written for wave64.
//
// Used by: voxelisation (sort points along the Z-order curve), superpoint pooling (group points by
// superpoint id) and post-processing (top-k / score ordering).  These are HBM-bound integer passes
// over <= a few million elements; 8-bit digits, one histogram + one scatter kernel per digit (the
// scatter derives its cursors from the raw [256][n_blocks] histogram itself); arrays of <= 4096 elements are ranked by one workgroup.  Stability (needed so that points inside one voxel /
// superpoint stay in ascending point order => deterministic fp32 sums downstream) comes from
// ranking each wave's elements in chunk order with ballot-built match masks.
#include "common.h"

#define RS_TILE 2048          // elements per workgroup (4 waves x 8 chunks x 64 lanes)
#define RS_THREADS 256
#define SCAN_TILE 2048

// ----------------------------------------------------------------------------------------------
// block-wide exclusive scan of one int per thread (256 threads)
// ----------------------------------------------------------------------------------------------
__device__ static inline int block_excl_scan_256(int v, int* total, int* smem4) {
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

__global__ __launch_bounds__(256) void scan_tile_sums(const int* __restrict__ in, int64_t n_cap,
                                                      const int* __restrict__ n_dev, int* __restrict__ sums) {
    __shared__ int sm[4];
    const int64_t n = n_dev ? min((int64_t)*n_dev, n_cap) : n_cap;
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + threadIdx.x * 8;
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (base + i < n) ? in[base + i] : 0;
    int total;
    block_excl_scan_256(s, &total, sm);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// single workgroup: exclusive scan of `sums[0..nb)` in place, total -> *total_out
__global__ __launch_bounds__(256) void scan_sums_inplace(int* __restrict__ sums, int nb, int* __restrict__ total_out) {
    __shared__ int sm[4];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int start = 0; start < nb; start += 256 * 8) {
        const int base = start + threadIdx.x * 8;
        int v[8], s = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[i] = (base + i < nb) ? sums[base + i] : 0; s += v[i]; }
        int total;
        int ex = block_excl_scan_256(s, &total, sm) + carry_s;
#pragma unroll
        for (int i = 0; i < 8; ++i) { if (base + i < nb) sums[base + i] = ex; ex += v[i]; }
        __syncthreads();
        if (threadIdx.x == 0) carry_s += total;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry_s;
}

__global__ __launch_bounds__(256) void scan_apply(const int* __restrict__ in, int64_t n_cap, const int* __restrict__ n_dev,
                                                  const int* __restrict__ sums, int* __restrict__ out) {
    __shared__ int sm[4];
    const int64_t n = n_dev ? min((int64_t)*n_dev, n_cap) : n_cap;
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + threadIdx.x * 8;
    int v[8], s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = (base + i < n) ? in[base + i] : 0; s += v[i]; }
    int total;
    int ex = block_excl_scan_256(s, &total, sm) + sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < 8; ++i) { if (base + i < n_cap) out[base + i] = ex; ex += v[i]; }
}

// Second (and last) kernel of the tiled scan: every workgroup adds up the tile sums before it by itself (<= a few hundred
// ints, one wave-reduction) instead of waiting for a third, single-workgroup kernel to scan them; the last one stores the total.
__global__ __launch_bounds__(256) void scan_apply_v2(const int* __restrict__ in, int64_t n_cap, const int* __restrict__ n_dev,
                                                     const int* __restrict__ sums, int nb, int* __restrict__ out,
                                                     int* __restrict__ total_out) {
    __shared__ int sm[4];
    __shared__ int base_s;
    const int64_t n = n_dev ? min((int64_t)*n_dev, n_cap) : n_cap;
    {
        int part = 0;
        for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) part += sums[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = part;
        __syncthreads();
        if (threadIdx.x == 0) base_s = sm[0] + sm[1] + sm[2] + sm[3];
        __syncthreads();
    }
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + threadIdx.x * 8;
    int v[8], s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = (base + i < n) ? in[base + i] : 0; s += v[i]; }
    int total;
    int ex = block_excl_scan_256(s, &total, sm) + base_s;
#pragma unroll
    for (int i = 0; i < 8; ++i) { if (base + i < n_cap) out[base + i] = ex; ex += v[i]; }
    if (total_out && blockIdx.x == nb - 1 && threadIdx.x == 0) *total_out = base_s + total;
}

// One workgroup scans the whole array (n <= SCAN_SMALL_MAX = two 8 k-element rounds): a single launch.  (A single workgroup
// walking 150 k elements measured 113 us - 19 dependent rounds - against 14 us for the tiled path, so longer inputs stay tiled.)
#define SCAN_SMALL_MAX (1 << 14)
__global__ __launch_bounds__(1024) void scan_small_kernel(const int* __restrict__ in, int64_t n_cap, const int* __restrict__ n_dev,
                                                          int* __restrict__ out, int* __restrict__ total_out) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int64_t n = n_dev ? min((int64_t)*n_dev, n_cap) : n_cap;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t start = 0; start < n_cap; start += 1024 * 8) {
        const int64_t base = start + (int64_t)threadIdx.x * 8;
        int v[8], s = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[i] = (base + i < n) ? in[base + i] : 0; s += v[i]; }
        int inc = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(inc, d);
            if (lane >= d) inc += t;
        }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int wb = carry_s, tot = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int x = wsum[i]; if (i < w) wb += x; tot += x; }
        int ex = wb + inc - s;
#pragma unroll
        for (int i = 0; i < 8; ++i) { if (base + i < n_cap) out[base + i] = ex; ex += v[i]; }
        __syncthreads();
        if (threadIdx.x == 0) carry_s += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry_s;
}

size_t scan_ws_bytes(int64_t n) { return align_up((size_t)cdiv(n, SCAN_TILE) * sizeof(int), 256); }

// out may alias in.  n_dev (optional, device) = live length; elements beyond it count as zero.
int scan_exclusive_i32(const int* in, int* out, int64_t n_cap, const int* n_dev, int* total_dev, void* ws,
                       size_t ws_bytes, hipStream_t st) {
    if (n_cap <= 0) {
        if (total_dev) (void)hipMemsetAsync(total_dev, 0, sizeof(int), st);
        return SD3D_OK;
    }
    if (n_cap <= SCAN_SMALL_MAX) {
        hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(1024), 0, st, in, n_cap, n_dev, out, total_dev);
        SD3D_CHECK_LAUNCH();
        return SD3D_OK;
    }
    const int nb = (int)cdiv(n_cap, SCAN_TILE);
    if (ws_bytes < scan_ws_bytes(n_cap)) return sd3d_set_error(SD3D_ERR_WS, "scan workspace too small");
    int* sums = (int*)ws;
    hipLaunchKernelGGL(scan_tile_sums, dim3(nb), dim3(256), 0, st, in, n_cap, n_dev, sums);
    if (nb <= 4096) {
        hipLaunchKernelGGL(scan_apply_v2, dim3(nb), dim3(256), 0, st, in, n_cap, n_dev, sums, nb, out, total_dev);
    } else {
        hipLaunchKernelGGL(scan_sums_inplace, dim3(1), dim3(256), 0, st, sums, nb, total_dev);
        hipLaunchKernelGGL(scan_apply, dim3(nb), dim3(256), 0, st, in, n_cap, n_dev, sums, out);
    }
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// ----------------------------------------------------------------------------------------------
// radix sort
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RS_THREADS) void rs_hist(const uint64_t* __restrict__ keys, int64_t n, int shift,
                                                      int* __restrict__ hist, int nb) {
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int c = 0; c < RS_TILE / RS_THREADS; ++c) {
        const int64_t i = base + c * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(int)((keys[i] >> shift) & 0xFF)], 1);
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}

// Each wave owns 512 consecutive elements (8 chunks of 64) of the block's 2048-element tile.
__global__ __launch_bounds__(RS_THREADS) void rs_scatter(const uint64_t* __restrict__ keys_in,
                                                         const uint32_t* __restrict__ vals_in,
                                                         uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                         int64_t n, int shift, const int* __restrict__ hist_scanned, int nb) {
    __shared__ int cnt[4][256];     // per-wave digit counts, then running output cursors
    __shared__ int scan_sm[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 256; i += RS_THREADS) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const int64_t wbase = (int64_t)blockIdx.x * RS_TILE + (int64_t)w * 512;
    uint64_t k[8];
    uint32_t v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int64_t i = wbase + c * 64 + lane;
        const bool ok = i < n;
        k[c] = ok ? keys_in[i] : 0ull;
        v[c] = ok ? (vals_in ? vals_in[i] : (uint32_t)i) : 0u;
        if (ok) atomicAdd(&cnt[w][(int)((k[c] >> shift) & 0xFF)], 1);
    }
    __syncthreads();
    {   // thread d: turn counts into starting cursors: global base of (digit, block) + earlier waves.  The global base is the
        // exclusive prefix of the digit-major [256][nb] histogram at (d, this block) = (all blocks of smaller digits) + (earlier
        // blocks of this digit); every workgroup adds it up itself from the raw histogram (nb <= a few hundred loads per
        // thread, L2-resident) - that replaces a separate scan of the histogram, three launches per digit pass.
        const int d = threadIdx.x;
        const int* hrow = hist_scanned + (int64_t)d * nb;
        int before = 0, all = 0;
        for (int b0 = 0; b0 < nb; b0 += 8) {
            int h[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) h[i] = b0 + i < nb ? hrow[b0 + i] : 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) { all += h[i]; if (b0 + i < (int)blockIdx.x) before += h[i]; }
        }
        int tot;
        const int smaller = block_excl_scan_256(all, &tot, scan_sm);
        int run = smaller + before;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) { int c = cnt[ww][d]; cnt[ww][d] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int64_t i = wbase + c * 64 + lane;
        const bool ok = i < n;
        const int d = (int)((k[c] >> shift) & 0xFF);
        uint64_t same = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t bal = __ballot((d >> b) & 1);
            same &= ((d >> b) & 1) ? bal : ~bal;
        }
        const int rank = __popcll(same & ((1ull << lane) - 1ull));
        const int num = __popcll(same);
        int pos = 0;
        if (ok) pos = cnt[w][d];
        __builtin_amdgcn_wave_barrier();
        if (ok && rank == 0) cnt[w][d] = pos + num;
        __builtin_amdgcn_wave_barrier();
        if (ok) {
            keys_out[pos + rank] = k[c];
            vals_out[pos + rank] = v[c];
        }
    }
}

// n <= RANK_SORT_MAX: all key bits at once, no digit passes.  Every element counts the elements that sort before it (smaller
// key, or equal key and smaller index: stable) against the whole key array, which each workgroup copies into LDS; its rank IS
// its output position, so the workgroups (256 elements each) never communicate.  n^2 compares: 9 M for the 3000 superpoint
// scores of the query selection, 0.4 M for the 600 candidate instances, spread over n / 256 CUs - one launch instead of 4-7
// digit passes of two launches each.  (One workgroup for everything measured 350 us at n = 3000: 64-bit compares on one CU.)
#define RANK_SORT_MAX 4096
#define RANK_PARTS 16          // lanes per element: each counts over 1 / 16 of the keys (one lane per element left n / 256 workgroups - twelve
                               // CUs at n = 3000 - walking all n keys: 80 us; sixteen lanes per element: the same ranks in a fifth of the time)
__global__ __launch_bounds__(256) void rank_sort_kernel(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                        uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int n,
                                                        uint64_t mask) {
    __shared__ uint64_t k[RANK_SORT_MAX];
    for (int i = threadIdx.x; i < n; i += 256) k[i] = keys_in[i] & mask;
    __syncthreads();
    const int i = (blockIdx.x * 256 + threadIdx.x) / RANK_PARTS;
    const int part = threadIdx.x % RANK_PARTS;
    const bool live = i < n;
    const uint64_t mine = live ? k[i] : 0;
    const int chunk = (n + RANK_PARTS - 1) / RANK_PARTS;
    const int j0 = part * chunk, j1 = min(n, j0 + chunk);
    int rank = 0;
    if (live)
        for (int j = j0; j < j1; ++j) rank += (k[j] < mine || (k[j] == mine && j < i));
#pragma unroll
    for (int d = 1; d < RANK_PARTS; d <<= 1) rank += __shfl_xor(rank, d);
    if (live && part == 0) {
        keys_out[rank] = keys_in[i];
        vals_out[rank] = vals_in ? vals_in[i] : (uint32_t)i;
    }
}

size_t sort_ws_bytes(int64_t n) {
    const int64_t nb = cdiv(n > 0 ? n : 1, RS_TILE);
    return align_up((size_t)nb * 256 * sizeof(int), 256) + scan_ws_bytes(nb * 256);
}

// Sorts by bits [begin_bit, end_bit).  keys_in/vals_in are clobbered (ping-pong); the result is in
// keys_out/vals_out.  vals_in == NULL means "value = original index".
// landed_in_input != NULL: no padding pass - with an even pass count the result is left in (keys_in, vals_in | vals_scratch) and
// *landed_in_input = 1 (the caller swaps its buffer roles); NULL: an odd pass count is forced so that the result is always in *_out.
int sort_pairs_u64(uint64_t* keys_in, uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out, int64_t n,
                   int begin_bit, int end_bit, void* ws, size_t ws_bytes, hipStream_t st, uint32_t* vals_scratch, int* landed_in_input) {
    if (landed_in_input) *landed_in_input = 0;
    if (n <= 0) return SD3D_OK;
    if (ws_bytes < sort_ws_bytes(n)) return sd3d_set_error(SD3D_ERR_WS, "sort workspace too small");
    if (n <= RANK_SORT_MAX) {
        const int nbits = end_bit - begin_bit;
        const uint64_t mask = (nbits >= 64 ? ~0ull : ((1ull << (nbits > 0 ? nbits : 1)) - 1ull)) << begin_bit;
        hipLaunchKernelGGL(rank_sort_kernel, dim3((unsigned)cdiv(n * RANK_PARTS, 256)), dim3(256), 0, st, keys_in, vals_in, keys_out, vals_out, (int)n, mask);
        SD3D_CHECK_LAUNCH();
        return SD3D_OK;
    }
    int passes = (end_bit - begin_bit + 7) / 8;
    if (passes < 1) passes = 1;
    if ((passes & 1) == 0) {
        if (landed_in_input && (vals_in || vals_scratch)) *landed_in_input = 1;
        else ++passes;                        // odd => result lands in *_out after ping-pong (a padding pass over zero bits: a full pass of time)
    }
    const int nb = (int)cdiv(n, RS_TILE);
    int* hist = (int*)ws;
    // ping-pong: even passes read A (=*_in) and write B (=*_out), odd passes the other way round;
    // with an odd pass count the last pass writes *_out.  When vals_in is NULL pass 0 synthesises
    // value = index and the ping-pong partner of vals_out is vals_scratch.
    uint32_t* vother = vals_in ? vals_in : vals_scratch;
    if (!vother && passes > 1) return sd3d_set_error(SD3D_ERR_ARG, "sort: need vals_in or vals_scratch");
    for (int p = 0; p < passes; ++p) {
        const int shift = begin_bit + 8 * p;
        const bool even = (p & 1) == 0;
        const uint64_t* ksrc = even ? keys_in : keys_out;
        uint64_t* kdst = even ? keys_out : keys_in;
        const uint32_t* vsrc = (p == 0) ? vals_in : (even ? vother : vals_out);
        uint32_t* vdst = even ? vals_out : vother;
        hipLaunchKernelGGL(rs_hist, dim3(nb), dim3(RS_THREADS), 0, st, ksrc, n, shift, hist, nb);
        hipLaunchKernelGGL(rs_scatter, dim3(nb), dim3(RS_THREADS), 0, st, ksrc, vsrc, kdst, vdst, n, shift, hist, nb);
    }
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// order-preserving float -> u64 key; descending order when `desc`
__global__ void f32_to_sortkey(const float* __restrict__ x, int64_t n, int desc, uint64_t* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t u = __float_as_uint(x[i]);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    if (desc) u = ~u;
    keys[i] = (uint64_t)u;
}
__global__ void i64_to_sortkey(const int64_t* __restrict__ x, int64_t n, uint64_t* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = (uint64_t)x[i];
}

int launch_f32_to_sortkey(const float* x, int64_t n, int desc, uint64_t* keys, hipStream_t st) {
    if (n <= 0) return SD3D_OK;
    hipLaunchKernelGGL(f32_to_sortkey, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, x, n, desc, keys);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
__global__ void i64_to_sortkey_add(const int64_t* __restrict__ x, int64_t n, uint64_t add, uint64_t* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = (uint64_t)x[i] + add;
}
int launch_i64_to_sortkey_add(const int64_t* x, int64_t n, uint64_t add, uint64_t* keys, hipStream_t st) {
    if (n <= 0) return SD3D_OK;
    hipLaunchKernelGGL(i64_to_sortkey_add, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, x, n, add, keys);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
__global__ void i64_to_sortkey_checked(const int64_t* __restrict__ x, int64_t n, uint64_t* __restrict__ keys, int bits, int32_t* __restrict__ flag, int value) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint64_t k = (uint64_t)x[i];
        keys[i] = k;
        if (k >> bits) atomicOr(flag, value);
    }
}
int launch_i64_to_sortkey_checked(const int64_t* x, int64_t n, uint64_t* keys, int bits, int32_t* flag, int value, hipStream_t st) {
    if (n <= 0) return SD3D_OK;
    if (!flag || bits < 1 || bits > 63) return sd3d_set_error(SD3D_ERR_ARG, "keys_from_i64_checked: flag pointer and 1..63 bits");
    hipLaunchKernelGGL(i64_to_sortkey_checked, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, x, n, keys, bits, flag, value);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
// ... and *max_out = max(*max_out, largest id) (one atomic per wave): the number of superpoints of a scene rides in the scene's
// read-back without waiting for the sort of the ids
__global__ void i64_to_sortkey_checked_max(const int64_t* __restrict__ x, int64_t n, uint64_t* __restrict__ keys, int bits, int32_t* __restrict__ flag,
                                           int value, int32_t* __restrict__ max_out, uint64_t add) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int m = 0;
    if (i < n) {
        const uint64_t k = (uint64_t)x[i];
        keys[i] = k + add;                                      // (add: the scene's bits of a batch-wide key; the check and the maximum see the id)
        if (bits < 64 && (k >> bits)) atomicOr(flag, value);
        // an id that is negative (huge after the cast) or above INT32_MAX - 1 cannot be a row count of int32 row numbers: bit 8 of the flag word,
        // whatever `bits` is (the full-sort retry with bits = 64 must not pass it silently either)
        if (k > 0x7FFFFFFEull) atomicOr(flag, 8);
        m = k > 0x7FFFFFFFull ? 0x7FFFFFFF : (int)k;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int t = __shfl_xor(m, d); m = t > m ? t : m; }
    // (one atomic per wave on ONE address was 2300 serialised L2 round trips - 29 us for 150 k ids: one per workgroup, and only when the
    //  workgroup's maximum exceeds what is already there - the maximum only grows, a stale read costs a redundant atomic at worst)
    __shared__ int wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[(threadIdx.x >> 6) & 3] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (int)(blockDim.x + 63) >> 6;
        int bm = wmax[0];
        for (int w = 1; w < nw && w < 4; ++w) bm = wmax[w] > bm ? wmax[w] : bm;
        if (bm > 0 && bm > __builtin_nontemporal_load(max_out)) atomicMax(max_out, bm);
    }
}
int launch_i64_to_sortkey_checked_max(const int64_t* x, int64_t n, uint64_t* keys, int bits, int32_t* flag, int value, int32_t* max_out, hipStream_t st,
                                      uint64_t add) {
    if (n <= 0) return SD3D_OK;
    if (!flag || !max_out || bits < 1 || bits > 64) return sd3d_set_error(SD3D_ERR_ARG, "keys_from_i64_checked_max: flag / max pointers and 1..64 bits");
    hipLaunchKernelGGL(i64_to_sortkey_checked_max, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, x, n, keys, bits, flag, value, max_out, add);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
int launch_i64_to_sortkey(const int64_t* x, int64_t n, uint64_t* keys, hipStream_t st) {
    if (n <= 0) return SD3D_OK;
    hipLaunchKernelGGL(i64_to_sortkey, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, x, n, keys);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
