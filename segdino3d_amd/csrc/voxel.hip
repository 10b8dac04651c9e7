// Sparse-voxel coordinate machinery (reference: third-party MinkowskiEngine / spconv calls at
// segdino3d/models/backbone/minkunet.py:624-631 and spconvunet.py:283-315; SURVEY.md 2b K1, K2, K5,
// K10, K11, K13).  All kernels are integer / gather passes bound by HBM bandwidth and latency:
//   scene_stats   : min / max / sum of xyz                                  (1 read of N*3 floats)
//   voxel_keys    : floor-quantise + 48-bit Z-order key                     (N*3 f32 -> N u64 + N*3 i32)
//   mark/emit     : run-length unique over SORTED keys -> voxel ids, segment starts, inverse map
//   coarsen       : parent level = unique(key >> 3) (no new sort: Z-order keeps children adjacent)
//   hash_insert   : open-addressing table key -> voxel id (linear probing, u64 CAS)
//   kernel_map    : nbr[k][v] = id of voxel at coord(v) + offset_k (or -1), K*V hash probes
//   stride_maps   : 2x2x2 stride-2 down / transposed-up neighbour tables from the parent array
//   voxel_mean    : per-voxel unweighted mean of point features (wave per voxel, ascending point order)
//   pool          : fused devoxelise + superpoint mean of features and quantised positions
#include "common.h"
#include "../../include/segdino3d_hip.h"

// ---------------------------------------------------------------------------------------------
// scene statistics: stats[0:3]=min xyz, [3:6]=max xyz, [6:9]=sum xyz (fp32)
// ---------------------------------------------------------------------------------------------
#define STAT_BLOCKS 256
__device__ static inline float wave_min(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d));
    return v;
}
__device__ static inline float wave_max(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
    return v;
}
__device__ static inline float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

__global__ __launch_bounds__(256) void scene_stats_partial(const float* __restrict__ pts, int ld, int64_t n,
                                                           float* __restrict__ part /*[gridDim.x][9]*/) {
    __shared__ float sm[4][9];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, su[3] = {0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float x = pts[i * ld + a];
            lo[a] = fminf(lo[a], x);
            hi[a] = fmaxf(hi[a], x);
            su[a] += x;
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float l = wave_min(lo[a]), h = wave_max(hi[a]), s = wave_sum(su[a]);
        if (lane == 0) { sm[w][a] = l; sm[w][3 + a] = h; sm[w][6 + a] = s; }
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        const int c = threadIdx.x;
        float r = sm[0][c];
        for (int ww = 1; ww < 4; ++ww)
            r = c < 3 ? fminf(r, sm[ww][c]) : (c < 6 ? fmaxf(r, sm[ww][c]) : r + sm[ww][c]);
        part[blockIdx.x * 9 + c] = r;
    }
}
// one wave per statistic: lanes stride over the per-block partials, then a butterfly (fixed order; nine threads walking all the
// partials one after the other took 39 us)
__global__ __launch_bounds__(576) void scene_stats_final(const float* __restrict__ part, int nb, float* __restrict__ stats, int32_t* __restrict__ zero,
                                                         int n_zero) {
    if ((int)threadIdx.x < n_zero) zero[threadIdx.x] = 0;      // (the counters of the launches behind this one: saves their memset launch)
    const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float r = c < 3 ? INFINITY : (c < 6 ? -INFINITY : 0.f);
    for (int b = lane; b < nb; b += 64) {
        const float v = part[b * 9 + c];
        r = c < 3 ? fminf(r, v) : (c < 6 ? fmaxf(r, v) : r + v);
    }
    r = c < 3 ? wave_min(r) : (c < 6 ? wave_max(r) : wave_sum(r));
    if (lane == 0) stats[c] = r;
}

int launch_scene_stats(const float* pts, int ld, int64_t n, float* stats, void* ws, size_t ws_bytes, hipStream_t st, int32_t* zero, int n_zero) {
    if (n_zero < 0 || n_zero > 576 || (n_zero > 0 && !zero)) return sd3d_set_error(SD3D_ERR_ARG, "scene_stats: at most 576 counters to zero");
    if (n <= 0) return sd3d_set_error(SD3D_ERR_ARG, "scene_stats: empty scene");
    if (ws_bytes < STAT_BLOCKS * 9 * sizeof(float)) return sd3d_set_error(SD3D_ERR_WS, "scene_stats workspace");
    const int nb = (int)min((int64_t)STAT_BLOCKS, cdiv(n, 256));
    hipLaunchKernelGGL(scene_stats_partial, dim3(nb), dim3(256), 0, st, pts, ld, n, (float*)ws);
    hipLaunchKernelGGL(scene_stats_final, dim3(1), dim3(576), 0, st, (const float*)ws, nb, stats, zero, n_zero);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// ---------------------------------------------------------------------------------------------
// voxel keys.  c = floor(x * inv_voxel)  (reference runs `coords / voxel_size` on a CUDA tensor, where
// ATen's div-by-CPU-scalar kernel multiplies by the fp32 reciprocal; see DESIGN.md "quantisation").
// shift_to_min != 0 (spconv path, spconvunet.py:286): c = floor((x - min_x) * inv_voxel).
// The key origin is a multiple of 16 (the coarsest tensor stride) so that floor(c / 2^l) of absolute
// coordinates equals (c - origin) >> l + origin / 2^l at every level, plus 32 voxels of margin so that
// +-2 neighbour offsets at any level never underflow.
// ---------------------------------------------------------------------------------------------
__device__ static inline int floor_div16(int v) { return (v >= 0) ? (v / 16) : -((-v + 15) / 16); }

__global__ __launch_bounds__(256) void voxel_keys_kernel(const float* __restrict__ pts, int ld, int64_t n, float inv_voxel,
                                                         const float* __restrict__ stats, int shift_to_min, int batch,
                                                         int32_t* __restrict__ origin_out, uint64_t* __restrict__ keys,
                                                         int32_t* __restrict__ icoords, int32_t* __restrict__ err_flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int org[3];
    float mn[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        mn[a] = shift_to_min ? stats[a] : 0.f;
        const int cmin = (int)floorf((stats[a] - mn[a]) * inv_voxel);
        org[a] = floor_div16(cmin) * 16 - 32;
    }
    if (i == 0 && origin_out) { origin_out[0] = org[0]; origin_out[1] = org[1]; origin_out[2] = org[2]; }
    if (i >= n) return;
    uint32_t r[3];
    bool bad = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int c = (int)floorf((pts[i * ld + a] - mn[a]) * inv_voxel);
        if (icoords) icoords[i * 3 + a] = c;
        const int rel = c - org[a];
        bad |= (rel < 32) | (rel >= 65536 - 64);
        r[a] = (uint32_t)rel & 0xFFFFu;
    }
    if (bad) atomicOr(err_flag, 1);
    const uint64_t m = morton_encode(r[0], r[1], r[2]);
    if (m >> 32) atomicOr(err_flag, 2);                       // a 32-bit radix sort of the Morton part would not be a full sort (sparse.OPTIMISTIC_SORT)
    keys[i] = m | ((uint64_t)(batch & 0xFF) << SD3D_MORTON_BITS);
}

int launch_voxel_keys(const float* pts, int ld, int64_t n, float inv_voxel, const float* stats, int shift_to_min, int batch,
                      int32_t* origin, uint64_t* keys, int32_t* icoords, int32_t* err_flag, hipStream_t st) {
    if (n <= 0) return sd3d_set_error(SD3D_ERR_ARG, "voxel_keys: empty scene");
    hipLaunchKernelGGL(voxel_keys_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, pts, ld, n, inv_voxel, stats,
                       shift_to_min, batch, origin, keys, icoords, err_flag);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// ---------------------------------------------------------------------------------------------
// run-length unique over sorted keys (optionally on key >> 3 : parent level)
// ---------------------------------------------------------------------------------------------
__device__ static inline uint64_t level_key(uint64_t k, int shift) {
    return ((k & SD3D_MORTON_MASK) >> shift) | (k & ~SD3D_MORTON_MASK);
}

// Optional output-extent clip (spconv SparseConv3d k=2 s=2 p=0: output extent (D - 2) / 2 + 1, so the
// trailing slice of an odd-sized grid has no output voxel; spconvunet.py:156-171, 309-310).  clip.level
// is the index of the level being CREATED (>= 1); D_0 = max(cmax + 1, min_shape) per axis.
struct ExtentClip {
    const float* stats;   // NULL = no clipping (MinkowskiEngine semantics)
    float inv_voxel;
    int level;
    int min_shape;
};
__device__ static inline bool parent_valid(uint64_t pkey, const ExtentClip& c) {
    if (!c.stats) return true;
    uint32_t x[3];
    morton_decode(pkey & SD3D_MORTON_MASK, x[0], x[1], x[2]);
    const int off = 32 >> c.level;                       // key origin is -32 voxels of level 0
    const float* st = c.stats + 9 * (int)((pkey >> SD3D_MORTON_BITS) & 0xFF);   // stats row of the key's scene (batch bits)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        int D = (int)floorf((st[3 + a] - st[a]) * c.inv_voxel) + 1;
        D = D > c.min_shape ? D : c.min_shape;
        for (int i = 0; i < c.level; ++i) D = (D - 2) / 2 + 1;
        if ((int)x[a] - off >= D) return false;
    }
    return true;
}

__global__ __launch_bounds__(256) void mark_heads(const uint64_t* __restrict__ keys, int64_t n_cap,
                                                  const int* __restrict__ n_dev, int shift, ExtentClip clip,
                                                  int* __restrict__ flags) {
    const int64_t n = n_dev ? min((int64_t)*n_dev, n_cap) : n_cap;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n_cap) return;
    int f = 0;
    if (j < n) {
        const uint64_t pk = level_key(keys[j], shift);
        f = ((j == 0) || (pk != level_key(keys[j - 1], shift))) && parent_valid(pk, clip);
    }
    flags[j] = f;
}

// excl = exclusive scan of flags.  id(j) = excl[j] + flags[j] - 1.
//   ukeys[id] = level_key(keys[j])        at heads
//   seg_start[id] = j at heads ; seg_start[V] = n            (optional)
//   map[ src_idx ? src_idx[j] : j ] = id                      (inverse map / parent array; -1 if clipped)
__global__ __launch_bounds__(256) void emit_unique(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ src_idx,
                                                   int64_t n_cap, const int* __restrict__ n_dev, int shift, ExtentClip clip,
                                                   const int* __restrict__ flags, const int* __restrict__ excl,
                                                   uint64_t* __restrict__ ukeys, int32_t* __restrict__ seg_start,
                                                   int32_t* __restrict__ map) {
    const int64_t n = n_dev ? min((int64_t)*n_dev, n_cap) : n_cap;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int f = flags[j];
    const uint64_t pk = level_key(keys[j], shift);
    const bool ok = parent_valid(pk, clip);
    const int id = ok ? excl[j] + f - 1 : -1;
    if (f) {
        ukeys[id] = pk;
        if (seg_start) seg_start[id] = (int32_t)j;
    }
    if (map) map[src_idx ? (int64_t)src_idx[j] : j] = id;
    if (j == n - 1 && seg_start) seg_start[excl[j] + f] = (int32_t)n;
}

int scan_exclusive_i32(const int* in, int* out, int64_t n_cap, const int* n_dev, int* total_dev, void* ws,
                       size_t ws_bytes, hipStream_t st);
size_t scan_ws_bytes(int64_t n);

size_t unique_ws_bytes(int64_t n_cap) {
    return 2 * align_up((size_t)n_cap * sizeof(int), 256) + scan_ws_bytes(n_cap);
}

int launch_unique_sorted(const uint64_t* keys, const uint32_t* src_idx, int64_t n_cap, const int* n_dev, int shift,
                         uint64_t* ukeys, int32_t* seg_start, int32_t* map, int32_t* n_unique_dev, void* ws,
                         size_t ws_bytes, const float* clip_stats, float clip_inv_voxel, int clip_level, int clip_min_shape,
                         hipStream_t st) {
    ExtentClip clip;
    clip.stats = clip_stats; clip.inv_voxel = clip_inv_voxel; clip.level = clip_level; clip.min_shape = clip_min_shape;
    if (n_cap <= 0) return sd3d_set_error(SD3D_ERR_ARG, "unique_sorted: n_cap <= 0");
    if (ws_bytes < unique_ws_bytes(n_cap)) return sd3d_set_error(SD3D_ERR_WS, "unique_sorted workspace too small");
    const size_t a = align_up((size_t)n_cap * sizeof(int), 256);
    int* flags = (int*)ws;
    int* excl = (int*)((char*)ws + a);
    void* sws = (char*)ws + 2 * a;
    const unsigned nb = (unsigned)cdiv(n_cap, 256);
    hipLaunchKernelGGL(mark_heads, dim3(nb), dim3(256), 0, st, keys, n_cap, n_dev, shift, clip, flags);
    int rc = scan_exclusive_i32(flags, excl, n_cap, n_dev, n_unique_dev, sws, ws_bytes - 2 * a, st);
    if (rc) return rc;
    hipLaunchKernelGGL(emit_unique, dim3(nb), dim3(256), 0, st, keys, src_idx, n_cap, n_dev, shift, clip, flags, excl, ukeys,
                       seg_start, map);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// ---------------------------------------------------------------------------------------------
// All coarser levels of a scene from its sorted level-0 keys in FOUR launches (mark, the two scan launches over the levels' flag rows
// back to back, emit) instead of four per level: level l's voxels are the runs of (Morton >> 3 l); a run boundary at level l + 1 is one at
// level l, so id_l(i) = (number of level-l heads among rows <= i) - 1 and the parent of level-(l - 1) voxel id_{l-1}(i) is id_l(i).  The
// keys, parents and counts are those of launch_unique_sorted called level after level (shift 3 each); no extent clip (MinkowskiEngine
// semantics).  The chain before a scene's first host synchronisation is bound by its number of dependent launches, not by their work.
// ---------------------------------------------------------------------------------------------
#define UL_MAX 7
struct ULParams {
    const uint64_t* keys; int64_t cap; const int* n_dev; int nl;
    uint64_t* ukeys[UL_MAX]; int32_t* parent[UL_MAX]; int32_t* counts;
};
__global__ __launch_bounds__(256) void mark_levels(const ULParams P, int* __restrict__ flags) {
    const int64_t n = P.n_dev ? min((int64_t)*P.n_dev, P.cap) : P.cap;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int l = blockIdx.y + 1;
    if (i >= P.cap) return;
    if (i == 0 && n <= 0) P.counts[l - 1] = 0;                 // an empty / fully filtered scene: emit_levels has no row n - 1 to write the count from
    int f = 0;
    if (i < n) f = (i == 0) || level_key(P.keys[i], 3 * l) != level_key(P.keys[i - 1], 3 * l);
    flags[(int64_t)blockIdx.y * P.cap + i] = f;
}
__global__ __launch_bounds__(256) void emit_levels(const ULParams P, const int* __restrict__ flags, const int* __restrict__ excl) {
    const int64_t n = P.n_dev ? min((int64_t)*P.n_dev, P.cap) : P.cap;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int l = blockIdx.y + 1;
    if (i >= n) return;
    const int64_t row = (int64_t)blockIdx.y * P.cap;
    const int f = flags[row + i];
    const int id = excl[row + i] - excl[row] + f - 1;          // (the scan ran over all rows back to back: subtract the row's base)
    if (f) P.ukeys[l - 1][id] = level_key(P.keys[i], 3 * l);
    if (l == 1) {
        P.parent[0][i] = id;
    } else {
        const int64_t prow = row - P.cap;
        if (flags[prow + i]) P.parent[l - 1][excl[prow + i] - excl[prow]] = id;
    }
    if (i == n - 1) P.counts[l - 1] = id + 1;
}
size_t unique_levels_ws_bytes(int64_t n_cap, int n_extra) {
    const int64_t tot = n_cap * n_extra;
    return 2 * align_up((size_t)tot * sizeof(int), 256) + scan_ws_bytes(tot) + 256;
}
int launch_unique_levels(const uint64_t* keys, int64_t n_cap, const int* n_dev, int n_extra, uint64_t* const* ukeys, int32_t* const* parents,
                         int32_t* counts, void* ws, size_t ws_bytes, hipStream_t st) {
    if (n_cap <= 0 || n_extra < 1 || n_extra > UL_MAX) return sd3d_set_error(SD3D_ERR_ARG, "unique_levels: n_cap > 0 and 1..7 coarser levels");
    if (!keys || !ukeys || !parents || !counts) return sd3d_set_error(SD3D_ERR_ARG, "unique_levels: null pointer");
    if (ws_bytes < unique_levels_ws_bytes(n_cap, n_extra)) return sd3d_set_error(SD3D_ERR_WS, "unique_levels workspace too small");
    ULParams P;
    P.keys = keys; P.cap = n_cap; P.n_dev = n_dev; P.nl = n_extra; P.counts = counts;
    for (int l = 0; l < UL_MAX; ++l) { P.ukeys[l] = l < n_extra ? ukeys[l] : nullptr; P.parent[l] = l < n_extra ? parents[l] : nullptr; }
    for (int l = 0; l < n_extra; ++l) if (!P.ukeys[l] || !P.parent[l]) return sd3d_set_error(SD3D_ERR_ARG, "unique_levels: null output");
    const int64_t tot = n_cap * n_extra;
    const size_t a = align_up((size_t)tot * sizeof(int), 256);
    int* flags = (int*)ws;
    int* excl = (int*)((char*)ws + a);
    int* total = (int*)((char*)ws + 2 * a);
    void* sws = (char*)ws + 2 * a + 256;
    const dim3 grid((unsigned)cdiv(n_cap, 256), (unsigned)n_extra);
    hipLaunchKernelGGL(mark_levels, grid, dim3(256), 0, st, P, flags);
    const int rc = scan_exclusive_i32(flags, excl, tot, nullptr, total, sws, ws_bytes - 2 * a - 256, st);
    if (rc) return rc;
    hipLaunchKernelGGL(emit_levels, grid, dim3(256), 0, st, P, (const int*)flags, (const int*)excl);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// ---------------------------------------------------------------------------------------------
// Round 5: EVERY level of a scene - the level-0 unique (keys, segment starts, point -> voxel map) and all coarser levels (keys, parents) -
// from the sorted POINT keys in four launches (mark, two scan launches, emit) instead of four for level 0 plus four for the rest: a
// run boundary of (Morton >> 3 l) over the sorted point keys is a run boundary over the unique keys of any finer level, so the
// arrays are those of launch_unique_sorted(shift 0) followed by launch_unique_levels, entry for entry.  No extent clip.
// ---------------------------------------------------------------------------------------------
struct VLParams {
    const uint64_t* keys; const uint32_t* src_idx; int64_t n; int nl;                   // sorted point keys, their points; nl levels (level 0 included)
    uint64_t* ukeys[UL_MAX + 1]; int32_t* parent[UL_MAX]; int32_t* seg_start; int32_t* map; int32_t* counts;
};
__global__ __launch_bounds__(256) void mark_all_levels(const VLParams P, int* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int l = blockIdx.y;
    if (i >= P.n) return;
    flags[(int64_t)l * P.n + i] = (i == 0) || level_key(P.keys[i], 3 * l) != level_key(P.keys[i - 1], 3 * l);
}
__global__ __launch_bounds__(256) void emit_all_levels(const VLParams P, const int* __restrict__ flags, const int* __restrict__ excl) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int l = blockIdx.y;
    if (i >= P.n) return;
    const int64_t row = (int64_t)l * P.n;
    const int f = flags[row + i];
    const int id = excl[row + i] - excl[row] + f - 1;          // (the scan ran over all rows back to back: subtract the row's base)
    if (f) P.ukeys[l][id] = level_key(P.keys[i], 3 * l);
    if (l == 0) {
        if (f) P.seg_start[id] = (int32_t)i;
        P.map[P.src_idx ? (int64_t)P.src_idx[i] : i] = id;
        if (i == P.n - 1) P.seg_start[id + 1] = (int32_t)P.n;
    } else {
        const int64_t prow = row - P.n;
        if (flags[prow + i]) P.parent[l - 1][excl[prow + i] - excl[prow]] = id;      // written by the first point of the finer voxel
    }
    if (i == P.n - 1) P.counts[l] = id + 1;
}
int launch_voxel_levels_all(const uint64_t* keys, const uint32_t* src_idx, int64_t n, int n_levels, uint64_t* const* ukeys, int32_t* seg_start,
                            int32_t* map, int32_t* const* parents, int32_t* counts, void* ws, size_t ws_bytes, hipStream_t st) {
    if (n <= 0 || n_levels < 1 || n_levels > UL_MAX + 1) return sd3d_set_error(SD3D_ERR_ARG, "voxel_levels_all: n > 0 and 1..8 levels");
    if (!keys || !ukeys || !seg_start || !map || !counts || (n_levels > 1 && !parents)) return sd3d_set_error(SD3D_ERR_ARG, "voxel_levels_all: null pointer");
    if (ws_bytes < unique_levels_ws_bytes(n, n_levels)) return sd3d_set_error(SD3D_ERR_WS, "voxel_levels_all workspace too small");
    VLParams P;
    P.keys = keys; P.src_idx = src_idx; P.n = n; P.nl = n_levels; P.seg_start = seg_start; P.map = map; P.counts = counts;
    for (int l = 0; l <= UL_MAX; ++l) P.ukeys[l] = l < n_levels ? ukeys[l] : nullptr;
    for (int l = 0; l < UL_MAX; ++l) P.parent[l] = l + 1 < n_levels ? parents[l] : nullptr;
    const int64_t tot = n * n_levels;
    const size_t a = align_up((size_t)tot * sizeof(int), 256);
    int* flags = (int*)ws;
    int* excl = (int*)((char*)ws + a);
    int* total = (int*)((char*)ws + 2 * a);
    void* sws = (char*)ws + 2 * a + 256;
    const dim3 grid((unsigned)cdiv(n, 256), (unsigned)n_levels);
    hipLaunchKernelGGL(mark_all_levels, grid, dim3(256), 0, st, P, flags);
    const int rc = scan_exclusive_i32(flags, excl, tot, nullptr, total, sws, ws_bytes - 2 * a - 256, st);
    if (rc) return rc;
    hipLaunchKernelGGL(emit_all_levels, grid, dim3(256), 0, st, P, (const int*)flags, (const int*)excl);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}


// ---------------------------------------------------------------------------------------------
// hash table (keys u64, values i32), capacity = power of two, EMPTY = all ones
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hash_insert(const uint64_t* __restrict__ ukeys, int64_t n,
                                                   unsigned long long* __restrict__ tkeys, int32_t* __restrict__ tvals,
                                                   uint32_t mask) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= n) return;
    const uint64_t key = ukeys[v];
    uint32_t slot = hash_u64(key) & mask;
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        const unsigned long long prev = atomicCAS(&tkeys[slot], (unsigned long long)SD3D_EMPTY_KEY, (unsigned long long)key);
        if (prev == SD3D_EMPTY_KEY || prev == key) { tvals[slot] = (int32_t)v; return; }
        slot = (slot + 1) & mask;
    }
}

__device__ static inline int hash_lookup(const uint64_t* __restrict__ tkeys, const int32_t* __restrict__ tvals,
                                         uint32_t mask, uint64_t key) {
    uint32_t slot = hash_u64(key) & mask;
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        const uint64_t k = tkeys[slot];
        if (k == key) return tvals[slot];
        if (k == SD3D_EMPTY_KEY) return -1;
        slot = (slot + 1) & mask;
    }
    return -1;
}

int launch_hash_build(const uint64_t* ukeys, int64_t n, uint64_t* tkeys, int32_t* tvals, int64_t capacity, hipStream_t st) {
    if (capacity <= 0 || (capacity & (capacity - 1)) || capacity < n + 1)
        return sd3d_set_error(SD3D_ERR_ARG, "hash_build: capacity must be a power of two > n");
    (void)hipMemsetAsync(tkeys, 0xFF, (size_t)capacity * sizeof(uint64_t), st);
    if (n > 0)
        hipLaunchKernelGGL(hash_insert, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, ukeys, n,
                           (unsigned long long*)tkeys, tvals, (uint32_t)(capacity - 1));
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// nbr[k * n_out + v] = id (in the table's level) of the voxel at coord(v) + off[k], or -1.
// Coordinates are in units of the level's own stride (the keys of level l hold (c - origin) >> l).
__global__ __launch_bounds__(256) void kernel_map_kernel(const uint64_t* __restrict__ okeys, int64_t n_out,
                                                         const uint64_t* __restrict__ tkeys, const int32_t* __restrict__ tvals,
                                                         uint32_t mask, const int8_t* __restrict__ offs, int K,
                                                         int32_t* __restrict__ nbr, int32_t* __restrict__ pair_count) {
    __shared__ int wsum[4];
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int id = -1;
    if (t < (int64_t)K * n_out) {
        const int k = (int)(t / n_out);
        const int64_t v = t - (int64_t)k * n_out;
        const uint64_t key = okeys[v];
        uint32_t x, y, z;
        morton_decode(key & SD3D_MORTON_MASK, x, y, z);
        const int nx = (int)x + offs[k * 3 + 0], ny = (int)y + offs[k * 3 + 1], nz = (int)z + offs[k * 3 + 2];
        if (((nx | ny | nz) >= 0) && nx < 65536 && ny < 65536 && nz < 65536) {
            const uint64_t q = morton_encode((uint32_t)nx, (uint32_t)ny, (uint32_t)nz) | (key & ~SD3D_MORTON_MASK);
            id = hash_lookup(tkeys, tvals, mask, q);
        }
        nbr[t] = id;
    }
    if (pair_count) {
        // rulebook size = number of (in, out, offset) pairs.  One atomic per WORKGROUP, spread over 64
        // counter words (a single word saturates at ~90 atomics/us; the host sums the 64 partials).
        const int c = __popcll(__ballot(id >= 0));
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            if (tot) atomicAdd(&pair_count[blockIdx.x & 63], tot);
        }
    }
}

// Mirrored variant for a table of a voxel set onto ITSELF with a centred odd kernel whose offset list is symmetric
// under index reversal (off[K-1-k] = -off[k], true for both enumeration orders used here): if v has neighbour j at
// offset k then j has neighbour v at offset K-1-k, so only the first K/2 offsets are probed and each hit is written
// twice.  Half the hash probes and Morton arithmetic; the upper half of nbr is pre-filled with -1 by the launcher.
__global__ __launch_bounds__(256) void kernel_map_mirrored_kernel(const uint64_t* __restrict__ keys, int64_t n,
                                                                  const uint64_t* __restrict__ tkeys, const int32_t* __restrict__ tvals,
                                                                  uint32_t mask, const int8_t* __restrict__ offs, int K,
                                                                  int32_t* __restrict__ nbr, int32_t* __restrict__ pair_count) {
    __shared__ int wsum[4];
    const int half = K / 2;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int found = 0;
    if (t < (int64_t)(half + 1) * n) {
        const int k = (int)(t / n);
        const int64_t v = t - (int64_t)k * n;
        if (k == half) {
            nbr[(int64_t)half * n + v] = (int32_t)v;                 // the centre offset is the voxel itself
            found = 1;
        } else {
            const uint64_t key = keys[v];
            uint32_t x, y, z;
            morton_decode(key & SD3D_MORTON_MASK, x, y, z);
            const int nx = (int)x + offs[k * 3 + 0], ny = (int)y + offs[k * 3 + 1], nz = (int)z + offs[k * 3 + 2];
            int id = -1;
            if (((nx | ny | nz) >= 0) && nx < 65536 && ny < 65536 && nz < 65536) {
                const uint64_t q = morton_encode((uint32_t)nx, (uint32_t)ny, (uint32_t)nz) | (key & ~SD3D_MORTON_MASK);
                id = hash_lookup(tkeys, tvals, mask, q);
            }
            nbr[t] = id;
            if (id >= 0) {
                nbr[(int64_t)(K - 1 - k) * n + id] = (int32_t)v;
                found = 2;
            }
        }
    }
    if (pair_count) {
        int c = found;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            if (tot) atomicAdd(&pair_count[blockIdx.x & 63], tot);
        }
    }
}

int launch_kernel_map(const uint64_t* okeys, int64_t n_out, const uint64_t* tkeys, const int32_t* tvals, int64_t capacity,
                      const int8_t* offs_dev, int K, int mirrored, int32_t* nbr, int32_t* pair_count, hipStream_t st) {
    if (n_out <= 0 || K <= 0) return SD3D_OK;
    if (mirrored) {
        if (!(K & 1)) return sd3d_set_error(SD3D_ERR_ARG, "kernel_map: the mirrored variant needs an odd, centred kernel");
        const int half = K / 2;
        if (half > 0 && hipMemsetAsync(nbr + (int64_t)(half + 1) * n_out, 0xFF, (size_t)half * n_out * sizeof(int32_t), st) != hipSuccess)
            return sd3d_set_error(SD3D_ERR_LAUNCH, "kernel_map: memset failed");
        hipLaunchKernelGGL(kernel_map_mirrored_kernel, dim3((unsigned)cdiv((int64_t)(half + 1) * n_out, 256)), dim3(256), 0, st, okeys,
                           n_out, tkeys, tvals, (uint32_t)(capacity - 1), offs_dev, K, nbr, pair_count);
        SD3D_CHECK_LAUNCH();
        return SD3D_OK;
    }
    const int64_t total = (int64_t)K * n_out;
    hipLaunchKernelGGL(kernel_map_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, okeys, n_out, tkeys, tvals,
                       (uint32_t)(capacity - 1), offs_dev, K, nbr, pair_count);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// ---------------------------------------------------------------------------------------------
// Kernel maps WITHOUT a hash table: through the level hierarchy the sorted keys already carry.
// The voxels of every level are sorted along the Z-order curve and level l + 1 is unique(key >> 3), so (a) the children of a coarse
// voxel are CONSECUTIVE rows of the finer level, ordered by their low three Morton bits, and (b) the neighbour of voxel v at offset d
// (|d| <= 2 per axis) lies in the parent cell P(v) + pd with pd = floor((bit(v) + d) / 2) in {-1, 0, 1} per axis: ONE entry of the
// parent level's 3^3 map finds that cell, and its {first child, 8-bit child mask} record finds the row:
//     nbr[k][v] = first[Q] + popcount(mask[Q] & ((1 << cb) - 1))  if mask[Q] has bit cb,   Q = nbr3_parent[pd][P(v)],  cb = low bits of v + d
// Two dependent, cache-friendly loads per probe (a wave's 64 consecutive rows share one or two parents) instead of a random 64-byte
// line of an open-addressing table per probe and table slot - the hash probes fetched 0.5-1.1 GB per scene for ~0.1 GB of useful bytes
// (profiles/r04_pmc_fetch.md).  No Morton arithmetic, no coordinate range checks (a cell outside the key range has no parent entry), no
// insert pass, no table memsets, and every entry of nbr is written by its own thread: coalesced, nothing to pre-fill.  The coarsest level
// (a few thousand rows) is searched directly: binary search of the encoded neighbour key in its sorted keys.
// Same table as kernel_map_kernel (MinkowskiEngine kernel map, minkunet.py:146-162), entry for entry.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void kmap_top_body(const uint64_t* __restrict__ keys, int64_t n, const int8_t* __restrict__ offs, int K,
                                              int32_t* __restrict__ nbr, int32_t* __restrict__ pair_count, const int bx) {
    __shared__ int wsum[4];
    const int64_t t = (int64_t)bx * 256 + threadIdx.x;
    int id = -1;
    if (t < (int64_t)K * n) {
        const int k = (int)(t / n);
        const int64_t v = t - (int64_t)k * n;
        const uint64_t key = keys[v];
        uint32_t x, y, z;
        morton_decode(key & SD3D_MORTON_MASK, x, y, z);
        const int nx = (int)x + offs[k * 3 + 0], ny = (int)y + offs[k * 3 + 1], nz = (int)z + offs[k * 3 + 2];
        if (((nx | ny | nz) >= 0) && nx < 65536 && ny < 65536 && nz < 65536) {
            const uint64_t q = morton_encode((uint32_t)nx, (uint32_t)ny, (uint32_t)nz) | (key & ~SD3D_MORTON_MASK);
            int64_t lo = 0, hi = n;                             // first row with key >= q
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (keys[mid] < q) lo = mid + 1; else hi = mid;
            }
            if (lo < n && keys[lo] == q) id = (int)lo;
        }
        nbr[t] = id;
    }
    if (pair_count) {
        const int c = __popcll(__ballot(id >= 0));
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            if (tot) atomicAdd(&pair_count[bx & 63], tot);
        }
    }
}

// cinfo[p] = {first child row, child mask} of every coarse voxel: the thread of a parent's FIRST child walks its (<= 8) siblings.
// The same launch writes the level pair's stride-2 maps (launch_stride_maps' tables, entry for entry) when asked to: every fine voxel
// its column of nbr_up [8, n_fine] (its parent in the row of its own kernel offset, -1 in the other seven), every first child the
// column of its parent in nbr_down [8, n_coarse] - each entry written by the thread that owns it, nothing to pre-fill.
__device__ __forceinline__ void child_info_body(const uint64_t* __restrict__ fkeys, const int32_t* __restrict__ parent, int64_t n_fine,
                                                int64_t n_coarse, int2* __restrict__ cinfo, const int32_t* __restrict__ perm8,
                                                int32_t* __restrict__ nbr_down, int32_t* __restrict__ nbr_up, const int bx) {
    const int64_t j = (int64_t)bx * 256 + threadIdx.x;
    if (j >= n_fine) return;
    const int p = parent[j];
    if (nbr_up) {
        const int mine = perm8[(int)(fkeys[j] & 7ull)];
#pragma unroll
        for (int k = 0; k < 8; ++k) nbr_up[(int64_t)k * n_fine + j] = k == mine ? p : -1;
    }
    if (j > 0 && parent[j - 1] == p) return;
    int mask = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (j + c < n_fine && parent[j + c] == p) mask |= 1 << (int)(fkeys[j + c] & 7ull);
    cinfo[p] = make_int2((int)j, mask);
    if (nbr_down) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
            nbr_down[(int64_t)perm8[c] * n_coarse + p] = ((mask >> c) & 1) ? (int)j + __popc((unsigned)mask & ((1u << c) - 1u)) : -1;
    }
}

// Everything of the hierarchy that needs no map of another level, in ONE launch (round 5; before: one launch for the coarsest level's
// map and one child-info launch per level pair, each in the chain of dependent launches in front of the first convolution): the
// coarsest level's 3^3 map by binary search and the child records / stride-2 maps of EVERY level pair.
struct KRoots {
    const uint64_t* top_keys; int64_t top_n; const int8_t* offs3; int32_t* top_nbr; int32_t* top_count; int top_blocks;
    int n_pairs;                                                      // level pairs (fine level l = 0 .. n_pairs - 1)
    const uint64_t* fkeys[7]; const int32_t* parent[7]; int64_t n_fine[7], n_coarse[7]; int2* cinfo[7];
    const int32_t* perm8; int32_t* nbr_down[7]; int32_t* nbr_up[7]; int blk0[8];          // blk0[l]: first workgroup of pair l behind the top's
};
__global__ __launch_bounds__(256) void kmap_roots_kernel(const KRoots R) {
    const int bx = (int)blockIdx.x;
    if (bx < R.top_blocks) { kmap_top_body(R.top_keys, R.top_n, R.offs3, 27, R.top_nbr, R.top_count, bx); return; }
    int l = 0;
    for (int i = 1; i < R.n_pairs; ++i) if (bx >= R.blk0[i]) l = i;
    child_info_body(R.fkeys[l], R.parent[l], R.n_fine[l], R.n_coarse[l], R.cinfo[l], R.perm8, R.nbr_down[l], R.nbr_up[l], bx - R.blk0[l]);
}

#define KH_UNROLL 4                 // offsets per thread: their loads are all requested before the first is used
struct KHParams {
    const uint64_t* keys; const int32_t* parent; int64_t n;          // this level: keys, parent row of every voxel
    const int32_t* nbr3p; int64_t np;                                 // the parent level's 3^3 map [27, np]
    const int2* cinfo;                                                // per parent-level voxel: {first child, child mask}
    const int8_t* offs; int K;                                        // this table's offsets (|d| <= 2)
    int8_t inv27[27];                                                 // (pdx + 1) + 3 (pdy + 1) + 9 (pdz + 1) -> offset index of the parent level's 3^3 map
    int32_t* nbr; int32_t* pair_count;
    // a second table of the SAME level in the same launch (the finest level's 5^3 map next to its 3^3 map): y-blocks >= y_split
    const int8_t* offs_b; int K_b; int32_t* nbr_b; int32_t* pair_count_b; int y_split;
};
__global__ __launch_bounds__(256) void kmap_hier_kernel(const KHParams P) {
    __shared__ int wsum[4];
    // (the table of this y-block: selected field by field - a modified COPY of the argument struct would move inv27[] to scratch)
    const bool second = P.nbr_b != nullptr && (int)blockIdx.y >= P.y_split;
    const int by = second ? (int)blockIdx.y - P.y_split : (int)blockIdx.y;
    const int8_t* __restrict__ t_offs = second ? P.offs_b : P.offs;
    const int t_K = second ? P.K_b : P.K;
    int32_t* __restrict__ t_nbr = second ? P.nbr_b : P.nbr;
    int32_t* __restrict__ t_count = second ? P.pair_count_b : P.pair_count;
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int k0 = by * KH_UNROLL;
    int found = 0;
    if (v < P.n) {
        const int bits = (int)(P.keys[v] & 7ull);                     // Morton bit 0 = x, 1 = y, 2 = z
        const int par = P.parent[v];
        int q[KH_UNROLL], cb[KH_UNROLL];
#pragma unroll
        for (int u = 0; u < KH_UNROLL; ++u) {
            const int k = k0 + u < t_K ? k0 + u : t_K - 1;
            const int sx = (bits & 1) + t_offs[k * 3 + 0], sy = ((bits >> 1) & 1) + t_offs[k * 3 + 1], sz = ((bits >> 2) & 1) + t_offs[k * 3 + 2];
            const int pd = ((sx >> 1) + 1) + 3 * ((sy >> 1) + 1) + 9 * ((sz >> 1) + 1);     // arithmetic shift = floor
            cb[u] = (sx & 1) | ((sy & 1) << 1) | ((sz & 1) << 2);
            q[u] = pd == 13 ? par : P.nbr3p[(int64_t)P.inv27[pd] * P.np + par];
        }
        int2 ci[KH_UNROLL];
#pragma unroll
        for (int u = 0; u < KH_UNROLL; ++u) ci[u] = q[u] >= 0 ? P.cinfo[q[u]] : make_int2(0, 0);
#pragma unroll
        for (int u = 0; u < KH_UNROLL; ++u) {
            if (k0 + u < t_K) {
                const int hit = (ci[u].y >> cb[u]) & 1;
                t_nbr[(int64_t)(k0 + u) * P.n + v] = hit ? ci[u].x + __popc((unsigned)ci[u].y & ((1u << cb[u]) - 1u)) : -1;
                found += hit;
            }
        }
    }
    if (t_count) {
        int c = found;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            if (tot) atomicAdd(&t_count[(blockIdx.x + by) & 63], tot);
        }
    }
}

size_t kernel_maps_hier_ws_bytes(int n_levels, const int64_t* n) {
    size_t b = 256;
    for (int l = 1; l < n_levels; ++l) b += align_up((size_t)n[l] * sizeof(int2), 256);
    return b;
}
// levels[0] = finest.  nbr3[l]: [27, n_l] (every level), nbr5: [125, n_0] or NULL.  offs3 / offs5: device [K, 3] int8 offset tables (the same
// enumeration order for every level); inv27: HOST table (pd index as above -> row of offs3).  pair_counts: NULL or device int32
// [(n_levels + 1) * 64], zeroed: 64 partial counters per table (levels 0 .. n_levels - 1, then the 5^3 table).
// nbr_down[l] [8, n_{l+1}] / nbr_up[l] [8, n_l] (l < n_levels - 1): optional stride-2 maps of the level pair (NULL arrays or entries: not built).
int launch_kernel_maps_hier(int n_levels, const uint64_t* const* keys, const int32_t* const* parent, const int64_t* n, int32_t* const* nbr3,
                            int32_t* nbr5, const int8_t* offs3, const int8_t* offs5, const int8_t* inv27, int32_t* pair_counts,
                            const int32_t* perm8, int32_t* const* nbr_down, int32_t* const* nbr_up, void* ws, size_t ws_bytes, hipStream_t st) {
    if (n_levels < 1 || n_levels > 8) return sd3d_set_error(SD3D_ERR_ARG, "kernel_maps_hier: 1..8 levels");
    if (ws_bytes < kernel_maps_hier_ws_bytes(n_levels, n)) return sd3d_set_error(SD3D_ERR_WS, "kernel_maps_hier: workspace too small");
    for (int l = 0; l < n_levels; ++l)
        if (n[l] <= 0 || !keys[l] || !nbr3[l] || (l + 1 < n_levels && !parent[l])) return sd3d_set_error(SD3D_ERR_ARG, "kernel_maps_hier: empty level or null pointer");
    int2* cinfo[8] = {};
    size_t off = 0;
    for (int l = 1; l < n_levels; ++l) { cinfo[l] = (int2*)((char*)ws + off); off += align_up((size_t)n[l] * sizeof(int2), 256); }
    const int top = n_levels - 1;
    {
        KRoots R;
        R.top_keys = keys[top]; R.top_n = n[top]; R.offs3 = offs3; R.top_nbr = nbr3[top]; R.top_count = pair_counts ? pair_counts + 64 * top : nullptr;
        R.top_blocks = (int)cdiv(27 * n[top], 256);
        R.n_pairs = top; R.perm8 = perm8;
        int blocks = R.top_blocks;
        for (int l = 0; l < 7; ++l) {
            const bool live = l < top;
            R.fkeys[l] = live ? keys[l] : nullptr; R.parent[l] = live ? parent[l] : nullptr;
            R.n_fine[l] = live ? n[l] : 0; R.n_coarse[l] = live ? n[l + 1] : 0; R.cinfo[l] = live ? cinfo[l + 1] : nullptr;
            R.nbr_down[l] = live && perm8 && nbr_down ? nbr_down[l] : nullptr; R.nbr_up[l] = live && perm8 && nbr_up ? nbr_up[l] : nullptr;
            R.blk0[l] = blocks;
            if (live) blocks += (int)cdiv(n[l], 256);
        }
        R.blk0[7] = blocks;
        hipLaunchKernelGGL(kmap_roots_kernel, dim3((unsigned)blocks), dim3(256), 0, st, R);
    }
    for (int l = top - 1; l >= 0; --l) {
        KHParams P;
        P.keys = keys[l]; P.parent = parent[l]; P.n = n[l]; P.nbr3p = nbr3[l + 1]; P.np = n[l + 1]; P.cinfo = cinfo[l + 1];
        for (int i = 0; i < 27; ++i) P.inv27[i] = inv27[i];
        P.offs = offs3; P.K = 27; P.nbr = nbr3[l]; P.pair_count = pair_counts ? pair_counts + 64 * l : nullptr;
        P.offs_b = nullptr; P.K_b = 0; P.nbr_b = nullptr; P.pair_count_b = nullptr; P.y_split = (int)cdiv(27, KH_UNROLL);
        unsigned gy = (unsigned)P.y_split;
        if (l == 0 && nbr5) {                                  // the stem's 5^3 table rides in the finest level's launch
            P.offs_b = offs5; P.K_b = 125; P.nbr_b = nbr5; P.pair_count_b = pair_counts ? pair_counts + 64 * n_levels : nullptr;
            gy += (unsigned)cdiv(125, KH_UNROLL);
        }
        hipLaunchKernelGGL(kmap_hier_kernel, dim3((unsigned)cdiv(n[l], 256), gy), dim3(256), 0, st, P);
    }
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// Stride-2, kernel-2 maps from the parent array.  The child's position inside its parent is the low
// three Z-order bits (x | y<<1 | z<<2); perm8 maps it to the weight index of the library whose
// checkpoint is loaded (identity for MinkowskiEngine's x-fastest order, bit-reversal for spconv).
//   nbr_down[k][p] = child j of p sitting at kernel offset k (else -1)      (conv k=2 s=2)
//   nbr_up[k][j]   = parent p of j if j sits at offset k (else -1)          (transposed conv)
__global__ __launch_bounds__(256) void stride_maps_kernel(const uint64_t* __restrict__ fkeys, const int32_t* __restrict__ parent,
                                                          int64_t n_fine, int64_t n_coarse, const int32_t* __restrict__ perm8,
                                                          int32_t* __restrict__ nbr_down, int32_t* __restrict__ nbr_up) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n_fine) return;
    const int k = perm8[(int)(fkeys[j] & 7ull)];
    const int p = parent[j];
    if (p < 0) return;       // child dropped by the library's output-extent rule
    if (nbr_down) nbr_down[(int64_t)k * n_coarse + p] = (int32_t)j;
    if (nbr_up) nbr_up[(int64_t)k * n_fine + j] = p;
}

int launch_stride_maps(const uint64_t* fkeys, const int32_t* parent, int64_t n_fine, int64_t n_coarse, const int32_t* perm8,
                       int32_t* nbr_down, int32_t* nbr_up, hipStream_t st) {
    if (nbr_down) (void)hipMemsetAsync(nbr_down, 0xFF, (size_t)8 * n_coarse * sizeof(int32_t), st);
    if (nbr_up) (void)hipMemsetAsync(nbr_up, 0xFF, (size_t)8 * n_fine * sizeof(int32_t), st);
    if (n_fine > 0)
        hipLaunchKernelGGL(stride_maps_kernel, dim3((unsigned)cdiv(n_fine, 256)), dim3(256), 0, st, fkeys, parent, n_fine,
                           n_coarse, perm8, nbr_down, nbr_up);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// ---------------------------------------------------------------------------------------------
// per-voxel mean of the assembled point feature row (one wave per voxel)
//   mode 0: [rgb(3) | f2d(F)]              Res16UNet34C early_fusion (minkunet.py:616-618)
//   mode 1: [rgb(3)]                       only_rgb
//   mode 2: [rgb(3) | xyz - mean(3) | f2d] SpConvUNet early_fusion (spconvunet.py:287)
// Members of a voxel are visited in ascending point index (stable sort) => deterministic sum.
// Columns C..ld_out-1 are written as zero (K padding for the first convolution).
// ---------------------------------------------------------------------------------------------
// `poff` = index of the scene's first point in the batch-global point numbering of sidx (0 for a single scene)
__device__ __forceinline__ void voxel_mean_body(const float* __restrict__ pts, int ld_pts, const float* __restrict__ f2d, int F, int mode,
                                                const float* __restrict__ stats, float inv_n, int64_t poff,
                                                const uint32_t* __restrict__ sidx, const int32_t* __restrict__ seg_start, int64_t v,
                                                float* __restrict__ out, int ld_out) {
    const int lane = threadIdx.x & 63;
    const int j0 = seg_start[v], j1 = seg_start[v + 1];
    const int C = (mode == 0) ? 3 + F : (mode == 1 ? 3 : 6 + F);
    const float cnt = (float)(j1 - j0);
    if (ld_out <= 320) {
        // all five 64-column chunks of the row in flight at once, the voxel's points four at a time (a voxel holds 1.1 points on
        // average): one chunk after the other was a chain of ~10 dependent load latencies per wave (155 us for 150 k points)
        float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        for (int jb = j0; jb < j1; jb += 4) {
            int64_t pp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) pp[u] = jb + u < j1 ? (int64_t)sidx[jb + u] - poff : -1;
            float x[4][5];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int ch = 0; ch < 5; ++ch) {
                    const int c = lane + 64 * ch;
                    float val = 0.f;
                    if (pp[u] >= 0 && c < C) {
                        if (c < 3) val = pts[pp[u] * ld_pts + 3 + c];
                        else if (mode == 2 && c < 6) val = pts[pp[u] * ld_pts + (c - 3)] - stats[6 + (c - 3)] * inv_n;
                        else val = f2d[pp[u] * F + (c - (mode == 2 ? 6 : 3))];
                    }
                    x[u][ch] = val;
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)                         // ascending point order, as before
                if (pp[u] >= 0) {
#pragma unroll
                    for (int ch = 0; ch < 5; ++ch) acc[ch] += x[u][ch];
                }
        }
#pragma unroll
        for (int ch = 0; ch < 5; ++ch) {
            const int c = lane + 64 * ch;
            if (c < ld_out) out[v * ld_out + c] = c < C ? acc[ch] / cnt : 0.f;
        }
        return;
    }
    for (int c = lane; c < ld_out; c += 64) {
        float s = 0.f;
        if (c < C) {
            for (int j = j0; j < j1; ++j) {
                const int64_t p = (int64_t)sidx[j] - poff;
                float x;
                if (c < 3) x = pts[p * ld_pts + 3 + c];
                else if (mode == 2 && c < 6) x = pts[p * ld_pts + (c - 3)] - stats[6 + (c - 3)] * inv_n;
                else x = f2d[p * F + (c - (mode == 2 ? 6 : 3))];
                s += x;
            }
            s = s / cnt;                      // sum / count, like the reference's average pooling
        }
        out[v * ld_out + c] = s;
    }
}

__global__ __launch_bounds__(256) void voxel_mean_kernel(const float* __restrict__ pts, int ld_pts,
                                                         const float* __restrict__ f2d, int F, int mode,
                                                         const float* __restrict__ stats, float inv_n,
                                                         const uint32_t* __restrict__ sidx, const int32_t* __restrict__ seg_start,
                                                         int64_t n_vox, float* __restrict__ out, int ld_out) {
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n_vox) return;
    voxel_mean_body(pts, ld_pts, f2d, F, mode, stats, inv_n, 0, sidx, seg_start, v, out, ld_out);
}

// Several scenes voxelised as ONE block-diagonal tensor (batch index in the key bits above the Z-order code): a voxel finds its
// scene's source arrays from its key.  Per voxel the arithmetic is the single-scene kernel's (same point order, same sums).
struct VMBatch { int n; sd3d_scene_src s[SD3D_MAX_BATCH]; };
__global__ __launch_bounds__(256) void voxel_mean_batch_kernel(const VMBatch b, int F, int mode, const uint64_t* __restrict__ ukeys,
                                                               const uint32_t* __restrict__ sidx, const int32_t* __restrict__ seg_start,
                                                               int64_t n_vox, float* __restrict__ out, int ld_out) {
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n_vox) return;
    int bi = (int)((ukeys[v] >> SD3D_MORTON_BITS) & 0xFF);
    bi = __builtin_amdgcn_readfirstlane(bi < b.n ? bi : 0);
    const sd3d_scene_src& sc = b.s[bi];
    voxel_mean_body(sc.points, sc.ld_points, sc.feats2d, F, mode, sc.stats, 1.0f / (float)sc.n_points, sc.point_off, sidx, seg_start, v,
                    out, ld_out);
}

int launch_voxel_mean(const float* pts, int ld_pts, const float* f2d, int F, int mode, const float* stats, int64_t n_points,
                      const uint32_t* sidx, const int32_t* seg_start, int64_t n_vox, float* out, int ld_out, hipStream_t st) {
    if (n_vox <= 0) return SD3D_OK;
    if (mode != 1 && !f2d) return sd3d_set_error(SD3D_ERR_ARG, "voxel_mean: 2D features missing");
    hipLaunchKernelGGL(voxel_mean_kernel, dim3((unsigned)cdiv(n_vox, 4)), dim3(256), 0, st, pts, ld_pts, f2d, F, mode, stats,
                       1.0f / (float)n_points, sidx, seg_start, n_vox, out, ld_out);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

int launch_voxel_mean_batch(const sd3d_scene_src* scenes, int n_scenes, int F, int mode, const uint64_t* ukeys, const uint32_t* sidx,
                            const int32_t* seg_start, int64_t n_vox, float* out, int ld_out, hipStream_t st) {
    if (n_vox <= 0) return SD3D_OK;
    if (n_scenes <= 0 || n_scenes > SD3D_MAX_BATCH) return sd3d_set_error(SD3D_ERR_ARG, "voxel_mean_batch: 1..16 scenes per call");
    VMBatch b;
    b.n = n_scenes;
    for (int i = 0; i < n_scenes; ++i) {
        b.s[i] = scenes[i];
        if (!scenes[i].points || scenes[i].n_points <= 0 || !scenes[i].stats) return sd3d_set_error(SD3D_ERR_ARG, "voxel_mean_batch: scene without points / stats");
        if (mode != 1 && !scenes[i].feats2d) return sd3d_set_error(SD3D_ERR_ARG, "voxel_mean_batch: 2D features missing");
    }
    hipLaunchKernelGGL(voxel_mean_batch_kernel, dim3((unsigned)cdiv(n_vox, 4)), dim3(256), 0, st, b, F, mode, ukeys, sidx, seg_start, n_vox,
                       out, ld_out);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// ---------------------------------------------------------------------------------------------
// superpoint pooling: segment starts from sorted superpoint ids, then fused devoxelise
// (`x.slice(field)` = F[inverse[p]]) + scatter_mean of features and of floor-quantised positions
// (minkunet.py:631-656).  One workgroup per superpoint (see the kernel); a wave's two 32-lane halves take alternate
// members (C/4 lanes x float4 each, 3 more lanes for xyz) and are combined with one cross-half
// shuffle, so the summation order is fixed.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void segment_starts_kernel(const uint64_t* __restrict__ sorted_ids, int64_t n, int64_t S,
                                                             int32_t* __restrict__ start /*[S+1]*/) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j > n) return;
    const int64_t prev = (j == 0) ? -1 : (int64_t)sorted_ids[j - 1];
    int64_t cur = (j == n) ? S : (int64_t)sorted_ids[j];
    if (cur > S) cur = S;
    for (int64_t s = prev + 1; s <= cur; ++s) start[s] = (int32_t)j;
}

// Round 5: one WORKGROUP (four waves) per superpoint, the rows staged through LDS.  With one wave per superpoint the launch lasted as
// long as its largest superpoint (176 points on the benchmark scene against a mean of 50): three rounds of {sidx -> inverse -> two
// batches of feature rows}, twelve dependent memory latencies, 36 us for 58 MB.  Now the 256 threads resolve the sidx -> inverse chain
// of up to 256 points with ONE pair of dependent loads, the four waves request the feature rows of a 64-point pass together (16 rows
// per wave, all in flight) and park them in LDS, and wave 0 adds them up from there IN THE ORDER OF THE OLD KERNEL (its 32-lane half h
// takes the points h, h + 2, h + 4, ... of the superpoint in ascending order, the halves are combined by one shuffle at the end): the
// sums are the same bits as before, only the loads are no longer serialised behind the additions.
#define PS_PASS 64               // points per pass
#define PS_LD 100                // floats per staged row: C <= 96 feature columns + 3 coordinates (+ 1 pad)
__global__ __launch_bounds__(256) void pool_superpoints_kernel(const float* __restrict__ feat, int ld_feat, int C,
                                                               const int32_t* __restrict__ inverse,
                                                               const int32_t* __restrict__ icoords, float voxel_size,
                                                               const uint32_t* __restrict__ sidx, const int32_t* __restrict__ start,
                                                               int64_t S, float* __restrict__ out_feat, float* __restrict__ out_pos) {
    __shared__ __attribute__((aligned(16))) float rows[PS_PASS][PS_LD];
    __shared__ int vix[256], pix[256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t s = blockIdx.x;
    const int half = lane >> 5, li = lane & 31;
    const int nvec = C >> 2;
    const int j0 = start[s], j1 = start[s + 1];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float pacc = 0.f;
    for (int jp = j0; jp < j1; jp += 256) {
        const int jm = jp + tid;
        const uint32_t pm = jm < j1 ? sidx[jm] : 0u;
        const int vm = jm < j1 ? inverse[pm] : 0;
        __syncthreads();                                        // (the previous chunk's last pass is summed)
        vix[tid] = vm;
        pix[tid] = (int)pm;
        __syncthreads();
        const int npts = j1 - jp < 256 ? j1 - jp : 256;
        for (int p0 = 0; p0 < npts; p0 += PS_PASS) {
            const int np = npts - p0 < PS_PASS ? npts - p0 : PS_PASS;      // points of this pass
            f32x4 xx[8];
            int cc[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int q = 16 * w + 2 * r + half;            // slot of the pass
                const int pt = q < np ? p0 + q : p0;            // (a slot past the end re-reads the pass's first row: never added)
                if (li < nvec) xx[r] = __builtin_nontemporal_load((const f32x4*)(feat + (int64_t)vix[pt] * ld_feat + li * 4));
                else if (li < nvec + 3) cc[r] = icoords[(int64_t)pix[pt] * 3 + (li - nvec)];
            }
            __syncthreads();                                    // wave 0 is done with the previous pass's rows
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int q = 16 * w + 2 * r + half;
                if (li < nvec) *(f32x4*)&rows[q][li * 4] = xx[r];
                else if (li < nvec + 3) rows[q][96 + (li - nvec)] = (float)cc[r];
            }
            __syncthreads();
            if (w == 0) {
                const int ni = (np - half + 1) >> 1;            // this half's points of the pass
                if (li < nvec) {
                    for (int i = 0; i < ni; ++i) acc += *(const f32x4*)&rows[half + 2 * i][li * 4];
                } else if (li < nvec + 3) {
                    for (int i = 0; i < ni; ++i) pacc += rows[half + 2 * i][96 + (li - nvec)] * voxel_size;
                }
            }
        }
    }
    if (w != 0) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] += __shfl_xor(acc[q], 32);
    pacc += __shfl_xor(pacc, 32);
    const int cnt = j1 - j0;
    const float den = (float)(cnt > 0 ? cnt : 1);       // torch_scatter: count clamped to >= 1
    if (half == 0) {
        if (li < nvec) {
            f32x4 r = acc / den;
            *(f32x4*)(out_feat + s * C + li * 4) = r;
        } else if (li < nvec + 3 && out_pos) {
            out_pos[s * 3 + (li - nvec)] = pacc / den;
        }
    }
}

// sorted ids = (scene << 32) | superpoint id of the scene; dense id = off[scene] + id (scenes one after the other)
struct SegOff { int n; int32_t off[SD3D_MAX_BATCH]; };
__global__ __launch_bounds__(256) void segment_starts_batch_kernel(const uint64_t* __restrict__ sorted_ids, int64_t n, int64_t S,
                                                                   const SegOff so, int32_t* __restrict__ start /*[S+1]*/) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j > n) return;
    auto dense = [&](uint64_t key) -> int64_t {
        int b = (int)(key >> 32);
        b = b < so.n ? b : so.n - 1;
        return (int64_t)so.off[b] + (int64_t)(key & 0xFFFFFFFFull);
    };
    const int64_t prev = (j == 0) ? -1 : dense(sorted_ids[j - 1]);
    int64_t cur = (j == n) ? S : dense(sorted_ids[j]);
    if (cur > S) cur = S;
    for (int64_t s = prev + 1; s <= cur; ++s) start[s] = (int32_t)j;
}

int launch_segment_starts_batch(const uint64_t* sorted_ids, int64_t n, int64_t S, const int32_t* off_host, int n_scenes, int32_t* start,
                                hipStream_t st) {
    if (n_scenes <= 0 || n_scenes > SD3D_MAX_BATCH) return sd3d_set_error(SD3D_ERR_ARG, "segment_starts_batch: 1..16 scenes per call");
    SegOff so;
    so.n = n_scenes;
    for (int i = 0; i < n_scenes; ++i) so.off[i] = off_host[i];
    hipLaunchKernelGGL(segment_starts_batch_kernel, dim3((unsigned)cdiv(n + 1, 256)), dim3(256), 0, st, sorted_ids, n, S, so, start);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

int launch_segment_starts(const uint64_t* sorted_ids, int64_t n, int64_t S, int32_t* start, hipStream_t st) {
    hipLaunchKernelGGL(segment_starts_kernel, dim3((unsigned)cdiv(n + 1, 256)), dim3(256), 0, st, sorted_ids, n, S, start);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

int launch_pool_superpoints(const float* feat, int ld_feat, int C, const int32_t* inverse, const int32_t* icoords,
                            float voxel_size, const uint32_t* sidx, const int32_t* start, int64_t S, float* out_feat,
                            float* out_pos, hipStream_t st) {
    if (S <= 0) return SD3D_OK;
    if ((C & 3) || C > 96 || (ld_feat & 3))
        return sd3d_set_error(SD3D_ERR_ARG, "pool_superpoints: C must be a multiple of 4 and <= 96");
    hipLaunchKernelGGL(pool_superpoints_kernel, dim3((unsigned)S), dim3(256), 0, st, feat, ld_feat, C, inverse,
                       icoords, voxel_size, sidx, start, S, out_feat, out_pos);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
