/*
 * segdino3d_hip.h - C ABI of the MI355X-native (gfx950) SegDINO3D forward path.
 *
 * This is the drop-in boundary below the reference's Python operator interface: every entry point
 * replaces a call the reference makes into a third-party CUDA library (MinkowskiEngine, spconv,
 * torch_scatter) or into ATen for the eval-mode `Baseline3D.forward`
 * (segdino3d/models/architecture/baseline3d.py:308-346).  Citations are reference file:line.
 *
 * Conventions
 *   - All pointers are DEVICE pointers unless the name ends in `_host`; tensors are row-major fp32 /
 *     int32 / int64 / uint64 as typed; `ld*` are row strides in elements.
 *   - `stream` is a hipStream_t passed as void*; every function only enqueues work on it and returns
 *     (no device synchronisation, no allocation).  Scratch memory is caller-provided: `ws`/`ws_bytes`,
 *     sized with the matching `*_ws_bytes` query.
 *   - Return value: 0 on success, negative SD3D_ERR_* otherwise; `sd3d_last_error()` gives the text.
 *     The Python host (segdino3d_amd/_lib.py) maps a non-zero status to a RuntimeError.
 *   - No global state besides the last-error string; no internal threads.
 */
#ifndef SEGDINO3D_HIP_H
#define SEGDINO3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SD3D_ABI_VERSION 1

int sd3d_abi_version(void);
const char* sd3d_last_error(void);
/* Host-only self test of the Z-order key codec (runs without a GPU); returns 0 when consistent. */
int sd3d_selftest_host(void);
/* Scheduling hint (process-wide, default 1): how many independent scenes the caller keeps in flight on this GPU, each on its
 * own stream (dist_eval.PipelinedRunner).  With more than one, launchers prefer kernel variants with a small LDS footprint
 * that co-schedule with the other scenes' kernels over the ones that are fastest alone on an idle GPU (sd3d_pair_conv: the
 * weight-stationary pass 1 for >= 96 output columns).  Results are bit-identical either way.  Returns the previous value. */
int sd3d_set_scenes_in_flight(int n);

/* The shared tail of the lock-step pass 1 (csrc/pair_gemm.hip "The shared tail of a lock-step launch"): the last 3/16 of a launch's tiles
 * are drawn in small units from an epoch-stamped counter of a per-device ring of launch slots.  Results are the same bits with the tail
 * on or off.  sd3d_set_pair_pool(0 / 1) switches it at run time (default: SD3D_PAIR_POOL, 1) and returns the previous setting.
 * sd3d_pair_pool_check() synchronises the current device and returns the number of counter words of its ring in a state no sequence of
 * finished launches can leave behind (0 = healthy; < 0 = error, see sd3d_last_error).  sd3d_pair_pool_launches(&n): pooled launches handed
 * out on the current device so far.  sd3d_pair_pool_poison(word) is a TEST hook: it overwrites every counter word of the current
 * device's ring (what a launch that died half-way or a foreign write would leave) - later launches must still produce the same bits.
 * A launch on a stream that is being captured into a HIP graph never uses the tail. */
int sd3d_set_pair_pool(int on);
int sd3d_pair_pool_check(void);
int sd3d_pair_pool_launches(int64_t* launches_out);
int sd3d_pair_pool_poison(int64_t word);

/* ---------------------------------------------------------------------------------------------
 * Sort / scan primitives (used by voxelisation, superpoint pooling, top-k)
 * ------------------------------------------------------------------------------------------- */
size_t sd3d_sort_ws_bytes(int64_t n);
/* Stable LSD radix sort of (key, value) pairs on bits [begin_bit, end_bit).  keys_in / vals_in are
 * clobbered.  vals_in may be NULL (value = element index); then vals_scratch [n] must be given. */
int sd3d_sort_pairs_u64(uint64_t* keys_in, uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out,
                        uint32_t* vals_scratch, int64_t n, int begin_bit, int end_bit, void* ws, size_t ws_bytes,
                        void* stream);
/* The same without the padding pass: the radix passes ping-pong between (keys_in, vals_in | vals_scratch) and (keys_out, vals_out); an
 * EVEN number of 8-bit passes leaves the result in the former and sets *landed_in_input = 1 (sd3d_sort_pairs_u64 adds a pass over zero
 * bits instead, so that its result is always in *_out). */
int sd3d_sort_pairs_u64_ex(uint64_t* keys_in, uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out, uint32_t* vals_scratch,
                           int64_t n, int begin_bit, int end_bit, void* ws, size_t ws_bytes, int* landed_in_input, void* stream);
size_t sd3d_scan_ws_bytes(int64_t n);
int sd3d_scan_exclusive_i32(const int32_t* in, int32_t* out, int64_t n, int32_t* total_dev, void* ws, size_t ws_bytes,
                            void* stream);
/* order-preserving key builders */
int sd3d_keys_from_f32(const float* x, int64_t n, int descending, uint64_t* keys, void* stream);
int sd3d_keys_from_i64(const int64_t* x, int64_t n, uint64_t* keys, void* stream);
/* The same, and *flag |= flag_value when an id does not fit `bits` bits (a radix sort over fewer key bits is then not a full sort). */
int sd3d_keys_from_i64_checked(const int64_t* x, int64_t n, uint64_t* keys, int bits, int32_t* flag, int flag_value, void* stream);
/* ... and *max_out = max(*max_out, largest id, clamped to INT32_MAX) - the caller zeroes it; bits = 64: no width check.  An id that is negative or
 * larger than INT32_MAX - 1 ORs 8 into *flag whatever `bits` is (ids are row numbers: int32).  The superpoint count of a
 * scene (largest id + 1; `minkunet.py:631-639` scatter_mean sizes its output the same way) is known without waiting for the sort of the ids. */
int sd3d_keys_from_i64_checked_max(const int64_t* x, int64_t n, uint64_t* keys, int bits, int32_t* flag, int flag_value, int32_t* max_out,
                                   void* stream);

/* keys[i] = x[i] + add (sd3d_keys_from_i64_offset: the scene's bits of a batch-wide key) with the check and the maximum of
 * sd3d_keys_from_i64_checked_max taken on x[i]. */
int sd3d_keys_from_i64_offset_checked_max(const int64_t* x, int64_t n, int64_t add, uint64_t* keys, int bits, int32_t* flag, int flag_value,
                                          int32_t* max_out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Voxelisation and coordinate maps.
 * Replaces ME.utils.batch_sparse_collate + ME.TensorField(...).sparse() + inverse_mapping
 * (minkunet.py:624-630, spconvunet.py:285-315) and the coordinate-manager side of
 * ME.MinkowskiConvolution / spconv indice-pair generation (minkunet.py:146-162, 176-192;
 * spconvunet.py:45-74, 156-201).
 * ------------------------------------------------------------------------------------------- */
/* stats[9] = min xyz, max xyz, sum xyz of points[:, 0:3] (baseline3d.py:285-287 scene range). */
size_t sd3d_scene_stats_ws_bytes(void);
int sd3d_scene_stats(const float* points, int ld, int64_t n, float* stats, void* ws, size_t ws_bytes, void* stream);
/* keys[i] = Z-order key of floor((xyz - (shift_to_min ? min : 0)) * inv_voxel); icoords [n,3] (optional)
 * receives the floor-quantised integer coordinates; origin[3] the key origin; *err_flag |= 1 when
 * the scene exceeds the 16-bit-per-axis key range. */
int sd3d_voxel_keys(const float* points, int ld, int64_t n, float inv_voxel, const float* stats, int shift_to_min,
                    int batch_index, int32_t* origin, uint64_t* keys, int32_t* icoords, int32_t* err_flag, void* stream);
/* Run-length unique over SORTED keys compared after (morton >> shift):
 *   ukeys [<=n], seg_start [<=n+1] (optional), map[src_idx ? src_idx[j] : j] = unique id (optional),
 *   *n_unique_dev = number of unique keys.  n_dev (optional) = device-resident live length <= n_cap.
 * clip_stats != NULL applies spconv's SparseConv3d(k=2, s=2) output-extent rule when creating level
 * clip_level >= 1: parents outside (D - 2) / 2 + 1 per axis (D_0 = max(extent, clip_min_shape),
 * spconvunet.py:309-310) are not created and their children map to -1. */
size_t sd3d_unique_ws_bytes(int64_t n_cap);
int sd3d_unique_sorted(const uint64_t* keys, const uint32_t* src_idx, int64_t n_cap, const int32_t* n_dev, int shift,
                       uint64_t* ukeys, int32_t* seg_start, int32_t* map, int32_t* n_unique_dev, void* ws,
                       size_t ws_bytes, const float* clip_stats, float clip_inv_voxel, int clip_level, int clip_min_shape,
                       void* stream);
/* ALL coarser levels of a scene from its sorted level-0 unique keys in four launches: level l = 1..n_extra keeps the runs of
 * (Morton >> 3 l).  ukeys[l - 1] [<= n_cap], parents[l - 1] [<= n_cap]: the level-l id of every level-(l - 1) voxel, counts[l - 1] = voxels
 * of level l - the arrays sd3d_unique_sorted(shift 3, map) produces when called level after level, without the extent clip. */
size_t sd3d_unique_levels_ws_bytes(int64_t n_cap, int n_extra);
int sd3d_unique_levels(const uint64_t* keys, int64_t n_cap, const int32_t* n_dev, int n_extra, uint64_t* const* ukeys,
                       int32_t* const* parents, int32_t* counts, void* ws, size_t ws_bytes, void* stream);
/* One scene's voxelisation chain from ONE call (sd3d_scene_stats, sd3d_voxel_keys, the radix sort of the keys, sd3d_unique_sorted with
 * segment starts and the point -> voxel map, sd3d_unique_levels, the superpoint ids' sort keys): ~28 launches that take the GPU ~0.25 ms
 * but cost a Python host ~0.5 ms when issued one by one in front of everything else of the scene.  Same kernels, same order, same
 * outputs as the separate calls (`minkunet.py:624-630`: batch_sparse_collate + TensorField.sparse() + inverse_mapping).
 *   readback [n_levels + 2] is zeroed here and receives {voxels of level 0 .. n_levels - 1, flags, largest superpoint id}
 *     (flags: 1 = key range exceeded, 2 = a Morton key needs more than key_bits bits, 4 = a superpoint id needs more than sp_bits bits);
 *   the sorted (key, point) pairs ping-pong between (keys_a, vals_a) and (keys_b, vals_b): *sorted_in_a tells where they landed;
 *   superpoints NULL: no ids (sp_keys unused); sp_bits = 64: no check of the ids' width. */
typedef struct sd3d_voxelise_desc {
    const float* points; int64_t n; int ld; int shift_to_min; float inv_voxel; int key_bits; int n_levels; int sp_bits;
    float* stats;                           /* [9] */
    int32_t *origin, *icoords;              /* [3], [n, 3] */
    uint64_t *keys_a, *keys_b;              /* [n] each */
    uint32_t *vals_a, *vals_b;              /* [n] each */
    uint64_t* ukeys0;                       /* [n] */
    int32_t *seg_start, *inverse;           /* [n + 1], [n] */
    uint64_t* const* ukeys;                 /* n_levels - 1 pointers, [n] each */
    int32_t* const* parents;                /* n_levels - 1 pointers, [n] each */
    int32_t* readback;                      /* [n_levels + 2] */
    const int64_t* superpoints; uint64_t* sp_keys;   /* [n] each, optional */
    void* ws; size_t ws_bytes;              /* >= sd3d_voxelise_scene_ws_bytes(n, n_levels) */
} sd3d_voxelise_desc;
size_t sd3d_voxelise_scene_ws_bytes(int64_t n, int n_levels);
int sd3d_voxelise_scene(const sd3d_voxelise_desc* d, int* sorted_in_a, void* stream);
/* sd3d_unique_sorted(shift 0, seg_start, map) + sd3d_unique_levels in ONE set of four launches, from the sorted POINT keys: ukeys[l] [<= n]
 * (l = 0 .. n_levels - 1, level 0 included), seg_start [<= n + 1], map[src_idx ? src_idx[j] : j] = level-0 id, parents[l] [<= n] (l < n_levels - 1:
 * level-(l + 1) id of every level-l voxel), counts[l] = voxels of level l.  1 <= n_levels <= 8, no extent clip; entry for entry the arrays of the
 * two separate calls.  Workspace: sd3d_unique_levels_ws_bytes(n, n_levels). */
int sd3d_voxel_levels_all(const uint64_t* sorted_keys, const uint32_t* src_idx, int64_t n, int n_levels, uint64_t* const* ukeys, int32_t* seg_start,
                          int32_t* map, int32_t* const* parents, int32_t* counts, void* ws, size_t ws_bytes, void* stream);
/* Open-addressing hash table key -> voxel id; capacity = power of two > n. */
int sd3d_hash_build(const uint64_t* ukeys, int64_t n, uint64_t* table_keys, int32_t* table_vals, int64_t capacity,
                    void* stream);
/* nbr[k*n_out + v] = id of the voxel at coord(v) + offsets[k] in the hashed level, or -1.
 * offsets: int8 [K,3] in units of that level's stride.  pair_count (optional, device int32[64],
 * pre-zeroed) accumulates the rulebook size (number of hits) as 64 partial sums - the host adds them
 * and uses the density to pick the convolution kernel. 
 * mirrored = 1: out_keys are the table's own keys and offsets[K-1-k] == -offsets[k] (centred odd kernel): only half the
 * offsets are probed and every hit is written to both mirror slots. */
int sd3d_kernel_map(const uint64_t* out_keys, int64_t n_out, const uint64_t* table_keys, const int32_t* table_vals,
                    int64_t capacity, const int8_t* offsets, int K, int mirrored, int32_t* nbr, int32_t* pair_count, void* stream);
/* The 3^3 kernel maps of EVERY level of a scene (and the 5^3 map of the finest level) from one call, without hash tables: through the
 * hierarchy of the sorted keys (level l + 1 = unique(key >> 3): the children of a coarse voxel are consecutive rows, the neighbour of a
 * voxel lies in one of the 27 cells around its parent).  keys[l] / n[l]: the levels' sorted keys (finest first, as sd3d_unique_levels
 * leaves them), parent[l][j] = row of voxel j's parent on level l + 1 (NULL for the coarsest level; no extent clip: every parent
 * exists), nbr3[l] [27, n_l] outputs, nbr5 [125, n_0] output or NULL, offsets3 / offsets5 device int8 [K, 3] in the enumeration order
 * of the weights, inv27 (HOST) maps (dx + 1) + 3 (dy + 1) + 9 (dz + 1) to the row of offsets3.  pair_counts: NULL or device int32
 * [(n_levels + 1) x 64], zeroed - 64 partial rulebook counters per table (levels 0 .. n_levels - 1, then the 5^3 table).  The tables
 * are those of sd3d_kernel_map, entry for entry (MinkowskiEngine kernel map generation: minkunet.py:146-162).
 * perm8 + nbr_down[l] [8, n_{l+1}] / nbr_up[l] [8, n_l] for l < n_levels - 1 (all optional: NULL): the stride-2 maps of every level pair
 * (the tables of sd3d_stride_maps) from the same launches, every entry written by the thread that owns it (no pre-fill). */
size_t sd3d_kernel_maps_hier_ws_bytes(int n_levels, const int64_t* n);
int sd3d_kernel_maps_hier(int n_levels, const uint64_t* const* keys, const int32_t* const* parent, const int64_t* n,
                          int32_t* const* nbr3, int32_t* nbr5, const int8_t* offsets3, const int8_t* offsets5, const int8_t* inv27,
                          int32_t* pair_counts, const int32_t* perm8, int32_t* const* nbr_down, int32_t* const* nbr_up, void* ws,
                          size_t ws_bytes, void* stream);
/* 2x2x2 stride-2 maps from the parent array: nbr_down [8, n_coarse], nbr_up [8, n_fine]; perm8[8]
 * maps the child's Z-order position (x | y<<1 | z<<2) to the weight index. */
int sd3d_stride_maps(const uint64_t* fine_keys, const int32_t* parent, int64_t n_fine, int64_t n_coarse,
                     const int32_t* perm8, int32_t* nbr_down, int32_t* nbr_up, void* stream);
/* Unweighted per-voxel average of the assembled point feature row (ME quantisation mode used by the
 * reference).  mode 0: rgb|f2d, 1: rgb, 2: rgb|xyz-mean|f2d.  out [n_vox, ld_out], zero padded. */
int sd3d_voxel_mean(const float* points, int ld_points, const float* feats2d, int F, int mode, const float* stats,
                    int64_t n_points, const uint32_t* sorted_idx, const int32_t* seg_start, int64_t n_vox, float* out,
                    int ld_out, void* stream);
/* start[s] = first position of superpoint id s in the sorted id array, start[S] = n. */
int sd3d_segment_starts(const uint64_t* sorted_ids, int64_t n, int64_t S, int32_t* start, void* stream);

/* ---- several scenes as ONE block-diagonal sparse tensor (the collation of `utils/dataset_utils.py:215-230` collate_fn_3D +
 * ME.utils.batch_sparse_collate, minkunet.py:624-627): the scene index sits in the key bits above the 48-bit Z-order code
 * (sd3d_voxel_keys batch_index), so ONE sort / unique / hash / kernel-map pass serves every scene of the batch and rows of
 * different scenes never become neighbours.  Per voxel / superpoint the arithmetic is the single-scene kernels'. ---- */
#define SD3D_MAX_BATCH 16
typedef struct sd3d_scene_src {
    const float* points;     /* [n_points, ld_points] xyz rgb of this scene */
    const float* feats2d;    /* [n_points, F] or NULL (mode 1) */
    const float* stats;      /* the scene's sd3d_scene_stats row */
    int64_t point_off;       /* index of the scene's first point in the batch-global point numbering */
    int64_t n_points;
    int32_t ld_points, pad_;
} sd3d_scene_src;
/* sd3d_voxel_mean over the voxels of all scenes: voxel v belongs to scene (ukeys[v] >> 48) & 0xFF; sorted_idx holds
 * batch-global point numbers.  `scenes` is a HOST array (copied into the launch). */
int sd3d_voxel_mean_batch(const sd3d_scene_src* scenes, int n_scenes, int F, int mode, const uint64_t* ukeys,
                          const uint32_t* sorted_idx, const int32_t* seg_start, int64_t n_vox, float* out, int ld_out, void* stream);
/* keys[i] = x[i] + add: superpoint ids of scene b become (b << 32) | id before the batch-wide sort. */
int sd3d_keys_from_i64_offset(const int64_t* x, int64_t n, int64_t add, uint64_t* keys, void* stream);
/* sd3d_segment_starts for sorted (scene << 32 | id) keys: dense id = id_off[scene] + id, S = number of dense ids of the batch.
 * `id_off` is a HOST array of n_scenes entries. */
int sd3d_segment_starts_batch(const uint64_t* sorted_ids, int64_t n, int64_t S, const int32_t* id_off, int n_scenes, int32_t* start,
                              void* stream);
/* Fused `x.slice(field)` + torch_scatter.scatter_mean of features [S,C] and of the floor-quantised
 * coordinates * voxel_size [S,3]  (minkunet.py:631-656; spconvunet.py:390). */
/* C (feature columns) must be a multiple of 4 and <= 96 (round 5: a staged row of 96 columns + 3 coordinates per LDS slot; the shipped networks pool
 * 96 columns; SD3D_ERR_ARG otherwise). */
int sd3d_pool_superpoints(const float* feat, int ld_feat, int C, const int32_t* inverse, const int32_t* icoords,
                          float voxel_size, const uint32_t* sorted_idx, const int32_t* start, int64_t S, float* out_feat,
                          float* out_pos, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Gather-GEMM (fp32 MFMA): sparse convolution and dense Linear in one kernel.
 *   out[r][n] = act(scale[n] * sum_k sum_c in[nbr[k][r]][c] * wt[k][n][c] + shift[n] + res[r][n])
 * Replaces ME.MinkowskiConvolution / ConvolutionTranspose + MinkowskiBatchNorm + MinkowskiReLU
 * (minkunet.py:135-192, 28-38, 234-250), spconv SubMConv3d / SparseConv3d / SparseInverseConv3d
 * (spconvunet.py:45-74, 156-201) and torch.nn.functional.linear in the decoder.
 *   in0 [*, ld0] first C0 channels, in1 [*, ld1] the remaining Cin - C0 (skip concatenation, or NULL)
 *   nbr  [K, M] or NULL (identity rows, K == 1);  wt [K, Cout, Cin] (Cin % 32 == 0)
 *   act: 0 none, 1 ReLU, 2 GELU(erf), 3 sigmoid;  nt: tiling code (0 = auto, see csrc/gather_gemm.hip)
 *   ws / ws_bytes: optional scratch (>= 8 * M * Cout floats enables split-K on small launches)
 * ------------------------------------------------------------------------------------------- */
int sd3d_gather_gemm(const float* in0, int ld0, int C0, const float* in1, int ld1, const int32_t* nbr, const float* wt,
                     int K, int Cin, int Cout, int64_t M, const float* scale, const float* shift, const float* res,
                     int ld_res, float* out, int ld_out, int act, int nt, void* ws, size_t ws_bytes, void* stream);
/* Host-only helper: the value of `nt` that makes sd3d_gather_gemm run a plain Linear (nbr = NULL, K = 1) on ANY number of rows with the
 * kernel - hence the summation order and the bits - it picks by itself for `rows` rows (0 = the lock-step kernel, whose order does
 * not depend on the row count).  The batched evaluation forward runs the rows of several scenes in one launch this way. */
int sd3d_dense_plan_code(int64_t rows, int Cin, int Cout);
/* n <= 8 INDEPENDENT plain Linears in one launch: out_i = act_i([in0_i | in1_i] wt_i^T + shift_i + res_i), wt_i [Cout, Cin]
 * (nn.Linear layout).  For the decoder's chains of few-hundred-row Linears (instance_seg_3d_decoder.py:656-772: the two box MLPs
 * of a layer, projections that share an input) - the same kernel as sd3d_gather_gemm's small-M path, one dispatch. */
typedef struct sd3d_linear_job {
    const float *in0, *in1;            /* in1: optional second source (concatenated channels C0..Cin-1) or NULL */
    const float *wt, *shift, *res;     /* shift / res optional */
    float* out;
    int64_t M;
    int32_t ld0, C0, ld1, Cin, Cout, ld_res, ld_out, act;
} sd3d_linear_job;
int sd3d_linear_group(int n, const sd3d_linear_job* jobs, void* stream);

/* Opt-in variant of sd3d_gather_gemm that evaluates the fp32 products as sums of bf16 MFMA products
 * (csrc/gather_gemm_split.hip).  wt_split = the fp32 weights [K, Cout, Cin] split into bf16 terms,
 * w = t0 + t1 (+ t2) with t_i = bf16_rne(w - sum_{j<i} t_j), laid out [2 or 3][K][Cout][Cin];
 * terms = 3 (two terms per operand, error ~2^-16 per product) or 6 (three terms, ~2^-24: fp32-grade), or
 * terms = 1: plain bf16 operands with fp32 accumulation ([1][K][Cout][Cin]) - the "bf16 decoder" of BASELINE
 * config #3 (autocast(bf16) around the decoder's nn.Linear calls in the reference, train_engine_3d.py:88-100).
 * Not used unless the host asks for it (SD3D_GEMM_MODE / decoder compute_dtype); the default is exact fp32 MFMA. */
int sd3d_gather_gemm_split(const float* in0, int ld0, int C0, const float* in1, int ld1, const int32_t* nbr,
                           const uint16_t* wt_split, int terms, int K, int Cin, int Cout, int64_t M, const float* scale,
                           const float* shift, const float* res, int ld_res, float* out, int ld_out, int act, int nt,
                           void* ws, size_t ws_bytes, void* stream);

/* Pair-major sparse convolution (csrc/pair_gemm.hip) - the same contract as sd3d_gather_gemm with a
 * neighbour table (MinkowskiConvolution / SubMConv3d + folded BN + residual + activation), evaluated
 * over the rulebook laid out offset-major so that every MFMA row is a real (in, out) pair.
 * sd3d_pair_lists builds, once per neighbour table nbr [K, M]:
 *   pos [K, M]      position of pair (k, r) in the list, -1 where nbr[k][r] < 0
 *   in_idx [p_cap]  gathered input row of each list entry; every offset's segment is padded to a
 *                   multiple of 128 entries with -1
 *   tile_k [p_cap/128 + 1]  offset of each 128-entry tile, -1 past the end of the list; the last entry
 *                   receives the number of real tiles
 * p_cap: multiple of 128, >= (number of pairs) + 127 * K (pairs beyond the capacity are dropped:
 * size it from the pair count sd3d_kernel_map returns).
 * sd3d_pair_conv: part = caller scratch of >= p_cap * Cout floats.  Cin % 32 == 0, Cout % 4 == 0. */
size_t sd3d_pair_lists_ws_bytes(int K, int64_t M);
int sd3d_pair_lists(const int32_t* nbr, int K, int64_t M, int64_t p_cap, int32_t* pos, int32_t* in_idx, int32_t* tile_k,
                    void* ws, size_t ws_bytes, void* stream);
/* The same for n <= 16 tables in ONE launch set (three kernels, no memsets): arrays of n host-side entries; ws holds the
 * tables' scratch back to back, each rounded up to 256 bytes (sum of align256(sd3d_pair_lists_ws_bytes(K_i, M_i))). */
int sd3d_pair_lists_batch(int n, const int32_t* const* nbr, const int* K, const int64_t* M, const int64_t* p_cap,
                          int32_t* const* pos, int32_t* const* in_idx, int32_t* const* tile_k, void* ws, size_t ws_bytes,
                          void* stream);
int sd3d_pair_conv(const float* in0, int ld0, int C0, const float* in1, int ld1, const int32_t* in_idx,
                   const int32_t* tile_k, int64_t p_cap, const int32_t* pos, const float* wt, int K, int Cin, int Cout,
                   int64_t M, const float* scale, const float* shift, const float* res, int ld_res, float* out,
                   int ld_out, int act, float* part, size_t part_bytes, void* stream);

/* Round-3 products of the list builder, all optional, and the convolution entry that uses them (same contract as sd3d_pair_conv,
 * same fixed per-row summation order for a given table description, results reproducible bit for bit):
 *   rlist [M, rl_stride]   per output row {count, list positions of its pairs in offset order}; rl_stride a multiple of 4,
 *                          >= K + 4.  Pass 2 walks a row's own partial products instead of the K slots of pos[k][r].
 *   center                 -1 (plain lists) or SD3D_PAIR_CHAINED (below).  (Round 3 also took an offset number here - a dense
 *                          kernel for the centre offset of a stride-1 table; it measured slower and left the library, see
 *                          profiles/EXPERIMENTS.md.  A value >= 0 is refused with SD3D_ERR_ARG.)
 *   meta                   bit 0: tile_k has p_cap / 128 + 3 entries (two reserved slots, written as zero, behind the tile count);
 *                          bit 1: the capacity behind the last real tile (in_idx, out_idx, tile_k) is left UNWRITTEN - for callers
 *                          whose consumers walk tile_k[p_cap / 128] tiles and nothing else (sd3d_pair_conv_ex / sd3d_run_layers do);
 *                          with worst-case capacities that tail is tens of MB per table.
 *   pos == NULL            (round 5) no position table: needs rlist or out_idx, K <= 128.  The builder then takes the row-block form
 *                          (a workgroup owns 256 rows and walks all offsets; three launches; the per-row lists come out of the fill
 *                          launch).  What an evaluation forward uses: the stem's [125, V] table alone is 69 MB written and read back.
 *   out_idx [p_cap]        output row of every list entry (-1 on padding).  Handing it to the convolution promises ONE pair per
 *                          output row - the transposed k2s2 convolutions (minkunet.py:165-192 `conv_tr`: every fine voxel has one
 *                          parent) - and pass 1 writes act(scale * product + shift + res) to the row directly: no pass 2. */
/* center == SD3D_PAIR_CHAINED (stride-1 table of a voxel set onto itself, odd kernel with symmetric offsets off[K-1-k] == -off[k],
 * centre K / 2): CHAINED lists - the entries of an output row that belong to one mirror group {k, K-1-k}, plus the centre in the
 * row's first non-empty group, share ONE partial product (pass 1 accumulates across up to three consecutive sub-tiles and stores
 * once; tile_k carries bit 30 on all but the last sub-tile of a chain).  27-44 % fewer partial rows on surface-like scenes.  For
 * these tables pos is not written (may be NULL: a row's partial positions are its rlist), rlist is required (rl_stride >= K / 2 + 2), K <= 125, p_cap >=
 * pairs + 127 * (11 * (K / 2) + 1), and the same value goes to sd3d_pair_conv_ex / sd3d_run_layers as `center`. */
#define SD3D_PAIR_CHAINED (-2)
/* sd3d_pair_conv_ex only: `center` - 2 (i.e. -3 for plain, -4 for chained lists) = the same lists with the weight matrices taken in MIRRORED offset
 * order, W[K - 1 - k] for offset k.  The input gradient of a stride-1 convolution is the forward convolution on the same table with the offsets
 * mirrored and the matrices transposed (csrc/pair_wgrad.hip header): a parameter kept as [K, Cin, Cout] (ME.MinkowskiConvolution.kernel) IS that
 * transposed set, so the training step passes it as it lies - no flipped / transposed copy per layer and step. */
#define SD3D_PAIR_MIRROR_W(center) ((center) - 2)
typedef struct sd3d_pair_table_desc {
    const int32_t* nbr;                  /* [K, M] */
    int32_t *pos, *in_idx, *tile_k;      /* as sd3d_pair_lists */
    int32_t *rlist, *out_idx;            /* optional (NULL) */
    int64_t M, p_cap;
    int32_t K, center, rl_stride, meta;
} sd3d_pair_table_desc;
int sd3d_pair_lists_desc(int n, const sd3d_pair_table_desc* tables, void* ws, size_t ws_bytes, void* stream);
int sd3d_pair_conv_ex(const float* in0, int ld0, int C0, const float* in1, int ld1, const int32_t* in_idx,
                      const int32_t* tile_k, int64_t p_cap, const int32_t* pos, const int32_t* rlist, int rl_stride, int center,
                      const int32_t* out_idx, const float* wt, int K, int Cin, int Cout, int64_t M, const float* scale,
                      const float* shift, const float* res, int ld_res, float* out, int ld_out, int act, float* part,
                      size_t part_bytes, void* stream);

/* Layer-sequence executor (csrc/executor.hip): a whole sparse U-Net forward from one call.  Replaces the
 * Python-level module loop of Res16UNetBase.forward (minkunet.py:531-601) / UBlock.forward
 * (spconvunet.py:156-201): the plan is built once per model, per scene the caller supplies the
 * neighbour tables and the activation buffers (one arena, carved by the caller); the call only enqueues.
 *   SD3D_LAYER_PAIR_CONV        out = act(scale * conv(cat[src0, src1]; table) + shift + res)   (sd3d_pair_conv)
 *   SD3D_LAYER_DENSE            the same with identity rows, K = 1                                 (sd3d_gather_gemm)
 *   SD3D_LAYER_SCALE_SHIFT_ACT  out = act(cat[src0, src1] * scale + shift) + res, Cin = total channels; res (optional) is
 *                               added AFTER the activation here                                  (sd3d_scale_shift_act_add) */
enum { SD3D_LAYER_PAIR_CONV = 0, SD3D_LAYER_DENSE = 1, SD3D_LAYER_SCALE_SHIFT_ACT = 2 };
typedef struct sd3d_layer {
    int32_t kind, table;               /* table: index into tables[] (PAIR_CONV) */
    int32_t src0, src1, res, dst;      /* buffer ids; src1 / res = -1 when absent */
    int32_t K, Cin, C0, Cout, act;     /* C0 = channels taken from src0 (Cin when src1 is absent) */
    int32_t pad_;
    const float *wt, *scale, *shift;   /* wt [K, Cout, Cin]; scale / shift per output channel or NULL */
} sd3d_layer;
typedef struct sd3d_table {
    const int32_t *in_idx, *tile_k, *pos;   /* from sd3d_pair_lists */
    int64_t p_cap, M;                        /* M = output rows of the table */
    int32_t K, pad_;
    const int32_t *rlist, *out_idx;          /* optional products of sd3d_pair_lists_desc (NULL) */
    int32_t rl_stride, center;               /* -1 or SD3D_PAIR_CHAINED, as in sd3d_pair_table_desc */
} sd3d_table;
typedef struct sd3d_buf {
    float* ptr;
    int64_t rows;
    int32_t ld, pad_;
} sd3d_buf;
int sd3d_run_layers(const sd3d_layer* layers, int n_layers, const sd3d_table* tables, int n_tables, const sd3d_buf* bufs,
                    int n_bufs, float* part, size_t part_bytes, void* ws, size_t ws_bytes, void* stream);
/* The same with table_events[n_tables] (or NULL): entry t, when not NULL, is a hipEvent_t recorded on another stream after table t's
 * lists were built there; `stream` waits for it before the first layer that reads the table (fork / join inside a scene). */
int sd3d_run_layers_ev(const sd3d_layer* layers, int n_layers, const sd3d_table* tables, int n_tables, const sd3d_buf* bufs,
                       int n_bufs, float* part, size_t part_bytes, void* ws, size_t ws_bytes, const void* const* table_events,
                       void* stream);

/* ---------------------------------------------------------------------------------------------
 * Decoder kernels (segdino3d/models/decoder/instance_seg_3d_decoder.py:606-799,
 * segdino3d/models/module/attention.py:186-395, segdino3d/models/module/utils.py:53-105)
 * ------------------------------------------------------------------------------------------- */
/* ---- Row-chain executor (csrc/rowchain.hip): the row-local part of a decoder layer as ONE launch ------------------------------
 * Replaces the per-op launches of instance_seg_3d_decoder.py:606-799 between the superpoint cross-attention and the mask-logit
 * product: a workgroup owns 16 (program.tile_rows = 0 / 16) or 4 (tile_rows = 4: csrc/rowchain_narrow.hip, 8 waves) consecutive query
 * rows of one scene, keeps their activations in LDS slots ([rows][260] fp32 each; a buffer wider than 256 columns has row stride
 * width + 4 and spills into the following slots) and interprets `ops` over them.
 * All pointers are device pointers; global tensors hold the query rows of ALL scenes of the call back to back (row = scene.q0 +
 * row in scene).  Slot 0xFF = "none".  Per row the arithmetic does not depend on the other rows of the launch. */
#define SD3D_RC_MAX_OPS 44
#define SD3D_RC_MAX_PROGRAMS 4
enum { SD3D_RC_LOAD = 1,   /* slot dst[:, :cout] = p0[row, :cout] (row stride ld) */
       SD3D_RC_STORE = 2,  /* p0[row, :cout] = slot src0 */
       SD3D_RC_LINEAR = 3, /* dst = act([src0 (k0 ch) | src1 (k1 ch)] . W^T + bias (+ res)); p0 = W [cout, K = k0 + k1] PACKED in MFMA-fragment order
                            * P[tile = col / 16][g = ch / 16][lane = 16 * ((ch % 16) / 4) + col % 16][ch % 4] = W[min(col, cout - 1)][ch] (4-row tiles:
                            * P[tile = col / 64][quad = ch / 4][lane = col % 64][ch % 4]), p1 = bias | NULL,
                            * p2 = optional global copy (row stride ld); flag NO_LDS_DST: only the global copy */
       SD3D_RC_LN = 4,     /* dst = act(LayerNorm_256(src0 (+ res)) * p0 + p1), eps = f0; p2 = optional global copy (ld) */
       SD3D_RC_PE = 5,     /* dst = sine PE (utils.py:53-105) of p0[row, 0:3] in the scene's range; p1 = dim_t [256], p2 = axis int8 [256];
                            * src0 != none: times slot src0[:, a] / p3[row, a] (box modulation, :659-666) */
       SD3D_RC_BOX = 6,    /* box refinement (:735-759): p2[row] = p0[row] + src0[:, 0:3]; with src1: size p3 / metric size p4 from p1 (previous
                            * size) and src1[:, 0:3]; flag NORMALIZE */
       SD3D_RC_MERGE = 7,  /* dst = rows of the superpoint cross-attention: p1 (its output, stride ld) where the scene's key split is 1, else
                            * the combination of its partial states p0 + scene.part_off (sd3d_attention_parts) */
       SD3D_RC_BITS2D = 8, /* LDS bit rows: blocked2d = no superpoint open for the query (p0 + scene.bits_off: blocked bits [nq, nw]) and near
                            * the 2D query (p1 + scene.near_off: [nm - 1, nw]) (:722-726) */
       SD3D_RC_ATTN = 9    /* dst = 8-head attention (32 channels per head) of slot src0 over the scene's keys: K = p0, V = p1 (row stride ld),
                            * key rows scene.q0.. (nq) or, flag KEYS_2D, scene.m0.. (nm); flag MASK_BITS2D: the LDS bit rows; scale f0;
                            * aux = first of two scratch slots */ };
#define SD3D_RC_F_NO_LDS_DST 1
#define SD3D_RC_F_NORMALIZE 2
#define SD3D_RC_F_KEYS_2D 4
#define SD3D_RC_F_MASK_BITS2D 8
#define SD3D_RC_F_INPLACE 16           /* LINEAR: dst overlaps src0 / src1 (a barrier separates the contraction from the stores) */
typedef struct sd3d_rc_op {
    uint8_t type, act, src0, src1, dst, res, flag, aux;
    uint16_t k0, k1, cout, pad_;
    int32_t ld;
    float f0;
    const void *p0, *p1, *p2, *p3, *p4;
} sd3d_rc_op;                                                  /* 64 bytes */
typedef struct sd3d_rc_scene {
    int32_t q0, nq;                                            /* first query row, number of query rows */
    int32_t m0, nm;                                            /* 2D keys: first row, count (incl. the appended dummy key) */
    int32_t bits_off, nw;                                      /* blocked bits of the scene: offset in words, words per row */
    int32_t near_off, ksplit;                                  /* near table offset in words; key split of the cross-attention */
    int64_t part_off;                                          /* offset (floats) of the scene's partial attention states */
} sd3d_rc_scene;                                               /* 40 bytes */
typedef struct sd3d_rc_program {
    int32_t n_scenes, n_programs, n_slots, nw_max, nw2_max, tile_rows;
    const float* rng;                                          /* [n_scenes][6] scene ranges (lo, hi) for PE / BOX */
    int32_t tile0[SD3D_MAX_BATCH + 1];                         /* prefix sums of ceil(nq / tile rows) */
    int32_t prog_begin[SD3D_RC_MAX_PROGRAMS + 1];              /* op ranges of the programs (gridDim.y) */
    sd3d_rc_scene scenes[SD3D_MAX_BATCH];
    sd3d_rc_op ops[SD3D_RC_MAX_OPS];
} sd3d_rc_program;
/* program_host: HOST pointer to the program (passed to the kernel by value). */
int sd3d_row_chain(const sd3d_rc_program* program_host, void* stream);
size_t sd3d_row_chain_program_bytes(void);                     /* sizeof(sd3d_rc_program): bindings check their layout against it */

/* out = act(LayerNorm(x + res) * w + b); nn.LayerNorm (+ the residual adds at decoder :690-691,
 * :708-709, :82-84, :187-188).  act: 0 none, 1 ReLU (input_proj, :228-229). */
int sd3d_layernorm(const float* x, int ld_x, const float* res, int ld_res, const float* w, const float* b, float eps,
                   int64_t M, int D, float* out, int ld_out, int act, void* stream);
/* out = act(LayerNorm(x wt^T + bias + res) * ln_w + ln_b) in ONE launch for few rows (the 200-query tensors of the decoder): the
 * attention out-projection / second FFN Linear with the residual add and norm that follow it (decoder :690-691, :708-709, :82-84,
 * :187-188).  wt [Cout = 256, Cin], Cin a multiple of 16; bias / res optional; fp32 MFMA, two-pass row statistics. */
int sd3d_linear_layernorm(const float* x, int ld_x, int64_t M, int Cin, const float* wt, int Cout, const float* bias, const float* res, int ld_res,
                          const float* ln_w, const float* ln_b, float eps, int act, float* out, int ld_out, void* stream);
/* PositionEmbeddingCoordsSine.get_sine_embeddings (utils.py:53-105) incl. shift_scale_points
 * (pc_util.py:48-76).  range = (lo[3], hi[3]); dim_t / axis: per output channel (host tables);
 * optional box modulation out *= mod_num / mod_den (decoder :660-663; ld_den may be 0 = broadcast). */
int sd3d_sine_pe(const float* xyz, int ld_xyz, int64_t n, const float* range, const float* dim_t, const int8_t* axis,
                 int d_pos, const float* mod_num, int ld_num, const float* mod_den, int ld_den, float* out, int ld_out,
                 void* stream);
/* PositionEmbeddingCoordsSine.get_fourier_embeddings (utils.py:107-142; decoder `pos_type="fourier"`): out[:, :d/2] = sin(p),
 * out[:, d/2:] = cos(p), p = (2 pi * shift_scale(xyz)) @ gauss_b[:, :d/2]; gauss_b [3, >= d/2] is the module's buffer. */
int sd3d_fourier_pe(const float* xyz, int ld_xyz, int64_t n, const float* range, const float* gauss_b, int ld_b, int d_pos,
                    float* out, int ld_out, void* stream);
/* The same two encodings and sd3d_box_refine over the rows of SEVERAL scenes at once (the decoder of a batched evaluation forward):
 * ranges [n_scenes, 6], row_scene [n] = scene of every row; row r uses ranges[row_scene[r]].  Per row the arithmetic is unchanged. */
int sd3d_sine_pe_rows(const float* xyz, int ld_xyz, int64_t n, const float* ranges, const int32_t* row_scene, const float* dim_t,
                      const int8_t* axis, int d_pos, const float* mod_num, int ld_num, const float* mod_den, int ld_den, float* out,
                      int ld_out, void* stream);
int sd3d_fourier_pe_rows(const float* xyz, int ld_xyz, int64_t n, const float* ranges, const int32_t* row_scene, const float* gauss_b,
                         int ld_b, int d_pos, float* out, int ld_out, void* stream);
int sd3d_box_refine_rows(const float* ref_points, const float* d_center, const float* size_prev, int ld_size_prev,
                         const float* d_size, const float* ranges, const int32_t* row_scene, int normalize, int64_t Q, float* center,
                         float* size, float* size_metric, void* stream);
/* Fused multi-head attention (replaces bmm + masked_fill + softmax + bmm of attention.py:361-385 and
 * nn.MultiheadAttention's SDPA at decoder :79).  Heads are 32-channel slices; nsrc = 2 concatenates
 * [q0|q1] . [k0|k1] per head (decoder :681-687).  mask_bits [Lq, ceil(Lk/32)]: bit = 1 -> blocked.
 * ws / ws_bytes: optional scratch (sd3d_attention_ws_bytes) that lets few-query launches split the keys over several
 * workgroups and merge the softmax states in a second pass; NULL = single pass. */
size_t sd3d_attention_ws_bytes(int Lq, int H);
int sd3d_attention(const float* q0, int ldq0, const float* q1, int ldq1, const float* k0, int ldk0, const float* k1,
                   int ldk1, const float* v, int ldv, const uint32_t* mask_bits, int Lq, int Lk, int H, float scale,
                   float* out, int ldo, void* ws, size_t ws_bytes, void* stream);
/* Same contract with Q, K, P and V rounded to bf16 for the two contractions (v_mfma_f32_32x32x16_bf16, fp32 accumulate;
 * scores, mask, softmax in fp32): the attention of the "bf16 decoder", BASELINE config #3. */
int sd3d_attention_bf16(const float* q0, int ldq0, const float* q1, int ldq1, const float* k0, int ldk0, const float* k1,
                        int ldk1, const float* v, int ldv, const uint32_t* mask_bits, int Lq, int Lk, int H, float scale,
                        float* out, int ldo, void* ws, size_t ws_bytes, void* stream);
/* n <= SD3D_MAX_BATCH independent attentions of the same kind (heads, scale, one or two sources, fp32 / bf16 contractions) in ONE launch:
 * the decoder of a batched evaluation forward runs the scenes' cross- / self- / 2D-query attentions this way.  blockIdx.x runs over
 * the query tiles of all scenes; each scene keeps the waves per workgroup and the key split its own launch would have, so its rows are
 * the bits of sd3d_attention on that scene alone (scenes whose workgroup shapes differ are launched one by one inside the call).
 * ws: split workspace, scene i's sd3d_attention_ws_bytes(Lq_i, H) bytes back to back. */
typedef struct sd3d_attn_job {
    const float *q0, *q1, *k0, *k1, *v;      /* q1 / k1 NULL for one source (all jobs alike) */
    const uint32_t* mask_bits;               /* [Lq, ceil(Lk / 32)] or NULL */
    float* out;
    int32_t ldq0, ldq1, ldk0, ldk1, ldv, ldo, Lq, Lk;
} sd3d_attn_job;
int sd3d_attention_batch(int n, const sd3d_attn_job* jobs, int H, float scale, int bf16, void* ws, size_t ws_bytes, void* stream);
/* The same launch WITHOUT the pass that combines the key splits (its consumer does: sd3d_row_chain MERGE): on return scene i's rows
 * are final in its `out` where ksplit_out_host[i] == 1, and otherwise wait as partial softmax states [ceil(Lq/32)][H][ksplit][64 + 1024]
 * (m[32], l[32], O[32 dv][32 q], log2 domain) at ws + part_off_out_host[i] floats. */
int sd3d_attention_batch_parts(int n, const sd3d_attn_job* jobs, int H, float scale, int bf16, void* ws, size_t ws_bytes,
                               int32_t* ksplit_out_host, int64_t* part_off_out_host, void* stream);
/* _forward_head mask part (:567-572): bits = sigmoid(logits) < thr, dead rows reset to open. */
int sd3d_mask_bits(const float* logits, int ld, int64_t Q, int S, float thr, uint32_t* bits, int nwords, void* stream);
/* (dist < thr) of torch.cdist(p=1) (:721) as bits near[M, ceil(S/32)]. */
int sd3d_near_bits(const float* sp_pos, int64_t S, const float* centers, int64_t M, float thr, uint32_t* near, int nwords,
                   void* stream);
/* mask_ = ((~attn_mask).float() @ near.float()) == 0, plus the always-open dummy key (:722-726). */
int sd3d_dinox_mask_bits(const uint32_t* blocked, const uint32_t* near, int nwords, int64_t Q, int64_t M, uint32_t* out,
                         int nwords_out, void* stream);
/* sd3d_mask_bits / sd3d_dinox_mask_bits for the matrices of n <= SD3D_MAX_BATCH scenes in one launch each (host arrays of n entries). */
int sd3d_mask_bits_batch(int n, const float* const* logits, const int* ld, const int64_t* Q, const int* S, uint32_t* const* bits,
                         const int* nwords, float thr, void* stream);
int sd3d_dinox_mask_bits_batch(int n, const uint32_t* const* blocked, const uint32_t* const* near, const int* nwords, const int64_t* Q,
                               const int64_t* Mq, uint32_t* const* out, const int* nwords_out, void* stream);
/* centre / size refinement (:735-759, :768-772). d_size may be NULL (no size head). */
int sd3d_box_refine(const float* ref_points, const float* d_center, const float* size_prev, int ld_size_prev,
                    const float* d_size, const float* range, int normalize, int64_t Q, float* center, float* size,
                    float* size_metric, void* stream);
/* out = act(x * scale + shift), x = [x0 (C0 channels) | x1] : pre-activation BatchNorm1d + ReLU of the
 * spconv residual blocks (spconvunet.py:48-51, 154-156, 184-187, 227-229).  x1 may be NULL. */
int sd3d_scale_shift_act(const float* x0, int ld0, int C0, const float* x1, int ld1, const float* scale,
                         const float* shift, int act, int64_t M, int C, float* out, int ld_out, void* stream);
/* out = act(x * scale + shift) + add : the tail of a normalize_before=False residual block - conv -> BatchNorm1d -> ReLU,
 * then the identity branch summed in AFTER the activation (spconvunet.py:66-81, 95-97).  add may be NULL. */
int sd3d_scale_shift_act_add(const float* x0, int ld0, int C0, const float* x1, int ld1, const float* scale,
                             const float* shift, int act, int64_t M, int C, const float* add, int ld_add, float* out,
                             int ld_out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Post-processing (segdino3d/models/architecture/baseline3d.py)
 * ------------------------------------------------------------------------------------------- */
/* softmax(cls)[:, :C] flattened (:427) and/or its row maximum (:236-238). Either output may be NULL. */
int sd3d_class_scores(const float* cls, int ld, int64_t Q, int C, float* scores, float* rowmax, void* stream);
/* labels / query index of the selected flat indices + mask-quality rescoring (:436-446). */
int sd3d_mask_scores(const float* masks, int ld, int S, const uint32_t* flat_idx, const float* score_in, int n, int C,
                     int normalize, int32_t* labels, int32_t* qidx, float* score_out, void* stream);
/* Index glue of predict_by_feat_instance / mask_matrix_nms as single launches (pure data movement):
 *   take_f32: out[i] = src[idx[i]] (:434-435 `scores[topk_idx]`);  take_pair: labels / scores in the first sort's order (:71-76);
 *   nms_finish: final_scores = scores2[order2], final_labels = labels1[order2], record = order1[order2] (int64, :133-139) and, boxes != NULL,
 *   boxes[i] = [centers | sizes][qidx[record[i]]] ([n, 6], baseline3d.py:447-452; centers / sizes [Q, 3] contiguous). */
int sd3d_take_f32(const float* src, const uint32_t* idx, int n, float* out, void* stream);
/* idx[0..k) = the first k entries of the STABLE descending sort of x[0..n) (ties: lower index first) - what
 * sd3d_keys_from_f32(descending) + sd3d_sort_pairs_u64 give, from one launch of one workgroup (radix select + rank of the k survivors).
 * `scores.flatten(0, 1).topk(topk_insts)` of predict_by_feat_instance (baseline3d.py:434).  1 <= k <= 1024, k <= n <= 40 960 (forty scores per thread of the one workgroup, in registers). */
int sd3d_topk_desc_f32(const float* x, int64_t n, int k, uint32_t* idx, void* stream);
/* The data-dependent selections of predict_by_feat_instance (:470-476) for two score thresholds on the device (all outputs sized k):
 * keep / pkeep = rows with score > thr0 / thr1 and count > npoint_thr, ascending; union_rows = rows in either; keep_u / pkeep_u = where
 * the rows of keep / pkeep sit in union_rows; score_mask[i] = score[i] > thr0; npoint_mask = (count > npoint_thr) compacted over the
 * rows with score > thr0; counts[4] = {|keep|, |pkeep|, |union|, |score > thr0|} - the only thing the host has to read. */
int sd3d_select_instances(const float* scores, const int32_t* count, int k, float thr0, float thr1, int npoint_thr, int32_t* keep,
                          int32_t* pkeep, int32_t* union_rows, int32_t* keep_u, int32_t* pkeep_u, uint8_t* score_mask, uint8_t* npoint_mask,
                          int32_t* counts, void* stream);
/* labels_out (int64) / scores_out / boxes_out ([m, 6] or NULL) = rows keep[0..m) of labels / scores / boxes (baseline3d.py:470-476). */
int sd3d_take_instances(const int32_t* keep, int m, const int32_t* labels, const float* scores, const float* boxes, int64_t* labels_out,
                        float* scores_out, float* boxes_out, void* stream);
int sd3d_take_pair(const uint32_t* order, const int32_t* labels, const float* scores, int n, int32_t* labels_out, float* scores_out,
                   void* stream);
int sd3d_nms_finish(const uint32_t* order2, const float* scores2, const int32_t* labels1, const uint32_t* order1, const int32_t* qidx,
                    const float* centers, const float* sizes, int n, float* final_scores, int32_t* final_labels, int64_t* record,
                    float* boxes, void* stream);
/* sig[r] = sigmoid(masks[qidx[order[r]]]) zero-padded to ld_out, area[r] = sum (:441, :66, :80-81). */
int sd3d_gather_sigmoid(const float* masks, int ld, int S, const int32_t* qidx, const uint32_t* order, int n, float* sig,
                        int ld_out, float* area, void* stream);
/* matrix-NMS decay from inter = sig . sig^T (:85-119); comp_ws [n] scratch. */
int sd3d_nms_decay(const float* inter, int ld, const float* area, const int32_t* labels, int n, int gaussian,
                   float sigma, const float* score_in, float* comp_ws, float* score_out, void* stream);
/* superpoint -> point broadcast + threshold + point count + optional box filter (:453-454, :464, :348-371).
 * out [n, N] bytes (0/1); count[n] counts BEFORE the box filter; boxes [n,6] or NULL.  Superpoint ids must be
 * < ld_sig (the padded row width of sig); ws holds the thresholded rows as a bit table. */
size_t sd3d_expand_masks_ws_bytes(int n, int ld_sig);
int sd3d_expand_masks(const float* sig, int ld_sig, const uint32_t* src_row, int n, const int64_t* superpoints,
                      const float* points, int ld_points, int64_t N, float sp_thr, const float* boxes, float loose_ratio,
                      uint8_t* out, int32_t* count, void* ws, size_t ws_bytes, void* stream);
/* The same in two steps, for callers that apply the score / point-count thresholds BEFORE expanding (count[] comes from step 1):
 * sd3d_mask_rowbits thresholds the rows into the bit table in `ws` (sd3d_expand_masks_ws_bytes) and makes count[n];
 * sd3d_expand_rows writes out[j] = row rows[j] for a list of m rows from the SAME ws (boxes [n, 6] indexed by the row number, or NULL):
 * the [n, N] byte table of sd3d_expand_masks (90 MB at 600 x 150 k) is never made, only the kept rows. */
int sd3d_mask_rowbits(const float* sig, int ld_sig, const uint32_t* src_row, int n, const int64_t* superpoints, int64_t N, float sp_thr,
                      int32_t* count, void* ws, size_t ws_bytes, void* stream);
int sd3d_expand_rows(const void* ws, int n, int ld_sig, const int32_t* rows, int m, const int64_t* superpoints, const float* points,
                     int ld_points, int64_t N, const float* boxes, float loose_ratio, uint8_t* out, void* stream);
/* Bit-packed copy of selected rows of the [n, N] byte masks of sd3d_expand_masks, for the device -> host copy of
 * `pts_instance_mask[0]` (baseline3d.py:453-454; evaluator_3d.py:178 reads it on the host): out [n_rows, nb = ceil(N/8)] bytes,
 * bit j of byte b = masks[rows[i]][8 b + j] != 0 (numpy bitorder "little"); rows NULL = rows 0..n_rows-1. */
int sd3d_pack_mask_rows(const uint8_t* masks, int64_t N, const int32_t* rows, int n_rows, uint8_t* out, int64_t nb, void* stream);
/* HOST function (no GPU): packed_host [n_rows, nb] -> out_host [n_rows, N] bytes 0 / 1, the torch.bool / numpy bool layout. */
int sd3d_unpack_bits_host(const uint8_t* packed_host, int64_t n_rows, int64_t N, int64_t nb, uint8_t* out_host);
/* argmax over selected columns (:504) and table gather (:504-507). */
int sd3d_row_argmax(const float* x, int ld, int64_t Q, const int32_t* cols, int ncols, int64_t* out, void* stream);
int sd3d_gather_i64(const int64_t* table, const int64_t* idx, int64_t N, int use_index, int64_t* out, void* stream);
/* panoptic paint + small-segment removal + map composition (:532-556).  rows_desc[n]: mask rows in
 * descending score order; inst_ws [N], hist_ws [n + n_stuff + 1] scratch. */
int sd3d_panoptic(const uint8_t* masks, int64_t N, const int32_t* rows_desc, const int32_t* labels_desc, int n, int n_stuff,
                  int npoint_thr, const int64_t* sem_stuff, int32_t* inst_ws, int32_t* hist_ws, int64_t* sem_map,
                  int64_t* inst_map, void* stream);
/* GT instance centres / sizes attached to the targets before inference (:289-305).  masks: bool bytes,
 * row stride mask_stride; mode 0 = "mean" centre, 1 = "median" (= bbox centre, :299-300). */
size_t sd3d_instance_boxes_ws_bytes(int n_inst);
int sd3d_instance_boxes(const float* points, int ld, int64_t N, const uint8_t* masks, int64_t mask_stride, int n_inst,
                        int mode, float* centers, float* sizes, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * ScanNet AP association (the caller side of the path):
 * evaluation/utils_instance_seg_3d_eval.py:340-371 counts |pred_mask & (gt_ids == id)| per (prediction,
 * ground truth) pair with numpy.  counts[p][c] = number of points of prediction row p (mask byte != 0) whose
 * gt_index is c, for c in [0, n_cols); points with gt_index outside that range are not counted.  The caller
 * maps instance ids to columns (and void points to one extra column), so vert_count = the row sum.
 * masks [n, mask_stride] bytes; counts [n, n_cols] is zeroed by the call; n_cols <= 8192. */
int sd3d_mask_overlaps(const uint8_t* masks, int64_t mask_stride, int n, const int32_t* gt_index, int64_t N, int n_cols,
                       int32_t* counts, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training criterion, segdino3d/models/loss/loss_3d.py.  One (decoder layer, scene) per call.
 * Shapes: cls [Q, n_cls1] (n_cls1 = instance classes + "no object"), masks [Q, S] logits over superpoints,
 * scores [Q] / centers [Q,3] / sizes [Q,3] nullable; ground truth: labels [G] int64, masks as bit rows
 * gt_bits [G, words] + gt_count [G] (sd3d_pack_mask_bits), gt_centers / gt_sizes [G, >=3] nullable,
 * query_masks [G, Q] bytes nullable ("query q lies in object g", loss_3d.py:358). */

/* bits[r][w] bit b = masks[r][32 w + b] != 0; counts[r] = set bits of the row (nullable). */
int sd3d_pack_mask_bits(const uint8_t* masks, int64_t ld, int n_rows, int n_cols, uint32_t* bits, int words, int32_t* counts,
                        void* stream);

/* cost [Q, G] = w0 * QueryClassificationCost + w1 * MaskBCECost + w2 * MaskDiceCost + w3 * CenterL1Cost + w4 * SizeL1Cost
 * (loss_3d.py:139-271 with the helpers :63-97); entries with query_masks[g][q] == 0 are 1e8 (SparseMatcher, :358-359).
 * weights5 is a HOST array. */
int sd3d_match_costs(const float* cls, int ld_cls, int n_cls1, const float* masks, int ld_masks, int Q, int S, const float* centers,
                     const float* sizes, const int64_t* labels, const uint32_t* gt_bits, int words, const int32_t* gt_count, int G,
                     const float* gt_centers, int ld_gc, const float* gt_sizes, int ld_gs, const uint8_t* query_masks,
                     const float* weights5, float* cost, void* stream);

/* SparseMatcher.__call__ (loss_3d.py:360-365): match[q][g] = cost[q][g] < (topk+1)-th smallest cost of column g. */
int sd3d_sparse_match(const float* cost, int Q, int G, int topk, uint8_t* match, void* stream);

/* InstanceCriterion's per-scene terms of one layer (loss_3d.py:459-503 / :618-663) for a given match [Q, G] (from
 * sd3d_sparse_match, or scipy's linear_sum_assignment for the HungarianMatcher :311), and their gradients.
 * parts8 (device) = [class CE, mask BCE, mask dice, score MSE, centre L1, size L1, matched pairs, scores kept];
 * d_* = coef6[i] * d(parts[i]) / d(prediction): coef6 (HOST array) carries the loss weight and the batch-size factors
 * of :505-521 / :665-679, so the d_* buffers are the gradients of the total loss.  d_cls [Q, n_cls1], d_masks [Q, S],
 * d_scores [Q], d_centers / d_sizes [Q, 3] are fully overwritten (the nullable ones only when their prediction is given). */
size_t sd3d_instance_loss_ws_bytes(int Q);
int sd3d_instance_loss(const float* cls, int ld_cls, int n_cls1, const float* masks, int ld_masks, int Q, int S, const float* scores,
                       const float* centers, const float* sizes, const int64_t* labels, const uint32_t* gt_bits, int words,
                       const int32_t* gt_count, int G, const float* gt_centers, int ld_gc, const float* gt_sizes, int ld_gs,
                       const uint8_t* match, const float* class_weight, const float* coef6, float* d_cls, float* d_masks,
                       float* d_scores, float* d_centers, float* d_sizes, float* parts8, void* ws, size_t ws_bytes, void* stream);

/* ScanNetSemanticCriterion for one scene (loss_3d.py:37-60): sem [Q, ld] logits of which the first n_logits count,
 * sem_masks [n_rows, Q] bytes (target = first set row, 0 if none), rows with target == ignore_index are skipped.
 * loss[0] = mean NLL; d_sem [Q, ld_d] = coef * d loss / d sem (columns >= n_logits are zeroed). */
size_t sd3d_semantic_loss_ws_bytes(int Q);
int sd3d_semantic_loss(const float* sem, int ld, int Q, int n_rows, int n_logits, const uint8_t* sem_masks, int ignore_index, float coef,
                       float* d_sem, int ld_d, float* loss, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Backward of the pair-major sparse convolution (MinkowskiEngine / spconv autograd in the reference,
 * reached through train_engine_3d.py:88-122).  The input gradient is sd3d_pair_conv on the transposed rulebook with
 * transposed weights; the weight gradient is
 *     dw[k][co][ci] (+)= sum over pairs p of offset k of dy[out_idx[p]][co] * x[in_idx[p]][ci]
 * with in_idx / tile_k from sd3d_pair_lists and out_idx from sd3d_pair_out_rows (out_idx[pos[k][r]] = r, -1 on
 * padding).  dw is [K, Cout, Cin] like the forward weights; exact fp32 MFMA, fixed summation order.
 * flags: bit 0 = accumulate into dw, bit 1 = round both operands to bf16 (nearest even) before multiplying - the numbers
 * of a bf16-operand product with fp32 accumulation, for the bf16 training mode of the decoder; bit 2 (K = 1: a Linear) =
 * dw has Cout more floats after the [Cout, Cin] block and receives the bias gradient there (column sums of dy over the
 * listed rows), formed from the dy rows the kernel has staged anyway. */
#define SD3D_WGRAD_ACCUMULATE 1
#define SD3D_WGRAD_BF16_OPERANDS 2
#define SD3D_WGRAD_BIAS 4
int sd3d_pair_out_rows(const int32_t* pos, int K, int64_t M, int64_t p_cap, int32_t* out_idx, void* stream);
size_t sd3d_pair_wgrad_ws_bytes(int K, int Cin, int Cout);
int sd3d_pair_wgrad(const float* dy, int ld_dy, const float* x, int ld_x, const int32_t* in_idx, const int32_t* out_idx,
                    const int32_t* tile_k, int64_t p_cap, int K, int Cin, int Cout, float* dw, int flags, void* ws, size_t ws_bytes,
                    void* stream);

/* Weight and bias gradient of a Linear on a few thousand rows in ONE launch (the decoder's ~130 Linears per training step):
 * dw [Cout, Cin] (+)= g^T x, db [Cout] (+)= column sums of g (db may be NULL); g [M, ld_g >= round4(Cout)] (act'-scaled output gradient),
 * x [M, ld_x >= round4(Cin)], both strides multiples of 4 floats.  A workgroup owns a 32 x 32 block of dw for all rows (rounds of 128 rows staged
 * in LDS, sixteen waves, their accumulators summed in a fixed order: bit-reproducible).
 * flags: SD3D_WGRAD_ACCUMULATE, SD3D_WGRAD_BF16_OPERANDS.  For tens of thousands of rows sd3d_pair_wgrad (row ranges + reduce) is the one. */
int sd3d_linear_wgrad(const float* g, int ld_g, const float* x, int ld_x, int64_t M, int Cin, int Cout, float* dw, float* db, int flags,
                      void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training-mode BatchNorm over voxel rows (ME.MinkowskiBatchNorm = nn.BatchNorm1d, minkunet.py:302-304) with the
 * BasicBlock's residual add and ReLU folded in (:234-250), and the backward of sd3d_pool_superpoints (:668-676).
 * x, y, dy, dx, res, dres are [M, C] fp32 with row strides in floats; act: 0 none, 1 relu.
 *   sd3d_bn_stats:    mean[c], var[c] (biased), rstd[c] = 1 / sqrt(var + eps) over the M rows
 *   sd3d_bn_apply:    y = act((x - mean) * rstd * gamma + beta + res)            (res nullable)
 *   sd3d_bn_backward: g = dy masked by y > 0 when act == relu; dbeta = sum g; dgamma = sum g * xhat;
 *                     dx = gamma * rstd * (g - dbeta / M - xhat * dgamma / M); dres = g (nullable)
 * Reductions run in a fixed order (bit-reproducible).  ws: sd3d_bn_ws_bytes(M, C). */
size_t sd3d_bn_ws_bytes(int64_t M, int C);
int sd3d_bn_stats(const float* x, int ld, int64_t M, int C, float eps, float* mean, float* var, float* rstd, void* ws, size_t ws_bytes,
                  void* stream);
/* sd3d_bn_stats that also advances nn.BatchNorm1d's buffers in place (any of the three may be NULL):
 * running_mean = (1 - momentum) running_mean + momentum mean, running_var likewise with the UNBIASED batch variance
 * (var * M / (M - 1)), num_batches_tracked += 1 (torch/nn/modules/batchnorm.py semantics behind minkunet.py:302-304). */
int sd3d_bn_stats_running(const float* x, int ld, int64_t M, int C, float eps, float* mean, float* var, float* rstd,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, void* ws,
                          size_t ws_bytes, void* stream);
int sd3d_bn_apply(const float* x, int ld_x, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* res,
                  int ld_res, int64_t M, int C, int act, float* y, int ld_y, void* stream);
int sd3d_bn_backward(const float* dy, int ld_dy, const float* y, int ld_y, const float* x, int ld_x, const float* mean, const float* rstd,
                     const float* gamma, int64_t M, int C, int act, float* dx, int ld_dx, float* dres, int ld_dres, float* dgamma,
                     float* dbeta, void* ws, size_t ws_bytes, void* stream);
/* dfeat[v] = sum over the points p of voxel v (sidx[seg_start[v] .. seg_start[v+1])) of dout[sp[p]] / max(|sp[p]|, 1),
 * with |s| = sp_start[s+1] - sp_start[s]; dout [S, C], C <= 128. */
int sd3d_pool_superpoints_backward(const float* dout, int C, const int64_t* superpoints, const uint32_t* sidx, const int32_t* seg_start,
                                   const int32_t* sp_start, int64_t V, float* dfeat, int ld, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Train-time augmentation, segdino3d/datasets/transform/point_cloud_transforms.py.  The random draws are
 * the caller's (numpy.random in the reference's order); trans3 / color_* are HOST arrays.
 *   sd3d_augment_points:   in place on points[:, 0:3]: flip x / y (:112-117), rotate by `angle` about z (points @ rot_mat_T
 *                          of mmdet3d's rotation_3d_in_axis, :284-300), scale (:312), translate (:255); with color_mean3 also
 *                          points[:, 3:6] = (rgb - mean) / std (:382-387).  Also used for query2d_pos (ld = 3, no colour).
 *   sd3d_voxel_units:      coords [n, 3] = points[:, 0:3] / voxel_size (:417)
 *   sd3d_box_blur3:        scipy.ndimage.convolve(ones(3) / 3 along `axis`, mode="constant") on `grids` volumes (:455-459)
 *   sd3d_elastic_displace: coords += mag * trilinear(noise [3, D0, D1, D2]) on the grid linspace(-(b-1) gran, (b-1) gran, b)
 *                          per axis, zero outside (:461-470) */
int sd3d_augment_points(float* points, int ld, int64_t n, int flip_x, int flip_y, float angle, float scale, const float* trans3,
                        const float* color_mean3, const float* color_std3, void* stream);
int sd3d_voxel_units(const float* points, int ld, int64_t n, float voxel_size, float* coords, void* stream);
int sd3d_box_blur3(const float* in, float* out, int grids, int D0, int D1, int D2, int axis, void* stream);
int sd3d_elastic_displace(float* coords, int64_t n, const float* noise, int D0, int D1, int D2, float gran, float mag, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Backward pieces of the query decoder; torch autograd over nn.Linear / nn.LayerNorm / elementwise ops in
 * the reference (instance_seg_3d_decoder.py:640-797).
 *   sd3d_act_backward:       g[:, :C] = dy * act'(.), g[:, C:C_pad] = 0.  act as in sd3d_gather_gemm; ref = forward OUTPUT for
 *                            relu / sigmoid, PRE-activation for gelu.  The Linear's input / weight gradients are then
 *                            sd3d_gather_gemm(g, W^T) and sd3d_pair_wgrad on identity pair lists, its bias gradient sd3d_col_sums(g).
 *   sd3d_col_sums:           out[c] = sum_r x[r][c], fixed order.
 *   sd3d_layernorm_backward: for y = act(LayerNorm(x + res) * w + b): dxin = d/d(x + res) [M, D], dw, db (act 0 / 1 = relu).
 *   sd3d_sine_pe_mod_backward: gradient of sd3d_sine_pe's box modulation w.r.t. mod_num [n, 3] (positions and mod_den are
 *                            detached in the reference, :740, :753). */
/* ---------------------------------------------------------------------------------------------
 * Training step of a sparse U-Net from two calls (csrc/train_plan.hip; segdino3d_amd/train_plan.py records the plan): the forward of every
 * {sparse convolution -> batch-statistics BatchNorm (+ residual) -> ReLU} layer of Res16UNetBase.forward (minkunet.py:531-601), and the
 * backward of the same list in reverse (what `loss.backward()` of train_engine_3d.py:88-122 gets from MinkowskiEngine's autograd).
 * A buffer (sd3d_buf) may hold the outputs of several layers side by side (`*_col` = first column of a layer's slice): a skip concatenation
 * is the two producers writing into one buffer.  Every pointer is device memory owned by the caller; the calls only enqueue.
 * ------------------------------------------------------------------------------------------- */
typedef struct sd3d_train_table {
    const int32_t *in_idx, *tile_k, *pos, *rlist;   /* sd3d_pair_lists_desc products of the table (plain lists, with the position table) */
    const int32_t *out_rows;                        /* [p_cap] output row of every list entry (sd3d_pair_out_rows / the table's out_idx) */
    int64_t p_cap, M;
    int32_t K, rl_stride, center, direct;           /* direct = 1: one pair per output row (transposed k2s2 convolution): pass 1 writes the rows */
} sd3d_train_table;
typedef struct sd3d_train_layer {
    int32_t table, table_t, mirrored;               /* forward table; table of the transposed rulebook (-1: no input gradient); 1: stride-1 table (its own transpose, offsets mirrored) */
    int32_t src, src_col, res, res_col, dst, dst_col;   /* buffer ids + first column of the slice; res = -1: no residual */
    int32_t raw;                                    /* raw[] id of the convolution output in front of the BatchNorm, [rows, Cout] */
    int32_t K, Cin, Cout, act;                      /* Cin = columns read from src (the stem's are zero-padded to a multiple of 32) */
    int32_t need_dx, dx_accum;                      /* need_dx = 0: the input carries no gradient; dx_accum = 1: grad[src] already holds one when this layer's arrives */
    int32_t stats, pad_;                            /* offset (floats) of {mean, var, rstd} x Cout in the statistics arena */
    float eps, momentum;
    const float *wt_fwd, *kernel;                   /* [K, Cout, Cin] copy for the forward; the parameter [K, Cin, Cout] (input gradient) */
    float* dkernel;                                 /* [K, Cin, Cout] gradient of the parameter */
    const float *gamma, *beta;
    float *dgamma, *dbeta, *running_mean, *running_var;
    int64_t* num_batches;
} sd3d_train_layer;
int sd3d_unet_train_forward(const sd3d_train_layer* layers, int n_layers, const sd3d_train_table* tables, int n_tables, const sd3d_buf* act,
                            int n_bufs, const sd3d_buf* raw, int n_raw, float* stats, float* part, size_t part_bytes, void* ws, size_t ws_bytes,
                            void* stream);
/* grad[]: gradient buffers, ids / shapes of act[] (d loss / d output already in the output's buffer); graw: scratch of max(rows x Cout) floats;
 * ws >= max(sd3d_bn_ws_bytes, sd3d_pair_wgrad_ws_bytes) over the layers. */
int sd3d_unet_train_backward(const sd3d_train_layer* layers, int n_layers, const sd3d_train_table* tables, int n_tables, const sd3d_buf* act,
                             const sd3d_buf* grad, int n_bufs, const sd3d_buf* raw, int n_raw, const float* stats, float* graw, size_t graw_floats,
                             float* part, size_t part_bytes, void* ws, size_t ws_bytes, void* stream);

/* dst_i [cols_i, ld_dst_i] = src_i [rows_i, cols_i]^T, columns rows_i .. ld_dst_i - 1 zero, for n matrices in as few launches as
 * their descriptors fit (112 per launch): all the W^T the Linear input gradients of one backward pass need
 * (`x.grad = dy @ W` of every nn.Linear in instance_seg_3d_decoder.py), made at once instead of one copy kernel per weight. */
typedef struct sd3d_transpose_job {
    const float* src; float* dst;
    int32_t rows, cols, ld_dst, batch;   /* batch > 1: that many matrices back to back in src ([rows, cols] each) and dst ([cols, ld_dst] each): the
                                          * [K, Cin, Cout] -> [K, Cout, Cin] weights of a sparse convolution as ONE job (0 / 1: one matrix) */
} sd3d_transpose_job;
int sd3d_transpose_batch(int n, const sd3d_transpose_job* jobs, void* stream);
int sd3d_act_backward(const float* dy, int ld_dy, const float* ref, int ld_ref, int act, int64_t M, int C, int C_pad, float* g, int ld_g,
                      void* stream);
size_t sd3d_col_sums_ws_bytes(int64_t M, int C);
int sd3d_col_sums(const float* x, int ld, int64_t M, int C, float* out, void* ws, size_t ws_bytes, void* stream);
size_t sd3d_layernorm_backward_ws_bytes(int64_t M, int D);
int sd3d_layernorm_backward(const float* dy, int ld_dy, const float* y, int ld_y, const float* x, int ld_x, const float* res, int ld_res,
                            const float* w, float eps, int64_t M, int D, int act, float* dxin, int ld_dx, float* dw, float* db, void* ws,
                            size_t ws_bytes, void* stream);
/* backward of sd3d_box_refine w.r.t. the two head outputs (d_center, d_size_out nullable = zero gradient). */
int sd3d_box_refine_backward(const float* d_center, const float* d_size_out, const float* size, const float* range, int normalize, int64_t Q,
                             float* d_dc, float* d_ds, void* stream);
int sd3d_sine_pe_mod_backward(const float* d_out, int ld_do, const float* xyz, int ld_xyz, int64_t n, const float* range, const float* dim_t,
                              const int8_t* axis, int d_pos, const float* mod_den, int ld_den, float* d_num, void* stream);

/* Attention for training: sd3d_attention that also returns lse [H, Lq] = log sum exp of every masked,
 * scaled score row, and the backward pass (gradients of attention.py:361-385 / nn.MultiheadAttention, decoder :79):
 * given out, lse and d_out -> dq0 (dq1), dk0 (dk1), dv with the shapes / strides of their forward tensors.  Exact fp32
 * MFMA, P recomputed tile by tile, fixed summation order.  ws: sd3d_attention_backward_ws_bytes(Lq, H). */
int sd3d_attention_lse(const float* q0, int ldq0, const float* q1, int ldq1, const float* k0, int ldk0, const float* k1,
                       int ldk1, const float* v, int ldv, const uint32_t* mask_bits, int Lq, int Lk, int H, float scale,
                       float* out, int ldo, float* lse, void* ws, size_t ws_bytes, void* stream);
/* bf16 forward (section "bf16 decoder") that keeps lse: mixed-precision training runs this forward and the fp32 backward below. */
int sd3d_attention_lse_bf16(const float* q0, int ldq0, const float* q1, int ldq1, const float* k0, int ldk0, const float* k1,
                            int ldk1, const float* v, int ldv, const uint32_t* mask_bits, int Lq, int Lk, int H, float scale,
                            float* out, int ldo, float* lse, void* ws, size_t ws_bytes, void* stream);
size_t sd3d_attention_backward_ws_bytes(int Lq, int H);
int sd3d_attention_backward(const float* q0, int ldq0, const float* q1, int ldq1, const float* k0, int ldk0, const float* k1, int ldk1,
                            const float* v, int ldv, const uint32_t* mask_bits, int Lq, int Lk, int H, float scale, const float* out, int ldo,
                            const float* lse, const float* d_out, int ld_do, float* dq0, int ld_dq0, float* dq1, int ld_dq1, float* dk0,
                            int ld_dk0, float* dk1, int ld_dk1, float* dv, int ld_dv, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEGDINO3D_HIP_H */
