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

/* ---------------------------------------------------------------------------------------------
 * Sort / scan primitives (used by voxelisation, superpoint pooling, top-k)
 * ------------------------------------------------------------------------------------------- */
size_t sd3d_sort_ws_bytes(int64_t n);
/* Stable LSD radix sort of (key, value) pairs on bits [begin_bit, end_bit).  keys_in / vals_in are
 * clobbered.  vals_in may be NULL (value = element index); then vals_scratch [n] must be given. */
int sd3d_sort_pairs_u64(uint64_t* keys_in, uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out,
                        uint32_t* vals_scratch, int64_t n, int begin_bit, int end_bit, void* ws, size_t ws_bytes,
                        void* stream);
size_t sd3d_scan_ws_bytes(int64_t n);
int sd3d_scan_exclusive_i32(const int32_t* in, int32_t* out, int64_t n, int32_t* total_dev, void* ws, size_t ws_bytes,
                            void* stream);
/* order-preserving key builders */
int sd3d_keys_from_f32(const float* x, int64_t n, int descending, uint64_t* keys, void* stream);
int sd3d_keys_from_i64(const int64_t* x, int64_t n, uint64_t* keys, void* stream);

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
 *   *n_unique_dev = number of unique keys.  n_dev (optional) = device-resident live length <= n_cap. */
size_t sd3d_unique_ws_bytes(int64_t n_cap);
int sd3d_unique_sorted(const uint64_t* keys, const uint32_t* src_idx, int64_t n_cap, const int32_t* n_dev, int shift,
                       uint64_t* ukeys, int32_t* seg_start, int32_t* map, int32_t* n_unique_dev, void* ws,
                       size_t ws_bytes, void* stream);
/* Open-addressing hash table key -> voxel id; capacity = power of two > n. */
int sd3d_hash_build(const uint64_t* ukeys, int64_t n, uint64_t* table_keys, int32_t* table_vals, int64_t capacity,
                    void* stream);
/* nbr[k*n_out + v] = id of the voxel at coord(v) + offsets[k] in the hashed level, or -1.
 * offsets: int8 [K,3] in units of that level's stride. */
int sd3d_kernel_map(const uint64_t* out_keys, int64_t n_out, const uint64_t* table_keys, const int32_t* table_vals,
                    int64_t capacity, const int8_t* offsets, int K, int32_t* nbr, void* stream);
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
/* Fused `x.slice(field)` + torch_scatter.scatter_mean of features [S,C] and of the floor-quantised
 * coordinates * voxel_size [S,3]  (minkunet.py:631-656; spconvunet.py:390). */
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
 *   act: 0 none, 1 ReLU, 2 GELU(erf), 3 sigmoid;  nt: 32-column subtiles per wave (0 = auto)
 * ------------------------------------------------------------------------------------------- */
int sd3d_gather_gemm(const float* in0, int ld0, int C0, const float* in1, int ld1, const int32_t* nbr, const float* wt,
                     int K, int Cin, int Cout, int64_t M, const float* scale, const float* shift, const float* res,
                     int ld_res, float* out, int ld_out, int act, int nt, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEGDINO3D_HIP_H */
