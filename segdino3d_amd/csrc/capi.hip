// extern "C" entry points declared in include/segdino3d_hip.h.  Thin argument checking + launch.
#include "gg_common.h"
#include "../../include/segdino3d_hip.h"
#include <string.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

int sd3d_set_error(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "unknown error");
    return code;
}

// ---- internal launchers (defined in the other translation units)
size_t sort_ws_bytes(int64_t n);
int sort_pairs_u64(uint64_t*, uint32_t*, uint64_t*, uint32_t*, int64_t, int, int, void*, size_t, hipStream_t, uint32_t*, int*);
size_t scan_ws_bytes(int64_t n);
int scan_exclusive_i32(const int*, int*, int64_t, const int*, int*, void*, size_t, hipStream_t);
int launch_f32_to_sortkey(const float*, int64_t, int, uint64_t*, hipStream_t);
int launch_i64_to_sortkey(const int64_t*, int64_t, uint64_t*, hipStream_t);
int launch_i64_to_sortkey_checked(const int64_t*, int64_t, uint64_t*, int, int32_t*, int, hipStream_t);
int launch_i64_to_sortkey_checked_max(const int64_t*, int64_t, uint64_t*, int, int32_t*, int, int32_t*, hipStream_t, uint64_t);
int launch_scene_stats(const float*, int, int64_t, float*, void*, size_t, hipStream_t, int32_t*, int);
int launch_pack_mask_rows(const uint8_t*, int64_t, const int32_t*, int, uint8_t*, int64_t, hipStream_t);
int launch_row_chain(const sd3d_rc_program*, hipStream_t);
size_t unique_ws_bytes(int64_t);
int launch_unique_sorted(const uint64_t*, const uint32_t*, int64_t, const int*, int, uint64_t*, int32_t*, int32_t*, int32_t*,
                         void*, size_t, const float*, float, int, int, hipStream_t);
int launch_hash_build(const uint64_t*, int64_t, uint64_t*, int32_t*, int64_t, hipStream_t);
size_t unique_levels_ws_bytes(int64_t, int);
int launch_unique_levels(const uint64_t*, int64_t, const int*, int, uint64_t* const*, int32_t* const*, int32_t*, void*, size_t, hipStream_t);
int launch_voxel_levels_all(const uint64_t*, const uint32_t*, int64_t, int, uint64_t* const*, int32_t*, int32_t*, int32_t* const*, int32_t*, void*, size_t,
                            hipStream_t);
int launch_kernel_map(const uint64_t*, int64_t, const uint64_t*, const int32_t*, int64_t, const int8_t*, int, int, int32_t*, int32_t*, hipStream_t);
int launch_stride_maps(const uint64_t*, const int32_t*, int64_t, int64_t, const int32_t*, int32_t*, int32_t*, hipStream_t);
int launch_voxel_mean(const float*, int, const float*, int, int, const float*, int64_t, const uint32_t*, const int32_t*, int64_t,
                      float*, int, hipStream_t);
int launch_segment_starts(const uint64_t*, int64_t, int64_t, int32_t*, hipStream_t);
int launch_segment_starts_batch(const uint64_t*, int64_t, int64_t, const int32_t*, int, int32_t*, hipStream_t);
int launch_voxel_mean_batch(const sd3d_scene_src*, int, int, int, const uint64_t*, const uint32_t*, const int32_t*, int64_t, float*, int,
                            hipStream_t);
int launch_i64_to_sortkey_add(const int64_t*, int64_t, uint64_t, uint64_t*, hipStream_t);
int launch_pool_superpoints(const float*, int, int, const int32_t*, const int32_t*, float, const uint32_t*, const int32_t*,
                            int64_t, float*, float*, hipStream_t);
int launch_voxel_keys(const float*, int, int64_t, float, const float*, int, int, int32_t*, uint64_t*, int32_t*, int32_t*, hipStream_t);

int launch_gather_gemm(const GGParams&, int, void*, size_t, hipStream_t);
int dense_plan_code(int64_t, int, int);
int launch_gather_gemm_split(const GGParams&, int, int, const void*, void*, size_t, hipStream_t);
size_t pair_lists_ws_bytes(int K, int64_t M);
int launch_pair_lists(const int32_t*, int, int64_t, int64_t, int32_t*, int32_t*, int32_t*, void*, size_t, hipStream_t);
int launch_linear_group(int, const GGParams*, hipStream_t);
int launch_fourier_pe(const float*, int, int64_t, const float*, const float*, int, int, float*, int, const int32_t*, hipStream_t);
int launch_pair_lists_batch(int, const int32_t* const*, const int*, const int64_t*, const int64_t*, int32_t* const*, int32_t* const*,
                            int32_t* const*, void*, size_t, hipStream_t);
int launch_pair_conv(const float*, int, int, const float*, int, const int32_t*, const int32_t*, int64_t, const int32_t*, const int32_t*, int,
                     int, const int32_t*, const float*, int, int, int, int64_t, const float*, const float*, const float*, int, float*, int,
                     int, float*, size_t, hipStream_t);
int launch_pair_lists_desc(int, const sd3d_pair_table_desc*, void*, size_t, hipStream_t);
size_t kernel_maps_hier_ws_bytes(int, const int64_t*);
int launch_kernel_maps_hier(int, const uint64_t* const*, const int32_t* const*, const int64_t*, int32_t* const*, int32_t*, const int8_t*,
                            const int8_t*, const int8_t*, int32_t*, const int32_t*, int32_t* const*, int32_t* const*, void*, size_t, hipStream_t);

int launch_layernorm(const float*, int, const float*, int, const float*, const float*, float, int64_t, int, float*, int, int, hipStream_t);
int launch_linear_layernorm(const float*, int, int64_t, int, const float*, int, const float*, const float*, int, const float*, const float*, float, int,
                            float*, int, hipStream_t);
int launch_sine_pe(const float*, int, int64_t, const float*, const float*, const int8_t*, int, const float*, int, const float*, int, float*, int, const int32_t*, hipStream_t);
struct AttnParams {
    const float* q[2]; int ldq[2];
    const float* k[2]; int ldk[2];
    const float* v; int ldv;
    const uint32_t* bits; int nwords;
    float* out; int ldo;
    int Lq, Lk, H;
    float scale;
    int ksplit;
    float* part;
    int bf16;
    float* lse;
};
size_t attention_ws_bytes(int Lq, int H);
int launch_attention_batch(int, const AttnParams*, int, void*, size_t, hipStream_t, int32_t* = nullptr, int64_t* = nullptr);
int launch_mask_bits_batch(int, const float* const*, const int*, const int64_t*, const int*, uint32_t* const*, const int*, float, hipStream_t);
int launch_dinox_mask_bits_batch(int, const uint32_t* const*, const uint32_t* const*, const int*, const int64_t*, const int64_t*,
                                 uint32_t* const*, const int*, hipStream_t);
int launch_attention(const AttnParams&, int, void*, size_t, hipStream_t, bool = true);
int launch_mask_bits(const float*, int, int64_t, int, float, uint32_t*, int, hipStream_t);
int launch_near_bits(const float*, int64_t, const float*, int64_t, float, uint32_t*, int, hipStream_t);
int launch_dinox_mask_bits(const uint32_t*, const uint32_t*, int, int64_t, int64_t, uint32_t*, int, hipStream_t);
int launch_box_refine(const float*, const float*, const float*, int, const float*, const float*, int, int64_t, float*, float*, float*, const int32_t*, hipStream_t);
int launch_class_scores(const float*, int, int64_t, int, float*, float*, hipStream_t);
int launch_mask_scores(const float*, int, int, const uint32_t*, const float*, int, int, int, int32_t*, int32_t*, float*, hipStream_t);
int launch_take_f32(const float*, const uint32_t*, int, float*, hipStream_t);
int launch_topk_desc(const float*, int64_t, int, uint32_t*, hipStream_t);
int launch_select_instances(const float*, const int32_t*, int, float, float, int, int32_t*, int32_t*, int32_t*, int32_t*, int32_t*, uint8_t*, uint8_t*,
                            int32_t*, hipStream_t);
int launch_take_instances(const int32_t*, int, const int32_t*, const float*, const float*, int64_t*, float*, float*, hipStream_t);
int launch_take_pair(const uint32_t*, const int32_t*, const float*, int, int32_t*, float*, hipStream_t);
int launch_nms_finish(const uint32_t*, const float*, const int32_t*, const uint32_t*, const int32_t*, const float*, const float*, int, float*,
                      int32_t*, int64_t*, float*, hipStream_t);
int launch_gather_sigmoid(const float*, int, int, const int32_t*, const uint32_t*, int, float*, int, float*, hipStream_t);
int launch_nms_decay(const float*, int, const float*, const int32_t*, int, int, float, const float*, float*, float*, hipStream_t);
size_t expand_masks_ws_bytes(int n, int ld_sig);
int launch_mask_rowbits(const float*, int, const uint32_t*, int, const int64_t*, int64_t, float, int32_t*, void*, size_t, hipStream_t);
int launch_expand_rows(const void*, int, int, const int32_t*, int, const int64_t*, const float*, int, int64_t, const float*, float, uint8_t*, hipStream_t);
int launch_mask_overlaps(const uint8_t*, int64_t, int, const int32_t*, int64_t, int, int32_t*, hipStream_t);
int launch_expand_masks(const float*, int, const uint32_t*, int, const int64_t*, const float*, int, int64_t, float, const float*, float, uint8_t*, int32_t*, void*, size_t, hipStream_t);
int launch_row_argmax(const float*, int, int64_t, const int32_t*, int, int64_t*, hipStream_t);
int launch_gather_i64(const int64_t*, const int64_t*, int64_t, int, int64_t*, hipStream_t);
int launch_panoptic(const uint8_t*, int64_t, const int32_t*, const int32_t*, int, int, int, const int64_t*, int32_t*, int32_t*, int64_t*, int64_t*, hipStream_t);
int launch_instance_boxes(const float*, int, int64_t, const uint8_t*, int64_t, int, int, float*, float*, void*, size_t, hipStream_t);
int launch_scale_shift_act(const float*, int, int, const float*, int, const float*, const float*, int, int64_t, int, const float*, int, float*, int,
                           hipStream_t);

#define ST ((hipStream_t)stream)

extern "C" {

int sd3d_abi_version(void) { return SD3D_ABI_VERSION; }
const char* sd3d_last_error(void) { return g_err; }

int sd3d_selftest_host(void) {
    // Z-order codec round trip + parent relation (runs on the host, no GPU needed)
    uint32_t s = 12345u;
    for (int it = 0; it < 20000; ++it) {
        s = s * 1664525u + 1013904223u; const uint32_t x = (s >> 8) & 0xFFFF;
        s = s * 1664525u + 1013904223u; const uint32_t y = (s >> 8) & 0xFFFF;
        s = s * 1664525u + 1013904223u; const uint32_t z = (s >> 8) & 0xFFFF;
        const uint64_t m = morton_encode(x, y, z);
        uint32_t a, b, c;
        morton_decode(m, a, b, c);
        if (a != x || b != y || c != z) return sd3d_set_error(-100, "morton round trip failed");
        if ((m >> 3) != morton_encode(x >> 1, y >> 1, z >> 1)) return sd3d_set_error(-101, "morton parent relation failed");
        if ((m & 7ull) != ((x & 1) | ((y & 1) << 1) | ((z & 1) << 2))) return sd3d_set_error(-102, "morton child bits failed");
        if (m >> 48) return sd3d_set_error(-103, "morton exceeds 48 bits");
    }
    return 0;
}

size_t sd3d_sort_ws_bytes(int64_t n) { return sort_ws_bytes(n); }
int sd3d_sort_pairs_u64(uint64_t* keys_in, uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out, uint32_t* vals_scratch,
                        int64_t n, int begin_bit, int end_bit, void* ws, size_t ws_bytes, void* stream) {
    if (n < 0 || begin_bit < 0 || end_bit > 64 || end_bit <= begin_bit) return sd3d_set_error(SD3D_ERR_ARG, "sort: bad arguments");
    return sort_pairs_u64(keys_in, vals_in, keys_out, vals_out, n, begin_bit, end_bit, ws, ws_bytes, ST, vals_scratch, nullptr);
}
int sd3d_sort_pairs_u64_ex(uint64_t* keys_in, uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out, uint32_t* vals_scratch,
                           int64_t n, int begin_bit, int end_bit, void* ws, size_t ws_bytes, int* landed_in_input, void* stream) {
    if (n < 0 || begin_bit < 0 || end_bit > 64 || end_bit <= begin_bit || !landed_in_input) return sd3d_set_error(SD3D_ERR_ARG, "sort: bad arguments");
    return sort_pairs_u64(keys_in, vals_in, keys_out, vals_out, n, begin_bit, end_bit, ws, ws_bytes, ST, vals_scratch, landed_in_input);
}
size_t sd3d_scan_ws_bytes(int64_t n) { return scan_ws_bytes(n > 0 ? n : 1); }
int sd3d_scan_exclusive_i32(const int32_t* in, int32_t* out, int64_t n, int32_t* total_dev, void* ws, size_t ws_bytes, void* stream) {
    return scan_exclusive_i32(in, out, n, nullptr, total_dev, ws, ws_bytes, ST);
}
int sd3d_keys_from_f32(const float* x, int64_t n, int descending, uint64_t* keys, void* stream) {
    return launch_f32_to_sortkey(x, n, descending, keys, ST);
}
int sd3d_keys_from_i64(const int64_t* x, int64_t n, uint64_t* keys, void* stream) { return launch_i64_to_sortkey(x, n, keys, ST); }
int sd3d_keys_from_i64_checked(const int64_t* x, int64_t n, uint64_t* keys, int bits, int32_t* flag, int flag_value, void* stream) {
    return launch_i64_to_sortkey_checked(x, n, keys, bits, flag, flag_value, ST);
}
int sd3d_keys_from_i64_checked_max(const int64_t* x, int64_t n, uint64_t* keys, int bits, int32_t* flag, int flag_value, int32_t* max_out, void* stream) {
    return launch_i64_to_sortkey_checked_max(x, n, keys, bits, flag, flag_value, max_out, ST, 0);
}
int sd3d_keys_from_i64_offset_checked_max(const int64_t* x, int64_t n, int64_t add, uint64_t* keys, int bits, int32_t* flag, int flag_value,
                                          int32_t* max_out, void* stream) {
    return launch_i64_to_sortkey_checked_max(x, n, keys, bits, flag, flag_value, max_out, ST, (uint64_t)add);
}

size_t sd3d_scene_stats_ws_bytes(void) { return 256 * 9 * sizeof(float); }
int sd3d_scene_stats(const float* points, int ld, int64_t n, float* stats, void* ws, size_t ws_bytes, void* stream) {
    return launch_scene_stats(points, ld, n, stats, ws, ws_bytes, ST, nullptr, 0);
}
int sd3d_voxel_keys(const float* points, int ld, int64_t n, float inv_voxel, const float* stats, int shift_to_min,
                    int batch_index, int32_t* origin, uint64_t* keys, int32_t* icoords, int32_t* err_flag, void* stream) {
    return launch_voxel_keys(points, ld, n, inv_voxel, stats, shift_to_min, batch_index, origin, keys, icoords, err_flag, ST);
}
size_t sd3d_unique_ws_bytes(int64_t n_cap) { return unique_ws_bytes(n_cap > 0 ? n_cap : 1); }
int sd3d_unique_sorted(const uint64_t* keys, const uint32_t* src_idx, int64_t n_cap, const int32_t* n_dev, int shift,
                       uint64_t* ukeys, int32_t* seg_start, int32_t* map, int32_t* n_unique_dev, void* ws, size_t ws_bytes,
                       const float* clip_stats, float clip_inv_voxel, int clip_level, int clip_min_shape, void* stream) {
    return launch_unique_sorted(keys, src_idx, n_cap, n_dev, shift, ukeys, seg_start, map, n_unique_dev, ws, ws_bytes,
                                clip_stats, clip_inv_voxel, clip_level, clip_min_shape, ST);
}
size_t sd3d_unique_levels_ws_bytes(int64_t n_cap, int n_extra) { return unique_levels_ws_bytes(n_cap > 0 ? n_cap : 1, n_extra > 0 ? n_extra : 1); }
int sd3d_unique_levels(const uint64_t* keys, int64_t n_cap, const int32_t* n_dev, int n_extra, uint64_t* const* ukeys, int32_t* const* parents,
                       int32_t* counts, void* ws, size_t ws_bytes, void* stream) {
    return launch_unique_levels(keys, n_cap, n_dev, n_extra, ukeys, parents, counts, ws, ws_bytes, ST);
}
int sd3d_voxel_levels_all(const uint64_t* sorted_keys, const uint32_t* src_idx, int64_t n, int n_levels, uint64_t* const* ukeys, int32_t* seg_start,
                          int32_t* map, int32_t* const* parents, int32_t* counts, void* ws, size_t ws_bytes, void* stream) {
    return launch_voxel_levels_all(sorted_keys, src_idx, n, n_levels, ukeys, seg_start, map, parents, counts, ws, ws_bytes, ST);
}
size_t sd3d_voxelise_scene_ws_bytes(int64_t n, int n_levels) {
    n = n > 0 ? n : 1;
    size_t b = sd3d_scene_stats_ws_bytes();
    const size_t c[3] = {sort_ws_bytes(n), unique_ws_bytes(n), unique_levels_ws_bytes(n, n_levels > 1 ? n_levels : 1)};
    for (size_t v : c) b = v > b ? v : b;
    return b;
}
int sd3d_voxelise_scene(const sd3d_voxelise_desc* d, int* sorted_in_a, void* stream) {
    if (!d || !sorted_in_a || !d->points || !d->stats || !d->origin || !d->keys_a || !d->keys_b || !d->vals_a || !d->vals_b || !d->ukeys0 ||
        !d->seg_start || !d->inverse || !d->readback || !d->ws)
        return sd3d_set_error(SD3D_ERR_ARG, "voxelise_scene: null pointer");
    if (d->n <= 0 || d->n_levels < 1 || d->n_levels > 8 || (d->n_levels > 1 && (!d->ukeys || !d->parents)) || d->key_bits < 8 || d->key_bits > 64)
        return sd3d_set_error(SD3D_ERR_ARG, "voxelise_scene: n > 0, 1..8 levels, 8..64 key bits");
    if (d->ws_bytes < sd3d_voxelise_scene_ws_bytes(d->n, d->n_levels)) return sd3d_set_error(SD3D_ERR_ARG, "voxelise_scene: workspace too small");
    if (d->superpoints && !d->sp_keys) return sd3d_set_error(SD3D_ERR_ARG, "voxelise_scene: superpoints without sp_keys");
    hipStream_t st = (hipStream_t)stream;
    const int L = d->n_levels;
    int rc = launch_scene_stats(d->points, d->ld, d->n, d->stats, d->ws, d->ws_bytes, st, d->readback, L + 2);     // (zeroes the read-back array too)
    if (rc) return rc;
    rc = launch_voxel_keys(d->points, d->ld, d->n, d->inv_voxel, d->stats, d->shift_to_min, 0, d->origin, d->keys_a, d->icoords, d->readback + L, st);
    if (rc) return rc;
    int landed = 0;
    rc = sort_pairs_u64(d->keys_a, nullptr, d->keys_b, d->vals_b, d->n, 0, d->key_bits, d->ws, d->ws_bytes, st, d->vals_a, &landed);
    if (rc) return rc;
    *sorted_in_a = landed;
    const uint64_t* skeys = landed ? d->keys_a : d->keys_b;
    const uint32_t* sidx = landed ? d->vals_a : d->vals_b;
    {   // every level from the sorted point keys in four launches (level 0 and the coarser levels used to be four each)
        uint64_t* uk[9];
        uk[0] = d->ukeys0;
        for (int l = 1; l < L; ++l) uk[l] = d->ukeys[l - 1];
        rc = launch_voxel_levels_all(skeys, sidx, d->n, L, uk, d->seg_start, d->inverse, d->parents, d->readback, d->ws, d->ws_bytes, st);
        if (rc) return rc;
    }
    if (d->superpoints) {
        rc = launch_i64_to_sortkey_checked_max(d->superpoints, d->n, d->sp_keys, d->sp_bits, d->readback + L, 4, d->readback + L + 1, st, 0);
        if (rc) return rc;
    }
    return SD3D_OK;
}
int sd3d_hash_build(const uint64_t* ukeys, int64_t n, uint64_t* table_keys, int32_t* table_vals, int64_t capacity, void* stream) {
    return launch_hash_build(ukeys, n, table_keys, table_vals, capacity, ST);
}
int sd3d_kernel_map(const uint64_t* out_keys, int64_t n_out, const uint64_t* table_keys, const int32_t* table_vals,
                    int64_t capacity, const int8_t* offsets, int K, int mirrored, int32_t* nbr, int32_t* pair_count, void* stream) {
    if (capacity <= 0 || (capacity & (capacity - 1))) return sd3d_set_error(SD3D_ERR_ARG, "kernel_map: capacity must be a power of two");
    return launch_kernel_map(out_keys, n_out, table_keys, table_vals, capacity, offsets, K, mirrored, nbr, pair_count, ST);
}
size_t sd3d_kernel_maps_hier_ws_bytes(int n_levels, const int64_t* n) { return kernel_maps_hier_ws_bytes(n_levels, n); }
int sd3d_kernel_maps_hier(int n_levels, const uint64_t* const* keys, const int32_t* const* parent, const int64_t* n, int32_t* const* nbr3,
                          int32_t* nbr5, const int8_t* offsets3, const int8_t* offsets5, const int8_t* inv27, int32_t* pair_counts,
                          const int32_t* perm8, int32_t* const* nbr_down, int32_t* const* nbr_up, void* ws, size_t ws_bytes, void* stream) {
    if (!keys || !parent || !n || !nbr3 || !offsets3 || !inv27 || (nbr5 && !offsets5)) return sd3d_set_error(SD3D_ERR_ARG, "kernel_maps_hier: null pointer");
    return launch_kernel_maps_hier(n_levels, keys, parent, n, nbr3, nbr5, offsets3, offsets5, inv27, pair_counts, perm8, nbr_down, nbr_up, ws,
                                   ws_bytes, ST);
}
int sd3d_stride_maps(const uint64_t* fine_keys, const int32_t* parent, int64_t n_fine, int64_t n_coarse, const int32_t* perm8,
                     int32_t* nbr_down, int32_t* nbr_up, void* stream) {
    return launch_stride_maps(fine_keys, parent, n_fine, n_coarse, perm8, nbr_down, nbr_up, ST);
}
int sd3d_voxel_mean(const float* points, int ld_points, const float* feats2d, int F, int mode, const float* stats,
                    int64_t n_points, const uint32_t* sorted_idx, const int32_t* seg_start, int64_t n_vox, float* out,
                    int ld_out, void* stream) {
    return launch_voxel_mean(points, ld_points, feats2d, F, mode, stats, n_points, sorted_idx, seg_start, n_vox, out, ld_out, ST);
}
int sd3d_segment_starts(const uint64_t* sorted_ids, int64_t n, int64_t S, int32_t* start, void* stream) {
    return launch_segment_starts(sorted_ids, n, S, start, ST);
}
int sd3d_segment_starts_batch(const uint64_t* sorted_ids, int64_t n, int64_t S, const int32_t* id_off, int n_scenes, int32_t* start,
                              void* stream) {
    if (!id_off) return sd3d_set_error(SD3D_ERR_ARG, "segment_starts_batch: id_off is NULL");
    return launch_segment_starts_batch(sorted_ids, n, S, id_off, n_scenes, start, ST);
}
int sd3d_voxel_mean_batch(const sd3d_scene_src* scenes, int n_scenes, int F, int mode, const uint64_t* ukeys,
                          const uint32_t* sorted_idx, const int32_t* seg_start, int64_t n_vox, float* out, int ld_out, void* stream) {
    if (!scenes) return sd3d_set_error(SD3D_ERR_ARG, "voxel_mean_batch: scenes is NULL");
    return launch_voxel_mean_batch(scenes, n_scenes, F, mode, ukeys, sorted_idx, seg_start, n_vox, out, ld_out, ST);
}
int sd3d_keys_from_i64_offset(const int64_t* x, int64_t n, int64_t add, uint64_t* keys, void* stream) {
    return launch_i64_to_sortkey_add(x, n, (uint64_t)add, keys, ST);
}
int sd3d_pool_superpoints(const float* feat, int ld_feat, int C, const int32_t* inverse, const int32_t* icoords,
                          float voxel_size, const uint32_t* sorted_idx, const int32_t* start, int64_t S, float* out_feat,
                          float* out_pos, void* stream) {
    return launch_pool_superpoints(feat, ld_feat, C, inverse, icoords, voxel_size, sorted_idx, start, S, out_feat, out_pos, ST);
}

int sd3d_gather_gemm(const float* in0, int ld0, int C0, const float* in1, int ld1, const int32_t* nbr, const float* wt, int K,
                     int Cin, int Cout, int64_t M, const float* scale, const float* shift, const float* res, int ld_res,
                     float* out, int ld_out, int act, int nt, void* ws, size_t ws_bytes, void* stream) {
    GGParams p;
    p.in0 = in0; p.ld0 = ld0; p.C0 = C0; p.in1 = in1; p.ld1 = ld1; p.nbr = nbr; p.wt = wt; p.K = K; p.Cin = Cin; p.Cout = Cout;
    p.M = M; p.scale = scale; p.shift = shift; p.res = res; p.ld_res = ld_res; p.out = out; p.ld_out = ld_out; p.act = act;
    p.col_groups = 1;
    p.ksplit = 1;
    p.ws = nullptr;
    return launch_gather_gemm(p, nt, ws, ws_bytes, ST);
}

int sd3d_gather_gemm_split(const float* in0, int ld0, int C0, const float* in1, int ld1, const int32_t* nbr,
                           const uint16_t* wt_split, int terms, int K, int Cin, int Cout, int64_t M, const float* scale,
                           const float* shift, const float* res, int ld_res, float* out, int ld_out, int act, int nt,
                           void* ws, size_t ws_bytes, void* stream) {
    GGParams p;
    p.in0 = in0; p.ld0 = ld0; p.C0 = C0; p.in1 = in1; p.ld1 = ld1; p.nbr = nbr; p.wt = nullptr; p.K = K; p.Cin = Cin; p.Cout = Cout;
    p.M = M; p.scale = scale; p.shift = shift; p.res = res; p.ld_res = ld_res; p.out = out; p.ld_out = ld_out; p.act = act;
    p.col_groups = 1;
    p.ksplit = 1;
    p.ws = nullptr;
    return launch_gather_gemm_split(p, nt, terms, wt_split, ws, ws_bytes, ST);
}

int sd3d_fourier_pe(const float* xyz, int ld_xyz, int64_t n, const float* range, const float* gauss_b, int ld_b, int d_pos, float* out,
                    int ld_out, void* stream) {
    return launch_fourier_pe(xyz, ld_xyz, n, range, gauss_b, ld_b, d_pos, out, ld_out, nullptr, ST);
}
int sd3d_fourier_pe_rows(const float* xyz, int ld_xyz, int64_t n, const float* ranges, const int32_t* row_scene, const float* gauss_b, int ld_b,
                         int d_pos, float* out, int ld_out, void* stream) {
    return launch_fourier_pe(xyz, ld_xyz, n, ranges, gauss_b, ld_b, d_pos, out, ld_out, row_scene, ST);
}

int sd3d_dense_plan_code(int64_t rows, int Cin, int Cout) { return dense_plan_code(rows, Cin, Cout); }

int sd3d_linear_group(int n, const sd3d_linear_job* jobs, void* stream) {
    GGParams g[8];
    if (n > 8) return sd3d_set_error(SD3D_ERR_ARG, "linear_group: at most 8 jobs per launch");
    for (int i = 0; i < n; ++i) {
        const sd3d_linear_job& J = jobs[i];
        GGParams& p = g[i];
        p.in0 = J.in0; p.ld0 = J.ld0; p.C0 = J.C0; p.in1 = J.in1; p.ld1 = J.ld1; p.nbr = nullptr; p.wt = J.wt; p.K = 1; p.Cin = J.Cin;
        p.Cout = J.Cout; p.M = J.M; p.scale = nullptr; p.shift = J.shift; p.res = J.res; p.ld_res = J.ld_res; p.out = J.out;
        p.ld_out = J.ld_out; p.act = J.act; p.col_groups = 1; p.ksplit = 1; p.ws = nullptr;
    }
    return launch_linear_group(n, g, ST);
}

size_t sd3d_pair_lists_ws_bytes(int K, int64_t M) { return pair_lists_ws_bytes(K, M); }
int sd3d_pair_lists(const int32_t* nbr, int K, int64_t M, int64_t p_cap, int32_t* pos, int32_t* in_idx, int32_t* tile_k, void* ws,
                    size_t ws_bytes, void* stream) {
    return launch_pair_lists(nbr, K, M, p_cap, pos, in_idx, tile_k, ws, ws_bytes, ST);
}
int sd3d_pair_lists_batch(int n, const int32_t* const* nbr, const int* K, const int64_t* M, const int64_t* p_cap, int32_t* const* pos,
                          int32_t* const* in_idx, int32_t* const* tile_k, void* ws, size_t ws_bytes, void* stream) {
    return launch_pair_lists_batch(n, nbr, K, M, p_cap, pos, in_idx, tile_k, ws, ws_bytes, ST);
}
int sd3d_pair_conv(const float* in0, int ld0, int C0, const float* in1, int ld1, const int32_t* in_idx, const int32_t* tile_k,
                   int64_t p_cap, const int32_t* pos, const float* wt, int K, int Cin, int Cout, int64_t M, const float* scale,
                   const float* shift, const float* res, int ld_res, float* out, int ld_out, int act, float* part,
                   size_t part_bytes, void* stream) {
    return launch_pair_conv(in0, ld0, C0, in1, ld1, in_idx, tile_k, p_cap, pos, nullptr, 0, -1, nullptr, wt, K, Cin, Cout, M, scale, shift,
                            res, ld_res, out, ld_out, act, part, part_bytes, ST);
}
int sd3d_pair_lists_desc(int n, const sd3d_pair_table_desc* tables, void* ws, size_t ws_bytes, void* stream) {
    if (n > 0 && !tables) return sd3d_set_error(SD3D_ERR_ARG, "pair_lists_desc: tables is NULL");
    return launch_pair_lists_desc(n, tables, ws, ws_bytes, ST);
}
int sd3d_pair_conv_ex(const float* in0, int ld0, int C0, const float* in1, int ld1, const int32_t* in_idx, const int32_t* tile_k,
                      int64_t p_cap, const int32_t* pos, const int32_t* rlist, int rl_stride, int center, const int32_t* out_idx,
                      const float* wt, int K, int Cin, int Cout, int64_t M, const float* scale, const float* shift, const float* res,
                      int ld_res, float* out, int ld_out, int act, float* part, size_t part_bytes, void* stream) {
    return launch_pair_conv(in0, ld0, C0, in1, ld1, in_idx, tile_k, p_cap, pos, rlist, rl_stride, center, out_idx, wt, K, Cin, Cout, M,
                            scale, shift, res, ld_res, out, ld_out, act, part, part_bytes, ST);
}

int sd3d_layernorm(const float* x, int ld_x, const float* res, int ld_res, const float* w, const float* b, float eps, int64_t M,
                   int D, float* out, int ld_out, int act, void* stream) {
    return launch_layernorm(x, ld_x, res, ld_res, w, b, eps, M, D, out, ld_out, act, ST);
}
int sd3d_linear_layernorm(const float* x, int ld_x, int64_t M, int Cin, const float* wt, int Cout, const float* bias, const float* res, int ld_res,
                          const float* ln_w, const float* ln_b, float eps, int act, float* out, int ld_out, void* stream) {
    return launch_linear_layernorm(x, ld_x, M, Cin, wt, Cout, bias, res, ld_res, ln_w, ln_b, eps, act, out, ld_out, ST);
}
int sd3d_sine_pe(const float* xyz, int ld_xyz, int64_t n, const float* range, const float* dim_t, const int8_t* axis, int d_pos,
                 const float* mod_num, int ld_num, const float* mod_den, int ld_den, float* out, int ld_out, void* stream) {
    if (mod_num && !mod_den) return sd3d_set_error(SD3D_ERR_ARG, "sine_pe: mod_den missing");
    return launch_sine_pe(xyz, ld_xyz, n, range, dim_t, axis, d_pos, mod_num, ld_num, mod_den, ld_den, out, ld_out, nullptr, ST);
}
int sd3d_sine_pe_rows(const float* xyz, int ld_xyz, int64_t n, const float* ranges, const int32_t* row_scene, const float* dim_t,
                      const int8_t* axis, int d_pos, const float* mod_num, int ld_num, const float* mod_den, int ld_den, float* out,
                      int ld_out, void* stream) {
    if (mod_num && !mod_den) return sd3d_set_error(SD3D_ERR_ARG, "sine_pe_rows: mod_den missing");
    return launch_sine_pe(xyz, ld_xyz, n, ranges, dim_t, axis, d_pos, mod_num, ld_num, mod_den, ld_den, out, ld_out, row_scene, ST);
}
static int attention_batch_impl(int n, const sd3d_attn_job* jobs, int H, float scale, int bf16, void* ws, size_t ws_bytes, void* stream,
                                int32_t* ksplit_out, int64_t* part_off_out);
int sd3d_attention_batch(int n, const sd3d_attn_job* jobs, int H, float scale, int bf16, void* ws, size_t ws_bytes, void* stream) {
    return attention_batch_impl(n, jobs, H, scale, bf16, ws, ws_bytes, stream, nullptr, nullptr);
}
int sd3d_attention_batch_parts(int n, const sd3d_attn_job* jobs, int H, float scale, int bf16, void* ws, size_t ws_bytes,
                               int32_t* ksplit_out_host, int64_t* part_off_out_host, void* stream) {
    if (!ksplit_out_host || !part_off_out_host) return sd3d_set_error(SD3D_ERR_ARG, "attention_batch_parts: output arrays missing");
    return attention_batch_impl(n, jobs, H, scale, bf16, ws, ws_bytes, stream, ksplit_out_host, part_off_out_host);
}
static int attention_batch_impl(int n, const sd3d_attn_job* jobs, int H, float scale, int bf16, void* ws, size_t ws_bytes, void* stream,
                                int32_t* ksplit_out, int64_t* part_off_out) {
    if (n <= 0) return SD3D_OK;
    if (n > SD3D_MAX_BATCH || !jobs) return sd3d_set_error(SD3D_ERR_ARG, "attention_batch: 1..16 jobs");
    AttnParams p[SD3D_MAX_BATCH];
    const bool two = jobs[0].q1 != nullptr;
    for (int i = 0; i < n; ++i) {
        const sd3d_attn_job& j = jobs[i];
        if ((j.q1 == nullptr) != (j.k1 == nullptr) || (j.q1 != nullptr) != two) return sd3d_set_error(SD3D_ERR_ARG, "attention_batch: all jobs need the same sources");
        p[i].q[0] = j.q0; p[i].ldq[0] = j.ldq0; p[i].q[1] = j.q1; p[i].ldq[1] = j.ldq1;
        p[i].k[0] = j.k0; p[i].ldk[0] = j.ldk0; p[i].k[1] = j.k1; p[i].ldk[1] = j.ldk1;
        p[i].v = j.v; p[i].ldv = j.ldv; p[i].bits = j.mask_bits; p[i].nwords = (j.Lk + 31) / 32; p[i].out = j.out; p[i].ldo = j.ldo;
        p[i].Lq = j.Lq; p[i].Lk = j.Lk; p[i].H = H; p[i].scale = scale; p[i].ksplit = 1; p[i].part = nullptr; p[i].bf16 = bf16 ? 1 : 0; p[i].lse = nullptr;
    }
    return launch_attention_batch(n, p, two ? 2 : 1, ws, ws_bytes, ST, ksplit_out, part_off_out);
}
int sd3d_mask_bits_batch(int n, const float* const* logits, const int* ld, const int64_t* Q, const int* S, uint32_t* const* bits,
                         const int* nwords, float thr, void* stream) {
    return launch_mask_bits_batch(n, logits, ld, Q, S, bits, nwords, thr, ST);
}
int sd3d_dinox_mask_bits_batch(int n, const uint32_t* const* blocked, const uint32_t* const* near, const int* nwords, const int64_t* Q,
                               const int64_t* Mq, uint32_t* const* out, const int* nwords_out, void* stream) {
    return launch_dinox_mask_bits_batch(n, blocked, near, nwords, Q, Mq, out, nwords_out, ST);
}
size_t sd3d_attention_ws_bytes(int Lq, int H) { return attention_ws_bytes(Lq, H); }
int sd3d_attention(const float* q0, int ldq0, const float* q1, int ldq1, const float* k0, int ldk0, const float* k1, int ldk1,
                   const float* v, int ldv, const uint32_t* mask_bits, int Lq, int Lk, int H, float scale, float* out, int ldo,
                   void* ws, size_t ws_bytes, void* stream) {
    if ((q1 == nullptr) != (k1 == nullptr)) return sd3d_set_error(SD3D_ERR_ARG, "attention: q1 and k1 must be given together");
    AttnParams p;
    p.q[0] = q0; p.ldq[0] = ldq0; p.q[1] = q1; p.ldq[1] = ldq1;
    p.k[0] = k0; p.ldk[0] = ldk0; p.k[1] = k1; p.ldk[1] = ldk1;
    p.v = v; p.ldv = ldv; p.bits = mask_bits; p.nwords = (Lk + 31) / 32; p.out = out; p.ldo = ldo;
    p.Lq = Lq; p.Lk = Lk; p.H = H; p.scale = scale; p.ksplit = 1; p.part = nullptr; p.bf16 = 0; p.lse = nullptr;
    return launch_attention(p, q1 ? 2 : 1, ws, ws_bytes, ST);
}
int sd3d_attention_lse(const float* q0, int ldq0, const float* q1, int ldq1, const float* k0, int ldk0, const float* k1, int ldk1,
                   const float* v, int ldv, const uint32_t* mask_bits, int Lq, int Lk, int H, float scale, float* out, int ldo,
                   float* lse, void* ws, size_t ws_bytes, void* stream) {
    if ((q1 == nullptr) != (k1 == nullptr)) return sd3d_set_error(SD3D_ERR_ARG, "attention: q1 and k1 must be given together");
    AttnParams p;
    p.q[0] = q0; p.ldq[0] = ldq0; p.q[1] = q1; p.ldq[1] = ldq1;
    p.k[0] = k0; p.ldk[0] = ldk0; p.k[1] = k1; p.ldk[1] = ldk1;
    p.v = v; p.ldv = ldv; p.bits = mask_bits; p.nwords = (Lk + 31) / 32; p.out = out; p.ldo = ldo;
    p.Lq = Lq; p.Lk = Lk; p.H = H; p.scale = scale; p.ksplit = 1; p.part = nullptr; p.bf16 = 0; p.lse = lse;
    return launch_attention(p, q1 ? 2 : 1, ws, ws_bytes, ST);
}
int sd3d_attention_lse_bf16(const float* q0, int ldq0, const float* q1, int ldq1, const float* k0, int ldk0, const float* k1, int ldk1,
                   const float* v, int ldv, const uint32_t* mask_bits, int Lq, int Lk, int H, float scale, float* out, int ldo,
                   float* lse, void* ws, size_t ws_bytes, void* stream) {
    if ((q1 == nullptr) != (k1 == nullptr)) return sd3d_set_error(SD3D_ERR_ARG, "attention: q1 and k1 must be given together");
    AttnParams p;
    p.q[0] = q0; p.ldq[0] = ldq0; p.q[1] = q1; p.ldq[1] = ldq1;
    p.k[0] = k0; p.ldk[0] = ldk0; p.k[1] = k1; p.ldk[1] = ldk1;
    p.v = v; p.ldv = ldv; p.bits = mask_bits; p.nwords = (Lk + 31) / 32; p.out = out; p.ldo = ldo;
    p.Lq = Lq; p.Lk = Lk; p.H = H; p.scale = scale; p.ksplit = 1; p.part = nullptr; p.bf16 = 1; p.lse = lse;
    return launch_attention(p, q1 ? 2 : 1, ws, ws_bytes, ST);
}
int sd3d_attention_bf16(const float* q0, int ldq0, const float* q1, int ldq1, const float* k0, int ldk0, const float* k1, int ldk1,
                   const float* v, int ldv, const uint32_t* mask_bits, int Lq, int Lk, int H, float scale, float* out, int ldo,
                   void* ws, size_t ws_bytes, void* stream) {
    if ((q1 == nullptr) != (k1 == nullptr)) return sd3d_set_error(SD3D_ERR_ARG, "attention: q1 and k1 must be given together");
    AttnParams p;
    p.q[0] = q0; p.ldq[0] = ldq0; p.q[1] = q1; p.ldq[1] = ldq1;
    p.k[0] = k0; p.ldk[0] = ldk0; p.k[1] = k1; p.ldk[1] = ldk1;
    p.v = v; p.ldv = ldv; p.bits = mask_bits; p.nwords = (Lk + 31) / 32; p.out = out; p.ldo = ldo;
    p.Lq = Lq; p.Lk = Lk; p.H = H; p.scale = scale; p.ksplit = 1; p.part = nullptr; p.bf16 = 1; p.lse = nullptr;
    return launch_attention(p, q1 ? 2 : 1, ws, ws_bytes, ST);
}
int sd3d_mask_bits(const float* logits, int ld, int64_t Q, int S, float thr, uint32_t* bits, int nwords, void* stream) {
    return launch_mask_bits(logits, ld, Q, S, thr, bits, nwords, ST);
}
int sd3d_near_bits(const float* sp_pos, int64_t S, const float* centers, int64_t M, float thr, uint32_t* near, int nwords, void* stream) {
    return launch_near_bits(sp_pos, S, centers, M, thr, near, nwords, ST);
}
int sd3d_dinox_mask_bits(const uint32_t* blocked, const uint32_t* near, int nwords, int64_t Q, int64_t M, uint32_t* out,
                         int nwords_out, void* stream) {
    return launch_dinox_mask_bits(blocked, near, nwords, Q, M, out, nwords_out, ST);
}
int sd3d_box_refine(const float* ref_points, const float* d_center, const float* size_prev, int ld_size_prev, const float* d_size,
                    const float* range, int normalize, int64_t Q, float* center, float* size, float* size_metric, void* stream) {
    return launch_box_refine(ref_points, d_center, size_prev, ld_size_prev, d_size, range, normalize, Q, center, size, size_metric, nullptr, ST);
}
int sd3d_box_refine_rows(const float* ref_points, const float* d_center, const float* size_prev, int ld_size_prev, const float* d_size,
                         const float* ranges, const int32_t* row_scene, int normalize, int64_t Q, float* center, float* size,
                         float* size_metric, void* stream) {
    return launch_box_refine(ref_points, d_center, size_prev, ld_size_prev, d_size, ranges, normalize, Q, center, size, size_metric, row_scene, ST);
}
int sd3d_class_scores(const float* cls, int ld, int64_t Q, int C, float* scores, float* rowmax, void* stream) {
    return launch_class_scores(cls, ld, Q, C, scores, rowmax, ST);
}
int sd3d_take_f32(const float* src, const uint32_t* idx, int n, float* out, void* stream) { return launch_take_f32(src, idx, n, out, ST); }
int sd3d_topk_desc_f32(const float* x, int64_t n, int k, uint32_t* idx, void* stream) { return launch_topk_desc(x, n, k, idx, ST); }
int sd3d_select_instances(const float* scores, const int32_t* count, int k, float thr0, float thr1, int npoint_thr, int32_t* keep, int32_t* pkeep,
                          int32_t* union_rows, int32_t* keep_u, int32_t* pkeep_u, uint8_t* score_mask, uint8_t* npoint_mask, int32_t* counts,
                          void* stream) {
    return launch_select_instances(scores, count, k, thr0, thr1, npoint_thr, keep, pkeep, union_rows, keep_u, pkeep_u, score_mask, npoint_mask, counts, ST);
}
int sd3d_take_instances(const int32_t* keep, int m, const int32_t* labels, const float* scores, const float* boxes, int64_t* labels_out,
                        float* scores_out, float* boxes_out, void* stream) {
    return launch_take_instances(keep, m, labels, scores, boxes, labels_out, scores_out, boxes_out, ST);
}
int sd3d_take_pair(const uint32_t* order, const int32_t* labels, const float* scores, int n, int32_t* labels_out, float* scores_out, void* stream) {
    return launch_take_pair(order, labels, scores, n, labels_out, scores_out, ST);
}
int sd3d_nms_finish(const uint32_t* order2, const float* scores2, const int32_t* labels1, const uint32_t* order1, const int32_t* qidx,
                    const float* centers, const float* sizes, int n, float* final_scores, int32_t* final_labels, int64_t* record, float* boxes,
                    void* stream) {
    return launch_nms_finish(order2, scores2, labels1, order1, qidx, centers, sizes, n, final_scores, final_labels, record, boxes, ST);
}
int sd3d_mask_scores(const float* masks, int ld, int S, const uint32_t* flat_idx, const float* score_in, int n, int C, int normalize,
                     int32_t* labels, int32_t* qidx, float* score_out, void* stream) {
    return launch_mask_scores(masks, ld, S, flat_idx, score_in, n, C, normalize, labels, qidx, score_out, ST);
}
int sd3d_gather_sigmoid(const float* masks, int ld, int S, const int32_t* qidx, const uint32_t* order, int n, float* sig, int ld_out,
                        float* area, void* stream) {
    return launch_gather_sigmoid(masks, ld, S, qidx, order, n, sig, ld_out, area, ST);
}
int sd3d_nms_decay(const float* inter, int ld, const float* area, const int32_t* labels, int n, int gaussian, float sigma,
                   const float* score_in, float* comp_ws, float* score_out, void* stream) {
    return launch_nms_decay(inter, ld, area, labels, n, gaussian, sigma, score_in, comp_ws, score_out, ST);
}
size_t sd3d_expand_masks_ws_bytes(int n, int ld_sig) { return expand_masks_ws_bytes(n, ld_sig); }
int sd3d_mask_rowbits(const float* sig, int ld_sig, const uint32_t* src_row, int n, const int64_t* superpoints, int64_t N, float sp_thr,
                      int32_t* count, void* ws, size_t ws_bytes, void* stream) {
    return launch_mask_rowbits(sig, ld_sig, src_row, n, superpoints, N, sp_thr, count, ws, ws_bytes, ST);
}
int sd3d_expand_rows(const void* ws, int n, int ld_sig, const int32_t* rows, int m, const int64_t* superpoints, const float* points, int ld_points,
                     int64_t N, const float* boxes, float loose_ratio, uint8_t* out, void* stream) {
    return launch_expand_rows(ws, n, ld_sig, rows, m, superpoints, points, ld_points, N, boxes, loose_ratio, out, ST);
}
int sd3d_expand_masks(const float* sig, int ld_sig, const uint32_t* src_row, int n, const int64_t* superpoints, const float* points,
                      int ld_points, int64_t N, float sp_thr, const float* boxes, float loose_ratio, uint8_t* out, int32_t* count,
                      void* ws, size_t ws_bytes, void* stream) {
    return launch_expand_masks(sig, ld_sig, src_row, n, superpoints, points, ld_points, N, sp_thr, boxes, loose_ratio, out, count, ws,
                               ws_bytes, ST);
}
int sd3d_row_chain(const sd3d_rc_program* program_host, void* stream) { return launch_row_chain(program_host, ST); }
size_t sd3d_row_chain_program_bytes(void) { return sizeof(sd3d_rc_program); }
int sd3d_pack_mask_rows(const uint8_t* masks, int64_t N, const int32_t* rows, int n_rows, uint8_t* out, int64_t nb, void* stream) {
    return launch_pack_mask_rows(masks, N, rows, n_rows, out, nb, ST);
}
// Host side of sd3d_pack_mask_rows: packed_host [n_rows, nb] bits -> out_host [n_rows, N] bytes (0 / 1).  One table lookup per
// input byte (8 output bytes at a time); plain C, no GPU, no threads - the caller's thread does it with the GIL released.
int sd3d_unpack_bits_host(const uint8_t* packed_host, int64_t n_rows, int64_t N, int64_t nb, uint8_t* out_host) {
    if (n_rows < 0 || N < 0 || nb != (N + 7) / 8) return sd3d_set_error(SD3D_ERR_ARG, "unpack_bits_host: nb != ceil(N / 8)");
    static uint64_t lut[256];
    static bool ready = false;
    if (!ready) {                                              // idempotent: racing callers write the same values
        for (int v = 0; v < 256; ++v) {
            uint64_t w = 0;
            for (int j = 0; j < 8; ++j) if (v >> j & 1) w |= 1ull << (8 * j);
            lut[v] = w;
        }
        __atomic_store_n(&ready, true, __ATOMIC_RELEASE);
    }
    const int64_t full = N / 8;
    for (int64_t r = 0; r < n_rows; ++r) {
        const uint8_t* src = packed_host + r * nb;
        uint8_t* dst = out_host + r * N;
        for (int64_t b = 0; b < full; ++b) { const uint64_t w = lut[src[b]]; __builtin_memcpy(dst + 8 * b, &w, 8); }
        for (int64_t p = full * 8; p < N; ++p) dst[p] = (src[full] >> (p - full * 8)) & 1;
    }
    return SD3D_OK;
}
int sd3d_row_argmax(const float* x, int ld, int64_t Q, const int32_t* cols, int ncols, int64_t* out, void* stream) {
    return launch_row_argmax(x, ld, Q, cols, ncols, out, ST);
}
int sd3d_gather_i64(const int64_t* table, const int64_t* idx, int64_t N, int use_index, int64_t* out, void* stream) {
    return launch_gather_i64(table, idx, N, use_index, out, ST);
}
int sd3d_panoptic(const uint8_t* masks, int64_t N, const int32_t* rows_desc, const int32_t* labels_desc, int n, int n_stuff,
                  int npoint_thr, const int64_t* sem_stuff, int32_t* inst_ws, int32_t* hist_ws, int64_t* sem_map, int64_t* inst_map,
                  void* stream) {
    return launch_panoptic(masks, N, rows_desc, labels_desc, n, n_stuff, npoint_thr, sem_stuff, inst_ws, hist_ws, sem_map, inst_map, ST);
}
size_t sd3d_instance_boxes_ws_bytes(int n_inst) { return (size_t)(n_inst > 0 ? n_inst : 1) * 64 * 10 * sizeof(float); }
int sd3d_instance_boxes(const float* points, int ld, int64_t N, const uint8_t* masks, int64_t mask_stride, int n_inst, int mode,
                        float* centers, float* sizes, void* ws, size_t ws_bytes, void* stream) {
    return launch_instance_boxes(points, ld, N, masks, mask_stride, n_inst, mode, centers, sizes, ws, ws_bytes, ST);
}
int sd3d_scale_shift_act(const float* x0, int ld0, int C0, const float* x1, int ld1, const float* scale, const float* shift, int act,
                         int64_t M, int C, float* out, int ld_out, void* stream) {
    return launch_scale_shift_act(x0, ld0, C0, x1, ld1, scale, shift, act, M, C, nullptr, 0, out, ld_out, ST);
}
int sd3d_scale_shift_act_add(const float* x0, int ld0, int C0, const float* x1, int ld1, const float* scale, const float* shift, int act,
                             int64_t M, int C, const float* add, int ld_add, float* out, int ld_out, void* stream) {
    return launch_scale_shift_act(x0, ld0, C0, x1, ld1, scale, shift, act, M, C, add, ld_add, out, ld_out, ST);
}

int sd3d_mask_overlaps(const uint8_t* masks, int64_t mask_stride, int n, const int32_t* gt_index, int64_t N, int n_cols,
                       int32_t* counts, void* stream) {
    return launch_mask_overlaps(masks, mask_stride, n, gt_index, N, n_cols, counts, ST);
}

}  // extern "C"
