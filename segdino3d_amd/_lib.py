"""ctypes binding of libsegdino3d_hip.so (the C ABI of include/segdino3d_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a tensor is not on a
HIP device the call raises.  (The CPU restatement lives in oracle/ and is test infrastructure.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SD3D_LIB: another build of the same library (same-box A/B of kernel variants; tools/ab_lib.sh)
LIB_PATH = os.environ.get("SD3D_LIB") or os.path.join(_HERE, "libsegdino3d_hip.so")
ABI_VERSION = 1

_lib = None

_p = C.c_void_p
_i = C.c_int
_l = C.c_int64
_z = C.c_size_t
_f = C.c_float

# name -> (restype, argtypes).  Must list every symbol include/segdino3d_hip.h declares
# (tests/test_capi_symbols.py cross-checks this table against the header).
SIGNATURES = {
    "sd3d_abi_version": (_i, []),
    "sd3d_last_error": (C.c_char_p, []),
    "sd3d_set_scenes_in_flight": (_i, [_i]),
    "sd3d_set_pair_pool": (_i, [_i]),
    "sd3d_pair_pool_check": (_i, []),
    "sd3d_pair_pool_launches": (_i, [_p]),
    "sd3d_pair_pool_poison": (_i, [_l]),
    "sd3d_selftest_host": (_i, []),
    "sd3d_sort_ws_bytes": (_z, [_l]),
    "sd3d_sort_pairs_u64": (_i, [_p, _p, _p, _p, _p, _l, _i, _i, _p, _z, _p]),
    "sd3d_sort_pairs_u64_ex": (_i, [_p, _p, _p, _p, _p, _l, _i, _i, _p, _z, _p, _p]),
    "sd3d_scan_ws_bytes": (_z, [_l]),
    "sd3d_scan_exclusive_i32": (_i, [_p, _p, _l, _p, _p, _z, _p]),
    "sd3d_keys_from_f32": (_i, [_p, _l, _i, _p, _p]),
    "sd3d_keys_from_i64": (_i, [_p, _l, _p, _p]),
    "sd3d_keys_from_i64_checked": (_i, [_p, _l, _p, _i, _p, _i, _p]),
    "sd3d_keys_from_i64_checked_max": (_i, [_p, _l, _p, _i, _p, _i, _p, _p]),
    "sd3d_keys_from_i64_offset_checked_max": (_i, [_p, _l, _l, _p, _i, _p, _i, _p, _p]),
    "sd3d_voxel_levels_all": (_i, [_p, _p, _l, _i, _p, _p, _p, _p, _p, _p, _z, _p]),
    "sd3d_voxelise_scene_ws_bytes": (_z, [_l, _i]),
    "sd3d_voxelise_scene": (_i, [_p, _p, _p]),
    "sd3d_scene_stats_ws_bytes": (_z, []),
    "sd3d_scene_stats": (_i, [_p, _i, _l, _p, _p, _z, _p]),
    "sd3d_voxel_keys": (_i, [_p, _i, _l, _f, _p, _i, _i, _p, _p, _p, _p, _p]),
    "sd3d_unique_ws_bytes": (_z, [_l]),
    "sd3d_unique_levels_ws_bytes": (_z, [_l, _i]),
    "sd3d_unique_levels": (_i, [_p, _l, _p, _i, _p, _p, _p, _p, _z, _p]),
    "sd3d_unique_sorted": (_i, [_p, _p, _l, _p, _i, _p, _p, _p, _p, _p, _z, _p, _f, _i, _i, _p]),
    "sd3d_hash_build": (_i, [_p, _l, _p, _p, _l, _p]),
    "sd3d_kernel_map": (_i, [_p, _l, _p, _p, _l, _p, _i, _i, _p, _p, _p]),
    "sd3d_kernel_maps_hier_ws_bytes": (_z, [_i, _p]),
    "sd3d_kernel_maps_hier": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "sd3d_stride_maps": (_i, [_p, _p, _l, _l, _p, _p, _p, _p]),
    "sd3d_voxel_mean": (_i, [_p, _i, _p, _i, _i, _p, _l, _p, _p, _l, _p, _i, _p]),
    "sd3d_segment_starts": (_i, [_p, _l, _l, _p, _p]),
    "sd3d_voxel_mean_batch": (_i, [_p, _i, _i, _i, _p, _p, _p, _l, _p, _i, _p]),
    "sd3d_keys_from_i64_offset": (_i, [_p, _l, _l, _p, _p]),
    "sd3d_segment_starts_batch": (_i, [_p, _l, _l, _p, _i, _p, _p]),
    "sd3d_pool_superpoints": (_i, [_p, _i, _i, _p, _p, _f, _p, _p, _l, _p, _p, _p]),
    "sd3d_gather_gemm": (_i, [_p, _i, _i, _p, _i, _p, _p, _i, _i, _i, _l, _p, _p, _p, _i, _p, _i, _i, _i, _p, _z, _p]),
    "sd3d_dense_plan_code": (_i, [_l, _i, _i]),
    "sd3d_linear_group": (_i, [_i, _p, _p]),
    "sd3d_gather_gemm_split": (_i, [_p, _i, _i, _p, _i, _p, _p, _i, _i, _i, _i, _l, _p, _p, _p, _i, _p, _i, _i, _i, _p, _z, _p]),
    "sd3d_pair_lists_ws_bytes": (_z, [_i, _l]),
    "sd3d_pair_lists": (_i, [_p, _i, _l, _l, _p, _p, _p, _p, _z, _p]),
    "sd3d_pair_lists_batch": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "sd3d_pair_conv": (_i, [_p, _i, _i, _p, _i, _p, _p, _l, _p, _p, _i, _i, _i, _l, _p, _p, _p, _i, _p, _i, _i, _p, _z, _p]),
    "sd3d_pair_lists_desc": (_i, [_i, _p, _p, _z, _p]),
    "sd3d_pair_conv_ex": (_i, [_p, _i, _i, _p, _i, _p, _p, _l, _p, _p, _i, _i, _p, _p, _i, _i, _i, _l, _p, _p, _p, _i, _p, _i, _i, _p, _z, _p]),
    "sd3d_run_layers": (_i, [_p, _i, _p, _i, _p, _i, _p, _z, _p, _z, _p]),
    "sd3d_run_layers_ev": (_i, [_p, _i, _p, _i, _p, _i, _p, _z, _p, _z, _p, _p]),
    "sd3d_layernorm": (_i, [_p, _i, _p, _i, _p, _p, _f, _l, _i, _p, _i, _i, _p]),
    "sd3d_linear_layernorm": (_i, [_p, _i, _l, _i, _p, _i, _p, _p, _i, _p, _p, _f, _i, _p, _i, _p]),
    "sd3d_sine_pe": (_i, [_p, _i, _l, _p, _p, _p, _i, _p, _i, _p, _i, _p, _i, _p]),
    "sd3d_fourier_pe": (_i, [_p, _i, _l, _p, _p, _i, _i, _p, _i, _p]),
    "sd3d_sine_pe_rows": (_i, [_p, _i, _l, _p, _p, _p, _p, _i, _p, _i, _p, _i, _p, _i, _p]),
    "sd3d_fourier_pe_rows": (_i, [_p, _i, _l, _p, _p, _p, _i, _i, _p, _i, _p]),
    "sd3d_box_refine_rows": (_i, [_p, _p, _p, _i, _p, _p, _p, _i, _l, _p, _p, _p, _p]),
    "sd3d_attention_batch": (_i, [_i, _p, _i, _f, _i, _p, _z, _p]),
    "sd3d_mask_bits_batch": (_i, [_i, _p, _p, _p, _p, _p, _p, _f, _p]),
    "sd3d_dinox_mask_bits_batch": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "sd3d_attention_ws_bytes": (_z, [_i, _i]),
    "sd3d_attention": (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _f, _p, _i, _p, _z, _p]),
    "sd3d_attention_bf16": (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _f, _p, _i, _p, _z, _p]),
    "sd3d_mask_bits": (_i, [_p, _i, _l, _i, _f, _p, _i, _p]),
    "sd3d_near_bits": (_i, [_p, _l, _p, _l, _f, _p, _i, _p]),
    "sd3d_dinox_mask_bits": (_i, [_p, _p, _i, _l, _l, _p, _i, _p]),
    "sd3d_box_refine": (_i, [_p, _p, _p, _i, _p, _p, _i, _l, _p, _p, _p, _p]),
    "sd3d_scale_shift_act": (_i, [_p, _i, _i, _p, _i, _p, _p, _i, _l, _i, _p, _i, _p]),
    "sd3d_scale_shift_act_add": (_i, [_p, _i, _i, _p, _i, _p, _p, _i, _l, _i, _p, _i, _p, _i, _p]),
    "sd3d_class_scores": (_i, [_p, _i, _l, _i, _p, _p, _p]),
    "sd3d_mask_scores": (_i, [_p, _i, _i, _p, _p, _i, _i, _i, _p, _p, _p, _p]),
    "sd3d_take_f32": (_i, [_p, _p, _i, _p, _p]),
    "sd3d_topk_desc_f32": (_i, [_p, _l, _i, _p, _p]),
    "sd3d_select_instances": (_i, [_p, _p, _i, _f, _f, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "sd3d_take_instances": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p]),
    "sd3d_take_pair": (_i, [_p, _p, _p, _i, _p, _p, _p]),
    "sd3d_nms_finish": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p]),
    "sd3d_gather_sigmoid": (_i, [_p, _i, _i, _p, _p, _i, _p, _i, _p, _p]),
    "sd3d_nms_decay": (_i, [_p, _i, _p, _p, _i, _i, _f, _p, _p, _p, _p]),
    "sd3d_row_chain": (_i, [_p, _p]),
    "sd3d_row_chain_program_bytes": (_z, []),
    "sd3d_attention_batch_parts": (_i, [_i, _p, _i, _f, _i, _p, _z, _p, _p, _p]),
    "sd3d_pack_mask_rows": (_i, [_p, _l, _p, _i, _p, _l, _p]),
    "sd3d_unpack_bits_host": (_i, [_p, _l, _l, _l, _p]),
    "sd3d_expand_masks_ws_bytes": (_z, [_i, _i]),
    "sd3d_mask_rowbits": (_i, [_p, _i, _p, _i, _p, _l, _f, _p, _p, _z, _p]),
    "sd3d_expand_rows": (_i, [_p, _i, _i, _p, _i, _p, _p, _i, _l, _p, _f, _p, _p]),
    "sd3d_expand_masks": (_i, [_p, _i, _p, _i, _p, _p, _i, _l, _f, _p, _f, _p, _p, _p, _z, _p]),
    "sd3d_row_argmax": (_i, [_p, _i, _l, _p, _i, _p, _p]),
    "sd3d_gather_i64": (_i, [_p, _p, _l, _i, _p, _p]),
    "sd3d_panoptic": (_i, [_p, _l, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p]),
    "sd3d_mask_overlaps": (_i, [_p, _l, _i, _p, _l, _i, _p, _p]),
    "sd3d_instance_boxes_ws_bytes": (_z, [_i]),
    "sd3d_instance_boxes": (_i, [_p, _i, _l, _p, _l, _i, _i, _p, _p, _p, _z, _p]),
    "sd3d_pack_mask_bits": (_i, [_p, _l, _i, _i, _p, _i, _p, _p]),
    "sd3d_match_costs": (_i, [_p, _i, _i, _p, _i, _i, _i, _p, _p, _p, _p, _i, _p, _i, _p, _i, _p, _i, _p, _p, _p, _p]),
    "sd3d_sparse_match": (_i, [_p, _i, _i, _i, _p, _p]),
    "sd3d_instance_loss_ws_bytes": (_z, [_i]),
    "sd3d_instance_loss": (_i, [_p, _i, _i, _p, _i, _i, _i, _p, _p, _p, _p, _p, _i, _p, _i, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p,
                                _p, _p, _z, _p]),
    "sd3d_pair_out_rows": (_i, [_p, _i, _l, _l, _p, _p]),
    "sd3d_pair_wgrad_ws_bytes": (_z, [_i, _i, _i]),
    "sd3d_pair_wgrad": (_i, [_p, _i, _p, _i, _p, _p, _p, _l, _i, _i, _i, _p, _i, _p, _z, _p]),
    "sd3d_linear_wgrad": (_i, [_p, _i, _p, _i, _l, _i, _i, _p, _p, _i, _p]),
    "sd3d_bn_ws_bytes": (_z, [_l, _i]),
    "sd3d_bn_stats": (_i, [_p, _i, _l, _i, _f, _p, _p, _p, _p, _z, _p]),
    "sd3d_transpose_batch": (_i, [_i, _p, _p]),
    "sd3d_unet_train_forward": (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _p, _p, _z, _p, _z, _p]),
    "sd3d_unet_train_backward": (_i, [_p, _i, _p, _i, _p, _p, _i, _p, _i, _p, _p, _z, _p, _z, _p, _z, _p]),
    "sd3d_bn_stats_running": (_i, [_p, _i, _l, _i, _f, _p, _p, _p, _p, _p, _p, _f, _p, _z, _p]),
    "sd3d_bn_apply": (_i, [_p, _i, _p, _p, _p, _p, _p, _i, _l, _i, _i, _p, _i, _p]),
    "sd3d_bn_backward": (_i, [_p, _i, _p, _i, _p, _i, _p, _p, _p, _l, _i, _i, _p, _i, _p, _i, _p, _p, _p, _z, _p]),
    "sd3d_pool_superpoints_backward": (_i, [_p, _i, _p, _p, _p, _p, _l, _p, _i, _p]),
    "sd3d_augment_points": (_i, [_p, _i, _l, _i, _i, _f, _f, _p, _p, _p, _p]),
    "sd3d_voxel_units": (_i, [_p, _i, _l, _f, _p, _p]),
    "sd3d_box_blur3": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "sd3d_elastic_displace": (_i, [_p, _l, _p, _i, _i, _i, _f, _f, _p]),
    "sd3d_act_backward": (_i, [_p, _i, _p, _i, _i, _l, _i, _i, _p, _i, _p]),
    "sd3d_box_refine_backward": (_i, [_p, _p, _p, _p, _i, _l, _p, _p, _p]),
    "sd3d_col_sums_ws_bytes": (_z, [_l, _i]),
    "sd3d_col_sums": (_i, [_p, _i, _l, _i, _p, _p, _z, _p]),
    "sd3d_layernorm_backward_ws_bytes": (_z, [_l, _i]),
    "sd3d_layernorm_backward": (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _p, _f, _l, _i, _i, _p, _i, _p, _p, _p, _z, _p]),
    "sd3d_sine_pe_mod_backward": (_i, [_p, _i, _p, _i, _l, _p, _p, _p, _i, _p, _i, _p, _p]),
    "sd3d_attention_lse": (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _f, _p, _i, _p, _p, _z, _p]),
    "sd3d_attention_lse_bf16": (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _f, _p, _i, _p, _p, _z, _p]),
    "sd3d_attention_backward_ws_bytes": (_z, [_i, _i]),
    "sd3d_attention_backward": (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _f, _p, _i, _p, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i,
                                     _p, _i, _p, _z, _p]),
    "sd3d_semantic_loss_ws_bytes": (_z, [_i]),
    "sd3d_semantic_loss": (_i, [_p, _i, _i, _i, _i, _p, _i, _f, _p, _i, _p, _p, _z, _p]),
}


class HipExtensionMissing(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes library; raises HipExtensionMissing when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipExtensionMissing(
            f"{LIB_PATH} not found - the HIP extension is not built.  Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C segdino3d_amd/csrc`). "
            "There is no CPU fallback for the product path.")
    # PyDLL = the calls keep the GIL.  Every entry point only enqueues work (a few microseconds), and with
    # several host threads each driving a stream (dist_eval.PipelinedRunner) releasing and re-taking the GIL
    # around ~600 such calls per forward costs more than the calls themselves (lock convoy: 73.6 -> 87.8 scenes/s
    # in a same-box A/B).  SD3D_RELEASE_GIL=1 restores ctypes.CDLL behaviour.
    lib = (C.CDLL if os.environ.get("SD3D_RELEASE_GIL") == "1" else C.PyDLL)(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.sd3d_abi_version() != ABI_VERSION:
        raise HipExtensionMissing(f"ABI version mismatch: library {lib.sd3d_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


_lib_nogil = None


def load_nogil():
    """Second handle of the same library whose calls RELEASE the GIL (ctypes.CDLL): for the few long entry
    points (sd3d_run_layers enqueues ~250 kernels) so that other host threads keep issuing meanwhile."""
    global _lib_nogil
    if _lib_nogil is None:
        load()
        lib = C.CDLL(LIB_PATH)
        for name in ("sd3d_run_layers", "sd3d_run_layers_ev", "sd3d_unpack_bits_host", "sd3d_unet_train_forward", "sd3d_unet_train_backward"):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = SIGNATURES[name]
        _lib_nogil = lib
    return _lib_nogil


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().sd3d_last_error().decode(errors="replace")
        raise RuntimeError(f"segdino3d_hip {what} failed with status {rc}: {msg}")
