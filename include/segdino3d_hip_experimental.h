/* C ABI of libsegdino3d_hip_experimental.so - kernels OUTSIDE the product library (`make -C segdino3d_amd/csrc experimental`).
 * Everything here is parity-tested (tests marked `experimental`) and measured slower than the product path on the benchmark
 * scene; profiles/EXPERIMENTS.md is the record.  Same conventions as include/segdino3d_hip.h: device pointers + sizes +
 * hipStream_t (void*), int status (0 = ok; text from the PRODUCT library's sd3d_last_error()), caller-owned workspace. */
#ifndef SEGDINO3D_HIP_EXPERIMENTAL_H
#define SEGDINO3D_HIP_EXPERIMENTAL_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Output-stationary sparse convolution (csrc/experimental/slab_conv.hip) - the same contract again (MinkowskiConvolution /
 * SubMConv3d / their transposes + folded BN + residual + activation; minkunet.py:135-192, spconvunet.py:21-99), straight
 * from the neighbour table nbr [K, M]: a workgroup owns a slab of consecutive output rows, keeps their fp32 sums in LDS
 * while it walks the offsets in ascending order, and writes every output row once - no partial products in HBM, no
 * second pass, no pair lists.  n_pairs (number of entries >= 0, from sd3d_kernel_map) only guides the launch geometry.
 * Cin % 32 == 0, Cout % 16 == 0, K <= 128; sd3d_slab_conv_ws_bytes returns 0 for shapes it does not handle (the caller
 * keeps sd3d_pair_conv for those).  ws: per-workgroup rulebook scratch (+ partial slabs when the offsets are split). */
size_t sd3d_slab_conv_ws_bytes(int K, int Cin, int Cout, int64_t M, int64_t n_pairs);
int sd3d_slab_conv(const float* in0, int ld0, int C0, const float* in1, int ld1, const int32_t* nbr, int64_t n_pairs,
                   const float* wt, int K, int Cin, int Cout, int64_t M, const float* scale, const float* shift,
                   const float* res, int ld_res, float* out, int ld_out, int act, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
