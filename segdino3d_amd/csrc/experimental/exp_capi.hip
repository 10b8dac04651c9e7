// C ABI of the EXPERIMENTAL library (include/segdino3d_hip_experimental.h): kernels that were built, are parity-green and measured
// slower than the product path (profiles/EXPERIMENTS.md).  Built by `make experimental` into libsegdino3d_hip_experimental.so,
// which links against the product library (error reporting: sd3d_set_error / sd3d_last_error live there).
#include "../common.h"
#include "../../../include/segdino3d_hip_experimental.h"

size_t slab_conv_ws_bytes(int, int, int, int64_t, int64_t);
int launch_slab_conv(const float*, int, int, const float*, int, const int32_t*, int64_t, const float*, int, int, int, int64_t, const float*,
                     const float*, const float*, int, float*, int, int, void*, size_t, hipStream_t);

extern "C" {
size_t sd3d_slab_conv_ws_bytes(int K, int Cin, int Cout, int64_t M, int64_t n_pairs) { return slab_conv_ws_bytes(K, Cin, Cout, M, n_pairs); }
int sd3d_slab_conv(const float* in0, int ld0, int C0, const float* in1, int ld1, const int32_t* nbr, int64_t n_pairs, const float* wt, int K,
                   int Cin, int Cout, int64_t M, const float* scale, const float* shift, const float* res, int ld_res, float* out,
                   int ld_out, int act, void* ws, size_t ws_bytes, void* stream) {
    return launch_slab_conv(in0, ld0, C0, in1, ld1, nbr, n_pairs, wt, K, Cin, Cout, M, scale, shift, res, ld_res, out, ld_out, act, ws,
                            ws_bytes, (hipStream_t)stream);
}
}
