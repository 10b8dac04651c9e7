// Micro-benchmark: do the fp32 MFMA pipe and the packed-fp32 VALU run CONCURRENTLY on a SIMD?
// Workgroup = 8 waves (2 per SIMD): waves 0-3 run a v_mfma_f32_32x32x2_f32 loop, waves 4-7 a v_pk_fma_f32 loop.
// mode 0: MFMA waves only, 1: VALU waves only, 2: both.   Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void k(float* out, int iters, const float* in, int mode) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float s = 0.f;
    if (wave < 4) {
        if (mode == 1) return;
        f32x16 acc[4];
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        float a = in[lane], b[4] = {in[lane + 64], in[lane + 128], in[lane + 192], in[lane + 256]};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[t], a, acc[t], 0, 0, 0);
        }
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    } else {
        if (mode == 0) return;
        f32x2 acc[32];
        for (int j = 0; j < 32; ++j) acc[j] = f32x2{0.f, 0.f};
        f32x2 a = {in[lane], in[lane + 64]};
        f32x2 w[8];
        for (int j = 0; j < 8; ++j) w[j] = f32x2{in[j * 2 + 512], in[j * 2 + 513]};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u)
#pragma unroll
                for (int j = 0; j < 32; ++j) acc[j] = __builtin_elementwise_fma(a, w[(j + u) & 7], acc[j]);
        }
        for (int j = 0; j < 32; ++j) s += acc[j][0] + acc[j][1];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    const int cus = 256, iters = 4000;
    float *out, *in;
    (void)hipMalloc(&out, cus * 512 * 4);
    (void)hipMalloc(&in, 4096 * 4);
    float* h = (float*)malloc(4096 * 4);
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    (void)hipMemcpy(in, h, 4096 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(cus), dim3(512), 0, 0, out, 10, in, mode);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(cus), dim3(512), 0, 0, out, iters, in, mode);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double f_mfma = (mode != 1) ? (double)cus * 4 * iters * 64.0 * 4096.0 : 0.0;
        const double f_valu = (mode != 0) ? (double)cus * 4 * iters * 1024.0 * 64 * 4.0 : 0.0;      // 64 pk_fma x 64 lanes x 4 flops
        printf("mode %d: %.3f ms  MFMA %.1f TF/s  VALU %.1f TF/s  total %.1f TF/s\n", mode, ms, f_mfma / ms / 1e9, f_valu / ms / 1e9,
               (f_mfma + f_valu) / ms / 1e9);
    }
    return 0;
}
