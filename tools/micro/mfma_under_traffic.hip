// Micro-benchmark: does HBM traffic lower the fp32 MFMA rate (power / clock coupling)?
// Stream A: register-resident v_mfma_f32_32x32x2_f32 loop, one 4-wave workgroup per CU.  Stream B: a streaming copy
// (read 1 GiB + write 1 GiB per launch, looped).  Reports the MFMA kernel's TFLOP/s alone and with the copy running,
// and the shader clock seen by s_memtime in both cases.     Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_k(float* out, int iters, const float* in, unsigned long long* ticks) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int lane = threadIdx.x & 63;
    float a = in[lane], b[4] = {in[lane + 64], in[lane + 128], in[lane + 192], in[lane + 256]};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[t], a, acc[t], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void copy_k(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n, int reps) {
    for (int r = 0; r < reps; ++r)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

int main() {
    const int cus = 256, iters = 40000;                       // ~64 MFMAs x 64 cycles x 40000 = 164 M cycles ~ 70 ms
    float *out, *in;
    unsigned long long* ticks;
    (void)hipMalloc(&out, cus * 256 * 4);
    (void)hipMalloc(&in, 4096 * 4);
    (void)hipMalloc(&ticks, 64);
    float* h = (float*)malloc(4096 * 4);
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    (void)hipMemcpy(in, h, 4096 * 4, hipMemcpyHostToDevice);
    const size_t n = (size_t)1 << 26;                         // 64 Mi float4 = 1 GiB
    f32x4 *src, *dst;
    (void)hipMalloc(&src, n * 16);
    (void)hipMalloc(&dst, n * 16);
    (void)hipMemset(src, 1, n * 16);
    hipStream_t sa, sb;
    (void)hipStreamCreate(&sa); (void)hipStreamCreate(&sb);
    hipEvent_t e0, e1, c0, c1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&c0); (void)hipEventCreate(&c1);
    const double flops = (double)cus * 4 * iters * 64.0 * 4096.0;
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(mfma_k, dim3(cus), dim3(256), 0, sa, out, 100, in, ticks);
        (void)hipDeviceSynchronize();
        if (mode == 1) {
            (void)hipEventRecord(c0, sb);
            hipLaunchKernelGGL(copy_k, dim3(cus * 4), dim3(256), 0, sb, src, dst, n, 400);      // far longer than the MFMA kernel
        }
        (void)hipEventRecord(e0, sa);
        hipLaunchKernelGGL(mfma_k, dim3(cus), dim3(256), 0, sa, out, iters, in, ticks);
        (void)hipEventRecord(e1, sa);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long tk = 0;
        (void)hipMemcpyAsync(&tk, ticks, 8, hipMemcpyDeviceToHost, sa);
        (void)hipStreamSynchronize(sa);
        printf("%s: MFMA kernel %.2f ms = %.1f TFLOP/s; s_memtime ticks %llu -> %.3f GHz if a tick is a shader cycle\n",
               mode ? "with HBM copy running" : "alone", ms, flops / ms / 1e9, tk, tk / (ms * 1e6));
        if (mode == 1) {
            (void)hipEventRecord(c1, sb);
            (void)hipEventSynchronize(c1);
            float cms;
            (void)hipEventElapsedTime(&cms, c0, c1);
            printf("copy: %.1f ms for %d GiB moved -> %.2f TB/s\n", cms, 2 * 400, 2.0 * 400 * 1.073741824 / cms);
        }
    }
    return 0;
}
