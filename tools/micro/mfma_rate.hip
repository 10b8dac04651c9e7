// Micro-benchmark: sustained rate of v_mfma_f32_32x32x2_f32 per SIMD for NACC independent accumulators,
// WPS waves per SIMD, with and without LDS operand reads in the loop.  Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k(float* out, int iters, const float* in) {
    __shared__ __attribute__((aligned(16))) float sm[4 * 32 * 132];
    for (int i = threadIdx.x; i < 4 * 32 * 132; i += 256) sm[i] = in[i & 1023];
    __syncthreads();
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int lane = threadIdx.x & 63;
    f32x4 a[4];
    for (int q = 0; q < 4; ++q) a[q] = *(const f32x4*)(in + lane * 16 + q * 4);
    const float* wb = sm + (lane & 31) * 132 + (lane >> 5) * 16;
    f32x4 w[NACC];
    for (int t = 0; t < NACC; ++t) w[t] = *(const f32x4*)(wb + t * 32 * 132);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (LDS) {
#pragma unroll
                for (int t = 0; t < NACC; ++t) w[t] = *(const f32x4*)(wb + t * 32 * 132 + q * 4 + (it & 3) * 32);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < NACC; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t][e], a[q][e], acc[t], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(int wgs_per_cu, const char* name) {
    const int cus = 256, iters = 2000;
    float *out, *in;
    hipMalloc(&out, cus * 8 * 256 * 4);
    hipMalloc(&in, 4096 * 4);
    float* h = (float*)malloc(4096 * 4);
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h, 4096 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = cus * wgs_per_cu;
    hipLaunchKernelGGL((k<NACC, LDS>), dim3(grid), dim3(256), 0, 0, out, 10, in);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, LDS>), dim3(grid), dim3(256), 0, 0, out, iters, in);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 16.0 * NACC * 4096.0;
    printf("%-28s NACC=%d lds=%d wgs/cu=%d : %.3f ms  %.1f TFLOP/s\n", name, NACC, (int)LDS, wgs_per_cu, ms, flops / ms / 1e9);
    hipFree(out); hipFree(in); free(h);
}

int main() {
    run<4, false>(1, "regs only"); run<4, false>(2, "regs only"); run<4, false>(4, "regs only");
    run<2, false>(1, "regs only"); run<2, false>(2, "regs only");
    run<1, false>(1, "regs only"); run<1, false>(2, "regs only"); run<1, false>(4, "regs only");
    run<4, true>(1, "lds operands"); run<4, true>(2, "lds operands");
    run<3, true>(2, "lds operands"); run<2, true>(2, "lds operands"); run<1, true>(2, "lds operands"); run<1, true>(4, "lds operands");
    return 0;
}
