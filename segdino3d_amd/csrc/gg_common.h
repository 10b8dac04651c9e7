// Shared declarations of the gather-GEMM translation units.
#pragma once
#include "common.h"

struct GGParams {
    const float* in0; int ld0; int C0;        // input features, first C0 channels
    const float* in1; int ld1;                // optional second source for channels C0..Cin-1 (skip concat)
    const int32_t* nbr;                       // [K][M] gather indices (-1 = no neighbour) or NULL (identity, K == 1)
    const float* wt;                          // [K][Cout][Cin]
    int K, Cin, Cout;
    int64_t M;                                // output rows
    const float* scale; const float* shift;   // per-column, optional
    const float* res; int ld_res;             // optional residual
    float* out; int ld_out;
    int act;                                  // 0 none, 1 relu, 2 gelu(erf), 3 sigmoid
    int col_groups;                           // ceil(Cout / (32*NT))
    int ksplit;                               // lock-step kernel only: gridDim.z offset slices (partials -> ws)
    float* ws;                                // [ksplit][M][Cout] partial sums when ksplit > 1
};

__device__ __forceinline__ int next_active(uint64_t m0, uint64_t m1, int after) {
    // smallest set bit index > after in the 128-bit mask (m1:m0), or -1
    int s = after + 1;
    if (s < 64) {
        const uint64_t r = m0 >> s;
        if (r) return s + __builtin_ctzll(r);
        s = 64;
    }
    if (s < 128) {
        const uint64_t r = m1 >> (s - 64);
        if (r) return s + __builtin_ctzll(r);
    }
    return -1;
}

