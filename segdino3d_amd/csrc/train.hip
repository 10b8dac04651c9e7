// Training-mode pieces of the sparse U-Net: batch-statistics BatchNorm forward / backward with the
// residual add and ReLU folded in (ME.MinkowskiBatchNorm = nn.BatchNorm1d over the voxel rows, minkunet.py:302-304;
// BasicBlock :234-250), and the backward of the fused devoxelise + superpoint mean (minkunet.py:668-676).
// All of it is streaming work over [M, C] row-major fp32 tensors: HBM-bound, float4 accesses, one pass per tensor.
// Column reductions are two-level with a fixed order (per-block partial sums, then one block adds the partials in
// double precision), so statistics and gradients are reproducible bit for bit.
#include "common.h"
#include "../../include/segdino3d_hip.h"

#define CS_ROWS 256                  // rows per block of the column reductions (110 k voxel rows -> 430 workgroups)

// Column sums of two per-element quantities over a chunk of rows.  MODE 0: (x - pivot), (x - pivot)^2 with
// pivot = first row (keeps the variance from cancelling); MODE 1: g, g * xhat with g = dy masked by the ReLU.
struct ColParams {
    const float* a; int ld_a;          // MODE 0: x;  MODE 1: dy
    const float* y; int ld_y;          // MODE 1: forward output (ReLU mask) or null
    const float* x; int ld_x;          // MODE 1: BatchNorm input
    const float* mean; const float* rstd;
    int64_t M; int C; int act;
    float* partial;                    // [nblk][2][C]
};

template <int MODE>
__global__ __launch_bounds__(256) void col_partial_kernel(const ColParams p) {
    extern __shared__ float sm[];                              // [rpi][2][C]
    const int c4n = p.C >> 2, rpi = 256 / c4n;
    const int tid = threadIdx.x;
    const int col = (tid % c4n) * 4, rl = tid / c4n;
    const int64_t r0 = (int64_t)blockIdx.x * CS_ROWS;
    const int64_t r1 = r0 + CS_ROWS < p.M ? r0 + CS_ROWS : p.M;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (rl < rpi) {
        f32x4 piv = {0.f, 0.f, 0.f, 0.f}, mu = piv, rs = piv;
        if (MODE == 0) piv = *(const f32x4*)(p.a + col);
        else { mu = *(const f32x4*)(p.mean + col); rs = *(const f32x4*)(p.rstd + col); }
        for (int64_t r = r0 + rl; r < r1; r += rpi) {
            const f32x4 v = *(const f32x4*)(p.a + r * p.ld_a + col);
            if (MODE == 0) {
                const f32x4 d = v - piv;
                s1 += d; s2 += d * d;
            } else {
                f32x4 g = v;
                if (p.act == 1) {
                    const f32x4 yy = *(const f32x4*)(p.y + r * p.ld_y + col);
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] = yy[e] > 0.f ? g[e] : 0.f;
                }
                const f32x4 xh = (*(const f32x4*)(p.x + r * p.ld_x + col) - mu) * rs;
                s1 += g; s2 += g * xh;
            }
        }
        *(f32x4*)(sm + (rl * 2 + 0) * p.C + col) = s1;
        *(f32x4*)(sm + (rl * 2 + 1) * p.C + col) = s2;
    }
    __syncthreads();
    for (int i = tid; i < 2 * p.C; i += 256) {                 // fixed order over the row lanes
        float a = 0.f;
        for (int r = 0; r < rpi; ++r) a += sm[r * 2 * p.C + i];
        p.partial[(int64_t)blockIdx.x * 2 * p.C + i] = a;
    }
}

// MODE 0 -> mean, biased variance, rstd;  MODE 1 -> dbeta = sum g, dgamma = sum g xhat.
// One workgroup per 16 columns; the per-block partials of a column are dealt to 16 threads (interleaved), each adds its share
// in double, and the 16 shares are added in a fixed order - hundreds of partials per column without a serial chain of loads.
template <int MODE>
__global__ __launch_bounds__(256) void col_final_kernel(const float* __restrict__ partial, int nblk, int C, int64_t M, float eps,
                                                        const float* __restrict__ x_first, float* __restrict__ o0, float* __restrict__ o1,
                                                        float* __restrict__ o2, float* __restrict__ run_mean = nullptr,
                                                        float* __restrict__ run_var = nullptr, long long* __restrict__ n_tracked = nullptr,
                                                        float momentum = 0.f) {
    __shared__ double sa[16][16], sb[16][16];
    const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int i = part; i < nblk; i += 16) { a += (double)partial[(int64_t)i * 2 * C + c]; b += (double)partial[(int64_t)i * 2 * C + C + c]; }
    sa[part][cl] = a; sb[part][cl] = b;
    __syncthreads();
    if (part != 0 || c >= C) return;
    a = 0.0; b = 0.0;
    for (int i = 0; i < 16; ++i) { a += sa[i][cl]; b += sb[i][cl]; }
    if (MODE == 0) {
        const double m1 = a / (double)M, var = b / (double)M - m1 * m1;
        o0[c] = (float)((double)x_first[c] + m1);
        o1[c] = (float)(var > 0.0 ? var : 0.0);
        o2[c] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps));
        // nn.BatchNorm1d's running statistics (momentum form, unbiased variance), in the launch that makes the batch statistics:
        // six elementwise launches per BatchNorm otherwise
        if (run_mean) run_mean[c] = __builtin_fmaf(momentum, o0[c], run_mean[c] * (1.f - momentum));
        if (run_var) run_var[c] = __builtin_fmaf(momentum, o1[c] * ((float)M / (float)(M > 1 ? M - 1 : 1)), run_var[c] * (1.f - momentum));
        if (n_tracked && c == 0) *n_tracked += 1;
    } else {
        o0[c] = (float)a;                                      // dbeta
        o1[c] = (float)b;                                      // dgamma
    }
}

// y = act((x - mean) * rstd * gamma + beta + res)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, int ld_x, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ res, int ld_res,
                                                       int64_t M, int C, int act, float* __restrict__ y, int ld_y) {
    const int cv = C >> 2;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= M * cv) return;
    const int64_t r = t / cv;
    const int c = (int)(t - r * cv) * 4;
    const f32x4 v = *(const f32x4*)(x + r * ld_x + c);
    const f32x4 mu = *(const f32x4*)(mean + c), rs = *(const f32x4*)(rstd + c), ga = *(const f32x4*)(gamma + c), be = *(const f32x4*)(beta + c);
    f32x4 o = (v - mu) * rs * ga + be;
    if (res) o += *(const f32x4*)(res + r * ld_res + c);
    if (act == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
    }
    *(f32x4*)(y + r * ld_y + c) = o;
}

// dx = gamma rstd (g - dbeta / M - xhat dgamma / M);  dres = g
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, int ld_dy, const float* __restrict__ y, int ld_y,
                                                           const float* __restrict__ x, int ld_x, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ dbeta, const float* __restrict__ dgamma, int64_t M, int C,
                                                           int act, float* __restrict__ dx, int ld_dx, float* __restrict__ dres, int ld_dres) {
    const int cv = C >> 2;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= M * cv) return;
    const int64_t r = t / cv;
    const int c = (int)(t - r * cv) * 4;
    f32x4 g = *(const f32x4*)(dy + r * ld_dy + c);
    if (act == 1) {
        const f32x4 yy = *(const f32x4*)(y + r * ld_y + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = yy[e] > 0.f ? g[e] : 0.f;
    }
    const f32x4 mu = *(const f32x4*)(mean + c), rs = *(const f32x4*)(rstd + c), ga = *(const f32x4*)(gamma + c);
    const f32x4 db = *(const f32x4*)(dbeta + c), dg = *(const f32x4*)(dgamma + c);
    const f32x4 xh = (*(const f32x4*)(x + r * ld_x + c) - mu) * rs;
    const float inv_m = 1.f / (float)M;
    *(f32x4*)(dx + r * ld_dx + c) = ga * rs * (g - db * inv_m - xh * dg * inv_m);
    if (dres) *(f32x4*)(dres + r * ld_dres + c) = g;
}

// backward of pool_superpoints (voxel.hip): dfeat[v] = sum over the points p of voxel v of dout[sp[p]] / max(|sp[p]|, 1).
// Half a wave per voxel, lanes over the float4 column pieces, the voxel's points in their sorted order.
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ dout, int C, const int64_t* __restrict__ sp,
                                                       const uint32_t* __restrict__ sidx, const int32_t* __restrict__ seg_start,
                                                       const int32_t* __restrict__ sp_start, int64_t V, float* __restrict__ dfeat, int ld) {
    const int64_t v = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int li = threadIdx.x & 31;
    if (v >= V) return;
    const int nvec = C >> 2;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int j = seg_start[v]; j < seg_start[v + 1]; ++j) {
        const int64_t s = sp[sidx[j]];
        const int cnt = sp_start[s + 1] - sp_start[s];
        const float inv = 1.f / (float)(cnt > 0 ? cnt : 1);
        if (li < nvec) acc += *(const f32x4*)(dout + s * C + li * 4) * inv;
    }
    if (li < nvec) *(f32x4*)(dfeat + v * ld + li * 4) = acc;
}

static int col_blocks(int64_t M) { return (int)cdiv(M, CS_ROWS); }

#define ST ((hipStream_t)stream)
extern "C" {

size_t sd3d_bn_ws_bytes(int64_t M, int C) { return align_up((size_t)col_blocks(M) * 2 * C * sizeof(float), 256); }

int sd3d_bn_stats_running(const float* x, int ld, int64_t M, int C, float eps, float* mean, float* var, float* rstd, float* running_mean,
                          float* running_var, int64_t* num_batches_tracked, float momentum, void* ws, size_t ws_bytes, void* stream) {
    if (M <= 0 || C <= 0 || (C & 3) || C > 1024 || (ld & 3)) return sd3d_set_error(SD3D_ERR_ARG, "bn_stats: C must be a multiple of 4, <= 1024");
    if (ws_bytes < sd3d_bn_ws_bytes(M, C)) return sd3d_set_error(SD3D_ERR_WS, "bn_stats: workspace too small");
    ColParams p{}; p.a = x; p.ld_a = ld; p.M = M; p.C = C; p.partial = (float*)ws;
    const int nblk = col_blocks(M), rpi = 256 / (C >> 2);
    col_partial_kernel<0><<<nblk, 256, (size_t)rpi * 2 * C * sizeof(float), ST>>>(p);
    col_final_kernel<0><<<(unsigned)cdiv(C, 16), 256, 0, ST>>>(p.partial, nblk, C, M, eps, x, mean, var, rstd, running_mean, running_var,
                                                               (long long*)num_batches_tracked, momentum);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

int sd3d_bn_stats(const float* x, int ld, int64_t M, int C, float eps, float* mean, float* var, float* rstd, void* ws, size_t ws_bytes,
                  void* stream) {
    return sd3d_bn_stats_running(x, ld, M, C, eps, mean, var, rstd, nullptr, nullptr, nullptr, 0.f, ws, ws_bytes, stream);
}

int sd3d_bn_apply(const float* x, int ld_x, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* res,
                  int ld_res, int64_t M, int C, int act, float* y, int ld_y, void* stream) {
    if (M <= 0) return SD3D_OK;
    if ((C & 3) || (ld_x & 3) || (ld_y & 3) || (res && (ld_res & 3)) || act < 0 || act > 1)
        return sd3d_set_error(SD3D_ERR_ARG, "bn_apply: channels / strides must be multiples of 4, act none or relu");
    bn_apply_kernel<<<(unsigned)cdiv(M * (C >> 2), 256), 256, 0, ST>>>(x, ld_x, mean, rstd, gamma, beta, res, ld_res, M, C, act, y, ld_y);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

int sd3d_bn_backward(const float* dy, int ld_dy, const float* y, int ld_y, const float* x, int ld_x, const float* mean, const float* rstd,
                     const float* gamma, int64_t M, int C, int act, float* dx, int ld_dx, float* dres, int ld_dres, float* dgamma,
                     float* dbeta, void* ws, size_t ws_bytes, void* stream) {
    if (M <= 0 || C <= 0 || (C & 3) || C > 1024 || (ld_dy & 3) || (ld_x & 3) || (ld_dx & 3) || (dres && (ld_dres & 3)) || act < 0 || act > 1 ||
        (act == 1 && (!y || (ld_y & 3))))
        return sd3d_set_error(SD3D_ERR_ARG, "bn_backward: bad shape");
    if (ws_bytes < sd3d_bn_ws_bytes(M, C)) return sd3d_set_error(SD3D_ERR_WS, "bn_backward: workspace too small");
    ColParams p{}; p.a = dy; p.ld_a = ld_dy; p.y = y; p.ld_y = ld_y; p.x = x; p.ld_x = ld_x; p.mean = mean; p.rstd = rstd;
    p.M = M; p.C = C; p.act = act; p.partial = (float*)ws;
    const int nblk = col_blocks(M), rpi = 256 / (C >> 2);
    col_partial_kernel<1><<<nblk, 256, (size_t)rpi * 2 * C * sizeof(float), ST>>>(p);
    col_final_kernel<1><<<(unsigned)cdiv(C, 16), 256, 0, ST>>>(p.partial, nblk, C, M, 0.f, nullptr, dbeta, dgamma, nullptr);
    bn_bwd_apply_kernel<<<(unsigned)cdiv(M * (C >> 2), 256), 256, 0, ST>>>(dy, ld_dy, y, ld_y, x, ld_x, mean, rstd, gamma, dbeta, dgamma, M, C, act,
                                                                          dx, ld_dx, dres, ld_dres);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

int sd3d_pool_superpoints_backward(const float* dout, int C, const int64_t* superpoints, const uint32_t* sidx, const int32_t* seg_start,
                                   const int32_t* sp_start, int64_t V, float* dfeat, int ld, void* stream) {
    if (V <= 0) return SD3D_OK;
    if ((C & 3) || C > 128 || (ld & 3)) return sd3d_set_error(SD3D_ERR_ARG, "pool_superpoints_backward: C must be a multiple of 4 and <= 128");
    pool_bwd_kernel<<<(unsigned)cdiv(V, 8), 256, 0, ST>>>(dout, C, superpoints, sidx, seg_start, sp_start, V, dfeat, ld);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

}  // extern "C"
