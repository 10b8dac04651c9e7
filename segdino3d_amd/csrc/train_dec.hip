// Backward pieces of the query decoder: activation masks, bias (column) sums, LayerNorm, the box
// modulation of the sine encoding.  The reference gets all of them from torch autograd over nn.Linear / nn.LayerNorm /
// elementwise ops (instance_seg_3d_decoder.py:640-797, train_engine_3d.py:104).  The decoder's tensors are small
// ([<= 3000, <= 1024] fp32): every kernel here is a single streaming pass, reductions in a fixed order.
#include "common.h"
#include "../../include/segdino3d_hip.h"
#include <math.h>

__device__ __forceinline__ float wsum64(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// g = dy * act'(.) with ref = the forward OUTPUT for relu / sigmoid and the PRE-activation for gelu; columns [C, C_pad)
// of g are zeroed (the GEMMs that consume g want 32-column multiples)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, int ld_dy, const float* __restrict__ ref, int ld_ref, int act,
                                                      int64_t M, int C, int C_pad, float* __restrict__ g, int ld_g) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= M * C_pad) return;
    const int64_t r = t / C_pad;
    const int c = (int)(t - r * C_pad);
    float v = 0.f;
    if (c < C) {
        v = dy[r * ld_dy + c];
        if (act == 1) v = ref[r * ld_ref + c] > 0.f ? v : 0.f;
        else if (act == 2) {                                   // d/dz [0.5 z (1 + erf(z / sqrt 2))]
            const float z = ref[r * ld_ref + c];
            v *= 0.5f * (1.f + erff(z * 0.70710678118654752440f)) + z * 0.39894228040143267794f * expf(-0.5f * z * z);
        } else if (act == 3) { const float y = ref[r * ld_ref + c]; v *= y * (1.f - y); }
    }
    g[r * ld_g + c] = v;
}

// column sums: one workgroup per (64-column group, 64-row chunk) -> partial[chunk][C]; then a fixed-order final pass
#define CSUM_ROWS 64
__global__ __launch_bounds__(256) void col_sum_partial_kernel(const float* __restrict__ x, int ld, int64_t M, int C, float* __restrict__ partial) {
    __shared__ float sm[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * CSUM_ROWS, r1 = r0 + CSUM_ROWS < M ? r0 + CSUM_ROWS : M;
    float a = 0.f;
    if (c < C) for (int64_t r = r0 + rl; r < r1; r += 4) a += x[r * ld + c];
    sm[rl][threadIdx.x & 63] = a;
    __syncthreads();
    if (rl == 0 && c < C) partial[(int64_t)blockIdx.y * C + c] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void col_sum_final_kernel(const float* __restrict__ partial, int nchunk, int C, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double a = 0.0;
    for (int i = 0; i < nchunk; ++i) a += (double)partial[(int64_t)i * C + c];
    out[c] = (float)a;
}

// LayerNorm backward, one wave per row (D <= 1024): xin = x + res, xhat = (xin - mean) rstd, g = dy (masked by y > 0 for the
// fused ReLU);  dxin = rstd (g w - mean(g w) - xhat mean(g w xhat));  gw_out = g, gxh_out = g xhat (column-summed afterwards)
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, int ld_dy, const float* __restrict__ y, int ld_y,
                                                            const float* __restrict__ x, int ld_x, const float* __restrict__ res, int ld_res,
                                                            const float* __restrict__ w, float eps, int64_t M, int D, int act,
                                                            float* __restrict__ dxin, int ld_dx, float* __restrict__ g_out, float* __restrict__ gxh_out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    f32x4 v[4], g[4];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = (it * 64 + lane) * 4;
        v[it] = f32x4{0.f, 0.f, 0.f, 0.f}; g[it] = v[it];
        if (c < D) {
            v[it] = *(const f32x4*)(x + r * ld_x + c);
            if (res) v[it] += *(const f32x4*)(res + r * ld_res + c);
            g[it] = *(const f32x4*)(dy + r * ld_dy + c);
            if (act == 1) {
                const f32x4 yy = *(const f32x4*)(y + r * ld_y + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[it][e] = yy[e] > 0.f ? g[it][e] : 0.f;
            }
            s += v[it][0] + v[it][1] + v[it][2] + v[it][3];
        }
    }
    const float mean = wsum64(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = (it * 64 + lane) * 4;
        if (c < D) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float dlt = v[it][e] - mean; q += dlt * dlt; }
        }
    }
    const float rstd = 1.0f / sqrtf(wsum64(q) / (float)D + eps);
    float a = 0.f, b = 0.f;                                    // sum g w, sum g w xhat
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = (it * 64 + lane) * 4;
        if (c < D) {
            const f32x4 ww = *(const f32x4*)(w + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (v[it][e] - mean) * rstd;
                a += g[it][e] * ww[e]; b += g[it][e] * ww[e] * xh;
            }
        }
    }
    a = wsum64(a) / (float)D; b = wsum64(b) / (float)D;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = (it * 64 + lane) * 4;
        if (c < D) {
            const f32x4 ww = *(const f32x4*)(w + c);
            f32x4 dx, gx;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (v[it][e] - mean) * rstd;
                dx[e] = rstd * (g[it][e] * ww[e] - a - xh * b);
                gx[e] = g[it][e] * xh;
            }
            *(f32x4*)(dxin + r * ld_dx + c) = dx;
            *(f32x4*)(g_out + r * D + c) = g[it];
            *(f32x4*)(gxh_out + r * D + c) = gx;
        }
    }
}

// d(mod_num)[r][a] = sum over the channels c of axis a of  d_out[r][c] * pe[r][c] / mod_den[r or 0][a], where
// pe = out / (mod_num / mod_den) is recomputed as in sine_pe_kernel (dense.hip); positions and denominators are detached
// in the reference (instance_seg_3d_decoder.py:740, 753), so this is the only gradient of the modulated encoding.
__global__ __launch_bounds__(256) void sine_pe_mod_bwd_kernel(const float* __restrict__ d_out, int ld_do, const float* __restrict__ xyz, int ld_xyz,
                                                              int64_t n, const float* __restrict__ rng, const float* __restrict__ dim_t,
                                                              const int8_t* __restrict__ axis, int d_pos, const float* __restrict__ mod_den,
                                                              int ld_den, float* __restrict__ d_num) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int c = lane; c < d_pos; c += 64) {
        const int a = axis[c];
        const float lo = rng[a], hi = rng[3 + a];
        float p = ((xyz[r * ld_xyz + a] - lo) * 1.0f) / (hi - lo) + 0.0f;
        p = p * 6.283185307179586f;
        p = p / dim_t[c];
        const float pe = (c & 1) ? cosf(p) : sinf(p);
        const float v = d_out[r * ld_do + c] * pe;
        acc[0] += a == 0 ? v : 0.f; acc[1] += a == 1 ? v : 0.f; acc[2] += a == 2 ? v : 0.f;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float t = wsum64(acc[a]);
        if (lane == 0) d_num[r * 3 + a] = t / mod_den[(int64_t)r * ld_den + a];
    }
}

// backward of box_refine_kernel (dense.hip): d(d_center) = d(center); d(d_size) = d(size_out) * (hi - lo) * s (1 - s) when the
// sizes are normalised, d(size_out) otherwise.  Reference points and previous sizes are detached in the reference (:740, :753).
__global__ __launch_bounds__(256) void box_refine_bwd_kernel(const float* __restrict__ d_center, const float* __restrict__ d_size_out,
                                                             const float* __restrict__ size, const float* __restrict__ rng, int normalize, int64_t Q,
                                                             float* __restrict__ d_dc, float* __restrict__ d_ds) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= Q * 3) return;
    const int a = (int)(t % 3);
    if (d_dc) d_dc[t] = d_center ? d_center[t] : 0.f;
    if (d_ds) {
        float g = d_size_out ? d_size_out[t] : 0.f;
        if (normalize) { const float s = size[t]; g *= (rng[3 + a] - rng[a]) * s * (1.f - s); }
        d_ds[t] = g;
    }
}

// W [rows, cols] -> W^T [cols, ld_dst] (columns rows..ld_dst-1 zero) for MANY matrices in one launch: the input gradient of a
// Linear is a product with W^T, and autograd used to transpose every weight with its own copy kernel (145 launches per step).
// A workgroup = one 32 x 32 tile of one destination, found by its tile number in the jobs' prefix sums.
#define TB_MAX 112
struct TBatch { int n; int32_t tile0[TB_MAX + 1]; sd3d_transpose_job job[TB_MAX]; };
static __global__ __launch_bounds__(256) void transpose_batch_kernel(const TBatch b) {
    __shared__ float tile[32][33];
    int lo = 0, hi = b.n - 1;
    const int wg = blockIdx.x;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (b.tile0[mid] <= wg) lo = mid; else hi = mid - 1; }
    const sd3d_transpose_job& J = b.job[lo];
    const int tx = (J.ld_dst + 31) / 32, ty = (J.cols + 31) / 32;      // tiles along the destination's columns (= source rows, padded) / rows
    int t = wg - b.tile0[lo];
    const int m = t / (tx * ty);                               // matrix of the job (J.batch matrices back to back)
    t -= m * (tx * ty);
    const float* src = J.src + (int64_t)m * J.rows * J.cols;
    float* dst = J.dst + (int64_t)m * J.cols * J.ld_dst;
    const int r0 = (t % tx) * 32, c0 = (t / tx) * 32;          // source row / column of the tile's corner
    const int x = threadIdx.x & 31, y = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int r = r0 + y + i, c = c0 + x;
        tile[y + i][x] = (r < J.rows && c < J.cols) ? src[(int64_t)r * J.cols + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int c = c0 + y + i, r = r0 + x;                  // destination row c, destination column r
        if (c < J.cols && r < J.ld_dst) dst[(int64_t)c * J.ld_dst + r] = tile[x][y + i];
    }
}

#define ST ((hipStream_t)stream)
extern "C" {

int sd3d_transpose_batch(int n, const sd3d_transpose_job* jobs, void* stream) {
    for (int i0 = 0; i0 < n; i0 += TB_MAX) {
        TBatch b;
        b.n = n - i0 < TB_MAX ? n - i0 : TB_MAX;
        int tiles = 0;
        for (int i = 0; i < b.n; ++i) {
            const sd3d_transpose_job& J = jobs[i0 + i];
            if (!J.src || !J.dst || J.rows <= 0 || J.cols <= 0 || J.ld_dst < J.rows)
                return sd3d_set_error(SD3D_ERR_ARG, "transpose_batch: bad job");
            b.job[i] = J;
            b.tile0[i] = tiles;
            tiles += (J.batch > 1 ? J.batch : 1) * ((J.ld_dst + 31) / 32) * ((J.cols + 31) / 32);
        }
        b.tile0[b.n] = tiles;
        transpose_batch_kernel<<<tiles, 256, 0, ST>>>(b);
        SD3D_CHECK_LAUNCH();
    }
    return SD3D_OK;
}

int sd3d_box_refine_backward(const float* d_center, const float* d_size_out, const float* size, const float* range, int normalize, int64_t Q,
                             float* d_dc, float* d_ds, void* stream) {
    if (Q <= 0) return SD3D_OK;
    if (d_ds && normalize && !size) return sd3d_set_error(SD3D_ERR_ARG, "box_refine_backward: needs the forward sizes");
    box_refine_bwd_kernel<<<(unsigned)cdiv(Q * 3, 256), 256, 0, ST>>>(d_center, d_size_out, size, range, normalize, Q, d_dc, d_ds);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

int sd3d_act_backward(const float* dy, int ld_dy, const float* ref, int ld_ref, int act, int64_t M, int C, int C_pad, float* g, int ld_g,
                      void* stream) {
    if (M <= 0 || C <= 0) return SD3D_OK;
    if (act < 0 || act > 3 || C_pad < C || ld_g < C_pad || (act != 0 && !ref)) return sd3d_set_error(SD3D_ERR_ARG, "act_backward: bad arguments");
    act_bwd_kernel<<<(unsigned)cdiv(M * C_pad, 256), 256, 0, ST>>>(dy, ld_dy, ref, ld_ref, act, M, C, C_pad, g, ld_g);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// few rows (the decoder's <= 8192 query / superpoint rows): one launch, one workgroup per 64-column group with 16 row lanes whose
// fp32 partial sums meet in LDS in a fixed order (double) - the two-launch path above costs 15 us of launch latency for 2.5 MB of data
#define CSUM_SMALL_MAX 8192
static __global__ __launch_bounds__(1024) void col_sum_small_kernel(const float* __restrict__ x, int ld, int64_t M, int C, float* __restrict__ out) {
    __shared__ float sm[16][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float a = 0.f;
    if (c < C) {
        int64_t r = rl;
        for (; r + 48 < M; r += 64) a += (x[r * ld + c] + x[(r + 16) * ld + c]) + (x[(r + 32) * ld + c] + x[(r + 48) * ld + c]);
        for (; r < M; r += 16) a += x[r * ld + c];
    }
    sm[rl][cl] = a;
    __syncthreads();
    if (rl == 0 && c < C) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += (double)sm[i][cl];
        out[c] = (float)t;
    }
}

size_t sd3d_col_sums_ws_bytes(int64_t M, int C) { return align_up((size_t)cdiv(M, CSUM_ROWS) * C * sizeof(float), 256); }

int sd3d_col_sums(const float* x, int ld, int64_t M, int C, float* out, void* ws, size_t ws_bytes, void* stream) {
    if (M <= 0 || C <= 0) return sd3d_set_error(SD3D_ERR_ARG, "col_sums: empty input");
    if (ws_bytes < sd3d_col_sums_ws_bytes(M, C)) return sd3d_set_error(SD3D_ERR_WS, "col_sums: workspace too small");
    if (M <= CSUM_SMALL_MAX) {
        col_sum_small_kernel<<<(unsigned)cdiv(C, 64), 1024, 0, ST>>>(x, ld, M, C, out);
        SD3D_CHECK_LAUNCH();
        return SD3D_OK;
    }
    const int nchunk = (int)cdiv(M, CSUM_ROWS);
    col_sum_partial_kernel<<<dim3((unsigned)cdiv(C, 64), nchunk), 256, 0, ST>>>(x, ld, M, C, (float*)ws);
    col_sum_final_kernel<<<(unsigned)cdiv(C, 256), 256, 0, ST>>>((const float*)ws, nchunk, C, out);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

size_t sd3d_layernorm_backward_ws_bytes(int64_t M, int D) { return (size_t)2 * M * D * sizeof(float) + sd3d_col_sums_ws_bytes(M, D); }

int sd3d_layernorm_backward(const float* dy, int ld_dy, const float* y, int ld_y, const float* x, int ld_x, const float* res, int ld_res,
                            const float* w, float eps, int64_t M, int D, int act, float* dxin, int ld_dx, float* dw, float* db, void* ws,
                            size_t ws_bytes, void* stream) {
    if (M <= 0 || D <= 0 || (D & 3) || D > 1024 || (ld_dy & 3) || (ld_x & 3) || (ld_dx & 3) || (res && (ld_res & 3)) || act < 0 || act > 1 ||
        (act == 1 && (!y || (ld_y & 3))))
        return sd3d_set_error(SD3D_ERR_ARG, "layernorm_backward: D must be a multiple of 4, <= 1024; strides multiples of 4");
    if (ws_bytes < sd3d_layernorm_backward_ws_bytes(M, D)) return sd3d_set_error(SD3D_ERR_WS, "layernorm_backward: workspace too small");
    float* g = (float*)ws;
    float* gxh = g + (size_t)M * D;
    void* cws = gxh + (size_t)M * D;
    const size_t cws_bytes = sd3d_col_sums_ws_bytes(M, D);
    layernorm_bwd_kernel<<<(unsigned)cdiv(M, 4), 256, 0, ST>>>(dy, ld_dy, y, ld_y, x, ld_x, res, ld_res, w, eps, M, D, act, dxin, ld_dx, g, gxh);
    SD3D_CHECK_LAUNCH();
    int rc = sd3d_col_sums(g, D, M, D, db, cws, cws_bytes, stream);
    if (rc != SD3D_OK) return rc;
    return sd3d_col_sums(gxh, D, M, D, dw, cws, cws_bytes, stream);
}

int sd3d_sine_pe_mod_backward(const float* d_out, int ld_do, const float* xyz, int ld_xyz, int64_t n, const float* range, const float* dim_t,
                              const int8_t* axis, int d_pos, const float* mod_den, int ld_den, float* d_num, void* stream) {
    if (n <= 0) return SD3D_OK;
    if (!mod_den) return sd3d_set_error(SD3D_ERR_ARG, "sine_pe_mod_backward: needs the modulation denominator");
    sine_pe_mod_bwd_kernel<<<(unsigned)cdiv(n, 4), 256, 0, ST>>>(d_out, ld_do, xyz, ld_xyz, n, range, dim_t, axis, d_pos, mod_den, ld_den, d_num);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

}  // extern "C"
