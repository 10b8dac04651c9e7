// Decoder kernels (reference: segdino3d/models/decoder/instance_seg_3d_decoder.py:606-799 and
// segdino3d/models/module/{attention.py:186-395, utils.py:53-105}).
//   layernorm        : y = act(LN(x + res))                       one wave per row
//   sine_pe          : box-modulated sine positional encoding      elementwise, fp32 sin/cos
//   attention        : fused masked multi-head attention, fp32 MFMA, online softmax, bit-packed mask;
//                      never materialises the [H, Q, S] score tensor (K16, K17)
//   mask_bits        : sigmoid(logit) < thr -> bit-packed attention mask + dead-row reset (K18)
//   near_bits / dinox_mask_bits : boolean (mask . distance) product as AND/any over bit words (K19)
//   box_refine       : iterative centre / size refinement (:735-759)
#include "common.h"
#include "../../include/segdino3d_hip.h"

__device__ static inline float wsum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dim D (multiple of 4, <= 1024): two-pass mean / variance in fp32.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ld_x, const float* __restrict__ res,
                                                        int ld_res, const float* __restrict__ w, const float* __restrict__ b,
                                                        float eps, int64_t M, int D, float* __restrict__ out, int ld_out, int act) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    f32x4 v[4];                       // up to 1024 columns: 4 x (64 lanes x 4)
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = (it * 64 + lane) * 4;
        v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < D) {
            v[it] = *(const f32x4*)(x + r * ld_x + c);
            if (res) v[it] += *(const f32x4*)(res + r * ld_res + c);
            s += v[it][0] + v[it][1] + v[it][2] + v[it][3];
        }
    }
    const float mean = wsum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = (it * 64 + lane) * 4;
        if (c < D) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float dlt = v[it][e] - mean; q += dlt * dlt; }
        }
    }
    const float rstd = 1.0f / sqrtf(wsum(q) / (float)D + eps);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = (it * 64 + lane) * 4;
        if (c < D) {
            const f32x4 ww = *(const f32x4*)(w + c), bb = *(const f32x4*)(b + c);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = (v[it][e] - mean) * rstd * ww[e] + bb[e];
                if (act == 1) y[e] = fmaxf(y[e], 0.f);
            }
            *(f32x4*)(out + r * ld_out + c) = y;
        }
    }
}

int launch_layernorm(const float* x, int ld_x, const float* res, int ld_res, const float* w, const float* b, float eps,
                     int64_t M, int D, float* out, int ld_out, int act, hipStream_t st) {
    if (M <= 0) return SD3D_OK;
    if ((D & 3) || D > 1024 || (ld_x & 3) || (ld_out & 3) || (res && (ld_res & 3)))
        return sd3d_set_error(SD3D_ERR_ARG, "layernorm: D must be a multiple of 4 and <= 1024, strides multiples of 4");
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)cdiv(M, 4)), dim3(256), 0, st, x, ld_x, res, ld_res, w, b, eps, M, D, out,
                       ld_out, act);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// ---------------------------------------------------------------------------------------------
// Projection + residual + LayerNorm in one launch for the decoder's few-hundred-row tensors (200 queries):
//     out = act(LayerNorm(x W^T + bias + res) * g + b),   W [256, Cin]
// (attention out-projection / second FFN Linear followed by the residual add and the norm: decoder :690-691, :708-709, :82-84,
// :187-188).  One workgroup owns 16 rows x all 256 columns - the whole LayerNorm row - as four waves of 64 columns on
// v_mfma_f32_16x16x4_f32 (A = 16 rows x 4 channels, B = 4 channels x 16 columns); the row statistics meet in LDS (two-pass mean /
// variance like layernorm_kernel).  Both operands are read as dwordx4 along the channel axis: lane (i, kq) holds channels
// 16 g + 4 kq .. + 3 of its row / column, and MFMA e of the group contracts {16 g + 4 kq + e}.
// ---------------------------------------------------------------------------------------------
// NG > 0: Cin = 16 NG known at compile time - the contraction is straight-line code in quarters of four 16-channel groups, the operands of
// quarter q + 2 requested while quarter q + 1 multiplies (two quarters in flight from the start; the register roles are static, nothing is
// copied).  The runtime loop (NG = 0) requests a group's operands right before it multiplies them: a 13-workgroup launch that is nothing
// but exposed latency (17.5 us for 200 x 256 x 256).  The MFMA sequence per accumulator is the same in both: identical bits.
template <int NG>
__global__ __launch_bounds__(256) void linear_layernorm_kernel(const float* __restrict__ x, int ld_x, const float* __restrict__ wt, int Cin,
                                                               const float* __restrict__ bias, const float* __restrict__ res, int ld_res,
                                                               const float* __restrict__ g, const float* __restrict__ b, float eps, int64_t M,
                                                               float* __restrict__ out, int ld_out, int act) {
    __shared__ float red[2][4][16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c16 = lane & 15, kq = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * 16;
    const int64_t arow = row0 + c16 < M ? row0 + c16 : M - 1;
    const float* __restrict__ xa = x + arow * ld_x + 4 * kq;
    const float* __restrict__ wb = wt + (int64_t)(wv * 64 + c16) * Cin + 4 * kq;
    f32x4 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the epilogue's operands (bias, residual rows, LayerNorm weight / bias) are requested BEFORE the contraction: with one wave per SIMD
    // nothing else hides their latency behind the last MFMA (the same values, used in the same expressions)
    float e_bias[4], e_g[4], e_b[4], e_res[4][4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        const int col = wv * 64 + cb * 16 + c16;
        e_bias[cb] = bias ? bias[col] : 0.f;
        e_g[cb] = g[col];
        e_b[cb] = b[col];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t r = row0 + 4 * kq + i;
            e_res[cb][i] = (res && r < M) ? res[r * ld_res + col] : 0.f;
        }
    }
    if (NG > 0) {
        constexpr int NQ = NG > 0 ? NG / 4 : 1;
        f32x4 ab[2][4], wq[2][4][4];
        auto load_q = [&](int buf, int q) {
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const int gi = q * 4 + gg;
                ab[buf][gg] = *(const f32x4*)(xa + 16 * gi);
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) wq[buf][gg][cb] = *(const f32x4*)(wb + (int64_t)cb * 16 * Cin + 16 * gi);
            }
        };
        load_q(0, 0);
        if (NQ > 1) load_q(1, 1);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[q & 1][gg][e], wq[q & 1][gg][cb][e], acc[cb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (q + 2 < NQ) load_q(q & 1, q + 2);
        }
    } else {
    const int ngroups = Cin >> 4;
#pragma unroll 4
    for (int gi = 0; gi < ngroups; ++gi) {
        const f32x4 a = *(const f32x4*)(xa + 16 * gi);
        f32x4 w4[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) w4[cb] = *(const f32x4*)(wb + (int64_t)cb * 16 * Cin + 16 * gi);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], w4[cb][e], acc[cb], 0, 0, 0);
    }
    }
    // acc[cb][i] = (x W^T)[row0 + 4 kq + i][wv * 64 + cb * 16 + c16]
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        const float bv = e_bias[cb];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t r = row0 + 4 * kq + i;
            float t = acc[cb][i] + bv;
            if (res && r < M) t += e_res[cb][i];
            acc[cb][i] = t;
            s[i] += t;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s[i] += __shfl_xor(s[i], o, 64);         // the 16 lanes that share kq: this wave's 64 columns
    }
    if (c16 == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) red[0][wv][4 * kq + i] = s[i];
    }
    __syncthreads();
    float mean[4], q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = 4 * kq + i;
        mean[i] = ((red[0][0][rr] + red[0][1][rr]) + (red[0][2][rr] + red[0][3][rr])) * (1.0f / 256.0f);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) { const float d = acc[cb][i] - mean[i]; q[i] += d * d; }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) q[i] += __shfl_xor(q[i], o, 64);
    }
    if (c16 == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) red[1][wv][4 * kq + i] = q[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = 4 * kq + i;
        const int64_t r = row0 + rr;
        const float var = ((red[1][0][rr] + red[1][1][rr]) + (red[1][2][rr] + red[1][3][rr])) * (1.0f / 256.0f);
        const float rstd = 1.0f / sqrtf(var + eps);
        if (r < M) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const int col = wv * 64 + cb * 16 + c16;
                float y = (acc[cb][i] - mean[i]) * rstd * e_g[cb] + e_b[cb];
                if (act == 1) y = fmaxf(y, 0.f);
                out[r * ld_out + col] = y;
            }
        }
    }
}

int launch_linear_layernorm(const float* x, int ld_x, int64_t M, int Cin, const float* wt, int Cout, const float* bias, const float* res,
                            int ld_res, const float* g, const float* b, float eps, int act, float* out, int ld_out, hipStream_t st) {
    if (M <= 0) return SD3D_OK;
    if (Cout != 256 || Cin <= 0 || (Cin & 15) || (ld_x & 3) || act < 0 || act > 1)
        return sd3d_set_error(SD3D_ERR_ARG, "linear_layernorm: Cout must be 256, Cin a multiple of 16, ld_x a multiple of 4, act 0 / 1");
    const dim3 grid((unsigned)cdiv(M, 16));
    if (Cin == 256)
        hipLaunchKernelGGL(linear_layernorm_kernel<16>, grid, dim3(256), 0, st, x, ld_x, wt, Cin, bias, res, ld_res, g, b, eps, M, out, ld_out, act);
    else if (Cin == 1024)
        hipLaunchKernelGGL(linear_layernorm_kernel<64>, grid, dim3(256), 0, st, x, ld_x, wt, Cin, bias, res, ld_res, g, b, eps, M, out, ld_out, act);
    else
        hipLaunchKernelGGL(linear_layernorm_kernel<0>, grid, dim3(256), 0, st, x, ld_x, wt, Cin, bias, res, ld_res, g, b, eps, M, out, ld_out, act);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// ---------------------------------------------------------------------------------------------
// sine positional encoding (utils.py:53-105).  Per output channel c the host supplies axis[c] in
// {0,1,2} and dim_t[c]; even channels are sin, odd cos.  rng = (lo[3], hi[3]) device.
//   pos = ((x - lo) * 1 / (hi - lo) + 0) * 2pi / dim_t        (same operation order as the reference)
//   out = f(pos) * (mod_num / mod_den)     (box modulation, optional; mod_den row stride may be 0)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sine_pe_kernel(const float* __restrict__ xyz, int ld_xyz, int64_t n,
                                                      const float* __restrict__ rng, const float* __restrict__ dim_t,
                                                      const int8_t* __restrict__ axis, int d_pos,
                                                      const float* __restrict__ mod_num, int ld_num,
                                                      const float* __restrict__ mod_den, int ld_den,
                                                      float* __restrict__ out, int ld_out, const int32_t* __restrict__ row_scene) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * d_pos) return;
    const int64_t r = t / d_pos;
    const int c = (int)(t - r * d_pos);
    const int a = axis[c];
    if (row_scene) rng += 6 * row_scene[r];                   // rows of several scenes: each row normalises by ITS scene's range
    const float lo = rng[a], hi = rng[3 + a];
    float p = ((xyz[r * ld_xyz + a] - lo) * 1.0f) / (hi - lo) + 0.0f;
    p = p * 6.283185307179586f;
    p = p / dim_t[c];
    float y = (c & 1) ? cosf(p) : sinf(p);
    if (mod_num) y *= mod_num[r * ld_num + a] / mod_den[r * ld_den + a];
    out[r * ld_out + c] = y;
}

int launch_sine_pe(const float* xyz, int ld_xyz, int64_t n, const float* rng, const float* dim_t, const int8_t* axis, int d_pos,
                   const float* mod_num, int ld_num, const float* mod_den, int ld_den, float* out, int ld_out, const int32_t* row_scene,
                   hipStream_t st) {
    if (n <= 0) return SD3D_OK;
    hipLaunchKernelGGL(sine_pe_kernel, dim3((unsigned)cdiv(n * d_pos, 256)), dim3(256), 0, st, xyz, ld_xyz, n, rng, dim_t, axis,
                       d_pos, mod_num, ld_num, mod_den, ld_den, out, ld_out, row_scene);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// PositionEmbeddingCoordsSine.get_fourier_embeddings (utils.py:107-142, pos_type = "fourier"): the normalised coordinates
// times 2 pi projected by the fixed Gaussian matrix gauss_B [3, d_pos / 2], then [sin | cos].  Same operation order as the
// reference: (x_a * 2 pi) * B[a][c], summed a = 0, 1, 2.
__global__ __launch_bounds__(256) void fourier_pe_kernel(const float* __restrict__ xyz, int ld_xyz, int64_t n, const float* __restrict__ rng,
                                                         const float* __restrict__ gauss_b, int ld_b, int d_pos,
                                                         float* __restrict__ out, int ld_out, const int32_t* __restrict__ row_scene) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int dh = d_pos >> 1;
    if (t >= n * dh) return;
    const int64_t r = t / dh;
    const int c = (int)(t - r * dh);
    if (row_scene) rng += 6 * row_scene[r];
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = rng[a], hi = rng[3 + a];
        float p = ((xyz[r * ld_xyz + a] - lo) * 1.0f) / (hi - lo) + 0.0f;
        p = p * 6.283185307179586f;
        acc = fmaf(p, gauss_b[a * ld_b + c], acc);
    }
    out[r * ld_out + c] = sinf(acc);
    out[r * ld_out + dh + c] = cosf(acc);
}

int launch_fourier_pe(const float* xyz, int ld_xyz, int64_t n, const float* rng, const float* gauss_b, int ld_b, int d_pos, float* out,
                      int ld_out, const int32_t* row_scene, hipStream_t st) {
    if (n <= 0) return SD3D_OK;
    if (d_pos <= 0 || (d_pos & 1)) return sd3d_set_error(SD3D_ERR_ARG, "fourier_pe: d_pos must be even");
    hipLaunchKernelGGL(fourier_pe_kernel, dim3((unsigned)cdiv(n * (d_pos / 2), 256)), dim3(256), 0, st, xyz, ld_xyz, n, rng, gauss_b, ld_b,
                       d_pos, out, ld_out, row_scene);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// ---------------------------------------------------------------------------------------------
// Fused multi-head attention, head slices of 32 channels, NSRC concatenated sources per head.
//   score(q, key) = scale * sum_src  q_src[q, head*32 : +32] . k_src[key, head*32 : +32]
//   (NSRC = 2 is the reference's per-head [content | positional] concatenation, decoder :681-687,
//    without ever building the 512-wide tensors)
//   out[q, head*32 : +32] = softmax_keys(score masked by bits) @ v[key, head*32 : +32]
// One workgroup = (32-query tile, head); its NW waves split the key tiles and are merged through
// LDS in a fixed order.  Per wave and 32-key tile:
//   S^T  = K_tile . Q_tile^T        16*NSRC MFMA 32x32x2 : lane (col = query) holds 16 keys of ITS query
//   online softmax is therefore lane-local (+1 cross-half shuffle for the max)
//   O^T += V_tile^T . P^T           16 MFMA             : P registers feed the B operand unchanged
// mask bits: word [q][tile], bit b = 1 -> key tile*32+b is blocked.
// ---------------------------------------------------------------------------------------------
struct AttnParams {
    const float* q[2]; int ldq[2];
    const float* k[2]; int ldk[2];
    const float* v; int ldv;
    const uint32_t* bits; int nwords;
    float* out; int ldo;
    int Lq, Lk, H;
    float scale;
    int ksplit;                               // key tiles are dealt to gridDim.z workgroups; partials -> part
    float* part;                              // [qtile][head][ksplit][64 + 1024]: m[32], l[32], O[32 dv][32 q]
    int bf16;                                 // 1: Q, K, P, V rounded to bf16 for the two contractions (fp32 accumulate, fp32 softmax)
    float* lse;                               // optional [H][Lq]: log-sum-exp of every score row (kept for the backward pass)
};

typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));

// BF16 = the "bf16 decoder" of BASELINE config #3: both contractions run on v_mfma_f32_32x32x16_bf16 (8 contraction
// indices per lane instead of 1: 2 * NSRC + 2 MFMAs per key tile instead of 16 * NSRC + 16); scores, softmax and the
// accumulators stay fp32.  The (lane half, register) -> (channel | key) assignment is the fp32 kernel's, which both
// operands of each product share, so only the grouping of the contraction changes.
// Softmax in the log2 domain: the queries are pre-scaled by scale * log2(e), so a probability is ONE v_exp_f32
// (`__builtin_amdgcn_exp2f`) of score - max instead of a library expf (~10 VALU instructions x 16 scores per lane and key tile:
// the VALU, not the matrix pipe, bounded the kernel at one query per superpoint - 42 % pipe busy).  M / lse leave the kernel
// converted back to natural units.  The next tile's keys and mask word are requested while the current tile multiplies, and a
// key tile whose 32 x 32 mask block is all "blocked" is skipped before its keys are touched (wave-uniform).
#define SD3D_LOG2E 1.4426950408889634f
#define SD3D_LN2 0.6931471805599453f
template <int NSRC, bool BF16>
__device__ __forceinline__ void attention_body(const AttnParams& p, const int bx, float* smem) {
    if ((int)blockIdx.z >= p.ksplit) return;                   // (a batched launch: this scene splits its keys fewer ways than the widest)
    const int nw = blockDim.x >> 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int head = blockIdx.y;
    const int q0 = bx * 32;
    const int qi = min(q0 + i, p.Lq - 1);
    const int hc = head * 32 + h * 16;

    float qreg[NSRC][16];
#pragma unroll
    for (int s = 0; s < NSRC; ++s) {
        const float* src = p.q[s] + (int64_t)qi * p.ldq[s] + hc;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x4 t = *(const f32x4*)(src + e * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) qreg[s][e * 4 + c] = t[c] * (p.scale * SD3D_LOG2E);
        }
    }
    abf16x8 qb[NSRC][2];
    if (BF16) {
#pragma unroll
        for (int s = 0; s < NSRC; ++s)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int c = 0; c < 8; ++c) qb[s][g][c] = (__bf16)qreg[s][g * 8 + c];
    }
    f32x16 O;
#pragma unroll
    for (int r = 0; r < 16; ++r) O[r] = 0.f;
    float m = -INFINITY, l = 0.f;                       // running max (log2 units) and sum

    const int ntiles = (p.Lk + 31) >> 5;
    const int tstep = nw * p.ksplit;
    const uint32_t tail = (p.Lk & 31) ? ~0u << (p.Lk & 31) : 0u;          // key slots of the last tile that lie past Lk
    // bit (slot + 4 * h) of the word belongs to this lane's register r with slot = (r & 3) + 8 * (r >> 2): shift once per tile
    auto load_word = [&](int t) -> uint32_t {
        uint32_t w = (p.bits && t < ntiles) ? p.bits[(int64_t)qi * p.nwords + t] : 0u;
        if (t == ntiles - 1) w |= tail;
        return w;
    };
    auto load_k = [&](f32x4 (&kk)[NSRC][4], int t) {
        const int kr = min(t * 32 + i, p.Lk - 1);      // A-operand row = key (tiles past the end reload the last key: harmless)
#pragma unroll
        for (int s = 0; s < NSRC; ++s) {
            const float* src = p.k[s] + (int64_t)kr * p.ldk[s] + hc;
#pragma unroll
            for (int e = 0; e < 4; ++e) kk[s][e] = *(const f32x4*)(src + e * 4);
        }
    };
    constexpr bool KPF = !BF16;                         // next tile's keys in flight while this one multiplies: fp32 187 vs 198 us at Q = S = 3000;
                                                        // the bf16 kernel is better off with the registers (4 waves per SIMD: 91 vs 105 us)
    int t = blockIdx.z * nw + wave;
    uint32_t word = load_word(t);
    f32x4 kcur[NSRC][4], knxt[NSRC][4];
    if (KPF && t < ntiles) load_k(kcur, t);
    for (; t < ntiles; t += tstep) {
        const int kt0 = t * 32;
        const uint32_t word_nxt = load_word(t + tstep);
        if (KPF) load_k(knxt, t + tstep);
        if (__ballot(word != ~0u) != 0ull) {                              // some query of the tile sees some key of it
        if (!KPF) load_k(kcur, t);
        // the value rows of this tile are requested before the score MFMAs: their latency hides behind QK^T + softmax
        float vreg[16];
        {
            const float* vsrc = p.v + head * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = min(kt0 + (r & 3) + 8 * (r >> 2) + 4 * h, p.Lk - 1);
                vreg[r] = vsrc[(int64_t)key * p.ldv];
            }
        }
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NSRC; ++s) {
            if (BF16) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    abf16x8 kb;
#pragma unroll
                    for (int c = 0; c < 8; ++c) kb[c] = (__bf16)kcur[s][2 * g + (c >> 2)][c & 3];
                    S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb, qb[s][g], S, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        S = __builtin_amdgcn_mfma_f32_32x32x2f32(kcur[s][e][c], qreg[s][e * 4 + c], S, 0, 0, 0);
            }
        }
        // S[r] = log2-score(key = kt0 + (r&3) + 8*(r>>2) + 4*h, query = q0 + i)
        const uint32_t wh = word >> (4 * h);
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            S[r] = ((wh >> ((r & 3) + 8 * (r >> 2))) & 1u) ? -INFINITY : S[r];
            tmax = fmaxf(tmax, S[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float mn = fmaxf(m, tmax);
        float alpha = 1.f;
        float pr[16];
        if (mn == -INFINITY) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pr[r] = 0.f;
        } else {
            alpha = __builtin_amdgcn_exp2f(m - mn);     // m = -inf -> 0
            float ls = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { pr[r] = __builtin_amdgcn_exp2f(S[r] - mn); ls += pr[r]; }
            l = l * alpha + ls;
            m = mn;
        }
        if (__ballot(alpha != 1.f) != 0ull) {           // the running maxima settle after a few tiles: no rescale, no dependency on O
#pragma unroll
            for (int r = 0; r < 16; ++r) O[r] *= alpha;
        }
        // O^T[dv][query] += sum_key V[key][dv] * P[query][key];  A = V^T (row = dv = i), B = P^T
        if (BF16) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                abf16x8 vb, pb;
#pragma unroll
                for (int c = 0; c < 8; ++c) { vb[c] = (__bf16)vreg[g * 8 + c]; pb[c] = (__bf16)pr[g * 8 + c]; }
                O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb, pb, O, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) O = __builtin_amdgcn_mfma_f32_32x32x2f32(vreg[r], pr[r], O, 0, 0, 0);
        }
        }
        word = word_nxt;
        if (KPF) {
#pragma unroll
            for (int s = 0; s < NSRC; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) kcur[s][e] = knxt[s][e];
        }
    }
    l += __shfl_xor(l, 32);

    // ---- merge the NW partial results: smem layout per wave: m[32], l[32], O[32 dv][32 q]
    float* wm = smem + wave * (64 + 1024);
    float* wl = wm + 32;
    float* wo = wl + 32;
    if (h == 0) { wm[i] = m; wl[i] = l; }
#pragma unroll
    for (int r = 0; r < 16; ++r) wo[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = O[r];
    __syncthreads();
    for (int e = threadIdx.x; e < 1024; e += blockDim.x) {
        const int qq = e >> 5, dv = e & 31;             // consecutive threads -> consecutive dv (coalesced store)
        float M = -INFINITY;
        for (int w = 0; w < nw; ++w) M = fmaxf(M, smem[w * (64 + 1024) + qq]);
        float L = 0.f, acc = 0.f;
        for (int w = 0; w < nw; ++w) {
            const float* base = smem + w * (64 + 1024);
            const float mw = base[qq];
            const float f = (mw == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mw - M);
            L += base[32 + qq] * f;
            acc += base[64 + dv * 32 + qq] * f;
        }
        if (p.ksplit > 1) {                             // this workgroup's share of the keys: leave (M, L, sum) for the merge pass
            float* dst = p.part + (((int64_t)bx * p.H + head) * p.ksplit + blockIdx.z) * (64 + 1024);
            if (dv == 0) { dst[qq] = M; dst[32 + qq] = L; }
            dst[64 + dv * 32 + qq] = acc;
        } else if (q0 + qq < p.Lq) {
            p.out[(int64_t)(q0 + qq) * p.ldo + head * 32 + dv] = acc / L;
            if (p.lse && dv == 0) p.lse[(int64_t)head * p.Lq + q0 + qq] = M * SD3D_LN2 + logf(L);      // back to natural units
        }
    }
}

template <int NSRC, bool BF16>
__global__ __launch_bounds__(512) void attention_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    attention_body<NSRC, BF16>(p, blockIdx.x, smem);
}

// Several scenes' attentions in ONE launch (the decoder of a batched evaluation forward): blockIdx.x runs over the query tiles of all
// scenes, a workgroup finds its scene from the tiles' prefix sums and then IS that scene's workgroup - same code, same waves per
// workgroup, same key split, so every scene's rows are the bits of its own launch.
struct AttnBatch { int n; int tile0[SD3D_MAX_BATCH + 1]; AttnParams s[SD3D_MAX_BATCH]; };
template <int NSRC, bool BF16>
__global__ __launch_bounds__(512) void attention_batch_kernel(const AttnBatch b) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int si = 0;
    for (int k = 1; k < b.n; ++k) if ((int)blockIdx.x >= b.tile0[k]) si = k;
    attention_body<NSRC, BF16>(b.s[si], blockIdx.x - b.tile0[si], smem);
}

// second pass of the key-split attention: combine the ksplit partial softmax states of one (query tile, head)
__device__ __forceinline__ void attention_merge_body(const AttnParams& p, const int bx) {
    if (p.ksplit <= 1) return;
    const int head = blockIdx.y, q0 = bx * 32;
    const float* base = p.part + ((int64_t)bx * p.H + head) * p.ksplit * (64 + 1024);
    for (int e = threadIdx.x; e < 1024; e += 256) {
        const int qq = e >> 5, dv = e & 31;
        float M = -INFINITY, L = 0.f, acc = 0.f;
        if (p.ksplit <= 8) {
            // every split's three values requested before the first is used (splits past the end re-read the last one and contribute
            // an exact zero): the two dependent loops below ran at two memory latencies per split - 11 us for a 56-workgroup launch
            float mz[8], lz[8], oz[8];
#pragma unroll
            for (int z = 0; z < 8; ++z) {
                const float* b = base + (z < p.ksplit ? z : p.ksplit - 1) * (64 + 1024);
                mz[z] = b[qq]; lz[z] = b[32 + qq]; oz[z] = b[64 + dv * 32 + qq];
            }
#pragma unroll
            for (int z = 0; z < 8; ++z) M = fmaxf(M, mz[z]);
#pragma unroll
            for (int z = 0; z < 8; ++z) {
                const float f = (z >= p.ksplit || mz[z] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mz[z] - M);
                if (z < p.ksplit) { L = __builtin_fmaf(lz[z], f, L); acc = __builtin_fmaf(oz[z], f, acc); }
            }
        } else {
        for (int z = 0; z < p.ksplit; ++z) M = fmaxf(M, base[z * (64 + 1024) + qq]);
        for (int z = 0; z < p.ksplit; ++z) {
            const float* b = base + z * (64 + 1024);
            const float mz = b[qq];
            const float f = (mz == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mz - M);
            L = __builtin_fmaf(b[32 + qq], f, L);
            acc = __builtin_fmaf(b[64 + dv * 32 + qq], f, acc);
        }
        }
        if (q0 + qq < p.Lq) {
            p.out[(int64_t)(q0 + qq) * p.ldo + head * 32 + dv] = acc / L;
            if (p.lse && dv == 0) p.lse[(int64_t)head * p.Lq + q0 + qq] = M * SD3D_LN2 + logf(L);
        }
    }
}
__global__ __launch_bounds__(256) void attention_merge_kernel(const AttnParams p) { attention_merge_body(p, blockIdx.x); }
__global__ __launch_bounds__(256) void attention_merge_batch_kernel(const AttnBatch b) {
    int si = 0;
    for (int k = 1; k < b.n; ++k) if ((int)blockIdx.x >= b.tile0[k]) si = k;
    attention_merge_body(b.s[si], blockIdx.x - b.tile0[si]);
}

size_t attention_ws_bytes(int Lq, int H) { return (size_t)cdiv(Lq, 32) * H * 8 * (64 + 1024) * sizeof(float); }

// waves per workgroup and key split of one attention (Lq queries, Lk keys, H heads) given `ws_bytes` of split workspace
static void attention_config(int Lq, int Lk, int H, bool have_ws, size_t ws_bytes, int* nw_out, int* ks_out) {
    const int ntiles = (Lk + 31) / 32;
    // 4 - 7 key tiles (the 200-key self-attention, the 301-key 2D-query attention): four waves of one or two tiles each instead of two waves
    // of up to four (every tile is a dependent load -> MFMA round trip): decoder 1.863 -> 1.840 ms.  SD3D_ATTN_NW_SMALL=2 restores round 3.
    static const int nw_small = [] { const char* e = getenv("SD3D_ATTN_NW_SMALL"); return e ? atoi(e) : 4; }();
    int nw = ntiles >= 32 ? 8 : (ntiles >= 8 ? 4 : (ntiles >= 4 ? nw_small : (ntiles >= 2 ? 2 : 1)));
    // few query tiles x heads (200 queries: 56 workgroups on 256 CUs) and many key tiles: deal the key tiles to several
    // workgroups and merge their softmax states in a second, tiny pass (each wave walks its tiles serially, so the
    // single-pass kernel is bound by ~12 dependent load -> MFMA round trips per wave)
    const int64_t wgs = cdiv(Lq, 32) * H;
    int ks = 1;
    if (have_ws && wgs < 128 && ntiles >= 4 * nw) {
        nw = nw > 4 ? 4 : nw;                                  // 4 waves x more key splits: 200 queries x 3000 keys 44.4 -> 37.7 us (8 waves merge through LDS longer than they multiply)
        ks = (int)(512 / wgs);
        const int most = ntiles / (2 * nw);
        ks = ks > most ? most : ks;
        ks = ks > 8 ? 8 : ks;
        if (ks < 2 || ws_bytes < (size_t)wgs * ks * (64 + 1024) * sizeof(float)) ks = 1;
    }
    *nw_out = nw; *ks_out = ks;
}

int launch_attention(const AttnParams& p_in, int nsrc, void* ws, size_t ws_bytes, hipStream_t st, bool merge = true) {
    AttnParams p = p_in;
    if (p.Lq <= 0 || p.Lk <= 0) return sd3d_set_error(SD3D_ERR_ARG, "attention: empty query or key set");
    for (int s = 0; s < nsrc; ++s)
        if ((p.ldq[s] & 3) || (p.ldk[s] & 3)) return sd3d_set_error(SD3D_ERR_ARG, "attention: q/k strides must be multiples of 4");
    int nw, ks;
    attention_config(p.Lq, p.Lk, p.H, ws != nullptr, ws_bytes, &nw, &ks);
    p.ksplit = ks;
    p.part = (float*)ws;
    const dim3 grid((unsigned)cdiv(p.Lq, 32), (unsigned)p.H, (unsigned)ks), block(64 * nw);
    const size_t sm = (size_t)nw * (64 + 1024) * sizeof(float);
    if (nsrc == 1 && !p.bf16) hipLaunchKernelGGL((attention_kernel<1, false>), grid, block, sm, st, p);
    else if (nsrc == 2 && !p.bf16) hipLaunchKernelGGL((attention_kernel<2, false>), grid, block, sm, st, p);
    else if (nsrc == 1) hipLaunchKernelGGL((attention_kernel<1, true>), grid, block, sm, st, p);
    else if (nsrc == 2) hipLaunchKernelGGL((attention_kernel<2, true>), grid, block, sm, st, p);
    else return sd3d_set_error(SD3D_ERR_ARG, "attention: nsrc must be 1 or 2");
    if (ks > 1 && merge) hipLaunchKernelGGL(attention_merge_kernel, dim3((unsigned)cdiv(p.Lq, 32), (unsigned)p.H), dim3(256), 0, st, p);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// n <= SD3D_MAX_BATCH independent attentions (same heads, scale, sources, arithmetic type) in one launch - when every scene's own launch
// would use the same number of waves per workgroup; otherwise one launch per scene.  Scene i's split workspace is
// attention_ws_bytes(Lq_i, H) bytes, back to back in ws.
// ksplit_out / part_off_out (host arrays of n entries, optional): the launch then STOPS after the key-split pass - scene i's rows are
// final in its `out` where ksplit_out[i] == 1 and otherwise wait as partial softmax states at ws + part_off_out[i] floats for the
// consumer that combines them (rowchain.hip MERGE: the expression of attention_merge_body).
int launch_attention_batch(int n, const AttnParams* jobs, int nsrc, void* ws, size_t ws_bytes, hipStream_t st, int32_t* ksplit_out = nullptr,
                           int64_t* part_off_out = nullptr) {
    if (n <= 0) return SD3D_OK;
    const bool merge = ksplit_out == nullptr;
    if (n > SD3D_MAX_BATCH) return sd3d_set_error(SD3D_ERR_ARG, "attention_batch: at most 16 scenes per call");
    AttnBatch b;
    b.n = n;
    int nw0 = 0, ks_max = 1, tiles = 0;
    bool same = true;
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        AttnParams p = jobs[i];
        if (p.Lq <= 0 || p.Lk <= 0) return sd3d_set_error(SD3D_ERR_ARG, "attention_batch: empty query or key set");
        for (int s = 0; s < nsrc; ++s)
            if ((p.ldq[s] & 3) || (p.ldk[s] & 3)) return sd3d_set_error(SD3D_ERR_ARG, "attention_batch: q/k strides must be multiples of 4");
        const size_t need = attention_ws_bytes(p.Lq, p.H);
        const bool have = ws != nullptr && off + need <= ws_bytes;
        int nw, ks;
        attention_config(p.Lq, p.Lk, p.H, have, need, &nw, &ks);
        p.ksplit = ks;
        p.part = have ? (float*)((char*)ws + off) : nullptr;
        off += need;
        if (i == 0) nw0 = nw;
        same = same && nw == nw0;
        ks_max = ks > ks_max ? ks : ks_max;
        b.tile0[i] = tiles;
        tiles += (int)cdiv(p.Lq, 32);
        b.s[i] = p;
        if (!merge) {
            ksplit_out[i] = ks;
            if (part_off_out) part_off_out[i] = have ? (int64_t)((off - need) / sizeof(float)) : 0;
        }
    }
    b.tile0[n] = tiles;
    if (!same) {                                               // different workgroup shapes: each scene its own launch (same results)
        for (int i = 0; i < n; ++i) {
            const int rc = launch_attention(jobs[i], nsrc, b.s[i].part, b.s[i].part ? attention_ws_bytes(jobs[i].Lq, jobs[i].H) : 0, st, merge);
            if (rc != SD3D_OK) return rc;
        }
        return SD3D_OK;
    }
    const bool bf16 = b.s[0].bf16 != 0;
    const dim3 grid((unsigned)tiles, (unsigned)b.s[0].H, (unsigned)ks_max), block(64 * nw0);
    const size_t sm = (size_t)nw0 * (64 + 1024) * sizeof(float);
    if (nsrc == 1 && !bf16) hipLaunchKernelGGL((attention_batch_kernel<1, false>), grid, block, sm, st, b);
    else if (nsrc == 2 && !bf16) hipLaunchKernelGGL((attention_batch_kernel<2, false>), grid, block, sm, st, b);
    else if (nsrc == 1) hipLaunchKernelGGL((attention_batch_kernel<1, true>), grid, block, sm, st, b);
    else if (nsrc == 2) hipLaunchKernelGGL((attention_batch_kernel<2, true>), grid, block, sm, st, b);
    else return sd3d_set_error(SD3D_ERR_ARG, "attention_batch: nsrc must be 1 or 2");
    if (ks_max > 1 && merge) hipLaunchKernelGGL(attention_merge_batch_kernel, dim3((unsigned)tiles, (unsigned)b.s[0].H), dim3(256), 0, st, b);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// ---------------------------------------------------------------------------------------------
// mask head bits (_forward_head :567-572): blocked = sigmoid(logit) < thr; a row with every real
// column blocked is reset to all-open.  Bits beyond S are always 1 (blocked).  One wave per row.
// ---------------------------------------------------------------------------------------------
// One WORKGROUP per row: the four waves read the row coalesced (lane = column), a ballot turns 64 verdicts into two words;
// the row-wide "any column open" meets in LDS.  (The first version gave every lane a 32-bit word and walked its 32 columns
// one dependent, uncoalesced load at a time: 15.7 us for 200 x 3000 logits; this one is bound by the launch.)
__device__ __forceinline__ void mask_bits_body(const float* __restrict__ logits, int ld, int S, float thr, uint32_t* __restrict__ bits,
                                               int nwords, const int64_t q, int* open_s) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float* row = logits + q * ld;
    uint32_t* out = bits + q * nwords;
    bool any_open = false;
    for (int s0 = wv * 64; s0 < nwords * 32; s0 += 256) {
        const int s = s0 + lane;
        bool blk = true;                                       // columns beyond S stay blocked
        if (s < S) {
            const float sg = 1.0f / (1.0f + expf(-row[s]));
            blk = sg < thr;
            any_open |= !blk;
        }
        const uint64_t bal = __ballot(blk);
        if (lane == 0) {
            const int w = s0 >> 5;
            out[w] = (uint32_t)bal;
            if (w + 1 < nwords) out[w + 1] = (uint32_t)(bal >> 32);
        }
    }
    const bool wave_open = __ballot(any_open) != 0ull;
    if (lane == 0) open_s[wv] = wave_open ? 1 : 0;
    __syncthreads();
    if (!(open_s[0] | open_s[1] | open_s[2] | open_s[3])) {    // dead row -> attend everywhere (real columns only)
        for (int w = threadIdx.x; w < nwords; w += 256) {
            const int rem = S - w * 32;
            out[w] = rem >= 32 ? 0u : (0xFFFFFFFFu << rem);
        }
    }
}
__global__ __launch_bounds__(256) void mask_bits_kernel(const float* __restrict__ logits, int ld, int64_t Q, int S, float thr,
                                                        uint32_t* __restrict__ bits, int nwords) {
    __shared__ int open_s[4];
    mask_bits_body(logits, ld, S, thr, bits, nwords, blockIdx.x, open_s);
}
// the same for the logit matrices of several scenes (one workgroup per query row of any scene)
struct MaskBitsBatch { int n; int row0[SD3D_MAX_BATCH + 1]; const float* logits[SD3D_MAX_BATCH]; uint32_t* bits[SD3D_MAX_BATCH];
                       int ld[SD3D_MAX_BATCH], S[SD3D_MAX_BATCH], nwords[SD3D_MAX_BATCH]; };
__global__ __launch_bounds__(256) void mask_bits_batch_kernel(const MaskBitsBatch b, float thr) {
    __shared__ int open_s[4];
    int si = 0;
    for (int k = 1; k < b.n; ++k) if ((int)blockIdx.x >= b.row0[k]) si = k;
    mask_bits_body(b.logits[si], b.ld[si], b.S[si], thr, b.bits[si], b.nwords[si], blockIdx.x - b.row0[si], open_s);
}
int launch_mask_bits_batch(int n, const float* const* logits, const int* ld, const int64_t* Q, const int* S, uint32_t* const* bits,
                           const int* nwords, float thr, hipStream_t st) {
    if (n <= 0) return SD3D_OK;
    if (n > SD3D_MAX_BATCH) return sd3d_set_error(SD3D_ERR_ARG, "mask_bits_batch: at most 16 scenes per call");
    MaskBitsBatch b;
    b.n = n;
    int rows = 0;
    for (int i = 0; i < n; ++i) {
        if (nwords[i] != (S[i] + 31) / 32) return sd3d_set_error(SD3D_ERR_ARG, "mask_bits_batch: nwords != ceil(S/32)");
        b.row0[i] = rows; rows += (int)Q[i];
        b.logits[i] = logits[i]; b.bits[i] = bits[i]; b.ld[i] = ld[i]; b.S[i] = S[i]; b.nwords[i] = nwords[i];
    }
    b.row0[n] = rows;
    if (rows <= 0) return SD3D_OK;
    hipLaunchKernelGGL(mask_bits_batch_kernel, dim3((unsigned)rows), dim3(256), 0, st, b, thr);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

int launch_mask_bits(const float* logits, int ld, int64_t Q, int S, float thr, uint32_t* bits, int nwords, hipStream_t st) {
    if (Q <= 0) return SD3D_OK;
    if (nwords != (S + 31) / 32) return sd3d_set_error(SD3D_ERR_ARG, "mask_bits: nwords != ceil(S/32)");
    hipLaunchKernelGGL(mask_bits_kernel, dim3((unsigned)Q), dim3(256), 0, st, logits, ld, Q, S, thr, bits, nwords);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// near[m][w] bit b = 1  <=>  L1(pos[w*32+b], ctr[m]) < thr        (torch.cdist(p=1) < thr, :721-722)
__global__ __launch_bounds__(256) void near_bits_kernel(const float* __restrict__ pos, int64_t S, const float* __restrict__ ctr,
                                                        int64_t Mq, float thr, uint32_t* __restrict__ near, int nwords) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= Mq * nwords) return;
    const int64_t m = t / nwords;
    const int w = (int)(t - m * nwords);
    const float cx = ctr[m * 3], cy = ctr[m * 3 + 1], cz = ctr[m * 3 + 2];
    uint32_t word = 0u;
    for (int b = 0; b < 32; ++b) {
        const int64_t s = (int64_t)w * 32 + b;
        if (s < S) {
            const float d = fabsf(pos[s * 3] - cx) + fabsf(pos[s * 3 + 1] - cy) + fabsf(pos[s * 3 + 2] - cz);
            word |= (d < thr ? 1u : 0u) << b;
        }
    }
    near[t] = word;
}

// blocked2d[q] bit m = 1 <=> no superpoint is both open for query q and near 2D query m  (:722-726);
// key Mq is the appended dummy key (always open); bits beyond Mq are blocked.
// A workgroup takes QB consecutive queries (their open words ~blocked[q][:] in LDS) and its waves take the output words: for
// each of a word's 32 keys m the wave reads near[m][:] with lane = superpoint word (coalesced rows of the 113 KB table), ANDs it
// with the QB open rows and reduces with a ballot.  One query per workgroup and lane = m (the first version) re-read the whole
// table per query through 64 cache lines per load: 150 us at 3000 queries x 300 keys, the whole table 3000 times.
#define DINOX_QB_MAX 8
template <int QB>
__device__ __forceinline__ void dinox_mask_bits_body(const uint32_t* __restrict__ blocked, const uint32_t* __restrict__ near, int nwords,
                                                     int64_t Q, int64_t Mq, uint32_t* __restrict__ out, int nwords_out, const int64_t q0,
                                                     uint32_t* open_w) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint16_t* out16 = (uint16_t*)out;                           // a wave produces 16 keys = half an output word at a time
    const int nhalf = nwords_out * 2;
    if (nwords <= 128) {
        // up to 4096 superpoints: a lane keeps its two words of every open row in registers; the near rows of a unit's SIXTEEN keys are
        // requested before the first is used (one key at a time the loop ran at the latency of one load per key), and the few-hundred-query
        // launches run eight waves per workgroup: twenty units are three rounds of one memory latency each
        uint32_t o[QB][2];
#pragma unroll
        for (int qq = 0; qq < QB; ++qq)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int w = lane + 64 * i;
                o[qq][i] = (q0 + qq < Q && w < nwords) ? ~blocked[(q0 + qq) * nwords + w] : 0u;
            }
        for (int unit = wave; unit < nhalf; unit += nw) {
            uint32_t word[QB];
#pragma unroll
            for (int qq = 0; qq < QB; ++qq) word[qq] = 0u;
            {
                uint32_t nb[16][2];                                // all sixteen keys of the unit requested before the first is used
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int64_t m = (int64_t)unit * 16 + u;
                    const int64_t mc = m < Mq ? m : (Mq > 0 ? Mq - 1 : 0);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int w = lane + 64 * i;
                        nb[u][i] = (Mq > 0 && w < nwords) ? near[mc * nwords + w] : 0u;
                    }
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int64_t m = (int64_t)unit * 16 + u;
#pragma unroll
                    for (int qq = 0; qq < QB; ++qq) {
                        const bool hit = __ballot(((o[qq][0] & nb[u][0]) | (o[qq][1] & nb[u][1])) != 0u) != 0ull;
                        const uint32_t blk = m < Mq ? (hit ? 0u : 1u) : (m == Mq ? 0u : 1u);
                        word[qq] |= blk << u;
                    }
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int qq = 0; qq < QB; ++qq)
                    if (q0 + qq < Q) out16[(q0 + qq) * nhalf + unit] = (uint16_t)word[qq];
            }
        }
        return;
    }
    // more superpoints than two words per lane: open rows in LDS, one key at a time
    for (int e = threadIdx.x; e < QB * nwords; e += blockDim.x) {
        const int qq = e / nwords, w = e - qq * nwords;
        open_w[e] = q0 + qq < Q ? ~blocked[(q0 + qq) * nwords + w] : 0u;
    }
    __syncthreads();
    for (int unit = wave; unit < nhalf; unit += nw) {
        uint32_t word[QB];
#pragma unroll
        for (int qq = 0; qq < QB; ++qq) word[qq] = 0u;
        for (int jb = 0; jb < 16; ++jb) {
            const int64_t m = (int64_t)unit * 16 + jb;                          // wave-uniform
            const int64_t mc = m < Mq ? m : (Mq > 0 ? Mq - 1 : 0);
            uint32_t acc[QB];
#pragma unroll
            for (int qq = 0; qq < QB; ++qq) acc[qq] = 0u;
            if (Mq > 0)
                for (int w = lane; w < nwords; w += 64) {
                    const uint32_t nb = near[mc * nwords + w];
#pragma unroll
                    for (int qq = 0; qq < QB; ++qq) acc[qq] |= open_w[qq * nwords + w] & nb;
                }
#pragma unroll
            for (int qq = 0; qq < QB; ++qq) {
                const bool hit = __ballot(acc[qq] != 0u) != 0ull;
                const uint32_t blk = m < Mq ? (hit ? 0u : 1u) : (m == Mq ? 0u : 1u);
                word[qq] |= blk << jb;
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int qq = 0; qq < QB; ++qq)
                if (q0 + qq < Q) out16[(q0 + qq) * nhalf + unit] = (uint16_t)word[qq];
        }
    }
}
template <int QB>
__global__ __launch_bounds__(512) void dinox_mask_bits_kernel(const uint32_t* __restrict__ blocked, const uint32_t* __restrict__ near,
                                                              int nwords, int64_t Q, int64_t Mq, uint32_t* __restrict__ out,
                                                              int nwords_out) {
    extern __shared__ uint32_t open_w[];                       // ~blocked[q0 .. q0 + QB)[:]
    dinox_mask_bits_body<QB>(blocked, near, nwords, Q, Mq, out, nwords_out, (int64_t)blockIdx.x * QB, open_w);
}
struct DinoxBitsBatch { int n; int wg0[SD3D_MAX_BATCH + 1]; const uint32_t* blocked[SD3D_MAX_BATCH]; const uint32_t* near[SD3D_MAX_BATCH];
                        uint32_t* out[SD3D_MAX_BATCH]; int nwords[SD3D_MAX_BATCH], Q[SD3D_MAX_BATCH], Mq[SD3D_MAX_BATCH], nwords_out[SD3D_MAX_BATCH]; };
template <int QB>
__global__ __launch_bounds__(512) void dinox_mask_bits_batch_kernel(const DinoxBitsBatch b) {
    extern __shared__ uint32_t open_w[];
    int si = 0;
    for (int k = 1; k < b.n; ++k) if ((int)blockIdx.x >= b.wg0[k]) si = k;
    dinox_mask_bits_body<QB>(b.blocked[si], b.near[si], b.nwords[si], b.Q[si], b.Mq[si], b.out[si], b.nwords_out[si],
                             (int64_t)(blockIdx.x - b.wg0[si]) * QB, open_w);
}
// queries per workgroup: 8 where that still leaves >= 256 workgroups, 2 for the few hundred queries of the default mode
static int dinox_qb(int64_t rows) { return rows >= 2048 ? 8 : 2; }
int launch_dinox_mask_bits_batch(int n, const uint32_t* const* blocked, const uint32_t* const* near, const int* nwords, const int64_t* Q,
                                 const int64_t* Mq, uint32_t* const* out, const int* nwords_out, hipStream_t st) {
    if (n <= 0) return SD3D_OK;
    if (n > SD3D_MAX_BATCH) return sd3d_set_error(SD3D_ERR_ARG, "dinox_mask_bits_batch: at most 16 scenes per call");
    DinoxBitsBatch b;
    b.n = n;
    int64_t rows = 0;
    int wmax = 0;
    for (int i = 0; i < n; ++i) rows += Q[i];
    const int qb = dinox_qb(rows);
    int wgs = 0;
    for (int i = 0; i < n; ++i) {
        if (nwords_out[i] != (int)((Mq[i] + 1 + 31) / 32)) return sd3d_set_error(SD3D_ERR_ARG, "dinox_mask_bits_batch: nwords_out != ceil((M+1)/32)");
        b.wg0[i] = wgs; wgs += (int)cdiv(Q[i], qb);
        b.blocked[i] = blocked[i]; b.near[i] = near[i]; b.out[i] = out[i]; b.nwords[i] = nwords[i]; b.Q[i] = (int)Q[i]; b.Mq[i] = (int)Mq[i];
        b.nwords_out[i] = nwords_out[i];
        wmax = nwords[i] > wmax ? nwords[i] : wmax;
    }
    b.wg0[n] = wgs;
    if (wgs <= 0) return SD3D_OK;
    const size_t sm = (size_t)qb * wmax * sizeof(uint32_t);
    if (qb == 8) hipLaunchKernelGGL(dinox_mask_bits_batch_kernel<8>, dim3((unsigned)wgs), dim3(256), sm, st, b);
    else hipLaunchKernelGGL(dinox_mask_bits_batch_kernel<2>, dim3((unsigned)wgs), dim3(512), sm, st, b);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

int launch_near_bits(const float* pos, int64_t S, const float* ctr, int64_t Mq, float thr, uint32_t* near, int nwords, hipStream_t st) {
    if (Mq <= 0 || S <= 0) return SD3D_OK;
    hipLaunchKernelGGL(near_bits_kernel, dim3((unsigned)cdiv(Mq * nwords, 256)), dim3(256), 0, st, pos, S, ctr, Mq, thr, near, nwords);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
int launch_dinox_mask_bits(const uint32_t* blocked, const uint32_t* near, int nwords, int64_t Q, int64_t Mq, uint32_t* out,
                           int nwords_out, hipStream_t st) {
    if (Q <= 0) return SD3D_OK;
    if (nwords_out != (int)((Mq + 1 + 31) / 32)) return sd3d_set_error(SD3D_ERR_ARG, "dinox_mask_bits: nwords_out != ceil((M+1)/32)");
    const int qb = dinox_qb(Q);
    const size_t sm = (size_t)qb * nwords * sizeof(uint32_t);
    if (sm > 64 * 1024) return sd3d_set_error(SD3D_ERR_ARG, "dinox_mask_bits: more than 65536 superpoints");
    if (qb == 8) hipLaunchKernelGGL(dinox_mask_bits_kernel<8>, dim3((unsigned)cdiv(Q, 8)), dim3(256), sm, st, blocked, near, nwords, Q, Mq, out, nwords_out);
    else hipLaunchKernelGGL(dinox_mask_bits_kernel<2>, dim3((unsigned)cdiv(Q, 2)), dim3(512), sm, st, blocked, near, nwords, Q, Mq, out, nwords_out);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// ---------------------------------------------------------------------------------------------
// iterative box refinement (:735-759, 768-772), elementwise over [Q,3]:
//   center   = ref_point + d_center
//   size     = normalize ? sigmoid(inverse_sigmoid(size_prev) + d_size) : size_prev + d_size
//   size_out = normalize ? size * (hi - lo) : size           (metric size reported to the caller)
// size_prev may have row stride 0 (layer 0: one broadcast row).
// ---------------------------------------------------------------------------------------------
__global__ void box_refine_kernel(const float* __restrict__ ref, const float* __restrict__ dc, const float* __restrict__ sprev,
                                  int ld_sprev, const float* __restrict__ ds, const float* __restrict__ rng, int normalize,
                                  int64_t Q, float* __restrict__ center, float* __restrict__ size, float* __restrict__ size_out,
                                  const int32_t* __restrict__ row_scene) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= Q * 3) return;
    const int64_t q = t / 3;
    const int a = (int)(t - q * 3);
    if (row_scene) rng += 6 * row_scene[q];
    center[t] = ref[t] + dc[t];
    if (!ds) return;
    const float sp = sprev[q * ld_sprev + a];
    float s;
    if (normalize) {
        const float eps = 1e-5f;
        const float x = fminf(fmaxf(sp, 0.f), 1.f);
        const float x1 = fmaxf(x, eps), x2 = fmaxf(1.f - x, eps);
        const float z = logf(x1 / x2) + ds[t];
        s = 1.0f / (1.0f + expf(-z));
        size_out[t] = s * (rng[3 + a] - rng[a]);
    } else {
        s = sp + ds[t];
        size_out[t] = s;
    }
    size[t] = s;
}

int launch_box_refine(const float* ref, const float* dc, const float* sprev, int ld_sprev, const float* ds, const float* rng,
                      int normalize, int64_t Q, float* center, float* size, float* size_out, const int32_t* row_scene, hipStream_t st) {
    if (Q <= 0) return SD3D_OK;
    hipLaunchKernelGGL(box_refine_kernel, dim3((unsigned)cdiv(Q * 3, 256)), dim3(256), 0, st, ref, dc, sprev, ld_sprev, ds, rng,
                       normalize, Q, center, size, size_out, row_scene);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// ---------------------------------------------------------------------------------------------
// y = act(x * scale + shift) over [M, C] where x may be the concatenation [x0 (C0 ch) | x1 (C - C0 ch)]:
// the pre-activation BatchNorm + ReLU of the spconv residual blocks (spconvunet.py:48-51, 154-156,
// 184-187, 227-229).  With `add` the identity branch is summed in AFTER the activation - the tail of a
// normalize_before=False block, conv -> BN -> ReLU, then + i_branch (spconvunet.py:66-81, 95-97).
// HBM-bound elementwise pass, float4 per thread.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scale_shift_act_kernel(const float* __restrict__ x0, int ld0, int C0,
                                                              const float* __restrict__ x1, int ld1, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int act, int64_t M, int C,
                                                              const float* __restrict__ add, int ld_add,
                                                              float* __restrict__ out, int ld_out) {
    const int cv = C >> 2;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= M * cv) return;
    const int64_t r = t / cv;
    const int c = (int)(t - r * cv) * 4;
    const f32x4 v = (c < C0) ? *(const f32x4*)(x0 + r * ld0 + c) : *(const f32x4*)(x1 + r * ld1 + (c - C0));
    const f32x4 s = *(const f32x4*)(scale + c), b = *(const f32x4*)(shift + c);
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        y[e] = v[e] * s[e] + b[e];
        if (act == 1) y[e] = fmaxf(y[e], 0.f);
    }
    if (add) y += *(const f32x4*)(add + r * ld_add + c);      // AFTER the activation: the post-activation residual blocks
    *(f32x4*)(out + r * ld_out + c) = y;
}

int launch_scale_shift_act(const float* x0, int ld0, int C0, const float* x1, int ld1, const float* scale, const float* shift,
                           int act, int64_t M, int C, const float* add, int ld_add, float* out, int ld_out, hipStream_t st) {
    if (M <= 0) return SD3D_OK;
    if ((C & 3) || (C0 & 3) || (ld0 & 3) || (ld_out & 3) || (x1 && (ld1 & 3)) || (add && (ld_add & 3)))
        return sd3d_set_error(SD3D_ERR_ARG, "scale_shift_act: channels and strides must be multiples of 4");
    if (!x1) C0 = C;
    hipLaunchKernelGGL(scale_shift_act_kernel, dim3((unsigned)cdiv(M * (C >> 2), 256)), dim3(256), 0, st, x0, ld0, C0, x1, ld1, scale,
                       shift, act, M, C, add, ld_add, out, ld_out);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
