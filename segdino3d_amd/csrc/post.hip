// Post-processing kernels (reference: segdino3d/models/architecture/baseline3d.py:22-141 matrix-NMS,
// :348-371 box filter, :406-486 instance prediction, :488-556 semantic / panoptic maps;
// SURVEY.md 2b K20, K21).  All HBM-bound; the [600, N] point masks are produced once, bit-tested and
// written as bytes (torch.bool layout the evaluator reads), never as fp32.
#include "common.h"

__device__ static inline float wred_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ static inline float wred_max(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
    return v;
}

// scores[q*C + c] = softmax(cls[q, 0:C+1])[c], c < C   (:427)        one wave per query
__global__ __launch_bounds__(256) void class_scores_kernel(const float* __restrict__ cls, int ld, int64_t Q, int C,
                                                           float* __restrict__ scores, float* __restrict__ rowmax) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    const float* row = cls + q * ld;
    float mx = -INFINITY;
    for (int c = lane; c <= C; c += 64) mx = fmaxf(mx, row[c]);
    mx = wred_max(mx);
    float s = 0.f;
    for (int c = lane; c <= C; c += 64) s += expf(row[c] - mx);
    s = wred_sum(s);
    float best = -INFINITY;
    for (int c = lane; c < C; c += 64) {
        const float pr = expf(row[c] - mx) / s;
        if (scores) scores[q * C + c] = pr;
        best = fmaxf(best, pr);
    }
    best = wred_max(best);
    if (rowmax && lane == 0) rowmax[q] = best;
}

// per selected instance r (flat index f = flat_idx[r] into [Q, C]):
//   label = f % C, qidx = f / C, logit row = masks[qidx]
//   score_out[r] = score_in[r] * sum(sigmoid * [logit > 0]) / (sum([logit > 0]) + 1e-6)     (:443-446)
__global__ __launch_bounds__(256) void mask_scores_kernel(const float* __restrict__ masks, int ld, int S,
                                                          const uint32_t* __restrict__ flat_idx, const float* __restrict__ score_in,
                                                          int n, int C, int normalize, int32_t* __restrict__ labels,
                                                          int32_t* __restrict__ qidx, float* __restrict__ score_out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const uint32_t f = flat_idx[r];
    const int q = (int)(f / (uint32_t)C);
    const float* row = masks + (int64_t)q * ld;
    float num = 0.f, den = 0.f;
    for (int s = lane; s < S; s += 64) {
        const float x = row[s];
        if (x > 0.f) { num += 1.0f / (1.0f + expf(-x)); den += 1.f; }
    }
    num = wred_sum(num);
    den = wred_sum(den);
    if (lane == 0) {
        labels[r] = (int32_t)(f % (uint32_t)C);
        qidx[r] = q;
        score_out[r] = normalize ? score_in[r] * (num / (den + 1e-6f)) : score_in[r];
    }
}

// sig[r, :] = sigmoid(masks[qidx[order[r]], :]) (zero padded to ld_out), area[r] = sum      (:441, :66)
__global__ __launch_bounds__(256) void gather_sigmoid_kernel(const float* __restrict__ masks, int ld, int S,
                                                             const int32_t* __restrict__ qidx, const uint32_t* __restrict__ order,
                                                             int n, float* __restrict__ sig, int ld_out, float* __restrict__ area) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const float* row = masks + (int64_t)qidx[order[r]] * ld;
    float a = 0.f;
    for (int s = lane; s < ld_out; s += 64) {
        float y = 0.f;
        if (s < S) { y = 1.0f / (1.0f + expf(-row[s])); a += y; }
        sig[(int64_t)r * ld_out + s] = y;
    }
    a = wred_sum(a);
    if (lane == 0) area[r] = a;
}

// matrix-NMS decay (:85-119).  inter [n, ld] = sig . sig^T, rows already sorted by score (desc).
//   diou[i][j] = (i < j && label_i == label_j) ? inter / (area_i + area_j - inter) : 0
//   comp[j]    = max_i diou[i][j]
//   coef[j]    = min_i decay(diou[i][j]) / decay(comp[i])      (linear: 1 - x, gaussian: exp(-sigma x^2))
__device__ static inline float nms_diou(const float* inter, int ld, const float* area, const int32_t* lab, int i, int j) {
    if (i >= j || lab[i] != lab[j]) return 0.f;
    const float it = inter[(int64_t)i * ld + j];
    return it / (area[i] + area[j] - it);
}
// 64 columns x 16 row lanes per workgroup: row lane r walks the rows i = r, r + 16, ...; max / min / "saw a NaN" are order-free,
// so the result does not depend on the split (one thread per column walked all n rows alone: 110 - 130 us for n = 600)
#define NMS_RL 16
__global__ __launch_bounds__(64 * NMS_RL) void nms_comp_kernel(const float* __restrict__ inter, int ld, const float* __restrict__ area,
                                                               const int32_t* __restrict__ lab, int n, float* __restrict__ comp) {
    __shared__ float sm[NMS_RL][64];
    __shared__ int sn[NMS_RL][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + cl;
    float m = 0.f;                                // row j..n-1 of the column are zeros -> max >= 0
    bool nan = false;
    if (j < n)
        for (int i = rl; i < j; i += NMS_RL) {
            const float d = nms_diou(inter, ld, area, lab, i, j);
            nan |= (d != d);
            m = fmaxf(m, d);
        }
    sm[rl][cl] = m; sn[rl][cl] = nan;
    __syncthreads();
    if (rl == 0 && j < n) {
        for (int r = 1; r < NMS_RL; ++r) { m = fmaxf(m, sm[r][cl]); nan |= sn[r][cl] != 0; }
        comp[j] = nan ? NAN : m;
    }
}
__global__ __launch_bounds__(64 * NMS_RL) void nms_coef_kernel(const float* __restrict__ inter, int ld, const float* __restrict__ area,
                                                               const int32_t* __restrict__ lab, const float* __restrict__ comp, int n,
                                                               int gaussian, float sigma, const float* __restrict__ score_in,
                                                               float* __restrict__ score_out) {
    __shared__ float sm[NMS_RL][64];
    __shared__ int sn[NMS_RL][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + cl;
    float m = INFINITY;
    bool nan = false;
    if (j < n)
        for (int i = rl; i < n; i += NMS_RL) {
            const float d = nms_diou(inter, ld, area, lab, i, j);
            const float c = comp[i];
            const float v = gaussian ? expf(-sigma * d * d) / expf(-sigma * c * c) : (1.f - d) / (1.f - c);
            nan |= (v != v);
            m = fminf(m, v);
        }
    sm[rl][cl] = m; sn[rl][cl] = nan;
    __syncthreads();
    if (rl == 0 && j < n) {
        for (int r = 1; r < NMS_RL; ++r) { m = fminf(m, sm[r][cl]); nan |= sn[r][cl] != 0; }
        score_out[j] = score_in[j] * (nan ? NAN : m);
    }
}

// Point masks (:453-454, :464-465, :348-371).  For final row r (src row = src[r] of sig), point p:
//   bit = sig[src[r]][superpoints[p]] > sp_thr ; count[r] = number of bits BEFORE the box filter;
//   out[r][p] = bit && inside(points[p], center[r] +- size[r] * (1 + loose) / 2)   (if boxes given)
// One thread handles 4 consecutive points for EM_ROWS consecutive rows: the superpoint ids and the
// coordinates are read once per EM_ROWS rows, each row costs 4 gathered floats (L2-resident sig row)
// and one uchar4 store.  grid = (point quads / 256, row groups).
// Two passes.  (1) em_rowbits_kernel thresholds the [n, S] sigmoid rows into a bit table transposed to
// [S][W] words (bit r of word r/32 = row r is on for superpoint s): 226 KB for 600 x 3000, L2-resident.
// (2) em_expand_kernel: a thread owns 4 consecutive points and one 32-row word; it reads the word of
// each point's superpoint once and emits 32 uchar4 stores (one per row) - the kernel is bound by the
// n x N bytes it writes (the first version gathered one float per (row, point): 90 M random 4-byte loads,
// 0.62 ms for 600 x 150 k).
__global__ __launch_bounds__(256) void em_hist_kernel(const int64_t* __restrict__ superpoints, int64_t N, int S, int32_t* __restrict__ npts) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= N) return;
    const int64_t sp = superpoints[p];
    if (sp >= 0 && sp < S) atomicAdd(&npts[sp], 1);
}

// also count[r] = number of points whose superpoint is on in row r = sum_s bit[r][s] * npts[s]: done here at
// superpoint granularity (S / 64 atomics per row) instead of one atomic per (row, wave of points) in the expand
// pass (2344 per row, serialised on 600 addresses: it was most of that kernel's time)
__global__ __launch_bounds__(256) void em_rowbits_kernel(const float* __restrict__ sig, int ld_sig, const uint32_t* __restrict__ src,
                                                         int n, int S, float thr, const int32_t* __restrict__ npts,
                                                         uint32_t* __restrict__ bits, int W, int32_t* __restrict__ count) {
    const int sp = blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y;
    uint32_t w = 0;
    if (sp < S) {
        for (int rr = 0; rr < 32; ++rr) {
            const int r = g * 32 + rr;
            if (r < n && sig[(int64_t)src[r] * ld_sig + sp] > thr) w |= 1u << rr;
        }
        bits[(int64_t)sp * W + g] = w;
    }
    const float np = sp < S ? (float)npts[sp] : 0.f;       // exact in fp32 up to 2^24 points per wave sum
    for (int rr = 0; rr < 32; ++rr) {
        const int r = g * 32 + rr;
        if (r >= n) break;
        const float c = wred_sum(((w >> rr) & 1u) ? np : 0.f);
        if ((threadIdx.x & 63) == 0 && c > 0.f) atomicAdd(&count[r], (int)c);
    }
}

#define EM_WORDS 4               // 32-row words per workgroup: superpoint ids / coordinates are re-read W / 4 times
__global__ __launch_bounds__(256) void em_expand_kernel(const uint32_t* __restrict__ bits, int W, int n, int S,
                                                        const int64_t* __restrict__ superpoints, const float* __restrict__ pts,
                                                        int ld_pts, int64_t N, const float* __restrict__ boxes, float loose,
                                                        uint8_t* __restrict__ out) {
    const int g0 = blockIdx.y * EM_WORDS;
    const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const bool vec = ((N & 3) == 0);
    const int np = p0 < N ? (int)min((int64_t)4, N - p0) : 0;
    uint32_t w[EM_WORDS][4];
    uint32_t any_pt = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t sp = e < np ? superpoints[p0 + e] : -1;
        const bool ok = sp >= 0 && sp < S;
#pragma unroll
        for (int u = 0; u < EM_WORDS; ++u) {
            w[u][e] = (ok && g0 + u < W) ? bits[sp * W + g0 + u] : 0u;
            any_pt |= w[u][e];
        }
    }
    float xyz[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    if (boxes && any_pt) {                                  // coordinates only where some row is on
        for (int e = 0; e < np; ++e)
#pragma unroll
            for (int a = 0; a < 3; ++a) xyz[e][a] = pts[(p0 + e) * ld_pts + a];
    }
#pragma unroll
    for (int u = 0; u < EM_WORDS; ++u) {
        const uint32_t any_w = w[u][0] | w[u][1] | w[u][2] | w[u][3];
        for (int rr = 0; rr < 32; ++rr) {
            const int r = (g0 + u) * 32 + rr;
            if (r >= n) break;                              // uniform per block
            uint8_t res[4] = {0, 0, 0, 0};
            if ((any_w >> rr) & 1u) {
                float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
                if (boxes) {
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const float c = boxes[r * 6 + a], s = boxes[r * 6 + 3 + a] * (1.f + loose);
                        lo[a] = c - s / 2.f;
                        hi[a] = c + s / 2.f;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bool b = (w[u][e] >> rr) & 1u;
                    if (b && boxes) {
#pragma unroll
                        for (int a = 0; a < 3; ++a) b &= (xyz[e][a] >= lo[a]) & (xyz[e][a] <= hi[a]);
                    }
                    res[e] = b;
                }
            }
            if (np == 4 && vec) *(uchar4*)(out + (int64_t)r * N + p0) = make_uchar4(res[0], res[1], res[2], res[3]);
            else for (int e = 0; e < np; ++e) out[(int64_t)r * N + p0 + e] = res[e];
        }
    }
}

// The same expansion for a LIST of rows (the instances that survive the score / point-count thresholds: 100 - 250 of the 600 candidates):
// out[j][p] = row rows[j] of the bit table at point p (and inside its box).  The thresholds only need count[], which em_rowbits_kernel
// makes at superpoint granularity - so the [600, N] byte table (90 MB written, then 20 - 40 MB of it gathered) never has to exist.
// A thread owns 4 consecutive points and 32 listed rows; a row's word is re-read only when it differs from the previous row's.
__global__ __launch_bounds__(256) void em_expand_rows_kernel(const uint32_t* __restrict__ bits, int W, int S, const int32_t* __restrict__ rows, int m,
                                                             const int64_t* __restrict__ superpoints, const float* __restrict__ pts, int ld_pts,
                                                             int64_t N, const float* __restrict__ boxes, float loose, uint8_t* __restrict__ out) {
    const int j0 = blockIdx.y * 32;
    const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const bool vec = ((N & 3) == 0);
    const int np = p0 < N ? (int)min((int64_t)4, N - p0) : 0;
    if (np == 0) return;
    int64_t sp[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t v = e < np ? superpoints[p0 + e] : -1;
        sp[e] = (v >= 0 && v < S) ? v : -1;
    }
    float xyz[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    if (boxes) {
        for (int e = 0; e < np; ++e)
#pragma unroll
            for (int a = 0; a < 3; ++a) xyz[e][a] = pts[(p0 + e) * ld_pts + a];
    }
    int cur_w = -1;
    uint32_t w[4] = {0, 0, 0, 0};
    for (int jj = 0; jj < 32; ++jj) {
        const int j = j0 + jj;
        if (j >= m) break;                                    // uniform per block
        const int r = rows[j];
        const int wi = r >> 5, rr = r & 31;
        if (wi != cur_w) {
            cur_w = wi;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = sp[e] >= 0 ? bits[sp[e] * W + wi] : 0u;
        }
        uint8_t res[4] = {0, 0, 0, 0};
        if (((w[0] | w[1] | w[2] | w[3]) >> rr) & 1u) {
            float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
            if (boxes) {
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float c = boxes[r * 6 + a], sz = boxes[r * 6 + 3 + a] * (1.f + loose);
                    lo[a] = c - sz / 2.f;
                    hi[a] = c + sz / 2.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bool b = (w[e] >> rr) & 1u;
                if (b && boxes) {
#pragma unroll
                    for (int a = 0; a < 3; ++a) b &= (xyz[e][a] >= lo[a]) & (xyz[e][a] <= hi[a]);
                }
                res[e] = b;
            }
        }
        if (np == 4 && vec) *(uchar4*)(out + (int64_t)j * N + p0) = make_uchar4(res[0], res[1], res[2], res[3]);
        else for (int e = 0; e < np; ++e) out[(int64_t)j * N + p0 + e] = res[e];
    }
}

// Bit-packed copy of selected rows of the [n, N] byte masks for the host (baseline3d.py:453-454 hands the evaluator an [n, N] bool
// array: 90 MB per scene over PCIe as bytes, 11 MB as bits): out[i][b] bit j = masks[rows[i]][8 b + j] != 0 (little-endian bit
// order, numpy's `unpackbits(bitorder="little")`); bits past N are 0.  A thread makes one output byte from 8 consecutive mask
// bytes - one 8-byte load where the row base allows it.
__global__ __launch_bounds__(256) void pack_mask_rows_kernel(const uint8_t* __restrict__ masks, int64_t N, const int32_t* __restrict__ rows,
                                                             int n_rows, uint8_t* __restrict__ out, int64_t nb) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (b >= nb || i >= n_rows) return;
    const int64_t r = rows ? rows[i] : i;
    const uint8_t* src = masks + r * N + 8 * b;
    uint32_t v = 0;
    if (8 * b + 8 <= N && ((r * N) & 7) == 0) {
        const uint64_t w = *(const uint64_t*)src;
#pragma unroll
        for (int j = 0; j < 8; ++j) v |= ((w >> (8 * j)) & 0xFFull) ? (1u << j) : 0u;
    } else {
        for (int j = 0; j < 8 && 8 * b + j < N; ++j) v |= src[j] ? (1u << j) : 0u;
    }
    out[(int64_t)i * nb + b] = (uint8_t)v;
}
int launch_pack_mask_rows(const uint8_t* masks, int64_t N, const int32_t* rows, int n_rows, uint8_t* out, int64_t nb, hipStream_t st) {
    if (n_rows <= 0 || N <= 0) return SD3D_OK;
    if (nb != (N + 7) / 8) return sd3d_set_error(SD3D_ERR_ARG, "pack_mask_rows: nb != ceil(N / 8)");
    hipLaunchKernelGGL(pack_mask_rows_kernel, dim3((unsigned)cdiv(nb, 256), (unsigned)n_rows), dim3(256), 0, st, masks, N, rows, n_rows, out, nb);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// sem_q[q] = argmax_c sem[q, classes...]: classes = first n_cls columns (n_cls = C) or an explicit list.
__global__ __launch_bounds__(256) void row_argmax_kernel(const float* __restrict__ x, int ld, int64_t Q, const int32_t* __restrict__ cols,
                                                         int ncols, int64_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < ncols; c += 64) {
        const float v = x[q * ld + (cols ? cols[c] : c)];
        if (v > best || (v == best && c < bi)) { best = v; bi = c; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float ob = __shfl_xor(best, d);
        const int oi = __shfl_xor(bi, d);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) out[q] = bi;
}

// out[p] = table[use_index ? idx[p] : 0]      (:504-507)
__global__ void gather_i64_kernel(const int64_t* __restrict__ table, const int64_t* __restrict__ idx, int64_t N, int use_index,
                                  int64_t* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p < N) out[p] = table[use_index ? idx[p] : 0];
}

// Panoptic paint (:532-543).  rows[0..n) = mask rows in DESCENDING score order; the reference sorts
// ascending and takes max(inst_id * mask), i.e. the best-scoring instance covering the point wins.
//   inst[p] = n_stuff + (n - 1 - first_hit)  or 0 ;  sem[p] = label[first_hit] + n_stuff or ... (see finalize)
__global__ void pan_assign_kernel(const uint8_t* __restrict__ masks, int64_t N, const int32_t* __restrict__ rows, int n, int n_stuff,
                                  int32_t* __restrict__ inst, int32_t* __restrict__ hist) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= N) return;
    int id = 0;
    for (int r = 0; r < n; ++r) {
        if (masks[(int64_t)rows[r] * N + p]) { id = n_stuff + (n - 1 - r); break; }
    }
    inst[p] = id;
    if (id) atomicAdd(&hist[id], 1);
}
// (:545-555): drop painted instances with <= npoint_thr points, then compose the two maps
//   hit: asc position a = id - n_stuff -> label = labels_desc[n-1-a]
__global__ void pan_finalize_kernel(const int32_t* __restrict__ inst, const int32_t* __restrict__ hist, int npoint_thr,
                                    const int32_t* __restrict__ labels_desc, int n, int n_stuff, const int64_t* __restrict__ sem_stuff,
                                    int64_t N, int64_t* __restrict__ sem_map, int64_t* __restrict__ inst_map) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= N) return;
    int id = inst[p];
    if (id != 0 && hist[id] <= npoint_thr) id = 0;
    int64_t s = sem_stuff[p];
    int64_t things_sem = 0;
    if (id != 0) {
        things_sem = labels_desc[n - 1 - (id - n_stuff)] + n_stuff;
        s = 0;
    }
    inst_map[p] = s + id;
    sem_map[p] = s + things_sem;
}

// ---------------------------------------------------------------------------------------------
int launch_class_scores(const float* cls, int ld, int64_t Q, int C, float* scores, float* rowmax, hipStream_t st) {
    if (Q <= 0) return SD3D_OK;
    hipLaunchKernelGGL(class_scores_kernel, dim3((unsigned)cdiv(Q, 4)), dim3(256), 0, st, cls, ld, Q, C, scores, rowmax);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
int launch_mask_scores(const float* masks, int ld, int S, const uint32_t* flat_idx, const float* score_in, int n, int C,
                       int normalize, int32_t* labels, int32_t* qidx, float* score_out, hipStream_t st) {
    if (n <= 0) return SD3D_OK;
    hipLaunchKernelGGL(mask_scores_kernel, dim3((unsigned)cdiv(n, 4)), dim3(256), 0, st, masks, ld, S, flat_idx, score_in, n, C,
                       normalize, labels, qidx, score_out);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
int launch_gather_sigmoid(const float* masks, int ld, int S, const int32_t* qidx, const uint32_t* order, int n, float* sig,
                          int ld_out, float* area, hipStream_t st) {
    if (n <= 0) return SD3D_OK;
    hipLaunchKernelGGL(gather_sigmoid_kernel, dim3((unsigned)cdiv(n, 4)), dim3(256), 0, st, masks, ld, S, qidx, order, n, sig, ld_out,
                       area);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
int launch_nms_decay(const float* inter, int ld, const float* area, const int32_t* labels, int n, int gaussian, float sigma,
                     const float* score_in, float* comp_ws, float* score_out, hipStream_t st) {
    if (n <= 0) return SD3D_OK;
    hipLaunchKernelGGL(nms_comp_kernel, dim3((unsigned)cdiv(n, 64)), dim3(64 * NMS_RL), 0, st, inter, ld, area, labels, n, comp_ws);
    hipLaunchKernelGGL(nms_coef_kernel, dim3((unsigned)cdiv(n, 64)), dim3(64 * NMS_RL), 0, st, inter, ld, area, labels, comp_ws, n, gaussian,
                       sigma, score_in, score_out);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
// (+ n ints behind npts: a caller that places `count` there - ops.MaskBits - gets both zeroed by ONE memset launch)
size_t expand_masks_ws_bytes(int n, int ld_sig) { return ((size_t)ld_sig * ((n + 31) / 32 + 1) + (size_t)n) * sizeof(uint32_t) + 256; }
// bits [ld_sig][W] words + npts [ld_sig] ints live in `ws` (expand_masks_ws_bytes): phase 1 of expand_masks, and all of it when the
// caller expands a list of rows later (launch_expand_rows on the same ws)
int launch_mask_rowbits(const float* sig, int ld_sig, const uint32_t* src, int n, const int64_t* superpoints, int64_t N, float sp_thr,
                        int32_t* count, void* ws, size_t ws_bytes, hipStream_t st) {
    if (n <= 0 || N <= 0) return SD3D_OK;
    if (ws_bytes < expand_masks_ws_bytes(n, ld_sig)) return sd3d_set_error(SD3D_ERR_ARG, "expand_masks: workspace too small");
    const int W = (n + 31) / 32, S = ld_sig;
    uint32_t* bits = (uint32_t*)ws;
    int32_t* npts = (int32_t*)(bits + (size_t)S * W);
    if (count == npts + S) {
        (void)hipMemsetAsync(npts, 0, (size_t)(S + n) * sizeof(int32_t), st);
    } else {
        (void)hipMemsetAsync(count, 0, (size_t)n * sizeof(int32_t), st);
        (void)hipMemsetAsync(npts, 0, (size_t)S * sizeof(int32_t), st);
    }
    hipLaunchKernelGGL(em_hist_kernel, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, st, superpoints, N, S, npts);
    hipLaunchKernelGGL(em_rowbits_kernel, dim3((unsigned)cdiv(S, 256), (unsigned)W), dim3(256), 0, st, sig, ld_sig, src, n, S, sp_thr,
                       npts, bits, W, count);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
int launch_expand_rows(const void* ws, int n, int ld_sig, const int32_t* rows, int m, const int64_t* superpoints, const float* pts, int ld_pts,
                       int64_t N, const float* boxes, float loose, uint8_t* out, hipStream_t st) {
    if (m <= 0 || N <= 0) return SD3D_OK;
    if (!ws || !rows || !out || n <= 0) return sd3d_set_error(SD3D_ERR_ARG, "expand_rows: null pointer");
    const int W = (n + 31) / 32;
    hipLaunchKernelGGL(em_expand_rows_kernel, dim3((unsigned)cdiv(N, 1024), (unsigned)cdiv(m, 32)), dim3(256), 0, st, (const uint32_t*)ws, W, ld_sig,
                       rows, m, superpoints, pts, ld_pts, N, boxes, loose, out);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
int launch_expand_masks(const float* sig, int ld_sig, const uint32_t* src, int n, const int64_t* superpoints, const float* pts,
                        int ld_pts, int64_t N, float sp_thr, const float* boxes, float loose, uint8_t* out, int32_t* count,
                        void* ws, size_t ws_bytes, hipStream_t st) {
    if (n <= 0 || N <= 0) return SD3D_OK;
    const int rc = launch_mask_rowbits(sig, ld_sig, src, n, superpoints, N, sp_thr, count, ws, ws_bytes, st);
    if (rc) return rc;
    const int W = (n + 31) / 32, S = ld_sig;
    hipLaunchKernelGGL(em_expand_kernel, dim3((unsigned)cdiv(N, 1024), (unsigned)cdiv(W, EM_WORDS)), dim3(256), 0, st, (const uint32_t*)ws, W, n, S,
                       superpoints, pts, ld_pts, N, boxes, loose, out);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
int launch_row_argmax(const float* x, int ld, int64_t Q, const int32_t* cols, int ncols, int64_t* out, hipStream_t st) {
    if (Q <= 0) return SD3D_OK;
    hipLaunchKernelGGL(row_argmax_kernel, dim3((unsigned)cdiv(Q, 4)), dim3(256), 0, st, x, ld, Q, cols, ncols, out);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
int launch_gather_i64(const int64_t* table, const int64_t* idx, int64_t N, int use_index, int64_t* out, hipStream_t st) {
    if (N <= 0) return SD3D_OK;
    hipLaunchKernelGGL(gather_i64_kernel, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, st, table, idx, N, use_index, out);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
int launch_panoptic(const uint8_t* masks, int64_t N, const int32_t* rows, const int32_t* labels_desc, int n, int n_stuff,
                    int npoint_thr, const int64_t* sem_stuff, int32_t* inst_ws, int32_t* hist_ws, int64_t* sem_map,
                    int64_t* inst_map, hipStream_t st) {
    if (N <= 0) return SD3D_OK;
    (void)hipMemsetAsync(hist_ws, 0, (size_t)(n + n_stuff + 1) * sizeof(int32_t), st);
    const unsigned nb = (unsigned)cdiv(N, 256);
    hipLaunchKernelGGL(pan_assign_kernel, dim3(nb), dim3(256), 0, st, masks, N, rows, n, n_stuff, inst_ws, hist_ws);
    hipLaunchKernelGGL(pan_finalize_kernel, dim3(nb), dim3(256), 0, st, inst_ws, hist_ws, npoint_thr, labels_desc, n, n_stuff,
                       sem_stuff, N, sem_map, inst_map);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// ---------------------------------------------------------------------------------------------
// GT instance centres / sizes (get_extra_instance_data, baseline3d.py:289-305): per instance the
// min / max / mean of the points selected by its boolean mask.  One workgroup per instance.
//   mode 0 "mean": centre = mean ; mode 1 "median": centre = (max + min) / 2 ; size = max - min
// Instances without points keep centre = size = 0.
// ---------------------------------------------------------------------------------------------
#define IB_CHUNKS 64
__global__ __launch_bounds__(256) void instance_boxes_partial(const float* __restrict__ pts, int ld, int64_t N,
                                                              const uint8_t* __restrict__ masks, int64_t mask_stride,
                                                              float* __restrict__ part /*[n_inst][IB_CHUNKS][10]*/) {
    __shared__ float sm[4][10];
    const int inst = blockIdx.y;
    const uint8_t* m = masks + (int64_t)inst * mask_stride;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, su[3] = {0.f, 0.f, 0.f}, cnt = 0.f;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < N; p += (int64_t)IB_CHUNKS * 256) {
        if (m[p]) {
            cnt += 1.f;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float x = pts[p * ld + a];
                lo[a] = fminf(lo[a], x); hi[a] = fmaxf(hi[a], x); su[a] += x;
            }
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = lo[a], h = hi[a];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { l = fminf(l, __shfl_xor(l, d)); h = fmaxf(h, __shfl_xor(h, d)); }
        const float s = wred_sum(su[a]);
        if (lane == 0) { sm[w][a] = l; sm[w][3 + a] = h; sm[w][6 + a] = s; }
    }
    const float c = wred_sum(cnt);
    if (lane == 0) sm[w][9] = c;
    __syncthreads();
    if (threadIdx.x < 10) {
        const int a = threadIdx.x;
        float r = sm[0][a];
        for (int ww = 1; ww < 4; ++ww) r = a < 3 ? fminf(r, sm[ww][a]) : (a < 6 ? fmaxf(r, sm[ww][a]) : r + sm[ww][a]);
        part[((int64_t)inst * IB_CHUNKS + blockIdx.x) * 10 + a] = r;
    }
}
__global__ void instance_boxes_final(const float* __restrict__ part, int n_inst, int mode, float* __restrict__ centers,
                                     float* __restrict__ sizes) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_inst * 3) return;
    const int inst = t / 3, a = t - inst * 3;
    const float* pp = part + (int64_t)inst * IB_CHUNKS * 10;
    float l = INFINITY, h = -INFINITY, s = 0.f, n = 0.f;
    for (int c = 0; c < IB_CHUNKS; ++c) {
        l = fminf(l, pp[c * 10 + a]); h = fmaxf(h, pp[c * 10 + 3 + a]); s += pp[c * 10 + 6 + a]; n += pp[c * 10 + 9];
    }
    float ctr = 0.f, sz = 0.f;
    if (n > 0.f) { ctr = mode == 0 ? s / n : (h + l) / 2.f; sz = h - l; }
    centers[t] = ctr;
    sizes[t] = sz;
}
int launch_instance_boxes(const float* pts, int ld, int64_t N, const uint8_t* masks, int64_t mask_stride, int n_inst, int mode,
                          float* centers, float* sizes, void* ws, size_t ws_bytes, hipStream_t st) {
    if (n_inst <= 0) return SD3D_OK;
    if (ws_bytes < (size_t)n_inst * IB_CHUNKS * 10 * sizeof(float)) return sd3d_set_error(SD3D_ERR_WS, "instance_boxes workspace");
    hipLaunchKernelGGL(instance_boxes_partial, dim3(IB_CHUNKS, (unsigned)n_inst), dim3(256), 0, st, pts, ld, N, masks, mask_stride,
                       (float*)ws);
    hipLaunchKernelGGL(instance_boxes_final, dim3((unsigned)cdiv(n_inst * 3, 64)), dim3(64), 0, st, (const float*)ws, n_inst, mode,
                       centers, sizes);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// ---------------------------------------------------------------------------------------------
// ScanNet AP association (evaluation/utils_instance_seg_3d_eval.py:340-371): the reference
// counts |pred_mask & (gt_ids == id)| with one numpy pass over N per (prediction, ground-truth) pair.
// Here one pass over the [n, N] byte masks builds, per prediction row, the histogram of the ground-truth
// column of its points (column = instance index, or the "void" column): counts[p][c].
// Workgroup = one row x one chunk of points, histogram in LDS (integer ds_add), non-zero bins flushed with
// global atomics.  Masks are sparse, so gt_index is only read where a mask byte is set.
// ---------------------------------------------------------------------------------------------
#define MO_CHUNK 16384
__global__ __launch_bounds__(256) void mask_overlaps_kernel(const uint8_t* __restrict__ masks, int64_t mask_stride, const int32_t* __restrict__ gt_index,
                                                            int64_t N, int n_cols, int32_t* __restrict__ counts) {
    extern __shared__ int32_t mo_hist[];
    const int row = blockIdx.y;
    const int64_t p_begin = (int64_t)blockIdx.x * MO_CHUNK;
    const int64_t p_end = min(N, p_begin + MO_CHUNK);
    for (int c = threadIdx.x; c < n_cols; c += 256) mo_hist[c] = 0;
    __syncthreads();
    const uint8_t* m = masks + (int64_t)row * mask_stride;
    const bool vec = ((mask_stride & 15) == 0) && ((((uintptr_t)masks) & 15) == 0) && (p_end - p_begin == MO_CHUNK);
    auto tally = [&](const uint32_t (&w)[4], int64_t p) {
        if ((w[0] | w[1] | w[2] | w[3]) == 0u) return;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if ((w[e >> 2] >> (8 * (e & 3))) & 0xffu) {
                const int g = gt_index[p + e];
                if (g >= 0 && g < n_cols) atomicAdd(&mo_hist[g], 1);
            }
        }
    };
    if (vec) {
        // a full chunk: 4 x 16 bytes per thread, all four loads in flight before the first is looked at
        uint4 v[MO_CHUNK / (256 * 16)];
#pragma unroll
        for (int u = 0; u < MO_CHUNK / (256 * 16); ++u) v[u] = *(const uint4*)(m + p_begin + (int64_t)(u * 256 + threadIdx.x) * 16);
#pragma unroll
        for (int u = 0; u < MO_CHUNK / (256 * 16); ++u) {
            const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            tally(w, p_begin + (int64_t)(u * 256 + threadIdx.x) * 16);
        }
    } else {
        for (int64_t p = p_begin + (int64_t)threadIdx.x * 16; p < p_end; p += 256 * 16) {
            uint32_t w[4] = {0, 0, 0, 0};
            const int np = (int)min((int64_t)16, p_end - p);
            for (int e = 0; e < np; ++e) w[e >> 2] |= (uint32_t)m[p + e] << (8 * (e & 3));
            tally(w, p);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < n_cols; c += 256) {
        const int v = mo_hist[c];
        if (v) atomicAdd(&counts[(int64_t)row * n_cols + c], v);
    }
}

int launch_mask_overlaps(const uint8_t* masks, int64_t mask_stride, int n, const int32_t* gt_index, int64_t N, int n_cols, int32_t* counts,
                         hipStream_t st) {
    if (n <= 0 || n_cols <= 0) return SD3D_OK;
    if (n_cols > 8192) return sd3d_set_error(SD3D_ERR_ARG, "mask_overlaps: at most 8192 ground-truth columns");
    if (hipMemsetAsync(counts, 0, (size_t)n * n_cols * sizeof(int32_t), st) != hipSuccess)
        return sd3d_set_error(SD3D_ERR_LAUNCH, "mask_overlaps: memset failed");
    if (N <= 0) return SD3D_OK;
    hipLaunchKernelGGL(mask_overlaps_kernel, dim3((unsigned)cdiv(N, MO_CHUNK), (unsigned)n), dim3(256), (size_t)n_cols * sizeof(int32_t), st,
                       masks, mask_stride, gt_index, N, n_cols, counts);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Index glue of predict_by_feat_instance (baseline3d.py:434-476, mask_matrix_nms :71-139) as three launches instead of sixteen
// (`x.long()` + `y[x]` pairs, each a 4-6 us ATen launch in a phase that is a chain of ~50 dependent launches): pure data movement.
// ---------------------------------------------------------------------------------------------------------------------------
// ---- top-k of a score vector: the first k entries of the stable descending sort, without the sort ------------------------------------
// `predict_by_feat_instance` keeps the topk_insts = 600 best of Q x C = 39 600 (query, class) scores (:434).  Sorting all of them was
// four radix passes = eight dependent launches (60 us) for an answer of 600 numbers.  ONE workgroup: an 8-bit radix SELECT over the
// order-preserving keys (four histogram passes over the L2-resident scores) finds the key of the k-th entry and how many of its ties
// belong to the answer (the ones with the lowest indices - what a stable sort keeps), the k entries are collected and ranked by
// (key, index) in LDS.  Bit for bit the index vector `sort_pairs(keys_from_f32(x, descending))[:k]` returns.
#define TK_THREADS 1024
#define TK_MAX_K 1024
#define TK_NPT 40                                                   // scores per thread, held in registers: n <= 40 960
__device__ __forceinline__ uint32_t tk_key(float v) {              // f32_to_sortkey(desc = 1): ascending keys = descending scores
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~u;
}
__global__ __launch_bounds__(TK_THREADS) void topk_desc_kernel(const float* __restrict__ x, int n, int k, uint32_t* __restrict__ out) {
    __shared__ int hist[256];
    __shared__ int s_bucket, s_remaining, s_slot;
    __shared__ int tie[TK_NPT * (TK_THREADS / 64)];               // ties per (row of 1024 scores, wave), then their exclusive scan
    __shared__ int wsum[TK_THREADS / 64];
    __shared__ uint64_t sel[TK_MAX_K + 8];                        // (key << 32 | index) of the survivors
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // all scores of the thread requested at once (score j of thread t = x[1024 j + t]: coalesced), then everything runs on registers
    uint32_t uc[TK_NPT];
#pragma unroll
    for (int j = 0; j < TK_NPT; ++j) {
        const int i = j * TK_THREADS + tid;
        uc[j] = tk_key(x[i < n ? i : 0]);
    }
    uint32_t prefix = 0;
    int remaining = k;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const uint32_t hi_mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        // (scores are sigmoid products: their top byte - sign and exponent - takes a handful of values, and 39 600 LDS atomics on a
        //  handful of addresses serialise; a thread adds up the run of equal digits it sees and flushes when the digit changes)
        int last = -1, run = 0;
#pragma unroll
        for (int j = 0; j < TK_NPT; ++j) {
            const uint32_t u = uc[j];
            if (j * TK_THREADS + tid < n && (u & hi_mask) == prefix) {
                const int dgt = (int)((u >> shift) & 255);
                if (dgt == last) ++run;
                else {
                    if (run) atomicAdd(&hist[last], run);
                    last = dgt;
                    run = 1;
                }
            }
        }
        if (run) atomicAdd(&hist[last], run);
        __syncthreads();
        if (tid < 64) {                                             // one wave finds the bucket that holds the remaining-th smallest key
            int c[4], sum = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { c[q] = hist[tid * 4 + q]; sum += c[q]; }
            int inc = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (tid >= d) inc += t; }
            int before = inc - sum;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (before < remaining && remaining <= before + c[q]) { s_bucket = tid * 4 + q; s_remaining = remaining - before; }
                before += c[q];
            }
        }
        __syncthreads();
        prefix |= (uint32_t)s_bucket << shift;
        remaining = s_remaining;
        __syncthreads();
    }
    // prefix = key of the k-th entry; `remaining` of the entries with exactly that key belong to the answer: the ones with the lowest
    // indices.  Index order = (row j, wave, lane): ties per (row, wave) by ballot, an exclusive scan over the 640 counters.
#pragma unroll
    for (int j = 0; j < TK_NPT; ++j) {
        const uint64_t bal = __ballot(j * TK_THREADS + tid < n && uc[j] == prefix);
        if (lane == 0) tie[j * (TK_THREADS / 64) + wv] = __popcll(bal);
    }
    if (tid == 0) s_slot = 0;
    __syncthreads();
    {
        const int v = tid < TK_NPT * (TK_THREADS / 64) ? tie[tid] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wv; ++w) base += wsum[w];
        if (tid < TK_NPT * (TK_THREADS / 64)) tie[tid] = base + inc - v;
        __syncthreads();
    }
    const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < TK_NPT; ++j) {
        const bool valid = j * TK_THREADS + tid < n;
        const bool is_tie = valid && uc[j] == prefix;
        const uint64_t bal = __ballot(is_tie);
        bool take = valid && uc[j] < prefix;
        if (is_tie) take = tie[j * (TK_THREADS / 64) + wv] + __popcll(bal & lt) < remaining;
        if (take) {
            const int slot = atomicAdd(&s_slot, 1);
            sel[slot] = ((uint64_t)uc[j] << 32) | (uint32_t)(j * TK_THREADS + tid);
        }
    }
    if (tid < 8) sel[k + tid] = ~0ull;                             // (the rank loop below walks the list eight at a time)
    __syncthreads();
    // rank by (key, index): the order of the stable sort.  One 64-bit compare per pair, eight broadcast reads in flight.
    for (int e = tid; e < k; e += TK_THREADS) {
        const uint64_t me = sel[e];
        int rank = 0;
        for (int j0 = 0; j0 < k; j0 += 8) {
            uint64_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = sel[j0 + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) rank += v[u] < me;
        }
        out[rank] = (uint32_t)me;
    }
}
int launch_topk_desc(const float* x, int64_t n, int k, uint32_t* out, hipStream_t st) {
    if (k <= 0) return SD3D_OK;
    if (n < k || k > TK_MAX_K || n > TK_NPT * TK_THREADS) return sd3d_set_error(SD3D_ERR_ARG, "topk_desc: 1 <= k <= 1024, k <= n <= 40960");
    hipLaunchKernelGGL(topk_desc_kernel, dim3(1), dim3(TK_THREADS), 0, st, x, (int)n, k, out);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

__global__ void take_f32_kernel(const float* __restrict__ src, const uint32_t* __restrict__ idx, int n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}
int launch_take_f32(const float* src, const uint32_t* idx, int n, float* out, hipStream_t st) {
    if (n <= 0) return SD3D_OK;
    hipLaunchKernelGGL(take_f32_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, src, idx, n, out);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// labels1 = labels[order], scores1 = scores[order]            (mask_matrix_nms first sort, :71-76)
__global__ void take_pair_kernel(const uint32_t* __restrict__ order, const int32_t* __restrict__ labels, const float* __restrict__ scores,
                                 int n, int32_t* __restrict__ labels_out, float* __restrict__ scores_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const uint32_t o = order[i]; labels_out[i] = labels[o]; scores_out[i] = scores[o]; }
}
int launch_take_pair(const uint32_t* order, const int32_t* labels, const float* scores, int n, int32_t* labels_out, float* scores_out,
                     hipStream_t st) {
    if (n <= 0) return SD3D_OK;
    hipLaunchKernelGGL(take_pair_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, order, labels, scores, n, labels_out, scores_out);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// the final sort's selections (:133-139) and the kept queries' boxes (baseline3d.py:447-452):
//   final_scores = scores2[order2], final_labels = labels1[order2], record = order1[order2] (int64), boxes = [centers | sizes][qidx[record]]
__global__ void nms_finish_kernel(const uint32_t* __restrict__ order2, const float* __restrict__ scores2, const int32_t* __restrict__ labels1,
                                  const uint32_t* __restrict__ order1, const int32_t* __restrict__ qidx, const float* __restrict__ centers,
                                  const float* __restrict__ sizes, int n, float* __restrict__ final_scores, int32_t* __restrict__ final_labels,
                                  int64_t* __restrict__ record, float* __restrict__ boxes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t o2 = order2[i];
    final_scores[i] = scores2[o2];
    final_labels[i] = labels1[o2];
    const uint32_t rec = order1[o2];
    record[i] = (int64_t)rec;
    if (boxes) {
        const int64_t q = qidx[rec];
#pragma unroll
        for (int a = 0; a < 3; ++a) { boxes[i * 6 + a] = centers[q * 3 + a]; boxes[i * 6 + 3 + a] = sizes[q * 3 + a]; }
    }
}
int launch_nms_finish(const uint32_t* order2, const float* scores2, const int32_t* labels1, const uint32_t* order1, const int32_t* qidx,
                      const float* centers, const float* sizes, int n, float* final_scores, int32_t* final_labels, int64_t* record, float* boxes,
                      hipStream_t st) {
    if (n <= 0) return SD3D_OK;
    if (boxes && (!centers || !sizes)) return sd3d_set_error(SD3D_ERR_ARG, "nms_finish: boxes need centers and sizes");
    hipLaunchKernelGGL(nms_finish_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, order2, scores2, labels1, order1, qidx, centers, sizes, n,
                       final_scores, final_labels, record, boxes);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// the kept instances' labels (int64, as the reference returns them), scores and boxes in one launch (baseline3d.py:470-476)
__global__ void take_instances_kernel(const int32_t* __restrict__ keep, int m, const int32_t* __restrict__ labels, const float* __restrict__ scores,
                                      const float* __restrict__ boxes, int64_t* __restrict__ labels_out, float* __restrict__ scores_out,
                                      float* __restrict__ boxes_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int r = keep[i];
    labels_out[i] = (int64_t)labels[r];
    scores_out[i] = scores[r];
    if (boxes_out) {
#pragma unroll
        for (int a = 0; a < 6; ++a) boxes_out[i * 6 + a] = boxes[r * 6 + a];
    }
}
int launch_take_instances(const int32_t* keep, int m, const int32_t* labels, const float* scores, const float* boxes, int64_t* labels_out,
                          float* scores_out, float* boxes_out, hipStream_t st) {
    if (m <= 0) return SD3D_OK;
    if (boxes_out && !boxes) return sd3d_set_error(SD3D_ERR_ARG, "take_instances: boxes_out without boxes");
    hipLaunchKernelGGL(take_instances_kernel, dim3((unsigned)cdiv(m, 256)), dim3(256), 0, st, keep, m, labels, scores, boxes, labels_out, scores_out,
                       boxes_out);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

// The data-dependent selections of predict_by_feat_instance (baseline3d.py:470-476) for the two score thresholds (instances, panoptic)
// on the device: one workgroup, k <= a few thousand candidates.  The host used to read the k scores and point counts, build the row lists
// with numpy and copy them back - a device -> host -> device round trip with the GPU idle; now it reads four counts.
//   keep_t = rows with score > thr_t and count > npoint_thr (ascending);  union_ = rows in either;  keep_u / pkeep_u = where keep_0 / keep_1
//   sit in union_;  score_mask = score > thr_0 (bytes);  npoint_mask = (count > npoint_thr) over the rows with score > thr_0 (bytes);
//   counts = {|keep_0|, |keep_1|, |union|, |score > thr_0|}
__global__ __launch_bounds__(1024) void select_instances_kernel(const float* __restrict__ scores, const int32_t* __restrict__ count, int k, float thr0,
                                                                float thr1, int npoint_thr, int32_t* __restrict__ keep, int32_t* __restrict__ pkeep,
                                                                int32_t* __restrict__ union_, int32_t* __restrict__ keep_u, int32_t* __restrict__ pkeep_u,
                                                                uint8_t* __restrict__ score_mask, uint8_t* __restrict__ npoint_mask,
                                                                int32_t* __restrict__ counts) {
    __shared__ int sc[4][1024];
    __shared__ int carry[4];
    const int t = threadIdx.x;
    if (t < 4) carry[t] = 0;
    __syncthreads();
    for (int base = 0; base < k; base += 1024) {
        const int i = base + t;
        int f[4] = {0, 0, 0, 0};                               // keep_0, keep_1, union, score_mask_0
        bool np_ok = false;
        if (i < k) {
            const float s = scores[i];
            np_ok = count[i] > npoint_thr;
            f[0] = (s > thr0) && np_ok;
            f[1] = (s > thr1) && np_ok;
            f[2] = f[0] | f[1];
            f[3] = s > thr0;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) sc[a][t] = f[a];
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {                   // inclusive scans of the four flag rows
            int v[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) v[a] = t >= d ? sc[a][t - d] : 0;
            __syncthreads();
#pragma unroll
            for (int a = 0; a < 4; ++a) sc[a][t] += v[a];
            __syncthreads();
        }
        int pos[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) pos[a] = carry[a] + sc[a][t] - f[a];
        if (i < k) {
            if (f[0]) { keep[pos[0]] = i; keep_u[pos[0]] = pos[2]; }
            if (f[1]) { pkeep[pos[1]] = i; pkeep_u[pos[1]] = pos[2]; }
            if (f[2]) union_[pos[2]] = i;
            score_mask[i] = (uint8_t)f[3];
            if (f[3]) npoint_mask[pos[3]] = (uint8_t)np_ok;
        }
        __syncthreads();
        if (t < 4) carry[t] += sc[t][1023];
        __syncthreads();
    }
    if (t < 4) counts[t] = carry[t];
}
int launch_select_instances(const float* scores, const int32_t* count, int k, float thr0, float thr1, int npoint_thr, int32_t* keep, int32_t* pkeep,
                            int32_t* union_, int32_t* keep_u, int32_t* pkeep_u, uint8_t* score_mask, uint8_t* npoint_mask, int32_t* counts,
                            hipStream_t st) {
    if (k < 0) return sd3d_set_error(SD3D_ERR_ARG, "select_instances: k < 0");
    hipLaunchKernelGGL(select_instances_kernel, dim3(1), dim3(1024), 0, st, scores, count, k, thr0, thr1, npoint_thr, keep, pkeep, union_, keep_u, pkeep_u,
                       score_mask, npoint_mask, counts);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}
