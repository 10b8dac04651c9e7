// Training criterion on the device: matching costs, SparseMatcher, and the per-layer instance /
// semantic losses WITH their gradients with respect to the predictions, following the reference's
// segdino3d/models/loss/loss_3d.py (costs :63-97, 139-271; SparseMatcher :352-365; layer loss :459-555 / :618-710;
// semantic loss :37-60).  One (decoder layer, scene) per call; everything is a row-per-workgroup streaming kernel
// over [Q, S] logits (Q <= ~3000 queries, S ~ 3000 superpoints): HBM / LDS bound, no matrix cores involved.
// All reductions run in a fixed order, so losses and gradients are reproducible bit for bit.
//
// Layout: ground-truth masks travel as bit rows gt_bits[G][words] (sd3d_pack_mask_bits), the match as a byte matrix
// match[Q][G].  A query row's logits are staged once in LDS together with their sigmoids; a lane reads consecutive
// superpoints (bank-conflict free) and the 32 lanes that share a mask word get it by broadcast.
#include "common.h"
#include "../../include/segdino3d_hip.h"
#include <math.h>

#define LOSS_MAX_S 12288             // superpoints per scene the LDS staging holds (3 rows of floats = 144 KB)
#define SPARSE_INF 1e8f              // loss_3d.py:326

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block-wide sum over 256 threads, same value returned to every thread; `red` holds >= 4 floats
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// ------------------------------------------------------------------ bit rows of boolean masks
__global__ __launch_bounds__(256) void pack_bits_kernel(const uint8_t* __restrict__ m, int64_t ld, int n_cols, uint32_t* __restrict__ bits,
                                                        int words, int32_t* __restrict__ counts) {
    const int r = blockIdx.x;
    __shared__ float red[4];
    int cnt = 0;
    for (int w = threadIdx.x; w < words; w += 256) {
        uint32_t v = 0;
        const int c0 = w * 32;
        for (int b = 0; b < 32 && c0 + b < n_cols; ++b) v |= (m[(int64_t)r * ld + c0 + b] ? 1u : 0u) << b;
        bits[(int64_t)r * words + w] = v;
        cnt += __popc(v);
    }
    const float tot = block_sum((float)cnt, red);              // exact: counts < 2^24
    if (threadIdx.x == 0 && counts) counts[r] = (int)tot;
}

// ------------------------------------------------------------------ matching costs
struct CostParams {
    const float* cls; int ld_cls, n_cls1;
    const float* masks; int ld_masks, Q, S;
    const float* centers; const float* sizes;                 // [Q, 3] or null
    const int64_t* labels; const uint32_t* gt_bits; int words; const int32_t* gt_count; int G;
    const float* gt_centers; int ld_gc; const float* gt_sizes; int ld_gs;
    const uint8_t* query_masks;                               // [G, Q] or null
    float w_cls, w_bce, w_dice, w_ctr, w_size;
    float* cost;
};

__global__ __launch_bounds__(256) void loss_cost_kernel(const CostParams p) {
    extern __shared__ float smem[];
    float* xs = smem;                                          // logits of the row
    float* sg = smem + p.S;                                    // their sigmoids
    __shared__ float red[4];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* xrow = p.masks + (int64_t)q * p.ld_masks;
    float n_sum = 0.f, s_sum = 0.f;
    for (int s = tid; s < p.S; s += 256) {
        const float x = xrow[s], sig = sigmoid_f(x);
        xs[s] = x; sg[s] = sig;
        n_sum += softplus_f(x); s_sum += sig;
    }
    n_sum = block_sum(n_sum, red);
    s_sum = block_sum(s_sum, red);
    // softmax statistics of the class row
    const float* crow = p.cls + (int64_t)q * p.ld_cls;
    float mx = -INFINITY;
    for (int c = tid; c < p.n_cls1; c += 256) mx = fmaxf(mx, crow[c]);
    mx = block_max(mx, red);
    float se = 0.f;
    for (int c = tid; c < p.n_cls1; c += 256) se += expf(crow[c] - mx);
    se = block_sum(se, red);
    for (int g = wv; g < p.G; g += 4) {
        float c;
        if (p.query_masks && !p.query_masks[(int64_t)g * p.Q + q]) {
            c = SPARSE_INF;                                    // the query does not lie in this object (loss_3d.py:358-359)
        } else {
            const uint32_t* bits = p.gt_bits + (int64_t)g * p.words;
            float a = 0.f, b = 0.f;
            for (int s = lane; s < p.S; s += 64) {
                const bool t = (bits[s >> 5] >> (s & 31)) & 1u;
                a += t ? xs[s] : 0.f;
                b += t ? sg[s] : 0.f;
            }
            a = wave_sum(a); b = wave_sum(b);
            const float T = (float)p.gt_count[g];
            c = -p.w_cls * expf(crow[p.labels[g]] - mx) / se;
            // sum_s softplus(-x) t + softplus(x) (1 - t) = sum_s softplus(x) - sum_s x t
            c += p.w_bce * (n_sum - a) / (float)p.S;
            c += p.w_dice * (1.f - (2.f * b + 1.f) / (s_sum + T + 1.f));
            if (p.centers && p.w_ctr != 0.f) {
                float l1 = 0.f;
                for (int d = 0; d < 3; ++d) l1 += fabsf(p.centers[q * 3 + d] - p.gt_centers[(int64_t)g * p.ld_gc + d]);
                c += p.w_ctr * l1;
            }
            if (p.sizes && p.w_size != 0.f) {
                float l1 = 0.f;
                for (int d = 0; d < 3; ++d) l1 += fabsf(p.sizes[q * 3 + d] - p.gt_sizes[(int64_t)g * p.ld_gs + d]);
                c += p.w_size * l1;
            }
        }
        if (lane == 0) p.cost[(int64_t)q * p.G + g] = c;
    }
}

// ------------------------------------------------------------------ SparseMatcher: one wave per object column
// kth = the (topk + 1)-th smallest cost of the column (with multiplicity); match = cost < kth.
__global__ __launch_bounds__(64) void loss_sparse_match_kernel(const float* __restrict__ cost, int Q, int G, int topk,
                                                              uint8_t* __restrict__ match) {
    const int g = blockIdx.x, lane = threadIdx.x;
    float last_v = -INFINITY; int last_i = -1;
    for (int it = 0; it <= topk; ++it) {
        float bv = INFINITY; int bi = 0x7fffffff;
        for (int q = lane; q < Q; q += 64) {
            const float v = cost[(int64_t)q * G + g];
            const bool after = v > last_v || (v == last_v && q > last_i);       // not selected in an earlier round
            if (after && (v < bv || (v == bv && q < bi))) { bv = v; bi = q; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        last_v = bv; last_i = bi;
    }
    for (int q = lane; q < Q; q += 64) match[(int64_t)q * G + g] = cost[(int64_t)q * G + g] < last_v ? 1 : 0;
}

// ------------------------------------------------------------------ instance loss of one layer / scene
struct InstParams {
    const float* cls; int ld_cls, n_cls1;
    const float* masks; int ld_masks, Q, S;
    const float* scores;                                      // [Q] or null
    const float* centers; const float* sizes;                 // [Q, 3] or null
    const int64_t* labels; const uint32_t* gt_bits; int words; const int32_t* gt_count; int G;
    const float* gt_centers; int ld_gc; const float* gt_sizes; int ld_gs;
    const uint8_t* match;                                     // [Q, G]
    const float* class_weight;                                // [n_cls1]
    float c_cls, c_bce, c_dice, c_score, c_ctr, c_size;       // d(total loss) / d(this scene's term)
    float* d_cls; float* d_masks; float* d_scores; float* d_centers; float* d_sizes;
    int32_t* tgt; int32_t* row_n; float* rowsum; float* stats; float* parts;
};
#define RS_COLS 8     // per-row sums: w*nll, bce, dice, score sq. error, kept scores, centre L1, size L1
// stats: [0] = matched pairs, [1] = sum of class weights of the targets

// class target of every query (the LAST matched object wins, as index_put does on the CPU, loss_3d.py:463) + totals
__global__ __launch_bounds__(1024) void loss_targets_kernel(const InstParams p) {
    __shared__ float red_n[16], red_w[16];
    float n = 0.f, w = 0.f;
    for (int q = threadIdx.x; q < p.Q; q += 1024) {
        int t = p.n_cls1 - 1, cnt = 0;
        for (int g = 0; g < p.G; ++g)
            if (p.match[(int64_t)q * p.G + g]) { t = (int)p.labels[g]; ++cnt; }
        p.tgt[q] = t; p.row_n[q] = cnt;
        n += (float)cnt; w += p.class_weight[t];
    }
    // fixed-order tree: lanes, then waves
    n = wave_sum(n); w = wave_sum(w);
    if ((threadIdx.x & 63) == 0) { red_n[threadIdx.x >> 6] = n; red_w[threadIdx.x >> 6] = w; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tn = 0.f, tw = 0.f;
        for (int i = 0; i < 16; ++i) { tn += red_n[i]; tw += red_w[i]; }
        p.stats[0] = tn; p.stats[1] = tw;
    }
}

__global__ __launch_bounds__(256) void loss_rows_kernel(const InstParams p) {
    extern __shared__ float smem[];
    float* xs = smem; float* sg = smem + p.S; float* gr = smem + 2 * p.S;
    __shared__ float red[4];
    const int q = blockIdx.x, tid = threadIdx.x;
    const float n_m = p.stats[0], w_tot = p.stats[1];
    float* rs = p.rowsum + (int64_t)q * RS_COLS;
    // ---- weighted cross entropy of the class row (loss_3d.py:459-467)
    {
        const float* crow = p.cls + (int64_t)q * p.ld_cls;
        const int t = p.tgt[q];
        const float w = p.class_weight[t];
        float mx = -INFINITY;
        for (int c = tid; c < p.n_cls1; c += 256) mx = fmaxf(mx, crow[c]);
        mx = block_max(mx, red);
        float se = 0.f;
        for (int c = tid; c < p.n_cls1; c += 256) se += expf(crow[c] - mx);
        se = block_sum(se, red);
        const float lse = mx + logf(se);
        const float k = p.c_cls * w / w_tot;
        for (int c = tid; c < p.n_cls1; c += 256)
            p.d_cls[(int64_t)q * p.n_cls1 + c] = k * (expf(crow[c] - lse) - (c == t ? 1.f : 0.f));
        if (tid == 0) rs[0] = w * (lse - crow[t]);
    }
    const int cnt = p.row_n[q];
    float* drow = p.d_masks + (int64_t)q * p.S;
    if (cnt == 0) {                                            // unmatched query: no mask / box / score terms
        for (int s = tid; s < p.S; s += 256) drow[s] = 0.f;
        if (tid < 3) { if (p.d_centers) p.d_centers[q * 3 + tid] = 0.f; if (p.d_sizes) p.d_sizes[q * 3 + tid] = 0.f; }
        if (tid == 0) { if (p.d_scores) p.d_scores[q] = 0.f; for (int i = 1; i < RS_COLS; ++i) rs[i] = 0.f; }
        return;
    }
    const float* xrow = p.masks + (int64_t)q * p.ld_masks;
    float s_sum = 0.f, nb = 0.f;
    for (int s = tid; s < p.S; s += 256) {
        const float x = xrow[s], sig = sigmoid_f(x);
        xs[s] = x; sg[s] = sig; gr[s] = 0.f;
        s_sum += sig; nb += sig >= 0.5f ? 1.f : 0.f;
    }
    s_sum = block_sum(s_sum, red);
    nb = block_sum(nb, red);
    float bce_acc = 0.f, dice_acc = 0.f, sq_acc = 0.f, keep_acc = 0.f, ctr_acc = 0.f, size_acc = 0.f;
    float dscore = 0.f, dctr[3] = {0.f, 0.f, 0.f}, dsize[3] = {0.f, 0.f, 0.f};
    for (int g = 0; g < p.G; ++g) {                            // matched objects in ascending order
        if (!p.match[(int64_t)q * p.G + g]) continue;
        const uint32_t* bits = p.gt_bits + (int64_t)g * p.words;
        float bce = 0.f, inter = 0.f, ni = 0.f;
        for (int s = tid; s < p.S; s += 256) {
            const bool t = (bits[s >> 5] >> (s & 31)) & 1u;
            const float x = xs[s];
            bce += softplus_f(x) - (t ? x : 0.f);              // BCE with logits (loss_3d.py:479-480)
            inter += t ? sg[s] : 0.f;
            ni += (t && sg[s] >= 0.5f) ? 1.f : 0.f;
        }
        bce = block_sum(bce, red); inter = block_sum(inter, red); ni = block_sum(ni, red);
        const float T = (float)p.gt_count[g];
        const float den = s_sum + T + 1.f, num = 2.f * inter + 1.f;
        bce_acc += bce / (float)p.S;
        dice_acc += 1.f - num / den;                          // loss_3d.py:120-137
        // d/dx_s: BCE mean over (n_m, S) -> (sig - t) / (n_m S); dice mean over n_m -> -(2 t den - num) / den^2 * sig (1 - sig) / n_m
        const float kb = p.c_bce / (n_m * (float)p.S), kd = p.c_dice / (n_m * den * den);
        for (int s = tid; s < p.S; s += 256) {
            const bool t = (bits[s >> 5] >> (s & 31)) & 1u;
            const float sig = sg[s];
            gr[s] += kb * (sig - (t ? 1.f : 0.f)) - kd * ((t ? 2.f * den : 0.f) - num) * sig * (1.f - sig);
        }
        if (p.centers) {                                       // loss_3d.py:483-486
            for (int d = 0; d < 3; ++d) {
                const float diff = p.centers[q * 3 + d] - p.gt_centers[(int64_t)g * p.ld_gc + d];
                ctr_acc += fabsf(diff); dctr[d] += (diff > 0.f) - (diff < 0.f);
            }
        }
        if (p.sizes) {
            for (int d = 0; d < 3; ++d) {
                const float diff = p.sizes[q * 3 + d] - p.gt_sizes[(int64_t)g * p.ld_gs + d];
                size_acc += fabsf(diff); dsize[d] += (diff > 0.f) - (diff < 0.f);
            }
        }
        if (p.scores) {                                        // objectness: MSE against the IoU where IoU > 0.5 (:495-503)
            const float iou = ni / (T + nb - ni + 1e-6f);
            if (iou > 0.5f) { const float e = p.scores[q] - iou; sq_acc += e * e; keep_acc += 1.f; dscore += 2.f * e; }
        }
    }
    for (int s = tid; s < p.S; s += 256) drow[s] = gr[s];
    if (tid < 3) {
        if (p.d_centers) p.d_centers[q * 3 + tid] = p.c_ctr / n_m * dctr[tid];
        if (p.d_sizes) p.d_sizes[q * 3 + tid] = p.c_size / n_m * dsize[tid];
    }
    if (tid == 0) {
        if (p.d_scores) p.d_scores[q] = dscore;                // scaled by c_score / kept once the total is known
        rs[1] = bce_acc; rs[2] = dice_acc; rs[3] = sq_acc; rs[4] = keep_acc; rs[5] = ctr_acc; rs[6] = size_acc; rs[7] = 0.f;
    }
}

// column sums of rowsum in a fixed order -> parts[0..5] = cls, bce, dice, score, centre, size; [6] = matched, [7] = kept
__global__ __launch_bounds__(1024) void loss_final_kernel(const InstParams p) {
    __shared__ float red[16][RS_COLS];
    __shared__ float tot[RS_COLS];
    float acc[RS_COLS];
    for (int i = 0; i < RS_COLS; ++i) acc[i] = 0.f;
    for (int q = threadIdx.x; q < p.Q; q += 1024)
        for (int i = 0; i < RS_COLS; ++i) acc[i] += p.rowsum[(int64_t)q * RS_COLS + i];
    for (int i = 0; i < RS_COLS; ++i) acc[i] = wave_sum(acc[i]);
    if ((threadIdx.x & 63) == 0) for (int i = 0; i < RS_COLS; ++i) red[threadIdx.x >> 6][i] = acc[i];
    __syncthreads();
    if (threadIdx.x < RS_COLS) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w][threadIdx.x];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    const float n_m = p.stats[0], w_tot = p.stats[1], kept = tot[4];
    if (threadIdx.x == 0) {
        p.parts[0] = tot[0] / w_tot;
        p.parts[1] = tot[1] / n_m;                             // 0 / 0 = nan for a scene without matches, like the reference's mean of nothing
        p.parts[2] = tot[2] / n_m;
        p.parts[3] = kept > 0.f ? tot[3] / kept : 0.f;
        p.parts[4] = p.centers ? tot[5] / n_m : 0.f;
        p.parts[5] = p.sizes ? tot[6] / n_m : 0.f;
        p.parts[6] = n_m; p.parts[7] = kept;
    }
    if (p.d_scores) {
        const float k = kept > 0.f ? p.c_score / kept : 0.f;
        for (int q = threadIdx.x; q < p.Q; q += 1024) p.d_scores[q] *= k;
    }
}

// ------------------------------------------------------------------ semantic loss (loss_3d.py:37-60)
// target of query q = first class r in [0, n] whose mask row r is set at q (argmax of a 0/1 column; 0 if none);
// rows whose target is `ignore_index` do not count.  Logit columns: the first n of sem[q] when ignore_index >= 0.
struct SemParams {
    const float* sem; int ld, Q, n_rows, n_logits, ignore_index;
    const uint8_t* sem_masks;                                 // [n_rows, Q]
    float coef; float* d_sem; int ld_d; int32_t* tgt; float* nll; float* stats; float* loss;
};
__global__ __launch_bounds__(1024) void sem_targets_kernel(const SemParams p) {
    __shared__ float red[16];
    float n = 0.f;
    for (int q = threadIdx.x; q < p.Q; q += 1024) {
        int t = 0;
        for (int r = 0; r < p.n_rows; ++r) if (p.sem_masks[(int64_t)r * p.Q + q]) { t = r; break; }
        p.tgt[q] = t;
        n += t != p.ignore_index ? 1.f : 0.f;
    }
    n = wave_sum(n);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) { float t = 0.f; for (int i = 0; i < 16; ++i) t += red[i]; p.stats[0] = t; }
}
__global__ __launch_bounds__(256) void sem_rows_kernel(const SemParams p) {
    __shared__ float red[4];
    const int q = blockIdx.x, tid = threadIdx.x;
    const float* row = p.sem + (int64_t)q * p.ld;
    float* drow = p.d_sem + (int64_t)q * p.ld_d;
    const int t = p.tgt[q];
    if (t == p.ignore_index) {
        for (int c = tid; c < p.ld_d; c += 256) drow[c] = 0.f;
        if (tid == 0) p.nll[q] = 0.f;
        return;
    }
    float mx = -INFINITY;
    for (int c = tid; c < p.n_logits; c += 256) mx = fmaxf(mx, row[c]);
    mx = block_max(mx, red);
    float se = 0.f;
    for (int c = tid; c < p.n_logits; c += 256) se += expf(row[c] - mx);
    se = block_sum(se, red);
    const float lse = mx + logf(se), k = p.coef / p.stats[0];
    for (int c = tid; c < p.ld_d; c += 256) drow[c] = c < p.n_logits ? k * (expf(row[c] - lse) - (c == t ? 1.f : 0.f)) : 0.f;
    if (tid == 0) p.nll[q] = lse - row[t];
}
__global__ __launch_bounds__(1024) void sem_final_kernel(const SemParams p) {
    __shared__ float red[16];
    float a = 0.f;
    for (int q = threadIdx.x; q < p.Q; q += 1024) a += p.nll[q];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) { float t = 0.f; for (int i = 0; i < 16; ++i) t += red[i]; p.loss[0] = t / p.stats[0]; }
}

// ------------------------------------------------------------------ C ABI
#define ST ((hipStream_t)stream)
extern "C" {

int sd3d_pack_mask_bits(const uint8_t* masks, int64_t ld, int n_rows, int n_cols, uint32_t* bits, int words, int32_t* counts,
                        void* stream) {
    if (n_rows < 0 || n_cols < 0 || words < (n_cols + 31) / 32) return sd3d_set_error(SD3D_ERR_ARG, "pack_mask_bits: bad shape");
    if (n_rows == 0) return SD3D_OK;
    pack_bits_kernel<<<n_rows, 256, 0, ST>>>(masks, ld, n_cols, bits, words, counts);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

int sd3d_match_costs(const float* cls, int ld_cls, int n_cls1, const float* masks, int ld_masks, int Q, int S, const float* centers,
                     const float* sizes, const int64_t* labels, const uint32_t* gt_bits, int words, const int32_t* gt_count, int G,
                     const float* gt_centers, int ld_gc, const float* gt_sizes, int ld_gs, const uint8_t* query_masks,
                     const float* weights5, float* cost, void* stream) {
    if (Q <= 0 || G <= 0) return SD3D_OK;
    if (S <= 0 || S > LOSS_MAX_S || words < (S + 31) / 32) return sd3d_set_error(SD3D_ERR_ARG, "match_costs: bad superpoint count");
    if ((centers && weights5[3] != 0.f && !gt_centers) || (sizes && weights5[4] != 0.f && !gt_sizes))
        return sd3d_set_error(SD3D_ERR_ARG, "match_costs: predicted boxes need ground-truth boxes");
    CostParams p{cls, ld_cls, n_cls1, masks, ld_masks, Q, S, centers, sizes, labels, gt_bits, words, gt_count, G, gt_centers, ld_gc,
                 gt_sizes, ld_gs, query_masks, weights5[0], weights5[1], weights5[2], weights5[3], weights5[4], cost};
    const size_t lds = (size_t)2 * S * sizeof(float);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)loss_cost_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * LOSS_MAX_S * 4); attr = true; }
    loss_cost_kernel<<<Q, 256, lds, ST>>>(p);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

int sd3d_sparse_match(const float* cost, int Q, int G, int topk, uint8_t* match, void* stream) {
    if (Q <= 0 || G <= 0) return SD3D_OK;
    if (topk < 0 || topk + 1 > Q) return sd3d_set_error(SD3D_ERR_ARG, "sparse_match: needs topk + 1 <= number of queries");
    loss_sparse_match_kernel<<<G, 64, 0, ST>>>(cost, Q, G, topk, match);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

size_t sd3d_instance_loss_ws_bytes(int Q) { return align_up((size_t)Q * (RS_COLS + 2) * 4 + 64, 256); }

int sd3d_instance_loss(const float* cls, int ld_cls, int n_cls1, const float* masks, int ld_masks, int Q, int S, const float* scores,
                       const float* centers, const float* sizes, const int64_t* labels, const uint32_t* gt_bits, int words,
                       const int32_t* gt_count, int G, const float* gt_centers, int ld_gc, const float* gt_sizes, int ld_gs,
                       const uint8_t* match, const float* class_weight, const float* coef6, float* d_cls, float* d_masks,
                       float* d_scores, float* d_centers, float* d_sizes, float* parts8, void* ws, size_t ws_bytes, void* stream) {
    if (Q <= 0) return sd3d_set_error(SD3D_ERR_ARG, "instance_loss: no queries");
    if (S <= 0 || S > LOSS_MAX_S || words < (S + 31) / 32) return sd3d_set_error(SD3D_ERR_ARG, "instance_loss: bad superpoint count");
    if (ws_bytes < sd3d_instance_loss_ws_bytes(Q)) return sd3d_set_error(SD3D_ERR_WS, "instance_loss: workspace too small");
    if ((centers && !gt_centers) || (sizes && !gt_sizes)) return sd3d_set_error(SD3D_ERR_ARG, "instance_loss: predicted boxes need ground-truth boxes");
    InstParams p{};
    p.cls = cls; p.ld_cls = ld_cls; p.n_cls1 = n_cls1; p.masks = masks; p.ld_masks = ld_masks; p.Q = Q; p.S = S;
    p.scores = scores; p.centers = centers; p.sizes = sizes; p.labels = labels; p.gt_bits = gt_bits; p.words = words;
    p.gt_count = gt_count; p.G = G; p.gt_centers = gt_centers; p.ld_gc = ld_gc; p.gt_sizes = gt_sizes; p.ld_gs = ld_gs;
    p.match = match; p.class_weight = class_weight;
    p.c_cls = coef6[0]; p.c_bce = coef6[1]; p.c_dice = coef6[2]; p.c_score = coef6[3]; p.c_ctr = coef6[4]; p.c_size = coef6[5];
    p.d_cls = d_cls; p.d_masks = d_masks; p.d_scores = scores ? d_scores : nullptr;
    p.d_centers = centers ? d_centers : nullptr; p.d_sizes = sizes ? d_sizes : nullptr;
    char* w = (char*)ws;
    p.stats = (float*)w; w += 64;
    p.tgt = (int32_t*)w; w += (size_t)Q * 4;
    p.row_n = (int32_t*)w; w += (size_t)Q * 4;
    p.rowsum = (float*)w;
    p.parts = parts8;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)loss_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * LOSS_MAX_S * 4); attr = true; }
    loss_targets_kernel<<<1, 1024, 0, ST>>>(p);
    loss_rows_kernel<<<Q, 256, (size_t)3 * S * sizeof(float), ST>>>(p);
    loss_final_kernel<<<1, 1024, 0, ST>>>(p);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

size_t sd3d_semantic_loss_ws_bytes(int Q) { return align_up((size_t)Q * 8 + 64, 256); }

int sd3d_semantic_loss(const float* sem, int ld, int Q, int n_rows, int n_logits, const uint8_t* sem_masks, int ignore_index, float coef,
                       float* d_sem, int ld_d, float* loss, void* ws, size_t ws_bytes, void* stream) {
    if (Q <= 0 || n_rows <= 0 || n_logits <= 0 || n_logits > ld || ld_d < n_logits) return sd3d_set_error(SD3D_ERR_ARG, "semantic_loss: bad shape");
    if (ws_bytes < sd3d_semantic_loss_ws_bytes(Q)) return sd3d_set_error(SD3D_ERR_WS, "semantic_loss: workspace too small");
    char* w = (char*)ws;
    SemParams p{sem, ld, Q, n_rows, n_logits, ignore_index, sem_masks, coef, d_sem, ld_d, nullptr, nullptr, nullptr, loss};
    p.stats = (float*)w; w += 64;
    p.tgt = (int32_t*)w; w += (size_t)Q * 4;
    p.nll = (float*)w;
    sem_targets_kernel<<<1, 1024, 0, ST>>>(p);
    sem_rows_kernel<<<Q, 256, 0, ST>>>(p);
    sem_final_kernel<<<1, 1024, 0, ST>>>(p);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

}  // extern "C"
