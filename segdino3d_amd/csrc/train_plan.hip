// Training step of a sparse U-Net from TWO C calls (round 6): the forward of every {sparse convolution -> batch-statistics
// BatchNorm (+ residual) -> ReLU} layer, and the whole backward of the same list in reverse - input gradients, weight gradients, BatchNorm
// gradients - where rounds 1-5 ran ~250 torch.autograd nodes with their own ctypes calls, tensor allocations, `torch.cat`s of the skip
// connections and `add_` kernels for every tensor with two consumers (`train_ops.TrainBackend`; ~28 ms of host time per forward for 19 ms
// of kernels, 2 800 launches per step).  What the reference gets from MinkowskiEngine's autograd (`train_engine_3d.py:88-122`:
// `loss.backward()` through `Res16UNetBase.forward`, `minkunet.py:531-601`).
//
// The plan (sd3d_train_layer[]) is recorded once per model by `segdino3d_amd/train_plan.py`; per step the caller supplies the scene's pair
// lists (sd3d_train_table[]) and carves three arenas into buffers (sd3d_buf[]): activations, their gradients, raw convolution outputs.
//   * a skip concatenation is never made: both producers write their column slice of ONE buffer (every kernel here takes a row stride),
//     the consumer reads the whole width, and its input gradient lands in both producers' slices with one launch;
//   * a tensor with several consumers receives its gradients in reverse layer order: the first writer stores, the later ones add through
//     the residual input of the convolution's pass 2 (`out = sum + res`, in place) - which one a layer is, the plan knows (`dx_accum`);
//   * the weight gradient is written in the parameter's own [K, Cin, Cout] layout and the input gradient reads the parameter as it lies
//     (mirrored offsets, SD3D_PAIR_MIRROR_W), so no transposed / flipped copy exists in the backward pass.
// Same kernels, same order per tensor as the autograd-node path: the two are compared at 1e-6 in tests/test_gpu_train_ops.py.
#include "common.h"
#include "../../include/segdino3d_hip.h"

int launch_pair_conv(const float*, int, int, const float*, int, const int32_t*, const int32_t*, int64_t, const int32_t*, const int32_t*, int,
                     int, const int32_t*, const float*, int, int, int, int64_t, const float*, const float*, const float*, int, float*, int,
                     int, float*, size_t, hipStream_t);

#define ST ((hipStream_t)stream)

static int check_layer(const sd3d_train_layer& L, int n_tables, int n_bufs, int n_raw, const char* what) {
    if (L.table < 0 || L.table >= n_tables || L.table_t >= n_tables || L.src < 0 || L.src >= n_bufs || L.dst < 0 || L.dst >= n_bufs ||
        L.res >= n_bufs || L.raw < 0 || L.raw >= n_raw)
        return sd3d_set_error(SD3D_ERR_ARG, what);
    return SD3D_OK;
}

extern "C" int sd3d_unet_train_forward(const sd3d_train_layer* layers, int n_layers, const sd3d_train_table* tables, int n_tables,
                                       const sd3d_buf* act, int n_bufs, const sd3d_buf* raw, int n_raw, float* stats, float* part,
                                       size_t part_bytes, void* ws, size_t ws_bytes, void* stream) {
    for (int i = 0; i < n_layers; ++i) {
        const sd3d_train_layer& L = layers[i];
        int rc = check_layer(L, n_tables, n_bufs, n_raw, "unet_train_forward: table / buffer id out of range");
        if (rc) return rc;
        const sd3d_train_table& T = tables[L.table];
        const sd3d_buf &a = act[L.src], &o = act[L.dst], &y0 = raw[L.raw];
        if (T.K != L.K || o.rows != T.M || y0.rows != T.M) return sd3d_set_error(SD3D_ERR_ARG, "unet_train_forward: table / layer shape mismatch");
        rc = launch_pair_conv(a.ptr + L.src_col, a.ld, L.Cin, nullptr, 0, T.in_idx, T.tile_k, T.p_cap, T.pos, T.rlist, T.rl_stride, T.center,
                              T.direct ? T.out_rows : nullptr, L.wt_fwd, L.K, L.Cin, L.Cout, T.M, nullptr, nullptr, nullptr, 0, y0.ptr, y0.ld, 0,
                              part, part_bytes, ST);
        if (rc) return rc;
        float* mean = stats + L.stats;
        float* var = mean + L.Cout;
        float* rstd = var + L.Cout;
        rc = sd3d_bn_stats_running(y0.ptr, y0.ld, T.M, L.Cout, L.eps, mean, var, rstd, L.running_mean, L.running_var, L.num_batches, L.momentum, ws,
                                   ws_bytes, stream);
        if (rc) return rc;
        const sd3d_buf* r = L.res >= 0 ? &act[L.res] : nullptr;
        rc = sd3d_bn_apply(y0.ptr, y0.ld, mean, rstd, L.gamma, L.beta, r ? r->ptr + L.res_col : nullptr, r ? r->ld : 0, T.M, L.Cout, L.act,
                           o.ptr + L.dst_col, o.ld, stream);
        if (rc) return rc;
    }
    return SD3D_OK;
}

// grad[]: the gradient buffers, same ids / shapes as act[] (the caller has put d(loss)/d(output) into the output's buffer); graw: scratch
// of max(rows x Cout) floats for the gradient of a raw convolution output.
extern "C" int sd3d_unet_train_backward(const sd3d_train_layer* layers, int n_layers, const sd3d_train_table* tables, int n_tables,
                                        const sd3d_buf* act, const sd3d_buf* grad, int n_bufs, const sd3d_buf* raw, int n_raw,
                                        const float* stats, float* graw, size_t graw_floats, float* part, size_t part_bytes, void* ws,
                                        size_t ws_bytes, void* stream) {
    for (int i = n_layers - 1; i >= 0; --i) {
        const sd3d_train_layer& L = layers[i];
        int rc = check_layer(L, n_tables, n_bufs, n_raw, "unet_train_backward: table / buffer id out of range");
        if (rc) return rc;
        const sd3d_train_table& T = tables[L.table];
        const sd3d_buf &a = act[L.src], &o = act[L.dst], &y0 = raw[L.raw], &go = grad[L.dst];
        if ((size_t)T.M * L.Cout > graw_floats) return sd3d_set_error(SD3D_ERR_WS, "unet_train_backward: raw-gradient scratch too small");
        const float* mean = stats + L.stats;
        const float* rstd = mean + 2 * L.Cout;
        const sd3d_buf* gr = L.res >= 0 ? &grad[L.res] : nullptr;
        // BatchNorm (+ residual) + ReLU: d(raw), d(residual) - the first gradient its tensor receives in reverse order -, d(gamma), d(beta)
        rc = sd3d_bn_backward(go.ptr + L.dst_col, go.ld, o.ptr + L.dst_col, o.ld, y0.ptr, y0.ld, mean, rstd, L.gamma, T.M, L.Cout, L.act, graw, L.Cout,
                              gr ? gr->ptr + L.res_col : nullptr, gr ? gr->ld : 0, L.dgamma, L.dbeta, ws, ws_bytes, stream);
        if (rc) return rc;
        if (L.need_dx) {
            // the forward convolution on the transposed rulebook with the parameter as it lies ([K, Cin, Cout] = the transposed matrices)
            if (L.table_t < 0) return sd3d_set_error(SD3D_ERR_ARG, "unet_train_backward: layer needs an input gradient but has no transposed table");
            const sd3d_train_table& Tt = tables[L.table_t];
            const sd3d_buf& ga = grad[L.src];
            if (Tt.K != L.K || ga.rows != Tt.M) return sd3d_set_error(SD3D_ERR_ARG, "unet_train_backward: transposed table / layer shape mismatch");
            float* dst = ga.ptr + L.src_col;
            rc = launch_pair_conv(graw, L.Cout, L.Cout, nullptr, 0, Tt.in_idx, Tt.tile_k, Tt.p_cap, Tt.pos, Tt.rlist, Tt.rl_stride,
                                  L.mirrored ? SD3D_PAIR_MIRROR_W(Tt.center) : Tt.center, Tt.direct ? Tt.out_rows : nullptr, L.kernel, L.K, L.Cout,
                                  L.Cin, Tt.M, nullptr, nullptr, L.dx_accum ? dst : nullptr, ga.ld, dst, ga.ld, 0, part, part_bytes, ST);
            if (rc) return rc;
        }
        // weight gradient in the parameter's layout: the operands (and their index lists) exchanged
        rc = sd3d_pair_wgrad(a.ptr + L.src_col, a.ld, graw, L.Cout, T.out_rows, T.in_idx, T.tile_k, T.p_cap, L.K, L.Cout, L.Cin, L.dkernel, 0, ws,
                             ws_bytes, stream);
        if (rc) return rc;
    }
    return SD3D_OK;
}
