// Layer-sequence executor: runs a whole sparse U-Net (or any straight-line list of the layer kinds
// below) from ONE C call.  The Python host used to issue every convolution itself - ~110 ctypes
// calls and as many tensor allocations per scene, all under the GIL (~2 ms of the ~6.5 ms a forward costs on
// the host).  The plan (sd3d_layer[]) is built once per model;
// per scene the caller provides the neighbour tables and ONE arena carved into the activation
// buffers (sd3d_buf[]), so this call neither allocates nor synchronises: it only enqueues.
#include "gg_common.h"
#include "../../include/segdino3d_hip.h"

int launch_gather_gemm(const GGParams&, int, void*, size_t, hipStream_t);
int launch_pair_conv(const float*, int, int, const float*, int, const int32_t*, const int32_t*, int64_t, const int32_t*, const int32_t*, int,
                     int, const int32_t*, const float*, int, int, int, int64_t, const float*, const float*, const float*, int, float*, int,
                     int, float*, size_t, hipStream_t);
int launch_scale_shift_act(const float*, int, int, const float*, int, const float*, const float*, int, int64_t, int, const float*, int,
                           float*, int, hipStream_t);

// table_events[t] (optional): a hipEvent_t recorded on ANOTHER stream after table t's lists were built there (fork / join inside a
// scene: the neighbour tables of the deeper levels are built on a side stream while the stem convolves); `stream` waits for it
// before the first layer that reads table t.  NULL entries / a NULL array: the tables are already ordered before this call.
extern "C" int sd3d_run_layers_ev(const sd3d_layer* layers, int n_layers, const sd3d_table* tables, int n_tables, const sd3d_buf* bufs,
                                  int n_bufs, float* part, size_t part_bytes, void* ws, size_t ws_bytes, const void* const* table_events,
                                  void* stream) {
    hipStream_t st = (hipStream_t)stream;
    uint64_t waited[4] = {0, 0, 0, 0};
    if (table_events && n_tables > 256) return sd3d_set_error(SD3D_ERR_ARG, "run_layers: at most 256 tables with events");
    for (int i = 0; i < n_layers; ++i) {
        const sd3d_layer& L = layers[i];
        if (L.src0 < 0 || L.src0 >= n_bufs || L.dst < 0 || L.dst >= n_bufs || L.src1 >= n_bufs || L.res >= n_bufs)
            return sd3d_set_error(SD3D_ERR_ARG, "run_layers: buffer id out of range");
        const sd3d_buf& a = bufs[L.src0];
        const sd3d_buf* b = L.src1 >= 0 ? &bufs[L.src1] : nullptr;
        const sd3d_buf* r = L.res >= 0 ? &bufs[L.res] : nullptr;
        const sd3d_buf& o = bufs[L.dst];
        int rc = SD3D_OK;
        if (L.kind == SD3D_LAYER_PAIR_CONV) {
            if (L.table < 0 || L.table >= n_tables) return sd3d_set_error(SD3D_ERR_ARG, "run_layers: table id out of range");
            const sd3d_table& T = tables[L.table];
            if (T.K != L.K) return sd3d_set_error(SD3D_ERR_ARG, "run_layers: table / weight offset count mismatch");
            if (o.rows != T.M) return sd3d_set_error(SD3D_ERR_ARG, "run_layers: output buffer rows != table rows");
            if (table_events && table_events[L.table] && !(waited[L.table >> 6] >> (L.table & 63) & 1)) {
                if (hipStreamWaitEvent(st, (hipEvent_t)table_events[L.table], 0) != hipSuccess)
                    return sd3d_set_error(SD3D_ERR_LAUNCH, "run_layers: hipStreamWaitEvent failed");
                for (int t = 0; t < n_tables; ++t)               // one wait per EVENT: the tables of a group share theirs
                    if (table_events[t] == table_events[L.table]) waited[t >> 6] |= 1ull << (t & 63);
            }
            rc = launch_pair_conv(a.ptr, a.ld, L.C0, b ? b->ptr : nullptr, b ? b->ld : 0, T.in_idx, T.tile_k, T.p_cap, T.pos, T.rlist,
                                  T.rl_stride, T.center, T.out_idx, L.wt, L.K,
                                  L.Cin, L.Cout, T.M, L.scale, L.shift, r ? r->ptr : nullptr, r ? r->ld : 0, o.ptr, o.ld, L.act, part,
                                  part_bytes, st);
        } else if (L.kind == SD3D_LAYER_DENSE) {
            GGParams p;
            p.in0 = a.ptr; p.ld0 = a.ld; p.C0 = L.C0; p.in1 = b ? b->ptr : nullptr; p.ld1 = b ? b->ld : 0; p.nbr = nullptr; p.wt = L.wt;
            p.K = 1; p.Cin = L.Cin; p.Cout = L.Cout; p.M = o.rows; p.scale = L.scale; p.shift = L.shift; p.res = r ? r->ptr : nullptr;
            p.ld_res = r ? r->ld : 0; p.out = o.ptr; p.ld_out = o.ld; p.act = L.act; p.col_groups = 1; p.ksplit = 1; p.ws = nullptr;
            rc = launch_gather_gemm(p, 0, ws, ws_bytes, st);
        } else if (L.kind == SD3D_LAYER_SCALE_SHIFT_ACT) {
            rc = launch_scale_shift_act(a.ptr, a.ld, L.C0, b ? b->ptr : nullptr, b ? b->ld : 0, L.scale, L.shift, L.act, o.rows, L.Cin,
                                        r ? r->ptr : nullptr, r ? r->ld : 0, o.ptr, o.ld, st);
        } else {
            return sd3d_set_error(SD3D_ERR_ARG, "run_layers: unknown layer kind");
        }
        if (rc != SD3D_OK) return rc;
    }
    return SD3D_OK;
}

extern "C" int sd3d_run_layers(const sd3d_layer* layers, int n_layers, const sd3d_table* tables, int n_tables, const sd3d_buf* bufs,
                               int n_bufs, float* part, size_t part_bytes, void* ws, size_t ws_bytes, void* stream) {
    return sd3d_run_layers_ev(layers, n_layers, tables, n_tables, bufs, n_bufs, part, part_bytes, ws, ws_bytes, nullptr, stream);
}
