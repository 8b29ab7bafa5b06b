// gm_tower.hip -- host-side driver that enqueues one denoiser tower forward:
//   GNNTower.forward (model/gnn.py:146-176) and MLPTower.forward (model/gnn.py:440-463), eval mode (dropout = id).
// The concatenated hidden state h_cat = [h_X^0..h_X^L | h_Y^0..h_Y^L | h_t] (model/gnn.py:166) is the layout every
// kernel reads from and writes into directly (column slices with a leading dimension), so no concat is ever run.
#include <stdarg.h>
#include <stdio.h>

#include "gm_common.h"

namespace gm {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace gm

using namespace gm;

extern "C" int gm_version(void) { return 0x00010000; }
extern "C" const char* gm_last_error(void) { return g_err; }
extern "C" int gm_device_cus(int* cus) {
    GM_REQUIRE(cus, "null pointer");
    int dev = 0;
    GM_HIP(hipGetDevice(&dev));
    GM_HIP(hipDeviceGetAttribute(cus, hipDeviceAttributeMultiprocessorCount, dev));
    return GM_OK;
}

static inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct TowerWs {
    size_t hcat, agg, mid, tmpX, ht0, ht1, total;
};
static TowerWs tower_layout(const gm_tower_dims* d, int64_t n) {
    const int64_t Hc = (int64_t)(d->num_layers + 1) * (d->hidden_X + d->hidden_Y) + d->hidden_t;
    const int64_t ka = d->hidden_X + d->hidden_Y + d->hidden_t;
    TowerWs w;
    size_t off = 0;
    w.hcat = off; off += align_up(sizeof(float) * (size_t)(n * Hc));
    w.agg = off;  off += align_up(sizeof(float) * (size_t)(n * ka));
    w.mid = off;  off += align_up(sizeof(float) * (size_t)(n * Hc));
    w.tmpX = off; off += align_up(sizeof(float) * (size_t)(n * d->hidden_X));
    w.ht0 = off;  off += align_up(sizeof(float) * (size_t)d->hidden_t);
    w.ht1 = off;  off += align_up(sizeof(float) * (size_t)d->hidden_t);
    w.total = off;
    return w;
}

static int check_dims(const gm_tower_dims* d, int64_t n) {
    GM_REQUIRE(d, "null dims");
    GM_REQUIRE(n > 0 && d->in_X > 0 && d->num_classes_Y > 0 && d->hidden_t > 0 && d->hidden_X > 0 && d->hidden_Y > 0 &&
                   d->num_layers >= 0 && d->out_size > 0,
               "bad tower dims");
    return GM_OK;
}

extern "C" int gm_tower_workspace_bytes(const gm_tower_dims* dims, int64_t n, size_t* bytes) {
    if (int rc = check_dims(dims, n)) return rc;
    GM_REQUIRE(bytes, "null pointer");
    *bytes = tower_layout(dims, n).total;
    return GM_OK;
}

#define GM_TRY(call)                 \
    do {                             \
        int _rc = (call);            \
        if (_rc != GM_OK) return _rc; \
    } while (0)

extern "C" int gm_tower_forward(const gm_tower_dims* d, const float* const* P, int64_t n, const float* t_float,
                                const float* X_one_hot, const int32_t* Y, const int32_t* rowptr, const int32_t* colidx,
                                float* out, float* hcat_out, int64_t out_row_begin, int64_t out_row_end, void* workspace,
                                size_t workspace_bytes, gm_stream_t stream) {
    if (int rc = check_dims(d, n)) return rc;
    GM_REQUIRE(P && t_float && X_one_hot && Y && out && workspace, "null pointer");
    GM_REQUIRE(d->is_mlp || (rowptr && (colidx || true)), "GNN tower needs a CSR adjacency");
    GM_REQUIRE(0 <= out_row_begin && out_row_begin <= out_row_end && out_row_end <= n, "bad output row range [%lld,%lld)",
               (long long)out_row_begin, (long long)out_row_end);
    const TowerWs L = tower_layout(d, n);
    GM_REQUIRE(workspace_bytes >= L.total, "workspace too small: need %zu bytes, got %zu", L.total, workspace_bytes);
    GM_REQUIRE(aligned(workspace, 256), "workspace must be 256-byte aligned");
    const int nparams = 9 + 8 * d->num_layers + 4;
    for (int i = 0; i < nparams; ++i) GM_REQUIRE(P[i], "parameter %d is null", i);

    hipStream_t s = as_stream(stream);
    char* ws = static_cast<char*>(workspace);
    float* hcat = reinterpret_cast<float*>(ws + L.hcat);
    float* agg = reinterpret_cast<float*>(ws + L.agg);
    float* mid = reinterpret_cast<float*>(ws + L.mid);
    float* tmpX = reinterpret_cast<float*>(ws + L.tmpX);
    float* ht0 = reinterpret_cast<float*>(ws + L.ht0);
    float* ht1 = reinterpret_cast<float*>(ws + L.ht1);

    const int ht = d->hidden_t, hX = d->hidden_X, hY = d->hidden_Y, nl = d->num_layers;
    const int64_t Hc = (int64_t)(nl + 1) * (hX + hY) + ht;
    const int64_t ka = hX + hY + ht;
    const int64_t offY = (int64_t)(nl + 1) * hX, offT = (int64_t)(nl + 1) * (hX + hY);

    // mlp_in_t (model/gnn.py:116-120,153)
    GM_TRY(gm_linear(1, ht, 1, t_float, 1, P[0], 1, P[1], ht0, ht, 1, stream));
    GM_TRY(gm_linear(1, ht, ht, ht0, ht, P[2], ht, P[3], ht1, ht, 1, stream));
    GM_TRY(launch_broadcast_row(n, ht, ht1, hcat + offT, Hc, s));
    // mlp_in_X (:121-126,154)
    GM_TRY(gm_linear(n, hX, d->in_X, X_one_hot, d->in_X, P[4], d->in_X, P[5], tmpX, hX, 1, stream));
    GM_TRY(gm_linear(n, hX, hX, tmpX, hX, P[6], hX, P[7], hcat, Hc, 1, stream));
    // emb_Y (:127,155)
    GM_TRY(launch_gather_rows(n, hY, Y, P[8], hY, hcat + offY, Hc, s));

    GM_TRY(launch_broadcast_row(n, ht, ht1, agg + hX + hY, ka, s));
    for (int l = 0; l < nl; ++l) {
        const float* const* Q = P + 9 + 8 * l;
        const float* hXl = hcat + (int64_t)l * hX;
        const float* hYl = hcat + offY + (int64_t)l * hY;
        if (!d->is_mlp) {
            // GNNLayer.forward (:60-61): A @ [h_X | h_Y]; A @ h_Y is its last hY columns
            GM_TRY(gm_spmm_mean(n, rowptr, colidx, hXl, Hc, hX, agg, ka, stream));
            GM_TRY(gm_spmm_mean(n, rowptr, colidx, hYl, Hc, hY, agg + hX, ka, stream));
        } else {
            // MLPLayer.forward (:387-392): no aggregation
            GM_TRY(launch_copy2d(n, hX, hXl, Hc, agg, ka, s));
            GM_TRY(launch_copy2d(n, hY, hYl, Hc, agg + hX, ka, s));
        }
        float* hXn = hcat + (int64_t)(l + 1) * hX;
        float* hYn = hcat + offY + (int64_t)(l + 1) * hY;
        GM_TRY(gm_linear(n, hX, ka, agg, ka, Q[0], ka, Q[1], hXn, Hc, 1, stream));       // update_X (:27-32,67)
        GM_TRY(gm_layernorm(n, hX, hXn, Hc, Q[2], Q[3], 1e-5f, stream));
        GM_TRY(gm_linear(n, hY, hY, agg + hX, ka, Q[4], hY, Q[5], hYn, Hc, 1, stream));   // update_Y (:33-38,68)
        GM_TRY(gm_layernorm(n, hY, hYn, Hc, Q[6], Q[7], 1e-5f, stream));
    }
    // mlp_out (:138-142,165-176)
    const float* const* O = P + 9 + 8 * nl;
    const int64_t rb = out_row_begin, rows = out_row_end - out_row_begin;
    if (rows > 0) {
        GM_TRY(gm_linear(rows, Hc, Hc, hcat + rb * Hc, Hc, O[0], Hc, O[1], mid + rb * Hc, Hc, 1, stream));
        GM_TRY(gm_linear(rows, d->out_size, Hc, mid + rb * Hc, Hc, O[2], Hc, O[3], out + rb * d->out_size, d->out_size, 0,
                         stream));
    }
    if (hcat_out) GM_HIP(hipMemcpyAsync(hcat_out, hcat, sizeof(float) * (size_t)(n * Hc), hipMemcpyDeviceToDevice, s));
    return GM_OK;
}
