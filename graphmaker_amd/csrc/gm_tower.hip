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
// Row stride of h_cat in the workspace: the width rounded up to 32 floats, so that every row starts on a 128-byte line.
// The mean aggregation gathers 128-byte pieces of [h_X | h_Y] rows straight out of h_cat; with the natural stride of the
// ht = 16 configurations (1744 floats = 54.5 lines) every other row's pieces straddled two lines and the d = 576 launch
// fetched 125 MB across the L2 boundary instead of 79 MB (tools/spmm_traffic.py, profiles/r3_spmm_traffic.txt).
static inline int64_t hcat_stride(int64_t Hc) { return (Hc + 31) & ~(int64_t)31; }

struct TowerWs {
    size_t hcat, agg, mid, tmpX, ht0, ht1, xu8, x01, total;
};
static TowerWs tower_layout(const gm_tower_dims* d, int64_t n) {
    const int64_t Hc = (int64_t)(d->num_layers + 1) * (d->hidden_X + d->hidden_Y) + d->hidden_t;
    const int64_t ldc = hcat_stride(Hc);
    const int64_t ka = d->hidden_X + d->hidden_Y + d->hidden_t;
    TowerWs w;
    size_t off = 0;
    w.hcat = off; off += align_up(sizeof(float) * (size_t)(n * ldc));
    w.agg = off;  off += align_up(sizeof(float) * (size_t)(n * hcat_stride(ka)));   // rows on 128-byte lines, like hcat
    w.mid = off;  off += align_up(sizeof(float) * (size_t)(n * Hc));
    w.tmpX = off; off += align_up(sizeof(float) * (size_t)(n * d->hidden_X));
    w.ht0 = off;  off += align_up(sizeof(float) * (size_t)d->hidden_t);
    w.ht1 = off;  off += align_up(sizeof(float) * (size_t)d->hidden_t);
    const int64_t F = d->in_X / 2;
    w.xu8 = off;  off += align_up((size_t)(n * F));                                   // attributes recovered from a one-hot input
    w.x01 = off;  off += align_up(sizeof(float) * (size_t)(n * ((F + 3) & ~(int64_t)3)));   // 0/1 matrix of the dense first layer
    w.total = off;
    return w;
}

static int check_dims(const gm_tower_dims* d, int64_t n) {
    GM_REQUIRE(d, "null dims");
    GM_REQUIRE(d->in_X % 2 == 0, "in_X = %d: two classes per attribute expected (binary attributes)", d->in_X);
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

namespace {

struct TowerCtx {
    const gm_tower_dims* d;
    const float* const* P;
    int64_t n;
    gm_stream_t stream;
    hipStream_t s;
    float *hcat, *agg, *mid, *tmpX, *ht0, *ht1, *x01;
    uint8_t* xu8;
    int ht, hX, hY, nl;
    int64_t Hc, ldc, ka, lda, offY, offT;   // ldc / lda: row strides of hcat / agg (>= Hc / ka, hcat_stride)
};

int tower_ctx(TowerCtx& c, const gm_tower_dims* d, const float* const* P, int64_t n, void* workspace, size_t workspace_bytes,
              gm_stream_t stream) {
    if (int rc = check_dims(d, n)) return rc;
    GM_REQUIRE(P && workspace, "null pointer");
    const TowerWs L = tower_layout(d, n);
    GM_REQUIRE(workspace_bytes >= L.total, "workspace too small: need %zu bytes, got %zu", L.total, workspace_bytes);
    GM_REQUIRE(aligned(workspace, 256), "workspace must be 256-byte aligned");
    const int nparams = 9 + 8 * d->num_layers + 4;
    for (int i = 0; i < nparams; ++i) GM_REQUIRE(P[i], "parameter %d is null", i);
    char* ws = static_cast<char*>(workspace);
    c.d = d; c.P = P; c.n = n; c.stream = stream; c.s = as_stream(stream);
    c.hcat = reinterpret_cast<float*>(ws + L.hcat);
    c.agg = reinterpret_cast<float*>(ws + L.agg);
    c.mid = reinterpret_cast<float*>(ws + L.mid);
    c.tmpX = reinterpret_cast<float*>(ws + L.tmpX);
    c.ht0 = reinterpret_cast<float*>(ws + L.ht0);
    c.ht1 = reinterpret_cast<float*>(ws + L.ht1);
    c.xu8 = reinterpret_cast<uint8_t*>(ws + L.xu8);
    c.x01 = reinterpret_cast<float*>(ws + L.x01);
    c.ht = d->hidden_t; c.hX = d->hidden_X; c.hY = d->hidden_Y; c.nl = d->num_layers;
    c.Hc = (int64_t)(c.nl + 1) * (c.hX + c.hY) + c.ht;
    c.ldc = hcat_stride(c.Hc);
    c.ka = c.hX + c.hY + c.ht;
    c.lda = hcat_stride(c.ka);
    c.offY = (int64_t)(c.nl + 1) * c.hX;
    c.offT = (int64_t)(c.nl + 1) * (c.hX + c.hY);
    return GM_OK;
}

// input projections for node rows [r0,r1): mlp_in_t, mlp_in_X, emb_Y (model/gnn.py:116-127,153-155) -> h^0 slices of hcat
int tower_input(const TowerCtx& c, const float* t_float, const float* X_one_hot, const uint8_t* x_u8, const float* prep,
                int sparse_X, const int32_t* Y, int64_t r0, int64_t r1) {
    const float* const* P = c.P;
    const int64_t rows = r1 - r0;
    if (rows <= 0) return GM_OK;
    const int64_t in_X = c.d->in_X;
    if (c.ht <= 256) {   // one launch: both layers of mlp_in_t and the two broadcasts of h_t
        GM_TRY(launch_time_embed(c.ht, t_float, P[0], P[1], P[2], P[3], rows, c.hcat + r0 * c.ldc + c.offT, c.ldc,
                                 c.agg + r0 * c.lda + c.hX + c.hY, c.lda, c.s));
    } else {
        GM_TRY(gm_linear(1, c.ht, 1, t_float, 1, P[0], 1, P[1], c.ht0, c.ht, 1, c.stream));
        GM_TRY(gm_linear(1, c.ht, c.ht, c.ht0, c.ht, P[2], c.ht, P[3], c.ht1, c.ht, 1, c.stream));
        GM_TRY(launch_broadcast_row(rows, c.ht, c.ht1, c.hcat + r0 * c.ldc + c.offT, c.ldc, c.s));
        GM_TRY(launch_broadcast_row(rows, c.ht, c.ht1, c.agg + r0 * c.lda + c.hX + c.hY, c.lda, c.s));
    }
    if (X_one_hot || x_u8) {   // both NULL: h_X^0 of these rows is still in the workspace from the previous call with the same X
        // first layer on the attributes themselves (gm_onehot.hip): a one-hot input is reduced to them first
        const int64_t F = in_X / 2;
        GM_REQUIRE(prep, "the tower needs its prepared first layer (gm_tower_prepare)");
        if (!x_u8) {
            GM_TRY(gm_x_from_onehot(rows, F, X_one_hot + r0 * in_X, c.xu8 + r0 * F, c.stream));
            x_u8 = c.xu8;
        }
        const int64_t ldF = (F + 3) & ~(int64_t)3;
        const bool walk = sparse_X && c.hX % 4 == 0 && c.hX <= 2048 && F < 65536;
        GM_TRY(gm_onehot_linear(rows, F, c.hX, x_u8 + r0 * F, prep, walk ? nullptr : c.x01 + r0 * ldF, c.tmpX + r0 * c.hX, c.hX, 1,
                                c.stream));
        GM_TRY(gm_linear(rows, c.hX, c.hX, c.tmpX + r0 * c.hX, c.hX, P[6], c.hX, P[7], c.hcat + r0 * c.ldc, c.ldc, 1, c.stream));
    }
    GM_TRY(launch_gather_rows(rows, c.hY, Y + r0, P[8], c.hY, c.hcat + r0 * c.ldc + c.offY, c.ldc, c.s));
    return GM_OK;
}

// layer l for node rows [r0,r1).  The aggregation reads h^l of ALL nodes: from hcat (src == NULL: single GPU, every
// row is local) or from `src`, an (n, hX+hY) buffer [h_X^l | h_Y^l] that an all-gather assembled.
int tower_layer(const TowerCtx& c, int l, const int32_t* rowptr, const int32_t* colidx, const float* src, int64_t r0, int64_t r1) {
    const int64_t rows = r1 - r0;
    if (rows <= 0) return GM_OK;
    const float* const* Q = c.P + 9 + 8 * l;
    const float* hXl = src ? src : c.hcat + (int64_t)l * c.hX;
    const float* hYl = src ? src + c.hX : c.hcat + c.offY + (int64_t)l * c.hY;
    const int64_t ld = src ? (int64_t)(c.hX + c.hY) : c.ldc;
    float* agg = c.agg + r0 * c.lda;
    if (!c.d->is_mlp) {
        // GNNLayer.forward (:60-61): A @ [h_X | h_Y]; A @ h_Y is its last hY columns
        if (src) {   // gathered buffer: [h_X | h_Y] is one contiguous row, one launch covers both products
            GM_TRY(launch_spmm_mean_rows(rows, c.n, rowptr + r0, colidx, src, ld, c.hX + c.hY, agg, c.lda, c.s));
        } else {
            GM_TRY(launch_spmm_mean_two(rows, rowptr + r0, colidx, hXl, c.hX, hYl, c.hY, ld, agg, c.lda, c.s));   // one launch
        }
    } else {
        // MLPLayer.forward (:387-392): no aggregation, row-local
        GM_TRY(launch_copy2d(rows, c.hX, c.hcat + r0 * c.ldc + (int64_t)l * c.hX, c.ldc, agg, c.lda, c.s));
        GM_TRY(launch_copy2d(rows, c.hY, c.hcat + r0 * c.ldc + c.offY + (int64_t)l * c.hY, c.ldc, agg + c.hX, c.lda, c.s));
    }
    float* hXn = c.hcat + r0 * c.ldc + (int64_t)(l + 1) * c.hX;
    float* hYn = c.hcat + r0 * c.ldc + c.offY + (int64_t)(l + 1) * c.hY;
    GM_TRY(gm_linear(rows, c.hX, c.ka, agg, c.lda, Q[0], c.ka, Q[1], hXn, c.ldc, 1, c.stream));        // update_X (:27-32,67)
    GM_TRY(gm_layernorm(rows, c.hX, hXn, c.ldc, Q[2], Q[3], 1e-5f, c.stream));
    GM_TRY(gm_linear(rows, c.hY, c.hY, agg + c.hX, c.lda, Q[4], c.hY, Q[5], hYn, c.ldc, 1, c.stream));  // update_Y (:33-38,68)
    GM_TRY(gm_layernorm(rows, c.hY, hYn, c.ldc, Q[6], Q[7], 1e-5f, c.stream));
    return GM_OK;
}

// mlp_out (:138-142,165-176) for rows [r0,r1)
int tower_output(const TowerCtx& c, float* out, int64_t r0, int64_t r1) {
    const int64_t rows = r1 - r0;
    if (rows <= 0) return GM_OK;
    const float* const* O = c.P + 9 + 8 * c.nl;
    const int64_t os = c.d->out_size;
    GM_TRY(gm_linear(rows, c.Hc, c.Hc, c.hcat + r0 * c.ldc, c.ldc, O[0], c.Hc, O[1], c.mid + r0 * c.Hc, c.Hc, 1, c.stream));
    GM_TRY(gm_linear(rows, os, c.Hc, c.mid + r0 * c.Hc, c.Hc, O[2], c.Hc, O[3], out + r0 * os, os, 0, c.stream));
    return GM_OK;
}

}  // namespace

extern "C" int64_t gm_tower_prep_floats(const gm_tower_dims* d) {
    if (!d || d->in_X <= 0 || d->in_X % 2 || d->hidden_X <= 0) return -1;
    return gm_onehot_prep_floats(d->in_X / 2, d->hidden_X);
}

extern "C" int gm_tower_prepare(const gm_tower_dims* d, const float* const* P, float* prep, gm_stream_t stream) {
    GM_REQUIRE(d && P && prep && P[4] && P[5], "null pointer");
    GM_REQUIRE(d->in_X > 0 && d->in_X % 2 == 0 && d->hidden_X > 0, "bad tower dims");
    return gm_onehot_prepare(d->in_X / 2, d->hidden_X, P[4], P[5], prep, stream);
}

extern "C" int gm_tower_forward(const gm_tower_dims* d, const float* const* P, int64_t n, const float* t_float,
                                const float* X_one_hot, const uint8_t* x_u8, const float* prep, int sparse_X, const int32_t* Y,
                                const int32_t* rowptr, const int32_t* colidx, float* out, float* hcat_out,
                                int64_t out_row_begin, int64_t out_row_end, void* workspace, size_t workspace_bytes,
                                gm_stream_t stream) {
    TowerCtx c;
    if (int rc = tower_ctx(c, d, P, n, workspace, workspace_bytes, stream)) return rc;
    GM_REQUIRE(t_float && Y && out, "null pointer");
    GM_REQUIRE(d->is_mlp || d->num_layers == 0 || (rowptr && colidx), "GNN tower needs a CSR adjacency");
    GM_REQUIRE(0 <= out_row_begin && out_row_begin <= out_row_end && out_row_end <= n, "bad output row range [%lld,%lld)",
               (long long)out_row_begin, (long long)out_row_end);
    GM_TRY(tower_input(c, t_float, X_one_hot, x_u8, prep, sparse_X, Y, 0, n));
    for (int l = 0; l < c.nl; ++l) GM_TRY(tower_layer(c, l, rowptr, colidx, nullptr, 0, n));
    GM_TRY(tower_output(c, out, out_row_begin, out_row_end));
    if (hcat_out)
        GM_HIP(hipMemcpy2DAsync(hcat_out, sizeof(float) * (size_t)c.Hc, c.hcat, sizeof(float) * (size_t)c.ldc,
                                sizeof(float) * (size_t)c.Hc, (size_t)n, hipMemcpyDeviceToDevice, c.s));
    return GM_OK;
}

extern "C" int gm_tower_stage(const gm_tower_dims* d, const float* const* P, int64_t n, int stage, int64_t row_begin,
                              int64_t row_end, const float* t_float, const float* X_one_hot, const uint8_t* x_u8,
                              const float* prep, int sparse_X, const int32_t* Y, const int32_t* rowptr, const int32_t* colidx,
                              const float* layer_src, float* layer_dst, float* out, void* workspace, size_t workspace_bytes,
                              gm_stream_t stream) {
    TowerCtx c;
    if (int rc = tower_ctx(c, d, P, n, workspace, workspace_bytes, stream)) return rc;
    GM_REQUIRE(0 <= row_begin && row_begin <= row_end && row_end <= n, "bad row range [%lld,%lld)", (long long)row_begin,
               (long long)row_end);
    GM_REQUIRE(stage >= -1 && stage <= c.nl, "stage %d out of range [-1,%d]", stage, c.nl);
    if (stage == -1) {
        GM_REQUIRE(t_float && Y, "null pointer");
        GM_TRY(tower_input(c, t_float, X_one_hot, x_u8, prep, sparse_X, Y, row_begin, row_end));
    } else if (stage < c.nl) {
        GM_REQUIRE(d->is_mlp || (rowptr && colidx), "GNN tower needs a CSR adjacency");
        GM_TRY(tower_layer(c, stage, rowptr, colidx, layer_src, row_begin, row_end));
    } else {
        GM_REQUIRE(out, "null pointer");
        GM_TRY(tower_output(c, out, row_begin, row_end));
    }
    // hand this rank's rows of [h_X^s | h_Y^s] (s = stage+1, the slices just written) to the caller for the all-gather
    if (layer_dst && stage < c.nl && row_end > row_begin) {
        const int sl = stage + 1;
        const int64_t rows = row_end - row_begin, w = c.hX + c.hY;
        GM_TRY(launch_copy2d(rows, c.hX, c.hcat + row_begin * c.ldc + (int64_t)sl * c.hX, c.ldc, layer_dst + row_begin * w, w, c.s));
        GM_TRY(launch_copy2d(rows, c.hY, c.hcat + row_begin * c.ldc + c.offY + (int64_t)sl * c.hY, c.ldc,
                             layer_dst + row_begin * w + c.hX, w, c.s));
    }
    return GM_OK;
}
