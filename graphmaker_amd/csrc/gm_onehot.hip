// gm_onehot.hip -- the first layer of the denoiser towers, mlp_in_X.0 (model/gnn.py:121-126,154), on one-hot attributes.
//
// X_t_one_hot (n, 2F) has exactly one 1 per attribute, at column 2f + x_vf (model/diffusion.py:232-235), so
//     X W^T + b  =  B + sum_{f : x_vf = 1} D[:, f],     D[o][f] = W[o][2f+1] - W[o][2f],     B[o] = (sum_f W[o][2f]) + b[o]
// (D and B depend on the weights only: gm_onehot_prepare, once per weight version).  Two forms with identical bits, because a
// k-ascending chain fmaf(x_f, D[o][f], acc) leaves acc unchanged where x_f = 0:
//   * dense : an f32 MFMA GEMM with K = F (half the reference's K = 2F) on the 0/1 matrix            -- MFMA-bound
//   * sparse: one wave per node walks the node's set attributes in ascending order and adds rows of D^T -- L2-gather-bound;
//             bag-of-words attributes have ~1% ones (Cora: 18 of 1433), which makes this 5-10x cheaper than the GEMM.
// oracle: gmo_onehot_linear.
#include "gm_common.h"

namespace gm {

static inline int64_t up64(int64_t x) { return (x + 63) & ~(int64_t)63; }
struct OnehotPrep {
    int64_t oB, oDt, oD, ldF, total;
};
static OnehotPrep prep_layout(int64_t F, int64_t H) {
    OnehotPrep p;
    p.ldF = (F + 3) & ~(int64_t)3;
    p.oB = 0;
    p.oDt = up64(H);
    p.oD = p.oDt + up64(F * H);
    p.total = p.oD + up64(H * p.ldF);
    return p;
}

// B[o], row o of D (zero-padded to ldF) and column o of D^T.  Two launches: k_onehot_prepare_D gives every 64 x 64 tile of
// D / D^T its own workgroup (the W tile, 64 rows x 128 floats, is read coalesced into LDS and both layouts leave coalesced);
// k_onehot_prepare_B continues the chain B[o] += W[o][2f] in ascending f (the oracle's order) with one thread per output unit,
// 16 units per workgroup, the W tiles staged through LDS the same way.  (One workgroup per 64 units doing all of it tile after
// tile was 8 workgroups and 0.20 ms per call -- per training step and tower; one thread per row reading its own row 0.35 ms.)
constexpr int OP_T = 64, OP_BR = 16;
__global__ void __launch_bounds__(256)
k_onehot_prepare_D(int64_t F, int64_t H, int64_t ldF, const float* __restrict__ W, float* __restrict__ Dt, float* __restrict__ D) {
    __shared__ float tile[OP_T][2 * OP_T + 1];
    const int t = threadIdx.x;
    const int64_t o0 = (int64_t)blockIdx.x * OP_T, f0 = (int64_t)blockIdx.y * OP_T;
    for (int e = t; e < OP_T * 2 * OP_T; e += 256) {
        const int r = e / (2 * OP_T), c = e % (2 * OP_T);
        const int64_t o = o0 + r, col = 2 * f0 + c;
        tile[r][c] = (o < H && col < 2 * F) ? W[o * 2 * F + col] : 0.0f;
    }
    __syncthreads();
    for (int e = t; e < OP_T * OP_T; e += 256) {
        {   // D: k fastest
            const int r = e / OP_T, k = e % OP_T;
            const int64_t o = o0 + r, f = f0 + k;
            if (o < H && f < ldF) D[o * ldF + f] = f < F ? tile[r][2 * k + 1] - tile[r][2 * k] : 0.0f;
        }
        {   // D^T: o fastest
            const int r = e % OP_T, k = e / OP_T;
            const int64_t o = o0 + r, f = f0 + k;
            if (o < H && f < F) Dt[f * H + o] = tile[r][2 * k + 1] - tile[r][2 * k];
        }
    }
}
__global__ void __launch_bounds__(256)
k_onehot_prepare_B(int64_t F, int64_t H, const float* __restrict__ W, const float* __restrict__ b, float* __restrict__ B) {
    __shared__ float tile[OP_BR][2 * OP_T + 1];
    const int t = threadIdx.x;
    const int64_t o0 = (int64_t)blockIdx.x * OP_BR;
    float acc = 0.0f;
    for (int64_t f0 = 0; f0 < F; f0 += OP_T) {
        for (int e = t; e < OP_BR * 2 * OP_T; e += 256) {
            const int r = e / (2 * OP_T), c = e % (2 * OP_T);
            const int64_t o = o0 + r, col = 2 * f0 + c;
            tile[r][c] = (o < H && col < 2 * F) ? W[o * 2 * F + col] : 0.0f;
        }
        __syncthreads();
        if (t < OP_BR) {
            const int64_t nf = F - f0 < OP_T ? F - f0 : OP_T;
            for (int64_t k = 0; k < nf; ++k) acc = acc + tile[t][2 * k];
        }
        __syncthreads();
    }
    if (t < OP_BR && o0 + t < H) B[o0 + t] = acc + (b ? b[o0 + t] : 0.0f);
}

// X01[v][f] = x[v][f] != 0 as float, rows padded with zeros to ldF (the A operand of the dense form)
__global__ void k_u8_to_f32_rows(int64_t n, int64_t F, int64_t ldF, const uint8_t* __restrict__ x, float* __restrict__ X01) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n * ldF) return;
    const int64_t v = g / ldF, f = g - v * ldF;
    X01[g] = (f < F && x[v * F + f]) ? 1.0f : 0.0f;
}

// sparse form: one wave per node.  Lane l owns float4 chunks l, l + 64, ... of the H outputs.  Pass 1: the node's attribute
// bytes are read 256 at a time; a ballot turns each 64 of them into a wave-uniform mask and the set attributes are appended,
// in ascending order, to a per-wave list in LDS.  Pass 2: the list is consumed four entries at a time -- four row gathers of
// D^T in flight, then four adds in list order.  A full list is drained and refilled, so any number of set attributes works.
constexpr int OH_LIST = 512;

template <int CH>
__global__ void __launch_bounds__(256)
k_onehot_sparse(int64_t n, int64_t F, int64_t H, const uint8_t* __restrict__ x, const float* __restrict__ B,
                const float* __restrict__ Dt, float* __restrict__ out, int64_t ldo, int relu) {
    __shared__ uint16_t lists[4][OH_LIST];
    const int lane = threadIdx.x & 63;
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n) return;                                    // whole waves leave together: no barrier below
    uint16_t* list = lists[threadIdx.x >> 6];
    const int nchunks = (int)(H >> 2);
    const uint8_t* xr = x + v * F;
    float4 acc[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    int cnt = 0;                                           // wave-uniform
    auto drain = [&]() {
        for (int p0 = 0; p0 < cnt; p0 += 4) {
            float4 r[4][CH];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool on = p0 + q < cnt;
                const int64_t f = on ? (int64_t)list[p0 + q] : 0;
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const int c = lane + 64 * k;
                    r[q][k] = (on && c < nchunks) ? *reinterpret_cast<const float4*>(Dt + f * H + 4 * c)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (p0 + q < cnt) {
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        acc[k].x = acc[k].x + r[q][k].x;
                        acc[k].y = acc[k].y + r[q][k].y;
                        acc[k].z = acc[k].z + r[q][k].z;
                        acc[k].w = acc[k].w + r[q][k].w;
                    }
                }
        }
        cnt = 0;
    };
    for (int64_t base = 0; base < F; base += 256) {
        uint8_t byte[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t f = base + 64 * q + lane;
            byte[q] = f < F ? xr[f] : (uint8_t)0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool on = byte[q] != 0;
            const unsigned long long m = __ballot(on);
            if (on) list[cnt + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(base + 64 * q + lane);
            cnt += __popcll(m);
        }
        if (cnt > OH_LIST - 256) drain();                  // LDS serves a wave's own writes and reads in order
    }
    drain();
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int c = lane + 64 * k;
        if (c < nchunks) {
            const float4 bb = *reinterpret_cast<const float4*>(B + 4 * c);
            float4 r;
            r.x = acc[k].x + bb.x; r.y = acc[k].y + bb.y; r.z = acc[k].z + bb.z; r.w = acc[k].w + bb.w;
            if (relu) {
                if (!(r.x > 0.0f)) r.x = 0.0f;
                if (!(r.y > 0.0f)) r.y = 0.0f;
                if (!(r.z > 0.0f)) r.z = 0.0f;
                if (!(r.w > 0.0f)) r.w = 0.0f;
            }
            *reinterpret_cast<float4*>(out + v * ldo + 4 * c) = r;
        }
    }
}

}  // namespace gm

using namespace gm;

extern "C" int64_t gm_onehot_prep_floats(int64_t F, int64_t H) {
    if (F <= 0 || H <= 0) return -1;
    return prep_layout(F, H).total;
}

extern "C" int gm_onehot_prepare(int64_t F, int64_t H, const float* W, const float* b, float* prep, gm_stream_t stream) {
    GM_REQUIRE(F > 0 && H > 0 && W && prep, "bad arguments");
    GM_REQUIRE(aligned(prep, 16), "prep must be 16-byte aligned");
    const OnehotPrep p = prep_layout(F, H);
    GM_REQUIRE((p.ldF + OP_T - 1) / OP_T <= 65535, "F too large");
    hipLaunchKernelGGL(k_onehot_prepare_D, dim3((unsigned)((H + OP_T - 1) / OP_T), (unsigned)((p.ldF + OP_T - 1) / OP_T)), dim3(256), 0,
                       as_stream(stream), F, H, p.ldF, W, prep + p.oDt, prep + p.oD);
    GM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_onehot_prepare_B, dim3((unsigned)((H + OP_BR - 1) / OP_BR)), dim3(256), 0, as_stream(stream), F, H, W, b,
                       prep + p.oB);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_onehot_linear(int64_t n, int64_t F, int64_t H, const uint8_t* x, const float* prep, float* dense_ws,
                                float* out, int64_t ldo, int relu, gm_stream_t stream) {
    GM_REQUIRE(n > 0 && F > 0 && H > 0 && x && prep && out && ldo >= H, "bad arguments");
    const OnehotPrep p = prep_layout(F, H);
    hipStream_t s = as_stream(stream);
    const bool can_walk = H % 4 == 0 && H <= 2048 && F < 65536 && ldo % 4 == 0 && aligned(out, 16) && aligned(prep, 16);
    if (!dense_ws) GM_REQUIRE(can_walk, "the sparse form needs hidden_X %% 4 == 0, hidden_X <= 2048, F < 65536 and a 16-byte aligned output");
    if (!dense_ws) {
        const dim3 grid((unsigned)((n + 3) / 4)), block(256);
        const int ch = (int)((H / 4 + 63) / 64);
        if (ch <= 1) hipLaunchKernelGGL(k_onehot_sparse<1>, grid, block, 0, s, n, F, H, x, prep + p.oB, prep + p.oDt, out, ldo, relu);
        else if (ch == 2) hipLaunchKernelGGL(k_onehot_sparse<2>, grid, block, 0, s, n, F, H, x, prep + p.oB, prep + p.oDt, out, ldo, relu);
        else if (ch <= 4) hipLaunchKernelGGL(k_onehot_sparse<4>, grid, block, 0, s, n, F, H, x, prep + p.oB, prep + p.oDt, out, ldo, relu);
        else hipLaunchKernelGGL(k_onehot_sparse<8>, grid, block, 0, s, n, F, H, x, prep + p.oB, prep + p.oDt, out, ldo, relu);
        GM_LAUNCH_CHECK();
        return GM_OK;
    }
    GM_REQUIRE(aligned(dense_ws, 16), "dense workspace must be 16-byte aligned");
    GM_REQUIRE(n * p.ldF < (1ll << 40), "n*F too large");
    hipLaunchKernelGGL(k_u8_to_f32_rows, dim3((unsigned)((n * p.ldF + 255) / 256)), dim3(256), 0, s, n, F, p.ldF, x, dense_ws);
    GM_LAUNCH_CHECK();
    return gm_linear(n, H, F, dense_ws, p.ldF, prep + p.oD, p.ldF, prep + p.oB, out, ldo, relu, stream);
}
