// gm_attr.hip -- attribute posterior + draw, one-hot conversions and the three priors.  HBM-bound elementwise work.
//   gm_attr_posterior_sample : softmax over the class dim (model/diffusion.py:879), per-attribute posterior
//                              (:884-894 via BaseModel.posterior :351-406) and sample_X (:214-237) in one pass,
//                              writing the uint8 state and (optionally) the (n,2F) float one-hot the towers read.
//   gm_attr_posterior_sample_diff / gm_head_diff_prepare : the same draw from the logit DIFFERENCE l1 - l0 that a
//                              node tower with the half-width head W[2f+1] - W[2f] puts out (half the output GEMM).
//   gm_prior_Y / gm_prior_X / gm_prior_E : model/diffusion.py:831-844 (sample_E :188-212 keeps diagonal draws).
#include <mutex>

#include "gm_common.h"

namespace gm {

// DIFF: `logits` is (n,F), the difference l1 - l0 per attribute (softmax([l0,l1]) = softmax([0,d]), as in the pair kernel)
// grid: x = group of four nodes, y = 256-attribute chunk (lanes run over attributes: every access is a coalesced row piece).
// The draw of (v,f) is word idx&3 of the Philox block idx>>2, idx = f*n + v (reshape(-1, 2) of the (F, n, 2) tensor,
// model/diffusion.py:228-230), so a thread takes the four nodes v = 4*q' - s + k, s = (f*n)&3, whose indices share one
// block: one Philox call per four draws instead of four (the kernel was bound by its integer multiplies).
template <bool DIFF>
__global__ void k_attr_posterior_sample(int64_t n, int64_t F, const float* __restrict__ logits,
                                        const uint8_t* __restrict__ x_t, const float* __restrict__ tab,
                                        const float* __restrict__ u, uint64_t seed, uint32_t step, int64_t v_begin,
                                        int64_t v_end, uint8_t* __restrict__ x_s, float* __restrict__ onehot,
                                        float* __restrict__ post) {
    const int64_t f = (int64_t)blockIdx.y * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const int64_t base = f * n, s = base & 3;
    const int64_t v0 = ((v_begin >> 2) + (int64_t)blockIdx.x) * 4 - s;
    u32x4 o;
    if (!u) {
        const uint64_t q = (uint64_t)(base + v0) >> 2;            // base + v0 is a multiple of 4 (and >= 0 where a v is valid)
        o = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), step, STREAM_X, (uint32_t)seed, (uint32_t)(seed >> 32));
    }
    const float* tf = tab + 8 * f;                                // this attribute's posterior table, once for the four nodes
    const float t0 = tf[0], t1 = tf[1], t2 = tf[2], t3 = tf[3], t4 = tf[4], t5 = tf[5], t6 = tf[6], t7 = tf[7];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t v = v0 + k;
        if (v < v_begin || v >= v_end) continue;
        const int64_t g = v * F + f;
        float q0, q1, p0, p1;
        if (DIFF) {
            softmax2_diff(logits[g], q0, q1);
        } else {
            const float2 l = *reinterpret_cast<const float2*>(logits + 2 * g);
            softmax2(l.x, l.y, q0, q1);
        }
        const bool e = x_t[g] != 0;                               // posterior2 (gm_common.h) on the registers above
        const float a0 = q0 * (e ? t1 : t0), b0 = q1 * (e ? t5 : t4), a1 = q0 * (e ? t3 : t2), b1 = q1 * (e ? t7 : t6);
        p0 = a0 + b0;
        p1 = a1 + b1;
        const float uu = u ? u[base + v] : u01(o.v[k]);
        const bool b = draw2(uu, p0, p1);
        x_s[g] = b ? 1 : 0;
        if (onehot) *reinterpret_cast<float2*>(onehot + 2 * g) = b ? make_float2(0.f, 1.f) : make_float2(1.f, 0.f);
        if (post) *reinterpret_cast<float2*>(post + 2 * g) = make_float2(p0, p1);
    }
}

// Wd[f][k] = W[2f+1][k] - W[2f][k], bd[f] = b[2f+1] - b[2f]: the last Linear of a node tower folded to one output per
// attribute (one thread per element of Wd; the first F threads also write bd)
__global__ void k_head_diff(int64_t F, int64_t Hc, const float* __restrict__ W, const float* __restrict__ b,
                            float* __restrict__ Wd, float* __restrict__ bd) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= F * Hc) return;
    const int64_t f = g / Hc, k = g - f * Hc;
    Wd[g] = W[(2 * f + 1) * Hc + k] - W[2 * f * Hc + k];
    if (g < F) bd[g] = b[2 * g + 1] - b[2 * g];
}

__global__ void k_onehot_from_x(int64_t total, const uint8_t* __restrict__ x, float* __restrict__ onehot) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    *reinterpret_cast<float2*>(onehot + 2 * g) = x[g] ? make_float2(0.f, 1.f) : make_float2(1.f, 0.f);
}
__global__ void k_x_from_onehot(int64_t total, const float* __restrict__ onehot, uint8_t* __restrict__ x) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const float2 o = *reinterpret_cast<const float2*>(onehot + 2 * g);
    x[g] = o.y > o.x ? 1 : 0;
}

__global__ void k_prior_Y(int64_t n, int C, const float* __restrict__ Ym, const float* __restrict__ u, uint64_t seed,
                          int32_t* __restrict__ y) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    float sum = 0.0f;
    for (int c = 0; c < C; ++c) sum = sum + Ym[c];
    const float uu = u ? u[v] : u_label(seed, (uint32_t)v);
    const float us = uu * sum;
    float cum = 0.0f;
    int pick = C - 1;
    for (int c = 0; c < C; ++c) {
        cum = cum + Ym[c];
        if (us < cum) { pick = c; break; }
    }
    y[v] = pick;
}

// grid and draw sharing as in k_attr_posterior_sample: four nodes per thread, one Philox block per four draws
__global__ void k_prior_X(int64_t n, int64_t F, const float* __restrict__ Xm, const float* __restrict__ u, uint64_t seed,
                          uint8_t* __restrict__ x, float* __restrict__ onehot) {
    const int64_t f = (int64_t)blockIdx.y * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const int64_t base = f * n, s = base & 3;
    const int64_t v0 = (int64_t)blockIdx.x * 4 - s;
    u32x4 o;
    if (!u) {
        const uint64_t q = (uint64_t)(base + v0) >> 2;
        o = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), 0u, STREAM_X, (uint32_t)seed, (uint32_t)(seed >> 32));
    }
    const float m0 = Xm[2 * f], m1 = Xm[2 * f + 1];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t v = v0 + k;
        if (v < 0 || v >= n) continue;
        const int64_t g = v * F + f;
        const float uu = u ? u[base + v] : u01(o.v[k]);
        const bool b = draw2(uu, m0, m1);
        x[g] = b ? 1 : 0;
        if (onehot) *reinterpret_cast<float2*>(onehot + 2 * g) = b ? make_float2(0.f, 1.f) : make_float2(1.f, 0.f);
    }
}

// upper[i][w] bits j >= i (diagonal included), one thread per word
__global__ void k_prior_E_upper(int64_t n, int64_t wpr, float p0, float p1, const float* __restrict__ u, uint64_t seed,
                                uint32_t* __restrict__ upper) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n * wpr) return;
    const int64_t i = g / wpr, w = g % wpr;
    uint32_t word = 0;
    if (w >= (i >> 5)) {
        const int64_t base = i * n + w * 32;                      // draw index of bit 0 of this word
        if (u) {
            for (int b = 0; b < 32; ++b) {
                const int64_t j = w * 32 + b;
                if (j >= i && j < n && draw2(u[base + b], p0, p1)) word |= 1u << b;
            }
        } else {
            u_idx64_run32(seed, 0u, STREAM_E_PRIOR, (uint64_t)base, [&](int b, float uu) {
                const int64_t j = w * 32 + b;
                if (j >= i && j < n && draw2(uu, p0, p1)) word |= 1u << b;
            });
        }
    }
    upper[g] = word;
}

}  // namespace gm

using namespace gm;

static inline unsigned blocks_for(int64_t total) { return (unsigned)((total + 255) / 256); }

template <bool DIFF>
static int attr_posterior_sample(int64_t n, int64_t F, const float* logits, const uint8_t* x_t, const float* tab,
                                 const float* u, uint64_t seed, uint32_t step, int64_t v_begin, int64_t v_end,
                                 uint8_t* x_s, float* onehot_out, float* post_dbg, gm_stream_t stream) {
    GM_REQUIRE(n > 0 && F > 0 && logits && x_t && tab && x_s, "bad arguments");
    GM_REQUIRE(0 <= v_begin && v_begin <= v_end && v_end <= n, "bad node range [%lld,%lld)", (long long)v_begin, (long long)v_end);
    if (v_begin == v_end) return GM_OK;
    GM_REQUIRE(n * F < (1ll << 39), "n*F too large");
    GM_REQUIRE(aligned(logits, DIFF ? 4 : 8) && (!onehot_out || aligned(onehot_out, 8)) && (!post_dbg || aligned(post_dbg, 8)),
               "logits/onehot/post must be 8-byte aligned (a logit difference: 4-byte)");
    GM_REQUIRE(v_end - v_begin < (1ll << 31) && (F + 255) / 256 <= 65535, "grid too large");
    // groups of four nodes that can hold a node of the range for some shift s in 0..3: 4*q' - s + k in [v_begin, v_end)
    const int64_t groups = ((v_end + 2) >> 2) - (v_begin >> 2) + 1;
    hipLaunchKernelGGL(k_attr_posterior_sample<DIFF>, dim3((unsigned)groups, (unsigned)((F + 255) / 256)), dim3(256), 0,
                       as_stream(stream), n, F, logits, x_t, tab, u, seed, step, v_begin, v_end, x_s, onehot_out, post_dbg);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_attr_posterior_sample(int64_t n, int64_t F, const float* logits, const uint8_t* x_t, const float* tab,
                                        const float* u, uint64_t seed, uint32_t step, int64_t v_begin, int64_t v_end,
                                        uint8_t* x_s, float* onehot_out, float* post_dbg, gm_stream_t stream) {
    return attr_posterior_sample<false>(n, F, logits, x_t, tab, u, seed, step, v_begin, v_end, x_s, onehot_out, post_dbg, stream);
}

extern "C" int gm_attr_posterior_sample_diff(int64_t n, int64_t F, const float* diff, const uint8_t* x_t, const float* tab,
                                             const float* u, uint64_t seed, uint32_t step, int64_t v_begin, int64_t v_end,
                                             uint8_t* x_s, float* onehot_out, float* post_dbg, gm_stream_t stream) {
    return attr_posterior_sample<true>(n, F, diff, x_t, tab, u, seed, step, v_begin, v_end, x_s, onehot_out, post_dbg, stream);
}

extern "C" int gm_head_diff_prepare(int64_t F, int64_t Hc, const float* W, const float* b, float* Wd, float* bd,
                                    gm_stream_t stream) {
    GM_REQUIRE(F > 0 && Hc > 0 && W && b && Wd && bd, "bad arguments");
    GM_REQUIRE(F * Hc < (1ll << 39), "F*Hc too large");
    hipLaunchKernelGGL(k_head_diff, dim3(blocks_for(F * Hc)), dim3(256), 0, as_stream(stream), F, Hc, W, b, Wd, bd);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_onehot_from_x(int64_t n, int64_t F, const uint8_t* x, float* onehot, gm_stream_t stream) {
    GM_REQUIRE(n > 0 && F > 0 && x && onehot && aligned(onehot, 8), "bad arguments");
    hipLaunchKernelGGL(k_onehot_from_x, dim3(blocks_for(n * F)), dim3(256), 0, as_stream(stream), n * F, x, onehot);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_x_from_onehot(int64_t n, int64_t F, const float* onehot, uint8_t* x, gm_stream_t stream) {
    GM_REQUIRE(n > 0 && F > 0 && x && onehot && aligned(onehot, 8), "bad arguments");
    hipLaunchKernelGGL(k_x_from_onehot, dim3(blocks_for(n * F)), dim3(256), 0, as_stream(stream), n * F, onehot, x);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_prior_Y(int64_t n, int C, const float* Y_marginal, const float* u, uint64_t seed, int32_t* y,
                          gm_stream_t stream) {
    GM_REQUIRE(n > 0 && C > 0 && Y_marginal && y, "bad arguments");
    hipLaunchKernelGGL(k_prior_Y, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), n, C, Y_marginal, u, seed, y);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_prior_X(int64_t n, int64_t F, const float* X_marginal, const float* u, uint64_t seed, uint8_t* x,
                          float* onehot_out, gm_stream_t stream) {
    GM_REQUIRE(n > 0 && F > 0 && X_marginal && x, "bad arguments");
    GM_REQUIRE(!onehot_out || aligned(onehot_out, 8), "onehot must be 8-byte aligned");
    GM_REQUIRE(n < (1ll << 31) && (F + 255) / 256 <= 65535, "grid too large");
    hipLaunchKernelGGL(k_prior_X, dim3((unsigned)(((n + 2) >> 2) + 1), (unsigned)((F + 255) / 256)), dim3(256), 0, as_stream(stream), n, F,
                       X_marginal, u, seed, x, onehot_out);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_prior_E(int64_t n, int64_t wpr, float p0, float p1, const float* u, uint64_t seed, uint32_t* upper_ws,
                          uint32_t* bits, gm_stream_t stream) {
    GM_REQUIRE(n > 0 && wpr >= (n + 31) / 32 && wpr % 4 == 0 && upper_ws && bits && upper_ws != bits, "bad arguments");
    hipLaunchKernelGGL(k_prior_E_upper, dim3(blocks_for(n * wpr)), dim3(256), 0, as_stream(stream), n, wpr, p0, p1, u, seed,
                       upper_ws);
    GM_LAUNCH_CHECK();
    return gm_symmetrize_upper(n, wpr, upper_ws, bits, stream);
}

// ---- gm_prior_E without any allocation (include/graphmaker_hip.h: no counterpart in the reference) ----
// The upper-triangle scratch bitmap that gm_prior_E takes from its caller lives in a file-scope device array here.  Every
// GPU the library runs on holds its own copy at its own address until the process ends; host code never uses the array's
// name as a pointer (it is a host-side stand-in, not a device address): the address is looked up once per device with
// hipGetSymbolAddress and kept in g_static_upper[].
namespace {
constexpr int64_t STATIC_MAX_N = 16384;                                   // largest graph the static scratch serves
constexpr int64_t STATIC_WORDS = STATIC_MAX_N * (STATIC_MAX_N / 32);     // 8,388,608 words = 32 MiB per GPU
constexpr int STATIC_MAX_DEVICES = 64;
static __device__ __align__(256) uint32_t scratch[STATIC_WORDS];
uint32_t* g_static_upper[STATIC_MAX_DEVICES] = {};
std::mutex g_static_mutex;

// address of this device's copy of `scratch` (the current device must be the one the stream belongs to)
int static_upper_ptr(uint32_t** out) {
    int dev = -1;
    GM_HIP(hipGetDevice(&dev));
    GM_REQUIRE(dev >= 0 && dev < STATIC_MAX_DEVICES, "device ordinal %d outside the static scratch table", dev);
    std::lock_guard<std::mutex> lock(g_static_mutex);
    if (!g_static_upper[dev]) {
        void* p = nullptr;
        GM_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        GM_REQUIRE(p && aligned(p, 256), "hipGetSymbolAddress returned a null or misaligned scratch address");
        g_static_upper[dev] = static_cast<uint32_t*>(p);
    }
    *out = g_static_upper[dev];
    return GM_OK;
}
}  // namespace

extern "C" int gm_prior_E_static(int64_t n, int64_t wpr, float p0, float p1, const float* u, uint64_t seed, uint32_t* bits,
                                 gm_stream_t stream) {
    GM_REQUIRE(n > 0 && wpr >= (n + 31) / 32 && wpr % 4 == 0 && bits, "bad arguments");
    GM_REQUIRE(n <= STATIC_MAX_N && n * wpr <= STATIC_WORDS,
               "gm_prior_E_static serves graphs of at most %lld nodes (n*wpr <= %lld words of static scratch); got n=%lld wpr=%lld: "
               "use gm_prior_E with a caller-owned workspace", (long long)STATIC_MAX_N, (long long)STATIC_WORDS, (long long)n,
               (long long)wpr);
    uint32_t* upper = nullptr;
    if (int rc = static_upper_ptr(&upper)) return rc;
    // k_prior_E_upper writes every one of the n*wpr words it is given, so nothing of the previous call's contents survives
    // in the part this call reads: the scratch needs no clearing
    hipLaunchKernelGGL(k_prior_E_upper, dim3(blocks_for(n * wpr)), dim3(256), 0, as_stream(stream), n, wpr, p0, p1, u, seed, upper);
    GM_LAUNCH_CHECK();
    return gm_symmetrize_upper(n, wpr, upper, bits, stream);
}
