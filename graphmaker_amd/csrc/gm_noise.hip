// gm_noise.hip -- forward noising q(G^t | G) of the training / validation path (model/diffusion.py:559-604,
// :962-1033) and the converters from the reference's dense training tensors to the library's state layout.
//   gm_noise_E : prob_E = E_one_hot @ Q_bar_t (:592-593), sample_E (:188-212: one draw per (i, j >= i), diagonal kept,
//                upper copied to lower).  The real graph comes in as a bitmap; row e0 of Q_bar_t is the pair's
//                distribution, so the kernel is the E prior with a per-bit choice between two Bernoulli rates.
//   gm_noise_X : prob_X = bmm(X_one_hot_3d, Q_bar_t_X) (:597-599), sample_X (:214-237).
//   gm_bitmap_from_onehot_f32 : E_one_hot (N,N,2) float -> bitmap of channel 1 (data.py:73-76).
//   gm_x_from_onehot3d        : X_one_hot_3d (F,N,2) float -> uint8 x[N][F] (data.py:65-71).
// HBM-bound elementwise work; Philox streams 5 (E) and 6 (X), indices as for the priors.
#include "gm_common.h"

namespace gm {

enum : uint32_t { STREAM_E_NOISE = 5, STREAM_X_NOISE = 6 };

struct Q4 {
    float q[4];   // q[r*2+c]
};

__global__ void k_noise_E_upper(int64_t n, int64_t wpr, const uint32_t* __restrict__ bits0, Q4 Q,
                                const float* __restrict__ u, uint64_t seed, uint32_t step, uint32_t* __restrict__ upper) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n * wpr) return;
    const int64_t i = g / wpr, w = g % wpr;
    uint32_t word = 0;
    if (w >= (i >> 5)) {
        const uint32_t w0 = bits0[g];
        const int64_t base = i * n + w * 32;                      // draw index of bit 0 of this word
        auto one = [&](int b, float uu) {
            const int64_t j = w * 32 + b;
            if (j >= i && j < n) {
                const bool e0 = (w0 >> b) & 1u;
                if (draw2(uu, e0 ? Q.q[2] : Q.q[0], e0 ? Q.q[3] : Q.q[1])) word |= 1u << b;
            }
        };
        if (u) {
            for (int b = 0; b < 32; ++b) one(b, u[base + b < n * n ? base + b : 0]);
        } else {
            u_idx64_run32(seed, step, STREAM_E_NOISE, (uint64_t)base, one);
        }
    }
    upper[g] = word;
}

// grid: x = group of four nodes, y = 256-attribute chunk; a thread takes the four nodes whose draw indices f*n + v share one
// Philox block (as k_attr_posterior_sample in gm_attr.hip): one block per four draws, same values as u_idx64
__global__ void k_noise_X(int64_t n, int64_t F, const uint8_t* __restrict__ x0, const float* __restrict__ QX,
                          const float* __restrict__ u, uint64_t seed, uint32_t step, uint8_t* __restrict__ x_t,
                          float* __restrict__ onehot) {
    const int64_t f = (int64_t)blockIdx.y * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const int64_t base = f * n, s = base & 3;
    const int64_t v0 = (int64_t)blockIdx.x * 4 - s;
    u32x4 o;
    if (!u) {
        const uint64_t q = (uint64_t)(base + v0) >> 2;
        o = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), step, STREAM_X_NOISE, (uint32_t)seed, (uint32_t)(seed >> 32));
    }
    const float q00 = QX[4 * f], q01 = QX[4 * f + 1], q10 = QX[4 * f + 2], q11 = QX[4 * f + 3];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t v = v0 + k;
        if (v < 0 || v >= n) continue;
        const int64_t g = v * F + f;
        const float uu = u ? u[base + v] : u01(o.v[k]);
        const bool r = x0[g] != 0;
        const bool b = draw2(uu, r ? q10 : q00, r ? q11 : q01);
        x_t[g] = b ? 1 : 0;
        if (onehot) *reinterpret_cast<float2*>(onehot + 2 * g) = b ? make_float2(0.f, 1.f) : make_float2(1.f, 0.f);
    }
}

// one thread per output word; channel 1 > channel 0 marks an edge (one-hot input)
__global__ void k_bitmap_from_onehot(int64_t n, int64_t wpr, const float* __restrict__ E, uint32_t* __restrict__ bits) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n * wpr) return;
    const int64_t i = g / wpr, w = g % wpr;
    uint32_t word = 0;
    for (int b = 0; b < 32; ++b) {
        const int64_t j = w * 32 + b;
        if (j < n) {
            const float2 o = *reinterpret_cast<const float2*>(E + 2 * (i * n + j));
            if (o.y > o.x) word |= 1u << b;
        }
    }
    bits[g] = word;
}

__global__ void k_x_from_onehot3d(int64_t n, int64_t F, const float* __restrict__ X3, uint8_t* __restrict__ x) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n * F) return;
    const int64_t v = g / F, f = g % F;
    const float2 o = *reinterpret_cast<const float2*>(X3 + 2 * (f * n + v));
    x[g] = o.y > o.x ? 1 : 0;
}

}  // namespace gm

using namespace gm;

static inline unsigned blocks_for(int64_t total) { return (unsigned)((total + 255) / 256); }

extern "C" int gm_noise_E(int64_t n, int64_t wpr, const uint32_t* bits0, const float* Q_bar_t, const float* u, uint64_t seed,
                          uint32_t step, uint32_t* upper_ws, uint32_t* bits_t, gm_stream_t stream) {
    GM_REQUIRE(n > 0 && wpr >= (n + 31) / 32 && wpr % 4 == 0 && bits0 && Q_bar_t && upper_ws && bits_t, "bad arguments");
    GM_REQUIRE(upper_ws != bits_t && upper_ws != bits0, "upper_ws must not alias bits0 / bits_t");
    Q4 Q;
    for (int k = 0; k < 4; ++k) Q.q[k] = Q_bar_t[k];
    hipLaunchKernelGGL(k_noise_E_upper, dim3(blocks_for(n * wpr)), dim3(256), 0, as_stream(stream), n, wpr, bits0, Q, u, seed,
                       step, upper_ws);
    GM_LAUNCH_CHECK();
    return gm_symmetrize_upper(n, wpr, upper_ws, bits_t, stream);
}

extern "C" int gm_noise_X(int64_t n, int64_t F, const uint8_t* x0, const float* Q_bar_t_X, const float* u, uint64_t seed,
                          uint32_t step, uint8_t* x_t, float* onehot_out, gm_stream_t stream) {
    GM_REQUIRE(n > 0 && F > 0 && x0 && Q_bar_t_X && x_t, "bad arguments");
    GM_REQUIRE(!onehot_out || aligned(onehot_out, 8), "onehot must be 8-byte aligned");
    GM_REQUIRE(n * F < (1ll << 39), "n*F too large");
    GM_REQUIRE((n + 2) / 4 + 1 < (1ll << 31) && (F + 255) / 256 <= 65535, "grid too large");
    hipLaunchKernelGGL(k_noise_X, dim3((unsigned)(((n + 2) >> 2) + 1), (unsigned)((F + 255) / 256)), dim3(256), 0, as_stream(stream), n, F,
                       x0, Q_bar_t_X, u, seed, step, x_t, onehot_out);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_bitmap_from_onehot_f32(int64_t n, int64_t wpr, const float* E_one_hot, uint32_t* bits, gm_stream_t stream) {
    GM_REQUIRE(n > 0 && wpr >= (n + 31) / 32 && E_one_hot && bits && aligned(E_one_hot, 8), "bad arguments");
    hipLaunchKernelGGL(k_bitmap_from_onehot, dim3(blocks_for(n * wpr)), dim3(256), 0, as_stream(stream), n, wpr, E_one_hot, bits);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_x_from_onehot3d(int64_t n, int64_t F, const float* X_one_hot_3d, uint8_t* x, gm_stream_t stream) {
    GM_REQUIRE(n > 0 && F > 0 && X_one_hot_3d && x && aligned(X_one_hot_3d, 8), "bad arguments");
    hipLaunchKernelGGL(k_x_from_onehot3d, dim3(blocks_for(n * F)), dim3(256), 0, as_stream(stream), n, F, X_one_hot_3d, x);
    GM_LAUNCH_CHECK();
    return GM_OK;
}
