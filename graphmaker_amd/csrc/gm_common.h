// gm_common.h -- shared device helpers and host-side error plumbing for libgraphmaker_hip.so (gfx950 only).
// Built with -ffp-contract=off: a*b+c is two roundings unless written as fmaf(); the arithmetic orders below are
// the ones oracle/gm_oracle.c documents, which is what makes bit-for-bit comparisons possible.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "graphmaker_hip.h"

namespace gm {

void set_error(const char* fmt, ...);

#define GM_REQUIRE(cond, ...)               \
    do {                                    \
        if (!(cond)) {                      \
            gm::set_error(__VA_ARGS__);     \
            return GM_E_INVALID;            \
        }                                   \
    } while (0)

#define GM_HIP(expr)                                                                       \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            gm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return GM_E_HIP;                                                               \
        }                                                                                  \
    } while (0)

#define GM_LAUNCH_CHECK() GM_HIP(hipGetLastError())

static inline hipStream_t as_stream(gm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// small data movers (gm_nodeops.hip) used by the tower driver
int launch_gather_rows(int64_t n, int w, const int32_t* idx, const float* table, int64_t ldt, float* dst, int64_t ldd, hipStream_t s);
int launch_broadcast_row(int64_t n, int w, const float* row, float* dst, int64_t ldd, hipStream_t s);
int launch_spmm_mean_rows(int64_t n, int64_t hn, const int32_t* rowptr, const int32_t* colidx, const float* H, int64_t ldh,
                          int64_t d, float* out, int64_t ldo, hipStream_t s);
int launch_spmm_mean_two(int64_t n, const int32_t* rowptr, const int32_t* colidx, const float* HX, int64_t dX, const float* HY,
                         int64_t dY, int64_t ldh, float* out, int64_t ldo, hipStream_t s);
int launch_copy2d(int64_t n, int w, const float* src, int64_t lds_, float* dst, int64_t ldd, hipStream_t s);
int launch_time_embed(int ht, const float* t, const float* W0, const float* b0, const float* W1, const float* b1, int64_t rows,
                      float* dst1, int64_t ld1, float* dst2, int64_t ld2, hipStream_t s);

enum : uint32_t { STREAM_Y = 1, STREAM_X = 2, STREAM_E_PRIOR = 3, STREAM_E = 4 };

// ---- Philox4x32-10 (same rounds and constants as oracle gmo_philox4x32_10)
struct u32x4 {
    uint32_t v[4];
};
__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    u32x4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}
__device__ __forceinline__ float u01(uint32_t w) { return (float)(w >> 8) * 5.9604644775390625e-08f; }
__device__ __forceinline__ uint32_t pick4(const u32x4& o, uint32_t w) {
    uint32_t a = (w & 1) ? o.v[1] : o.v[0];
    uint32_t b = (w & 1) ? o.v[3] : o.v[2];
    return (w & 2) ? b : a;
}
__device__ __forceinline__ float u_idx64(uint64_t seed, uint32_t step, uint32_t stream, uint64_t idx) {
    uint64_t q = idx >> 2;
    u32x4 o = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), step, stream, (uint32_t)seed, (uint32_t)(seed >> 32));
    return u01(pick4(o, (uint32_t)idx & 3u));
}
// fn(b, u) for the 32 consecutive draw indices base .. base+31 of one stream (b = 0..31): the same values as u_idx64, one
// Philox block per four indices instead of one per index (9 blocks cover any alignment of base; control flow is uniform)
template <class Fn>
__device__ __forceinline__ void u_idx64_run32(uint64_t seed, uint32_t step, uint32_t stream, uint64_t base, Fn&& fn) {
    const uint64_t q0 = base >> 2;
    const int s = (int)(base & 3u);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const uint64_t q = q0 + (uint64_t)t;
        const u32x4 o = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), step, stream, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int b = 4 * t + c - s;
            if (b >= 0 && b < 32) fn(b, u01(o.v[c]));
        }
    }
}
__device__ __forceinline__ float u_pair(uint64_t seed, uint32_t step, uint32_t i, uint32_t j) {
    u32x4 o = philox4x32_10(j >> 2, i, step, STREAM_E, (uint32_t)seed, (uint32_t)(seed >> 32));
    return u01(pick4(o, j & 3u));
}
__device__ __forceinline__ float u_label(uint64_t seed, uint32_t v) {
    u32x4 o = philox4x32_10(v >> 2, 0u, 0u, STREAM_Y, (uint32_t)seed, (uint32_t)(seed >> 32));
    return u01(pick4(o, v & 3u));
}

// ---- exp for x <= 0 : instruction-for-instruction the oracle's gm_expf_nonpos
__device__ __forceinline__ float expf_nonpos(float x) {
    if (x < -80.0f) x = -80.0f;
    float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = __builtin_fmaf(p, r2, r) + 1.0f;
    return __builtin_ldexpf(y, (int)n);
}
// softmax over two classes (model/diffusion.py:457,879)
__device__ __forceinline__ void softmax2(float l0, float l1, float& q0, float& q1) {
    float m = l0 > l1 ? l0 : l1;
    float e0 = expf_nonpos(l0 - m);
    float e1 = expf_nonpos(l1 - m);
    float s = e0 + e1;
    q0 = e0 / s;
    q1 = e1 / s;
}
// softmax over two classes from the logit difference d = l1 - l0: instruction-for-instruction the oracle's gmo_softmax2_diff
__device__ __forceinline__ void softmax2_diff(float d, float& q0, float& q1) {
    const bool pos = d > 0.0f;
    const float e = expf_nonpos(pos ? -d : d);
    const float s = 1.0f + e;
    const float r = 1.0f / s;
    const float small = e * r;
    q0 = pos ? small : r;
    q1 = pos ? r : small;
}
// The same function, same bits for every finite d, in fewer vector-ALU instructions (in the pair kernels each one costs
// matrix-pipe time): -|d| through source modifiers, the clamp as one max, and 1/s for s in [1, 2] as the compiler's own
// division sequence without its scaling and fix-up steps (which are the identity in that range).
__device__ __forceinline__ float rcp_1to2(float s) {
    float r = __builtin_amdgcn_rcpf(s);
    float e = __builtin_fmaf(-s, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = r;
    e = __builtin_fmaf(-s, q, 1.0f);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-s, q, 1.0f);
    return __builtin_fmaf(e, r, q);
}
__device__ __forceinline__ void softmax2_diff_lean(float d, float& q0, float& q1) {
    float x = __builtin_fmaxf(-__builtin_fabsf(d), -80.0f);
    float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = __builtin_fmaf(p, r2, r) + 1.0f;
    const float e = __builtin_ldexpf(y, (int)n);
    const float s = 1.0f + e;
    const float rs = rcp_1to2(s);
    const float small = e * rs;
    const bool pos = d > 0.0f;
    q0 = pos ? small : rs;
    q1 = pos ? rs : small;
}
// posterior from the table tab[(i*2+k)*2+e] (model/diffusion.py:351-406 for a one-hot Z_t = e)
__device__ __forceinline__ void posterior2(const float* tab, int e, float q0, float q1, float& p0, float& p1) {
    float a0 = q0 * tab[0 + e];
    float b0 = q1 * tab[4 + e];
    float a1 = q0 * tab[2 + e];
    float b1 = q1 * tab[6 + e];
    p0 = a0 + b0;
    p1 = a1 + b1;
}
__device__ __forceinline__ bool draw2(float u, float p0, float p1) {
    float s = p0 + p1;
    float us = u * s;
    return us < p1;
}

}  // namespace gm
