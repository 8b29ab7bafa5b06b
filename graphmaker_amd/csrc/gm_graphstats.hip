// gm_graphstats.hip -- integer graph statistics of a generated graph on its bitmap + CSR, the
// slice of the evaluator (eval_utils.py:19-74, 506-596) that is impossible at 100k nodes through A.to_dense():
//   gm_triangles_per_node   2 * triangles through v = sum_{u in N(v)} |N(v) & N(u)|     (get_triangle_count :19-21,
//                           nx.clustering :585)
//   gm_same_class_degree    |{u in N(v): y_u = y_v}|                                    (linkx_homophily :52-59)
//   gm_bitmap_square        pattern of A @ A: row2[i] = OR_{j in N(i)} row[j]           (k_order_g :521-529)
//   gm_class_bitmaps        one bitmap row per class
//   gm_masked_degree        deg2[v] = popcount(row2[v]), same2[v] = popcount(row2[v] & classrow[y_v])
// HBM/L2-bound word streams: one wave per node, lanes stride over the row's 32-bit words.
#include "gm_common.h"

namespace gm {

__device__ __forceinline__ int64_t wave_sum_i64(int64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ void __launch_bounds__(256)
k_triangles_per_node(int64_t n, int64_t wpr, const uint32_t* __restrict__ bits, const int32_t* __restrict__ rowptr,
                     const int32_t* __restrict__ colidx, int64_t* __restrict__ tri2) {
    const int lane = threadIdx.x & 63;
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n) return;
    const uint32_t* rv = bits + v * wpr;
    int64_t acc = 0;
    for (int p = rowptr[v]; p < rowptr[v + 1]; ++p) {
        const uint32_t* ru = bits + (int64_t)colidx[p] * wpr;
        for (int64_t w = lane; w < wpr; w += 64) acc += __popc(rv[w] & ru[w]);
    }
    acc = wave_sum_i64(acc);
    if (lane == 0) tri2[v] = acc;
}

__global__ void k_same_class_degree(int64_t n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                    const int32_t* __restrict__ y, int32_t* __restrict__ out) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const int yv = y[v];
    int c = 0;
    for (int p = rowptr[v]; p < rowptr[v + 1]; ++p) c += (y[colidx[p]] == yv);
    out[v] = c;
}

__global__ void __launch_bounds__(256)
k_bitmap_square(int64_t n, int64_t wpr, const uint32_t* __restrict__ bits, const int32_t* __restrict__ rowptr,
                const int32_t* __restrict__ colidx, uint32_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int b = rowptr[i], e = rowptr[i + 1];
    for (int64_t w = lane; w < wpr; w += 64) {
        uint32_t acc = 0;
        for (int p = b; p < e; ++p) acc |= bits[(int64_t)colidx[p] * wpr + w];
        out[i * wpr + w] = acc;
    }
}

__global__ void k_class_bitmaps(int64_t n, int64_t wpr, int C, const int32_t* __restrict__ y, uint32_t* __restrict__ out) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (int64_t)C * wpr) return;
    const int c = (int)(g / wpr);
    const int64_t w = g % wpr;
    uint32_t word = 0;
    for (int b = 0; b < 32; ++b) {
        const int64_t v = w * 32 + b;
        if (v < n && y[v] == c) word |= 1u << b;
    }
    out[g] = word;
}

__global__ void __launch_bounds__(256)
k_masked_degree(int64_t n, int64_t wpr, const uint32_t* __restrict__ bits, const uint32_t* __restrict__ classrows,
                const int32_t* __restrict__ y, int32_t* __restrict__ deg, int32_t* __restrict__ same) {
    const int lane = threadIdx.x & 63;
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n) return;
    const uint32_t* r = bits + v * wpr;
    const uint32_t* m = classrows + (int64_t)y[v] * wpr;
    int64_t d = 0, s = 0;
    for (int64_t w = lane; w < wpr; w += 64) {
        d += __popc(r[w]);
        s += __popc(r[w] & m[w]);
    }
    d = wave_sum_i64(d);
    s = wave_sum_i64(s);
    if (lane == 0) {
        deg[v] = (int32_t)d;
        same[v] = (int32_t)s;
    }
}

}  // namespace gm

using namespace gm;

extern "C" int gm_triangles_per_node(int64_t n, int64_t wpr, const uint32_t* bits, const int32_t* rowptr, const int32_t* colidx,
                                     int64_t* tri2, gm_stream_t stream) {
    GM_REQUIRE(n > 0 && wpr >= (n + 31) / 32 && bits && rowptr && tri2, "bad arguments");
    hipLaunchKernelGGL(k_triangles_per_node, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, as_stream(stream), n, wpr, bits, rowptr,
                       colidx, tri2);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_same_class_degree(int64_t n, const int32_t* rowptr, const int32_t* colidx, const int32_t* y, int32_t* out,
                                    gm_stream_t stream) {
    GM_REQUIRE(n > 0 && rowptr && y && out, "bad arguments");
    hipLaunchKernelGGL(k_same_class_degree, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), n, rowptr, colidx,
                       y, out);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_bitmap_square(int64_t n, int64_t wpr, const uint32_t* bits, const int32_t* rowptr, const int32_t* colidx,
                                uint32_t* out, gm_stream_t stream) {
    GM_REQUIRE(n > 0 && wpr >= (n + 31) / 32 && bits && rowptr && out && out != bits, "bad arguments");
    hipLaunchKernelGGL(k_bitmap_square, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, as_stream(stream), n, wpr, bits, rowptr,
                       colidx, out);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_class_bitmaps(int64_t n, int64_t wpr, int C, const int32_t* y, uint32_t* out, gm_stream_t stream) {
    GM_REQUIRE(n > 0 && wpr >= (n + 31) / 32 && C > 0 && y && out, "bad arguments");
    hipLaunchKernelGGL(k_class_bitmaps, dim3((unsigned)(((int64_t)C * wpr + 255) / 256)), dim3(256), 0, as_stream(stream), n, wpr,
                       C, y, out);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_masked_degree(int64_t n, int64_t wpr, const uint32_t* bits, const uint32_t* classrows, const int32_t* y,
                                int32_t* deg, int32_t* same, gm_stream_t stream) {
    GM_REQUIRE(n > 0 && wpr >= (n + 31) / 32 && bits && classrows && y && deg && same, "bad arguments (labels must lie in [0, C))");
    hipLaunchKernelGGL(k_masked_degree, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, as_stream(stream), n, wpr, bits, classrows, y,
                       deg, same);
    GM_LAUNCH_CHECK();
    return GM_OK;
}
