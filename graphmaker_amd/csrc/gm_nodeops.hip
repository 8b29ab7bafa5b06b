// gm_nodeops.hip -- HBM-bound per-node kernels of the denoiser towers:
//   * CSR mean aggregation  (model/gnn.py:60-61  A @ H with A = D^-1 A, model/diffusion.py:255-256)
//   * LayerNorm             (model/gnn.py:30,36)
//   * small data movers used by the tower driver (embedding gather, row broadcast, strided copy)
#include "gm_common.h"

namespace gm {

// ---------------------------------------------------------------- SpMM (mean aggregation)
// A group of TPR consecutive threads owns one output row; each thread keeps CH float4 column chunks
// (chunk c = t + k*TPR, so one neighbour row is read as contiguous 16-byte pieces across the group) and walks the
// row's neighbours in ascending order, acc = fmaf(H[j], 1/deg, acc) -- the order the oracle uses.
// Neighbour ids are read 4 at a time so 4 independent row gathers are in flight per thread.
// TRANSPOSED = true is the backward of the same operator for a symmetric adjacency: the weight of neighbour j is
// 1/deg(j) instead of 1/deg(i)  ((D^-1 A)^T G, model/gnn.py:60-61 under autograd).
template <int CH, bool TRANSPOSED = false>
__global__ void __launch_bounds__(256)
k_spmm_mean(int64_t n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
            const float* __restrict__ H, int64_t ldh, int nchunks, int tpr, float* __restrict__ out, int64_t ldo) {
    const int rows_per_block = 256 / tpr;
    const int r = threadIdx.x / tpr;
    const int t = threadIdx.x % tpr;
    const int64_t i = (int64_t)blockIdx.x * rows_per_block + r;
    if (r >= rows_per_block || i >= n) return;
    const int b = rowptr[i], e = rowptr[i + 1];
    float4 acc[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e > b) {
        const float inv_row = 1.0f / (float)(e - b);
        int p = b;
        for (; p + 4 <= e; p += 4) {
            int j[4];
            float w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) j[q] = colidx[p + q];
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = TRANSPOSED ? 1.0f / (float)(rowptr[j[q] + 1] - rowptr[j[q]]) : inv_row;
            float4 v[4][CH];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const int c = t + k * tpr;
                    v[q][k] = c < nchunks ? *reinterpret_cast<const float4*>(H + (int64_t)j[q] * ldh + 4 * c)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    acc[k].x = __builtin_fmaf(v[q][k].x, w[q], acc[k].x);
                    acc[k].y = __builtin_fmaf(v[q][k].y, w[q], acc[k].y);
                    acc[k].z = __builtin_fmaf(v[q][k].z, w[q], acc[k].z);
                    acc[k].w = __builtin_fmaf(v[q][k].w, w[q], acc[k].w);
                }
        }
        for (; p < e; ++p) {
            const int j = colidx[p];
            const float inv = TRANSPOSED ? 1.0f / (float)(rowptr[j + 1] - rowptr[j]) : inv_row;
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const int c = t + k * tpr;
                if (c < nchunks) {
                    float4 v = *reinterpret_cast<const float4*>(H + (int64_t)j * ldh + 4 * c);
                    acc[k].x = __builtin_fmaf(v.x, inv, acc[k].x);
                    acc[k].y = __builtin_fmaf(v.y, inv, acc[k].y);
                    acc[k].z = __builtin_fmaf(v.z, inv, acc[k].z);
                    acc[k].w = __builtin_fmaf(v.w, inv, acc[k].w);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int c = t + k * tpr;
        if (c < nchunks) *reinterpret_cast<float4*>(out + i * ldo + 4 * c) = acc[k];
    }
}

// XCD-sliced variant for a dense operand that does not fit one XCD's 4 MB L2: the columns are cut into slabs of `sc` <= 8
// float4 chunks whose n rows fit an L2 (<= 2.5 MB), a group of 8 threads owns a row, and workgroup b works on slab (b % 8) + 8 * round.  Workgroups are dealt
// round-robin to the 8 XCDs, so every gather of a slab is issued from the XCD whose L2 holds that slab after first touch,
// instead of each XCD streaming the whole of H through its L2 from the Infinity Cache (measured before: 0.89 GB across
// the L2 boundary for a 58 MB operand).  A wrong guess about the placement costs speed only.  Same arithmetic per column.
template <int TPR, bool TRANSPOSED = false>
__global__ void __launch_bounds__(256)
k_spmm_mean_xcd(int64_t n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                const float* __restrict__ H, int64_t ldh, int nchunks, int sc, int64_t rowgroups, float* __restrict__ out,
                int64_t ldo) {
    constexpr int CH = 1;                          // sc <= TPR: one float4 chunk per thread
    const int64_t b = blockIdx.x;
    const int64_t round = b / (8 * rowgroups);
    const int slab = (int)(round * 8 + (b & 7));
    const int64_t rg = (b >> 3) % rowgroups;
    const int r = threadIdx.x / TPR, t = threadIdx.x % TPR;
    const int64_t i = rg * (256 / TPR) + r;
    if (i >= n) return;
    const int c0 = slab * sc;
    bool on[CH];
    int cc[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        cc[k] = c0 + t + k * TPR;
        on[k] = (t + k * TPR < sc) && (cc[k] < nchunks);
    }
    const int bgn = rowptr[i], end = rowptr[i + 1];
    float4 acc[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (end > bgn) {
        const float inv_row = 1.0f / (float)(end - bgn);
        int p = bgn;
        for (; p + 4 <= end; p += 4) {
            int j[4];
            float w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) j[q] = colidx[p + q];
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = TRANSPOSED ? 1.0f / (float)(rowptr[j[q] + 1] - rowptr[j[q]]) : inv_row;
            float4 v[4][CH];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int k = 0; k < CH; ++k)
                    v[q][k] = on[k] ? *reinterpret_cast<const float4*>(H + (int64_t)j[q] * ldh + 4 * cc[k])
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    acc[k].x = __builtin_fmaf(v[q][k].x, w[q], acc[k].x);
                    acc[k].y = __builtin_fmaf(v[q][k].y, w[q], acc[k].y);
                    acc[k].z = __builtin_fmaf(v[q][k].z, w[q], acc[k].z);
                    acc[k].w = __builtin_fmaf(v[q][k].w, w[q], acc[k].w);
                }
        }
        for (; p < end; ++p) {
            const int j = colidx[p];
            const float inv = TRANSPOSED ? 1.0f / (float)(rowptr[j + 1] - rowptr[j]) : inv_row;
#pragma unroll
            for (int k = 0; k < CH; ++k)
                if (on[k]) {
                    const float4 v = *reinterpret_cast<const float4*>(H + (int64_t)j * ldh + 4 * cc[k]);
                    acc[k].x = __builtin_fmaf(v.x, inv, acc[k].x);
                    acc[k].y = __builtin_fmaf(v.y, inv, acc[k].y);
                    acc[k].z = __builtin_fmaf(v.z, inv, acc[k].z);
                    acc[k].w = __builtin_fmaf(v.w, inv, acc[k].w);
                }
        }
    }
#pragma unroll
    for (int k = 0; k < CH; ++k)
        if (on[k]) *reinterpret_cast<float4*>(out + i * ldo + 4 * cc[k]) = acc[k];
}

// scalar fallback for widths / strides that are not 16-byte friendly
template <bool TRANSPOSED = false>
__global__ void k_spmm_mean_scalar(int64_t n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                   const float* __restrict__ H, int64_t ldh, int64_t d, float* __restrict__ out, int64_t ldo) {
    const int64_t i = blockIdx.x;
    const int b = rowptr[i], e = rowptr[i + 1];
    const float inv_row = e > b ? 1.0f / (float)(e - b) : 0.0f;
    for (int64_t c = threadIdx.x; c < d; c += blockDim.x) {
        float acc = 0.0f;
        for (int p = b; p < e; ++p) {
            const int j = colidx[p];
            const float inv = TRANSPOSED ? 1.0f / (float)(rowptr[j + 1] - rowptr[j]) : inv_row;
            acc = __builtin_fmaf(H[(int64_t)j * ldh + c], inv, acc);
        }
        out[i * ldo + c] = acc;
    }
}

// ---------------------------------------------------------------- LayerNorm
// One wave per row.  Sums are lane-strided partials followed by an xor butterfly (32,16,...,1): the order the
// oracle's wave_sum64 reproduces.  Three passes over a row that sits in L2; the op is a few MB.
__device__ __forceinline__ float wave_allsum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off);
    return v;
}
__global__ void __launch_bounds__(256)
k_layernorm(int64_t M, int64_t N, float* __restrict__ X, int64_t ldx, const float* __restrict__ gamma,
            const float* __restrict__ beta, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float* x = X + m * ldx;
    float s = 0.0f;
    for (int64_t c = lane; c < N; c += 64) s = s + x[c];
    const float mean = wave_allsum(s) / (float)N;
    float q = 0.0f;
    for (int64_t c = lane; c < N; c += 64) {
        float t = x[c] - mean;
        q = q + t * t;
    }
    const float var = wave_allsum(q) / (float)N;
    const float rstd = 1.0f / __builtin_sqrtf(var + eps);
    for (int64_t c = lane; c < N; c += 64) {
        float t = (x[c] - mean) * rstd;
        float g = t * gamma[c];
        x[c] = g + beta[c];
    }
}

// ---------------------------------------------------------------- movers
// dst[v][0..w) = table[idx[v]][0..w)      (emb_Y, model/gnn.py:155)
__global__ void k_gather_rows(int64_t n, int w, const int32_t* __restrict__ idx, const float* __restrict__ table,
                              int64_t ldt, float* __restrict__ dst, int64_t ldd) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n * w) return;
    const int64_t v = g / w;
    const int c = (int)(g % w);
    dst[v * ldd + c] = table[(int64_t)idx[v] * ldt + c];
}
// dst[v][0..w) = row[0..w)                (h_t.expand, model/gnn.py:64,165)
__global__ void k_broadcast_row(int64_t n, int w, const float* __restrict__ row, float* __restrict__ dst, int64_t ldd) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n * w) return;
    dst[(g / w) * ldd + (g % w)] = row[g % w];
}
// mlp_in_t (model/gnn.py:116-120,153) on the one time row, Linear(1,ht)+ReLU then Linear(ht,ht)+ReLU in gm_linear's
// arithmetic (fmaf chain over ascending k from 0, then + bias), recomputed by every block and written into the h_t
// columns of two row-major buffers: one launch instead of two one-row GEMMs and two broadcasts.
__global__ void __launch_bounds__(256)
k_time_embed(int ht, const float* __restrict__ t, const float* __restrict__ W0, const float* __restrict__ b0,
             const float* __restrict__ W1, const float* __restrict__ b1, int64_t rows, float* __restrict__ dst1, int64_t ld1,
             float* __restrict__ dst2, int64_t ld2) {
    __shared__ float a0[256], a1[256];
    const int tid = threadIdx.x;
    if (tid < ht) {
        float v = __builtin_fmaf(t[0], W0[tid], 0.0f) + b0[tid];
        if (!(v > 0.0f)) v = 0.0f;
        a0[tid] = v;
    }
    __syncthreads();
    if (tid < ht) {
        float acc = 0.0f;
        for (int k = 0; k < ht; ++k) acc = __builtin_fmaf(a0[k], W1[tid * ht + k], acc);
        float v = acc + b1[tid];
        if (!(v > 0.0f)) v = 0.0f;
        a1[tid] = v;
    }
    __syncthreads();
    const int64_t total = rows * ht;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + tid; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = g / ht;
        const int c = (int)(g - r * ht);
        dst1[r * ld1 + c] = a1[c];
        dst2[r * ld2 + c] = a1[c];
    }
}
__global__ void k_copy2d(int64_t n, int w, const float* __restrict__ src, int64_t lds_, float* __restrict__ dst, int64_t ldd) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n * w) return;
    dst[(g / w) * ldd + (g % w)] = src[(g / w) * lds_ + (g % w)];
}

// out[b][0..d) = h[src[b]][0..d) * h[dst[b]][0..d)   (model/gnn.py:246 for an explicit pair list)
__global__ void k_hadamard_gather(int64_t B, int d, const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                  const float* __restrict__ h, int64_t ldh, float* __restrict__ out, int64_t ldo) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= B * d) return;
    const int64_t b = g / d;
    const int c = (int)(g % d);
    out[b * ldo + c] = h[src[b] * ldh + c] * h[dst[b] * ldh + c];
}

int launch_gather_rows(int64_t n, int w, const int32_t* idx, const float* table, int64_t ldt, float* dst, int64_t ldd,
                       hipStream_t s) {
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((n * w + 255) / 256)), dim3(256), 0, s, n, w, idx, table, ldt, dst, ldd);
    GM_LAUNCH_CHECK();
    return GM_OK;
}
int launch_broadcast_row(int64_t n, int w, const float* row, float* dst, int64_t ldd, hipStream_t s) {
    hipLaunchKernelGGL(k_broadcast_row, dim3((unsigned)((n * w + 255) / 256)), dim3(256), 0, s, n, w, row, dst, ldd);
    GM_LAUNCH_CHECK();
    return GM_OK;
}
int launch_time_embed(int ht, const float* t, const float* W0, const float* b0, const float* W1, const float* b1, int64_t rows,
                      float* dst1, int64_t ld1, float* dst2, int64_t ld2, hipStream_t s) {
    GM_REQUIRE(ht > 0 && ht <= 256, "hidden_t out of range for the fused time embedding");
    int64_t blocks = (rows * ht + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_time_embed, dim3((unsigned)blocks), dim3(256), 0, s, ht, t, W0, b0, W1, b1, rows, dst1, ld1, dst2, ld2);
    GM_LAUNCH_CHECK();
    return GM_OK;
}
int launch_copy2d(int64_t n, int w, const float* src, int64_t lds_, float* dst, int64_t ldd, hipStream_t s) {
    hipLaunchKernelGGL(k_copy2d, dim3((unsigned)((n * w + 255) / 256)), dim3(256), 0, s, n, w, src, lds_, dst, ldd);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

}  // namespace gm

using namespace gm;

// n output rows (rowptr[0..n]); hn = rows of the dense operand H (>= n when a rank aggregates only its node block)
template <bool TRANSPOSED>
static int launch_spmm(int64_t n, const int32_t* rowptr, const int32_t* colidx, const float* H, int64_t ldh, int64_t d,
                       float* out, int64_t ldo, hipStream_t s, int64_t hn = 0) {
    if (hn < n) hn = n;
    const bool vec = (d % 4 == 0) && (ldh % 4 == 0) && (ldo % 4 == 0) && aligned(H, 16) && aligned(out, 16);
    const int nchunks = (int)(d / 4);
    // threads per row: a power of two in [4,64]; CH chunks per thread covers the row
    int tpr = 4;
    while (tpr < 64 && tpr < nchunks) tpr <<= 1;
    const int ch = (nchunks + tpr - 1) / tpr;
    if (!vec || ch > 4) {
        hipLaunchKernelGGL((k_spmm_mean_scalar<TRANSPOSED>), dim3((unsigned)n), dim3(vec ? 256 : 128), 0, s, n, rowptr, colidx, H,
                           ldh, d, out, ldo);
        GM_LAUNCH_CHECK();
        return GM_OK;
    }
#ifndef GM_SPMM_NO_XCD   // (exploration builds define it to time the plain kernel)
    // operand larger than an L2, rows few enough that an 8-chunk (128 B wide) slab of all of them fits one (n <= 20480):
    // the leading multiple of 64 chunks goes through the XCD-sliced kernel (8 slabs per round of the 8 XCDs, every lane
    // busy; 6- and 9-chunk slabs measured no better than the plain kernel), the remaining columns through the plain one
    // (and enough output rows to fill the chip with 32-row workgroups per slab: a rank's 1719-row block measured slower)
    if (hn * d * 4 > (6ll << 20) && hn * 128 <= (5ll << 19) && nchunks >= 64 && n >= 4096) {
        const int nx = (nchunks / 64) * 64;
        const int64_t rowgroups = (n + 31) / 32;
        const int64_t blocks = (int64_t)(nx / 8) * rowgroups;
        if (blocks < (1ll << 31)) {
            hipLaunchKernelGGL((k_spmm_mean_xcd<8, TRANSPOSED>), dim3((unsigned)blocks), dim3(256), 0, s, n, rowptr, colidx, H, ldh,
                               nx, 8, rowgroups, out, ldo);
            GM_LAUNCH_CHECK();
            if (nx == nchunks) return GM_OK;
            return launch_spmm<TRANSPOSED>(n, rowptr, colidx, H + 4 * nx, ldh, d - 4 * nx, out + 4 * nx, ldo, s, hn);
        }
    }
#endif
    const int rows_per_block = 256 / tpr;
    dim3 grid((unsigned)((n + rows_per_block - 1) / rows_per_block)), block(256);
    if (ch == 1) hipLaunchKernelGGL((k_spmm_mean<1, TRANSPOSED>), grid, block, 0, s, n, rowptr, colidx, H, ldh, nchunks, tpr, out, ldo);
    else if (ch == 2) hipLaunchKernelGGL((k_spmm_mean<2, TRANSPOSED>), grid, block, 0, s, n, rowptr, colidx, H, ldh, nchunks, tpr, out, ldo);
    else hipLaunchKernelGGL((k_spmm_mean<4, TRANSPOSED>), grid, block, 0, s, n, rowptr, colidx, H, ldh, nchunks, tpr, out, ldo);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

// tower driver: the aggregation of a rank's node block out of all hn rows of H
int gm::launch_spmm_mean_rows(int64_t n, int64_t hn, const int32_t* rowptr, const int32_t* colidx, const float* H, int64_t ldh,
                              int64_t d, float* out, int64_t ldo, hipStream_t s) {
    GM_REQUIRE(n > 0 && d > 0 && rowptr && H && out && ldh >= d && ldo >= d, "bad arguments");
    return launch_spmm<false>(n, rowptr, colidx, H, ldh, d, out, ldo, s, hn);
}

extern "C" int gm_spmm_mean(int64_t n, const int32_t* rowptr, const int32_t* colidx, const float* H, int64_t ldh, int64_t d,
                            float* out, int64_t ldo, gm_stream_t stream) {
    GM_REQUIRE(n > 0 && d > 0 && rowptr && H && out, "bad arguments");
    GM_REQUIRE(ldh >= d && ldo >= d, "leading dimensions smaller than d");
    return launch_spmm<false>(n, rowptr, colidx, H, ldh, d, out, ldo, as_stream(stream));
}

extern "C" int gm_spmm_mean_bwd(int64_t n, const int32_t* rowptr, const int32_t* colidx, const float* G, int64_t ldg, int64_t d,
                                float* out, int64_t ldo, gm_stream_t stream) {
    GM_REQUIRE(n > 0 && d > 0 && rowptr && G && out && ldg >= d && ldo >= d, "bad arguments");
    return launch_spmm<true>(n, rowptr, colidx, G, ldg, d, out, ldo, as_stream(stream));
}

extern "C" int gm_layernorm(int64_t M, int64_t N, float* X, int64_t ldx, const float* gamma, const float* beta, float eps,
                            gm_stream_t stream) {
    GM_REQUIRE(M > 0 && N > 0 && X && gamma && beta && ldx >= N, "bad arguments");
    hipLaunchKernelGGL(k_layernorm, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, as_stream(stream), M, N, X, ldx, gamma, beta, eps);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_hadamard_gather(int64_t B, int64_t d, const int64_t* src, const int64_t* dst, const float* h, int64_t ldh,
                                  float* out, int64_t ldo, gm_stream_t stream) {
    GM_REQUIRE(B > 0 && d > 0 && src && dst && h && out && ldh >= d && ldo >= d, "bad arguments");
    hipLaunchKernelGGL(k_hadamard_gather, dim3((unsigned)((B * d + 255) / 256)), dim3(256), 0, as_stream(stream), B, (int)d, src,
                       dst, h, ldh, out, ldo);
    GM_LAUNCH_CHECK();
    return GM_OK;
}
