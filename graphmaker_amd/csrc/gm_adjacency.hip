// gm_adjacency.hip -- adjacency state as a symmetric bitmap + CSR construction.
// Replaces the dense (N,N) int64 E_t of model/diffusion.py:422-424 and get_adj (model/diffusion.py:239-256):
// `E_t.nonzero()` enumerates row-major, so a CSR with ascending columns is the same edge order; the values of
// D^-1 A are 1/deg[row] and stay implicit.  All kernels are HBM-bound bit/integer work.
#include "gm_common.h"

namespace gm {

// ---------------------------------------------------------------- dense <-> bitmap
// one wave per 64 consecutive columns of one row; __ballot packs 64 compare results into two words.
__global__ void k_bitmap_from_dense(int64_t n, int64_t wpr, const int64_t* __restrict__ dense,
                                    uint32_t* __restrict__ bits) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t chunks = wpr >> 1;  // 64-column chunks per row (wpr is even)
    if (wave >= n * chunks) return;
    const int64_t i = wave / chunks, c = wave % chunks;
    const int64_t j = c * 64 + lane;
    bool nz = (j < n) && (dense[i * n + j] != 0);
    unsigned long long m = __ballot(nz);
    if (lane == 0) {
        bits[i * wpr + 2 * c] = (uint32_t)m;
        bits[i * wpr + 2 * c + 1] = (uint32_t)(m >> 32);
    }
}

__global__ void k_dense_from_bitmap(int64_t n, int64_t wpr, const uint32_t* __restrict__ bits,
                                    int64_t* __restrict__ dense) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    const int64_t i = idx / n, j = idx % n;
    dense[idx] = (bits[i * wpr + (j >> 5)] >> (j & 31)) & 1u;
}

// ---------------------------------------------------------------- symmetrize
// One block per 32-row tile I; each half-wave owns one output word (i = 32I + lane&31, w) per iteration.
// Bits j >= i come from upper[i][w]; bits j < i are the transpose of the 32x32 bit tile (rows 32w.., word I):
// every lane loads one word of that tile and 32 ballots transpose it in registers (lane c keeps ballot number c).
__global__ void __launch_bounds__(256)
k_symmetrize(int64_t n, int64_t wpr, const uint32_t* __restrict__ upper, uint32_t* __restrict__ sym) {
    const int lane = threadIdx.x & 63;
    const int li = lane & 31;
    const int half = lane >> 5;
    const int64_t I = blockIdx.x;  // row tile
    const int64_t i = I * 32 + li;
    const int wavesPerBlock = blockDim.x >> 6;
    const int wv = threadIdx.x >> 6;
    // blockIdx.y cuts the word range into gridDim.y contiguous pieces (a row tile alone is too little latency-bound work
    // per CU); a block walks its piece 8 words = 32 bytes of every row at a time, so the four visits to a 128-byte line
    // come from one CU back to back instead of from four workgroups on different XCDs
    const int64_t per_it = 2 * wavesPerBlock * (int64_t)gridDim.y;
    const int64_t wsteps = (wpr + per_it - 1) / per_it;
    for (int64_t it = 0; it < wsteps; ++it) {            // uniform trip count: the ballots need every lane
        const int64_t w = (blockIdx.y * wsteps + it) * 2 * wavesPerBlock + wv * 2 + half;
        const bool inw = w < wpr;
        uint32_t own = 0, src = 0;
        if (inw && i < n && w >= I && w * 32 < n) {
            own = upper[i * wpr + w];
            if (w == I) own &= ~((1u << li) - 1u);       // keep j >= i
        }
        const int64_t jr = w * 32 + li;                   // row of the source tile this lane fetches
        if (inw && w <= I && jr < n) src = upper[jr * wpr + I];
        uint32_t trans = 0;
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const unsigned long long m = __ballot((src >> c) & 1u);
            const uint32_t mine = half ? (uint32_t)(m >> 32) : (uint32_t)m;
            if (li == c) trans = mine;
        }
        if (w == I) trans &= (1u << li) - 1u;             // keep j < i (w < I: all 32 rows are below i)
        uint32_t res = own | trans;
        const int64_t rem = n - w * 32;                   // clear padding bits
        if (rem <= 0) res = 0;
        else if (rem < 32) res &= (1u << rem) - 1u;
        if (inw && i < n) sym[i * wpr + w] = res;
    }
}

// Large graphs: one block per 512x512-bit tile (RB, CB >= RB) of the upper triangle, staged through LDS so that both the
// tile and its transpose leave as 64-byte row segments (the kernel above reads and writes 8 bytes per row and wave).
// The 32x32 sub-tiles are transposed across the lanes of a half-wave by the 5-stage butterfly (exchange with lane^s the
// half-blocks of s bits, s = 16..1): lane c ends up with column c.
constexpr int SYM_T = 512, SYM_W = SYM_T / 32, SYM_LD = SYM_W + 1;

__device__ __forceinline__ uint32_t transpose32_across_lanes(uint32_t x, int li) {
    const uint32_t masks[5] = {0x0000FFFFu, 0x00FF00FFu, 0x0F0F0F0Fu, 0x33333333u, 0x55555555u};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int sft = 16 >> k;
        const uint32_t m = masks[k];
        const uint32_t y = (uint32_t)__shfl_xor((int)x, sft);
        x = (li & sft) ? (((y >> sft) & m) | (x & ~m)) : ((x & m) | ((y & m) << sft));
    }
    return x;
}

__global__ void __launch_bounds__(256)
k_symmetrize_tiled(int64_t n, int64_t wpr, int nT, const uint32_t* __restrict__ upper, uint32_t* __restrict__ sym) {
    __shared__ uint32_t Us[SYM_T * SYM_LD];
    __shared__ uint32_t Ts[SYM_T * SYM_LD];
    const int tid = threadIdx.x;
    // block -> (RB, CB): row RB owns nT - RB tiles
    const int64_t t = blockIdx.x;
    int lo = 0, hi = nT - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int64_t)mid * nT - (int64_t)mid * (mid - 1) / 2 <= t) lo = mid; else hi = mid - 1;
    }
    const int RB = lo, CB = lo + (int)(t - ((int64_t)lo * nT - (int64_t)lo * (lo - 1) / 2));
    const bool diag = RB == CB;
#pragma unroll
    for (int k = 0; k < SYM_T * SYM_W / 4 / 256; ++k) {
        const int q = tid + 256 * k, r = q >> 2, c4 = q & 3;
        const int64_t gi = (int64_t)RB * SYM_T + r, w0 = (int64_t)CB * SYM_W + 4 * c4;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        const bool in = gi < n && w0 < wpr;
        if (in) v = *reinterpret_cast<const uint4*>(upper + gi * wpr + w0);
        uint32_t wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (diag) {                                   // keep j >= i
                const int lw = 4 * c4 + e, rI = r >> 5;
                if (lw < rI) wd[e] = 0u;
                else if (lw == rI) wd[e] &= ~((1u << (r & 31)) - 1u);
            }
            const int64_t rem = n - (w0 + e) * 32;        // clear padding columns
            if (rem <= 0) wd[e] = 0u;
            else if (rem < 32) wd[e] &= (1u << rem) - 1u;
            Us[r * SYM_LD + 4 * c4 + e] = wd[e];
        }
        if (!diag && in) *reinterpret_cast<uint4*>(sym + gi * wpr + w0) = make_uint4(wd[0], wd[1], wd[2], wd[3]);
    }
    __syncthreads();
    const int hw = tid >> 5, li = tid & 31;
    for (int it = 0; it < SYM_W * SYM_W / 8; ++it) {
        const int st = it * 8 + hw, a = st / SYM_W, b = st % SYM_W;       // rows 32a.. of word b -> rows 32b.. of word a
        const uint32_t x = transpose32_across_lanes(Us[(32 * a + li) * SYM_LD + b], li);
        Ts[(32 * b + li) * SYM_LD + a] = x;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SYM_T * SYM_W / 4 / 256; ++k) {
        const int q = tid + 256 * k, j = q >> 2, c4 = q & 3;
        const int64_t gj = (int64_t)CB * SYM_T + j, w0 = (int64_t)RB * SYM_W + 4 * c4;
        if (gj >= n || w0 >= wpr) continue;
        uint32_t wd[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            wd[e] = Ts[j * SYM_LD + 4 * c4 + e];
            if (diag) wd[e] |= Us[j * SYM_LD + 4 * c4 + e];
        }
        *reinterpret_cast<uint4*>(sym + gj * wpr + w0) = make_uint4(wd[0], wd[1], wd[2], wd[3]);
    }
}

// ---------------------------------------------------------------- CSR
__device__ __forceinline__ uint32_t masked_word(const uint32_t* bits, int64_t wpr, int64_t i, int64_t w, int upper_only) {
    uint32_t word = bits[i * wpr + w];
    if (upper_only) {
        const int64_t I = i >> 5;
        if (w < I) word = 0;
        else if (w == I) {
            const int li = (int)(i & 31);
            word = (li == 31) ? 0u : (word & ~((2u << li) - 1u));  // keep j > i
        }
    }
    return word;
}

// degree per row: one wave per row
__global__ void k_row_degree(int64_t n, int64_t wpr, const uint32_t* __restrict__ bits, int upper_only,
                             int32_t* __restrict__ deg) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    int cnt = 0;
    for (int64_t w = lane; w < wpr; w += 64) cnt += __popc(masked_word(bits, wpr, i, w, upper_only));
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (lane == 0) deg[i] = cnt;
}

// exclusive scan of deg[0..n) into rowptr[0..n], in place, single block of 1024 threads: every thread sums a
// contiguous run of ceil(n/1024) entries, the 1024 sums are scanned once, and the run is rewritten (2 barriers in all).
// capacity >= 0: every entry is clamped to it and *overflow |= (total > capacity).  With a caller-owned colidx buffer of
// fixed capacity no host read of the edge count is needed: every consumer of (rowptr, colidx) stays inside the buffer,
// and the flag tells the host later that a result built on the truncated list must be discarded.
__global__ void __launch_bounds__(1024) k_exclusive_scan(int64_t n, int32_t* __restrict__ rowptr, int64_t capacity,
                                                         int32_t* __restrict__ overflow) {
    __shared__ int32_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t c = (n + 1023) / 1024;
    const int64_t b = tid * c < n ? tid * c : n, e = b + c < n ? b + c : n;
    int32_t sum = 0;
    for (int64_t i = b; i < e; i += 8) {               // 8 independent loads in flight, not one dependent load per entry
        int32_t v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = i + q < e ? rowptr[i + q] : 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) sum += v[q];
    }
    int32_t inc = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int32_t woff = 0, total = 0;
    for (int k = 0; k < 16; ++k) {
        if (k < wv) woff += wsum[k];
        total += wsum[k];
    }
    const int32_t cap = capacity >= 0 ? (int32_t)capacity : 0x7fffffff;
    int32_t run = woff + inc - sum;
    for (int64_t i = b; i < e; i += 8) {
        int32_t v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = i + q < e ? rowptr[i + q] : 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (i + q < e) rowptr[i + q] = run < cap ? run : cap;
            run += v[q];
        }
    }
    if (tid == 0) {
        rowptr[n] = total < cap ? total : cap;
        if (total > cap) atomicOr(overflow, 1);
    }
}

// column indices in ascending order: one wave per row, lanes take words, in-wave prefix of popcounts
__global__ void k_csr_fill(int64_t n, int64_t wpr, const uint32_t* __restrict__ bits, int upper_only,
                           const int32_t* __restrict__ rowptr, int64_t capacity, int32_t* __restrict__ colidx,
                           int32_t* __restrict__ rows, int32_t row_base = 0) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    int64_t base = rowptr[i];
    for (int64_t w0 = 0; w0 < wpr; w0 += 64) {
        const int64_t w = w0 + lane;
        uint32_t word = w < wpr ? masked_word(bits, wpr, i, w, upper_only) : 0u;
        int cnt = __popc(word);
        int inc = cnt;
        for (int off = 1; off < 64; off <<= 1) {
            int t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        int64_t pos = base + inc - cnt;
        while (word) {
            const int b = __ffs(word) - 1;
            word &= word - 1;
            if (pos < capacity) {
                colidx[pos] = (int32_t)(w * 32 + b);
                if (rows) rows[pos] = (int32_t)i + row_base;
            }
            ++pos;
        }
        base += __shfl(inc, 63);
    }
}

__global__ void k_bitmap_from_edges(int64_t n, int64_t wpr, int64_t m, const int32_t* __restrict__ ei,
                                    const int32_t* __restrict__ ej, uint32_t* __restrict__ bits) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m) return;
    const int64_t i = ei[e], j = ej[e];
    if (i < 0 || j < 0 || i >= n || j >= n) return;
    atomicOr(&bits[i * wpr + (j >> 5)], 1u << (j & 31));
    atomicOr(&bits[j * wpr + (i >> 5)], 1u << (i & 31));
}

// every rank's segment of the gathered edge lists -> symmetric bitmap: segment s = [ei (cap) | ej (cap) | count | overflowed | pad].
// A segment whose rank had more edges than `cap` arrives truncated WITH its rank's overflow word: every rank that rebuilds the
// bitmap raises its own sticky flag then, so all ranks reject the sample together (one rank raising alone would leave the
// others in the next collective).
__global__ void k_bitmap_from_edge_segments(int64_t n, int64_t wpr, int nseg, int64_t cap, int64_t stride,
                                            const int32_t* __restrict__ edges, uint32_t* __restrict__ bits,
                                            int32_t* __restrict__ overflow) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (int64_t)nseg * cap) return;
    const int64_t s = g / cap, e = g - s * cap;
    const int32_t* seg = edges + s * stride;
    int64_t cnt = seg[2 * cap];
    if (e == 0 && overflow && (seg[2 * cap + 1] != 0 || cnt > cap || cnt < 0)) atomicOr(overflow, 1);
    if (cnt > cap) cnt = cap;
    if (e >= cnt) return;
    const int64_t i = seg[e], j = seg[cap + e];
    if (i < 0 || j < 0 || i >= n || j >= n) return;
    atomicOr(&bits[i * wpr + (j >> 5)], 1u << (j & 31));
    atomicOr(&bits[j * wpr + (i >> 5)], 1u << (i & 31));
}

}  // namespace gm

using namespace gm;

static int check_bitmap_args(int64_t n, int64_t wpr) {
    GM_REQUIRE(n > 0 && n < (1ll << 31), "n=%lld out of range", (long long)n);
    GM_REQUIRE(wpr >= (n + 31) / 32 && (wpr % 4) == 0, "wpr=%lld must be a multiple of 4 and >= ceil(n/32)", (long long)wpr);
    return GM_OK;
}

extern "C" int gm_bitmap_from_dense_i64(int64_t n, int64_t wpr, const int64_t* dense, uint32_t* bits, gm_stream_t stream) {
    if (int rc = check_bitmap_args(n, wpr)) return rc;
    GM_REQUIRE(dense && bits, "null pointer");
    const int64_t waves = n * (wpr / 2);
    const int64_t blocks = (waves + 3) / 4;
    hipLaunchKernelGGL(k_bitmap_from_dense, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), n, wpr, dense, bits);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_dense_i64_from_bitmap(int64_t n, int64_t wpr, const uint32_t* bits, int64_t* dense, gm_stream_t stream) {
    if (int rc = check_bitmap_args(n, wpr)) return rc;
    GM_REQUIRE(dense && bits, "null pointer");
    const int64_t blocks = (n * n + 255) / 256;
    GM_REQUIRE(blocks < (1ll << 31), "n too large for a dense materialisation");
    hipLaunchKernelGGL(k_dense_from_bitmap, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), n, wpr, bits, dense);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_symmetrize_upper(int64_t n, int64_t wpr, const uint32_t* upper, uint32_t* sym, gm_stream_t stream) {
    if (int rc = check_bitmap_args(n, wpr)) return rc;
    GM_REQUIRE(upper && sym && upper != sym, "null or aliased pointer");
    const int64_t nT = (n + SYM_T - 1) / SYM_T;
    // >= 136 tiles, and a row no wider than the tiles cover: the LDS-staged kernel
    if (nT >= 16 && nT < 40000 && wpr <= nT * SYM_W && aligned(upper, 16) && aligned(sym, 16)) {
        hipLaunchKernelGGL(k_symmetrize_tiled, dim3((unsigned)(nT * (nT + 1) / 2)), dim3(256), 0, as_stream(stream), n, wpr,
                           (int)nT, upper, sym);
        GM_LAUNCH_CHECK();
        return GM_OK;
    }
    const int64_t tiles = (n + 31) / 32;
    int64_t split = (4096 + tiles - 1) / tiles;          // ~4096 workgroups
    if (split > (wpr + 7) / 8) split = (wpr + 7) / 8;
    if (split < 1) split = 1;
    hipLaunchKernelGGL(k_symmetrize, dim3((unsigned)tiles, (unsigned)split), dim3(256), 0, as_stream(stream), n, wpr, upper, sym);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_csr_rowptr(int64_t n, int64_t wpr, const uint32_t* bits, int upper_only, int32_t* rowptr, gm_stream_t stream) {
    if (int rc = check_bitmap_args(n, wpr)) return rc;
    GM_REQUIRE(bits && rowptr, "null pointer");
    hipLaunchKernelGGL(k_row_degree, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, as_stream(stream), n, wpr, bits, upper_only, rowptr);
    GM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, as_stream(stream), n, rowptr, (int64_t)-1, (int32_t*)nullptr);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_csr_rowptr_capped(int64_t n, int64_t wpr, const uint32_t* bits, int upper_only, int64_t capacity,
                                    int32_t* rowptr, int32_t* overflow, gm_stream_t stream) {
    GM_REQUIRE(capacity > 0 && capacity < (1ll << 31) && overflow, "bad capacity or null overflow flag");
    if (int rc = check_bitmap_args(n, wpr)) return rc;
    GM_REQUIRE(bits && rowptr, "null pointer");
    hipLaunchKernelGGL(k_row_degree, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, as_stream(stream), n, wpr, bits, upper_only, rowptr);
    GM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, as_stream(stream), n, rowptr, capacity, overflow);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_csr_fill(int64_t n, int64_t wpr, const uint32_t* bits, int upper_only, const int32_t* rowptr,
                           int64_t capacity, int32_t* colidx, int32_t* rows, gm_stream_t stream) {
    if (int rc = check_bitmap_args(n, wpr)) return rc;
    GM_REQUIRE(bits && rowptr && (colidx || capacity == 0), "null pointer");
    if (capacity == 0) return GM_OK;
    hipLaunchKernelGGL(k_csr_fill, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, as_stream(stream), n, wpr, bits, upper_only,
                       rowptr, capacity, colidx, rows);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_bitmap_from_upper_edges(int64_t n, int64_t wpr, int64_t m, const int32_t* ei, const int32_t* ej,
                                          uint32_t* bits, gm_stream_t stream) {
    if (int rc = check_bitmap_args(n, wpr)) return rc;
    GM_REQUIRE(bits && m >= 0 && (m == 0 || (ei && ej)), "bad edge list");
    GM_HIP(hipMemsetAsync(bits, 0, sizeof(uint32_t) * (size_t)(n * wpr), as_stream(stream)));
    if (m == 0) return GM_OK;
    hipLaunchKernelGGL(k_bitmap_from_edges, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, as_stream(stream), n, wpr, m, ei, ej, bits);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

// ---- multi-GPU: the sampled edges of a rank's rows as a compact list, and the bitmap back from every rank's list
extern "C" int gm_upper_edges_ranges(int64_t n, int64_t wpr, const uint32_t* upper, int nranges, const int64_t* row_ranges,
                                     int64_t capacity, int32_t* rowptr_ws, int32_t* ei, int32_t* ej, int32_t* count,
                                     int32_t* overflow, gm_stream_t stream) {
    if (int rc = check_bitmap_args(n, wpr)) return rc;
    GM_REQUIRE(upper && rowptr_ws && ei && ej && count && overflow && row_ranges, "null pointer");
    GM_REQUIRE(nranges >= 1 && nranges <= 4, "between 1 and 4 row ranges");
    GM_REQUIRE(capacity > 0 && capacity < (1ll << 31), "bad capacity");
    hipStream_t s = as_stream(stream);
    int64_t total = 0;
    for (int r = 0; r < nranges; ++r) {
        const int64_t b = row_ranges[2 * r], e = row_ranges[2 * r + 1];
        GM_REQUIRE(0 <= b && b <= e, "bad row range [%lld,%lld)", (long long)b, (long long)e);
        if (e == b) continue;
        // `upper` holds only bits j > i in the words the pair kernel writes and zeros elsewhere: plain row degrees
        hipLaunchKernelGGL(k_row_degree, dim3((unsigned)((e - b + 3) / 4)), dim3(256), 0, s, e - b, wpr, upper + b * wpr, 0,
                           rowptr_ws + total);
        GM_LAUNCH_CHECK();
        total += e - b;
    }
    if (total == 0) {
        GM_HIP(hipMemsetAsync(count, 0, sizeof(int32_t), s));
        GM_HIP(hipMemcpyAsync(count + 1, overflow, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        return GM_OK;
    }
    hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, s, total, rowptr_ws, capacity, overflow);
    GM_LAUNCH_CHECK();
    GM_HIP(hipMemcpyAsync(count, rowptr_ws + total, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    // the (sticky) overflow word travels with the segment: the count above is clamped to the capacity, so without it the
    // ranks that receive a truncated list could not tell
    GM_HIP(hipMemcpyAsync(count + 1, overflow, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    int64_t off = 0;
    for (int r = 0; r < nranges; ++r) {
        const int64_t b = row_ranges[2 * r], e = row_ranges[2 * r + 1];
        if (e == b) continue;
        hipLaunchKernelGGL(k_csr_fill, dim3((unsigned)((e - b + 3) / 4)), dim3(256), 0, s, e - b, wpr, upper + b * wpr, 0,
                           rowptr_ws + off, capacity, ej, ei, (int32_t)b);
        GM_LAUNCH_CHECK();
        off += e - b;
    }
    return GM_OK;
}

extern "C" int gm_bitmap_from_edge_segments(int64_t n, int64_t wpr, int nseg, int64_t cap, int64_t stride,
                                            const int32_t* edges, uint32_t* bits, int32_t* overflow, gm_stream_t stream) {
    if (int rc = check_bitmap_args(n, wpr)) return rc;
    GM_REQUIRE(nseg > 0 && cap > 0 && stride >= 2 * cap + 2 && edges && bits, "bad arguments");
    GM_REQUIRE((int64_t)nseg * cap < (1ll << 38), "edge buffer too large");
    GM_HIP(hipMemsetAsync(bits, 0, sizeof(uint32_t) * (size_t)(n * wpr), as_stream(stream)));
    hipLaunchKernelGGL(k_bitmap_from_edge_segments, dim3((unsigned)(((int64_t)nseg * cap + 255) / 256)), dim3(256), 0,
                       as_stream(stream), n, wpr, nseg, cap, stride, edges, bits, overflow);
    GM_LAUNCH_CHECK();
    return GM_OK;
}
