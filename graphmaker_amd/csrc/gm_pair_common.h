// gm_pair_common.h -- what the pair kernels (gm_pair.hip: exact f32; gm_pair_bf16.hip: opt-in bf16 MFMA) share: the launch
// arguments, the tile queue decoding, the bounded LDS flag wait, and the host-side argument check / queue set-up.
#pragma once
#include "gm_common.h"

namespace gm {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int HE = 128;
constexpr int TILE = 64;          // TI == TJ
constexpr int LDR = HE + 4;       // LDS row stride in floats: 132 -> ds_read_b128 of 16 consecutive rows is conflict-free
constexpr int PAIR_THREADS = 512;

struct PairArgs {
    int64_t n;
    const float* h;
    const float* W1;
    const float* b1;
    const float* W2;
    const float* b2;
    const uint32_t* bits_t;
    int64_t wpr;
    float tab[8];
    // up to MAXR row ranges of the pair matrix in one launch (a rank's folded chunks): range r covers rows
    // [row_begin[r], row_end[r]) and owns tiles [tile_base[r], tile_base[r+1]) of the shared tile counter
    int nranges;
    int64_t row_begin[4], row_end[4];
    int nti[4], tjb[4];
    int64_t tile_base[5];
    const float* u;
    uint64_t seed;
    uint32_t step;
    uint32_t* out_upper;
    float* logits_dbg;
    float* post_dbg;
    uint32_t* counter;
    uint32_t* stall;                // set (never cleared) when an LDS hand-shake below timed out: results are then void
    int ntj;
    int64_t ntiles;
    // work items of the shared queue: items [0, nfull) are whole 64x64 tiles; the remaining tiles (fewer than one per
    // workgroup) are handed out one row-unit per wave at a time (8 items per tile), so the last round of the queue
    // does not leave most CUs idle for a whole tile time
    int64_t nfull, nitems;
};
constexpr int MAXR = 4;

__host__ __device__ __forceinline__ int64_t tile_prefix(int64_t ti, int64_t width) { return ti * width - ti * (ti - 1) / 2; }

// tile index t of the launch -> first row i0 / first column j0 / column-tile tj / end row of the range it belongs to
__device__ __forceinline__ void decode_tile(const PairArgs& a, int64_t t, int64_t& i0, int64_t& j0, int& tj, int64_t& row_end) {
    int r = 0;
#pragma unroll
    for (int q = 1; q < MAXR; ++q)
        if (q < a.nranges && t >= a.tile_base[q]) r = q;
    const int64_t tl = t - a.tile_base[r];
    const int64_t width = (int64_t)a.ntj - a.tjb[r];
    int lo = 0, hi = a.nti[r] - 1;          // largest ti with prefix(ti) <= tl
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tile_prefix(mid, width) <= tl) lo = mid; else hi = mid - 1;
    }
    tj = a.tjb[r] + lo + (int)(tl - tile_prefix(lo, width));
    i0 = a.row_begin[r] + (int64_t)lo * TILE;
    j0 = (int64_t)tj * TILE;
    row_end = a.row_end[r];
}

// queue item -> (tile, first unit, one-past-last unit)
__device__ __forceinline__ void decode_item(const PairArgs& a, int64_t it, int64_t& tile, int& ub, int& ue) {
    if (it < a.nfull) {
        tile = it; ub = 0; ue = TILE / 8;
    } else {
        const int64_t r = it - a.nfull;
        tile = a.nfull + (r >> 3); ub = (int)(r & 7); ue = ub + 1;
    }
}

// Flags in LDS are read and written through LDS-address-space pointers: through a generic `volatile int*` hipcc emits FLAT
// loads and stores (sc0 sc1, each followed by s_waitcnt vmcnt(0)), which take the long way round and make the publishing wave
// wait for its outstanding global stores as well.
typedef __attribute__((address_space(3))) volatile int lds_flag;
__device__ __forceinline__ lds_flag* as_lds_flag(int* p) { return (lds_flag*)p; }

// Wait until the first `cnt` ints at `flags` are all >= need (every lane returns the same answer).  These waits guard
// LDS contents, so they are long; a wave that gives up reports it through PairArgs::stall instead of hanging the GPU.
__device__ __forceinline__ bool wait_all_ge(lds_flag* flags, int cnt, int need, int lane) {
    for (int spin = 0; spin < (1 << 22); ++spin) {
        const int v = lane < cnt ? flags[lane] : 0x7fffffff;
        if (__all(v >= need)) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // later LDS reads stay below the flag read
            return true;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

__device__ __forceinline__ float relu_bits(float x) {   // max(x, 0) for non-NaN x, one integer max
    return __int_as_float(max(__float_as_int(x), 0));
}


}  // namespace gm

// host: validate the arguments of gm_pair_score_sample_ranges and fill PairArgs + grid size (shared by both precisions).
// Returns GM_OK with grid == 0 when there is nothing to do.
int gm_pair_generic_launch(int64_t n, int64_t hE, const float* h, const float* W1, const float* b1, const float* W2,
                           const float* b2, const uint32_t* bits_t, int64_t wpr, const float* tab, int nranges,
                           const int64_t* row_ranges, const float* u, uint64_t seed, uint32_t step, uint32_t* out_upper,
                           float* logits_dbg, float* post_dbg, gm_stream_t stream);

static inline int gm_build_pair_args(int64_t n, int64_t hE, const float* h, const float* W1, const float* b1, const float* W2,
                                     const float* b2, const uint32_t* bits_t, int64_t wpr, const float* tab, int nranges,
                                     const int64_t* row_ranges, const float* u, uint64_t seed, uint32_t step,
                                     uint32_t* out_upper, float* logits_dbg, float* post_dbg, void* workspace,
                                     gm_stream_t stream, gm::PairArgs& a, unsigned& grid) {
    using namespace gm;
    grid = 0;
    GM_REQUIRE(hE == HE, "hidden_E=%lld: the matrix-core pair kernels are built for hidden_E=128", (long long)hE);
    GM_REQUIRE(n > 1 && n < (1ll << 31), "n out of range");
    GM_REQUIRE(wpr >= (n + 31) / 32 && wpr % 4 == 0, "bad wpr");
    GM_REQUIRE(h && W1 && b1 && W2 && b2 && bits_t && tab && out_upper && workspace, "null pointer");
    GM_REQUIRE(aligned(h, 16) && aligned(W1, 16), "h and W1 must be 16-byte aligned");
    GM_REQUIRE(nranges >= 1 && nranges <= MAXR && row_ranges, "between 1 and %d row ranges", MAXR);
    hipStream_t s = as_stream(stream);
    a.n = n; a.h = h; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.bits_t = bits_t; a.wpr = wpr;
    for (int k = 0; k < 8; ++k) a.tab[k] = tab[k];
    a.u = u; a.seed = seed; a.step = step;
    a.out_upper = out_upper; a.logits_dbg = logits_dbg; a.post_dbg = post_dbg;
    a.counter = static_cast<uint32_t*>(workspace);
    a.stall = a.counter + 16;
    a.ntj = (int)((n + TILE - 1) / TILE);
    a.nranges = 0;
    a.tile_base[0] = 0;
    for (int r = 0; r < MAXR; ++r) { a.row_begin[r] = a.row_end[r] = 0; a.nti[r] = 0; a.tjb[r] = 0; a.tile_base[r + 1] = 0; }
    int64_t prev_end = 0;
    for (int r = 0; r < nranges; ++r) {
        const int64_t rb = row_ranges[2 * r], re = row_ranges[2 * r + 1];
        GM_REQUIRE(0 <= rb && rb <= re && re <= n, "bad row range [%lld,%lld)", (long long)rb, (long long)re);
        GM_REQUIRE(rb >= prev_end, "row ranges must be ascending and disjoint");   // two tiles never write one row word
        prev_end = re;
        if (rb == re) continue;
        const int q = a.nranges++;
        a.row_begin[q] = rb; a.row_end[q] = re;
        // rows beyond the last one that has a partner (i = n-1) still get their (all-zero) words written
        a.nti[q] = (int)((re - rb + TILE - 1) / TILE);
        a.tjb[q] = (int)(rb / TILE);
        a.tile_base[q + 1] = a.tile_base[q] + tile_prefix(a.nti[q], (int64_t)a.ntj - a.tjb[q]);
    }
    if (a.nranges == 0) return GM_OK;
    for (int q = a.nranges; q < MAXR; ++q) a.tile_base[q + 1] = a.tile_base[a.nranges];
    a.ntiles = a.tile_base[a.nranges];
    GM_REQUIRE(a.ntiles > 0 && a.ntiles < (1ll << 31), "tile count out of range");
    int dev = 0, cus = 0;
    GM_HIP(hipGetDevice(&dev));
    GM_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    GM_REQUIRE(cus > 0, "no compute units reported");
    GM_HIP(hipMemsetAsync(a.counter, 0, sizeof(uint32_t), s));
    a.nfull = (a.ntiles / cus) * cus;
    a.nitems = a.nfull + 8 * (a.ntiles - a.nfull);
    grid = (unsigned)(a.nitems < cus ? a.nitems : cus);   // ~137 KB of LDS: one workgroup per CU
    return GM_OK;
}
