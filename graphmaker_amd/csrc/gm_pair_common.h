// gm_pair_common.h -- what the pair kernels (gm_pair.hip: exact f32; gm_pair_bf16.hip: opt-in bf16 MFMA) share: the launch
// arguments, the tile queue decoding, the bounded LDS flag wait, and the host-side argument check / queue set-up.
#pragma once
#include "gm_common.h"
#include "gm_pair_protocol.h"

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
    uint32_t* stall;                // set (cleared by the host only) when an LDS hand-shake timed out: results are then void, and
                                    // a launch that finds it set does nothing (gm_pair_protocol.h)
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

// The bounded hand-shake (gm_pair_protocol.h) over LDS flags.  satisfied(): the first `cnt` ints at `flags` are all >= need
// (every lane returns the same answer).  A wait that gives up raises the workgroup's abort word and the launch's sticky stall
// word; after that no wave of the workgroup waits again and every wave leaves the tile loop at its next check, so a lost
// hand-shake costs one timeout (~30 ms), not one per remaining wait (minutes), and the host learns of it from the stall word.
struct LdsWait {
    lds_flag* flags;
    int cnt, need, lane;
    lds_flag* abort;
    const PairArgs* args;          // the stall word's address is fetched (kernel arguments) only if a wait gives up
    __device__ __forceinline__ bool satisfied() const {
        const int v = lane < cnt ? flags[lane] : 0x7fffffff;
        if (__all(v >= need)) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // later LDS reads stay below the flag read
            return true;
        }
        return false;
    }
    __device__ __forceinline__ bool aborted() const { return *abort != 0; }
    __device__ __forceinline__ void raise() const {
        if (lane == 0) { *abort = 1; atomicOr(args->stall, 1u); }
    }
    __device__ __forceinline__ void pause() const { __builtin_amdgcn_s_sleep(1); }
};
__device__ __forceinline__ bool wait_all_ge(lds_flag* flags, int cnt, int need, int lane, lds_flag* abort, const PairArgs& a) {
    LdsWait w{flags, cnt, need, lane, abort, &a};
    return gm_pair_protocol::bounded_wait(w, gm_pair_protocol::WAIT_SPINS, true);
}
// has any launch that used this workspace reported a stall (and nobody cleared it since)?  Read past the caches: the word
// is written by other workgroups of this launch
__device__ __forceinline__ bool stall_word_set(const uint32_t* stall) {
    return __hip_atomic_load(stall, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}
constexpr int QUEUE_END = 0x7fffffff;   // a queue item no launch has: "stop" (host check: nitems < QUEUE_END)

__device__ __forceinline__ float relu_bits(float x) {   // max(x, 0) for non-NaN x, one integer max
    return __int_as_float(max(__float_as_int(x), 0));
}
// ReLU of two hidden units in ONE instruction (there is no packed f32 max; every vector-ALU instruction costs matrix-pipe time):
// clamp01(x * 2^-63) through the multiply's clamp modifier, against second-layer weights scaled by 2^63 -- oracle relu63: equal
// to the plain ReLU bit for bit for x <= 0 and 2^-63 <= x <= 2^63 (tools/micro/pk_clamp_relu.hip checks the hardware's clamp).
constexpr float RELU_DOWN = 0x1p-63f, RELU_UP = 0x1p63f;
typedef float floatx2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ floatx2 relu63_pk(floatx2 x) {
    floatx2 r;
    const floatx2 s = {RELU_DOWN, RELU_DOWN};
    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(x), "v"(s));
    return r;
}
__device__ __forceinline__ float relu63(float x) {        // the same function on one value (generic kernel)
    const float c = x * RELU_DOWN;
    return !(c > 0.0f) ? 0.0f : (c > 1.0f ? 1.0f : c);
}

// ---- what follows the 128x128 product of a row-unit in both matrix-core kernels (f32: gm_pair.hip, bf16: gm_pair_bf16.hip):
// second layer, softmax, posterior, draw, bitmap word.  acc[strip][o-tile] is the 32x32 MFMA accumulator layout: lane
// (l31 = column = pair, half) holds hidden units o = 32 mt + 8 g + 4 half + r in register 4 g + r of tile mt.
struct EpiCtx {
    const float* dws;       // LDS: (W2[1] - W2[0]) * 2^63 (RELU_UP)
    const float* W2s;       // LDS: both rows of W2, * 2^63 (DBG only)
    const float* tab_s;     // LDS: posterior table as two rows [E_t bit][4] = tab[e], tab[4+e], tab[2+e], tab[6+e]
    uint32_t* rng_mine;     // LDS: this wave's 256 Philox words, [4 row-units][64 columns]
    float db, b20, b21;     // b2[1] - b2[0], b2[0], b2[1]
};
template <bool DBG>
__device__ __forceinline__ void pair_epilogue(const PairArgs& a, const EpiCtx& c, floatx16 (&acc)[2][4], int64_t i, int64_t i0,
                                              int64_t j0, int tj, int unit, int ub, int wv, int lane) {
    const int l31 = lane & 31, half = lane >> 5;
    // ---- second layer as a logit difference d = l1 - l0 = (W2[1] - W2[0]) . relu(hid) + (b2[1] - b2[0])
    // (softmax over two classes needs nothing else): in-lane dot over this lane's 64 hidden units, both strips
    // Even and odd hidden units sit in adjacent accumulator registers, so one v_pk_fma_f32 advances two chains
    // (oracle: C[h][p] over ascending o with ((o>>2)&1)==h, (o&1)==p): 64 packed FMAs per strip instead of 128.
    float part[2];
    float dbg0[2] = {0.0f, 0.0f}, dbg1[2] = {0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        floatx2 dd = {0.0f, 0.0f};
        floatx2 d0 = {0.0f, 0.0f}, d1 = {0.0f, 0.0f};
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ob = mt * 32 + 8 * g + 4 * half;
                const float4 wd = *reinterpret_cast<const float4*>(c.dws + ob);
                const floatx2 wlo = {wd.x, wd.y}, whi = {wd.z, wd.w};
                const floatx2 xlo = relu63_pk(floatx2{acc[s][mt][4 * g + 0], acc[s][mt][4 * g + 1]});
                const floatx2 xhi = relu63_pk(floatx2{acc[s][mt][4 * g + 2], acc[s][mt][4 * g + 3]});
                asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(dd) : "v"(xlo), "v"(wlo));
                asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(dd) : "v"(xhi), "v"(whi));
                if (DBG) {
                    const float4 w0 = *reinterpret_cast<const float4*>(c.W2s + ob);
                    const float4 w1 = *reinterpret_cast<const float4*>(c.W2s + HE + ob);
                    const float w0v[4] = {w0.x, w0.y, w0.z, w0.w};
                    const float w1v[4] = {w1.x, w1.y, w1.z, w1.w};
                    const float xv[4] = {xlo[0], xlo[1], xhi[0], xhi[1]};
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        d0[r4 & 1] = __builtin_fmaf(xv[r4], w0v[r4], d0[r4 & 1]);
                        d1[r4 & 1] = __builtin_fmaf(xv[r4], w1v[r4], d1[r4 & 1]);
                    }
                }
            }
        part[s] = dd[0] + dd[1];
        if (DBG) { dbg0[s] = d0[0] + d0[1]; dbg1[s] = d1[0] + d1[1]; }
    }
    // One v_permlane32_swap: afterwards lanes 0-31 hold (own strip-0 partial, the upper half's strip-0 partial)
    // and lanes 32-63 (the lower half's strip-1 partial, own strip-1 partial), i.e. in both halves
    // (hidden units with bit 2 clear) + (bit 2 set); then the bias difference: the oracle's order
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(part[0]), __float_as_uint(part[1]), false, false);
    const float D = (__uint_as_float(sw[0]) + __uint_as_float(sw[1])) + c.db;
    float L0 = 0.0f, L1 = 0.0f;
    if (DBG) {
        const float r0 = __shfl_xor(half ? dbg0[0] : dbg0[1], 32), r1 = __shfl_xor(half ? dbg1[0] : dbg1[1], 32);
        const float o0 = half ? dbg0[1] : dbg0[0], o1 = half ? dbg1[1] : dbg1[0];
        L0 = ((half ? r0 : o0) + (half ? o0 : r0)) + c.b20;
        L1 = ((half ? r1 : o1) + (half ? o1 : r1)) + c.b21;
    }

    const int64_t j = j0 + lane;                       // lanes 0-31: strip 0, lanes 32-63: strip 1
    const bool valid = ((int)j0 + lane > (int)i) && ((int)j0 + lane < (int)a.n);   // n < 2^31 (host check)
    // words 2 tj and 2 tj + 1 of row i: below wpr for every tile (wpr is a multiple of 4 and >= ceil(n/32), host
    // check), 8-byte aligned, wave-uniform -> one scalar load through the constant address space
    typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(4))) const uintx2 const_uintx2;
    const uintx2 et2 = *(const_uintx2*)(a.bits_t + i * a.wpr + 2 * (int64_t)tj);
    const uint32_t etw = half ? et2[1] : et2[0];
    const int e = (etw >> l31) & 1u;
    float q0, q1;
    softmax2_diff_lean(D, q0, q1);
    const float4 tb = *reinterpret_cast<const float4*>(c.tab_s + 4 * e);   // tab[e], tab[4+e], tab[2+e], tab[6+e]
    const float a0 = q0 * tb.x;
    const float c0 = q1 * tb.y;
    const float a1 = q0 * tb.z;
    const float c1 = q1 * tb.w;
    const float p0 = a0 + c0, p1 = a1 + c1;
    // rank-indexed arrays (host uniforms, logits, posteriors) exist in the DBG instantiation only
    const int64_t rank = DBG ? i * (2 * a.n - i - 1) / 2 + (j - i - 1) : 0;
    float u;
    if (DBG && a.u) u = valid ? a.u[rank] : 1.0f;
    else {
        // One Philox block serves 4 consecutive columns, so per row-unit only 16 of 64 lanes would do
        // distinct work: every fourth unit the wave computes the blocks of its next FOUR units instead
        // (lane = 16 * unit + block) and parks the words in LDS; a wave reads back only its own words.
        uint32_t* mine = c.rng_mine;
        const int ph = (unit - ub) & 3;
        if (ph == 0) {
            const uint32_t ir = (uint32_t)(i0 + wv + 8 * (unit + (lane >> 4)));
            const u32x4 o = philox4x32_10((uint32_t)(j0 >> 2) + (uint32_t)(lane & 15), ir, a.step, STREAM_E,
                                          (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
            *reinterpret_cast<uint4*>(mine + 4 * lane) = make_uint4(o.v[0], o.v[1], o.v[2], o.v[3]);
        }
        u = u01(mine[ph * 64 + lane]);
    }
    const bool bit = valid && draw2(u, p0, p1);
    const unsigned long long m = __ballot(bit);
    if (lane == 0) *reinterpret_cast<unsigned long long*>(a.out_upper + i * a.wpr + 2 * (int64_t)tj) = m;
    if (DBG && valid) {
        if (a.logits_dbg) { a.logits_dbg[2 * rank] = L0; a.logits_dbg[2 * rank + 1] = L1; }
        if (a.post_dbg) { a.post_dbg[2 * rank] = p0; a.post_dbg[2 * rank + 1] = p1; }
    }
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
    GM_REQUIRE(aligned(bits_t, 8) && aligned(out_upper, 8), "bits_t and out_upper must be 8-byte aligned (two words of a row move at once)");
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
    GM_REQUIRE(a.ntiles > 0 && a.ntiles < (1ll << 31) - 8 * 4096, "tile count out of range");   // nitems < QUEUE_END
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
