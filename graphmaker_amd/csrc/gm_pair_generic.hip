// gm_pair_generic.hip -- the all-pairs structure denoiser for hidden_E != 128 (model/gnn.py:204-231 takes any hidden_E; every
// shipped configuration uses 128, which runs the MFMA kernel of gm_pair.hip).  Same loop (model/diffusion.py:444-470), same
// arithmetic orders as oracle gmo_pair_score_sample -- hidden[o] = fmaf chain over ascending k from b1[o]; logit difference =
// four fmaf chains on W2[1]-W2[0] over ascending o split by bits 2 and 0 of o, summed pairwise, + (b2[1]-b2[0]) -- on the vector ALU:
// one wave per (row i, 64 consecutive columns j), one pair per lane.  The wave's z = h_i * h_j values sit in LDS as
// z[k][lane]; W1 rows are wave-uniform, so they come through the scalar cache.  VALU-bound (f32 vector peak = f32 MFMA peak on
// this chip, but every FMA also needs its z operand from LDS): a correct path for unusual configurations, not a tuned one.
#include "gm_pair_common.h"

namespace gm {

constexpr int GEN_MAX_HE = 512;

struct GenArgs {
    int64_t n, wpr;
    int hE;
    const float* h;
    const float* W1;
    const float* b1;
    const float* W2;
    const float* b2;
    const uint32_t* bits_t;
    float tab[8];
    int nranges;
    int64_t row_begin[MAXR], row_end[MAXR], unit_base[MAXR + 1];   // units = (row, 64-column chunk) with chunk >= row / 64
    int64_t nchunks;
    const float* u;
    uint64_t seed;
    uint32_t step;
    uint32_t* out_upper;
    float* logits_dbg;
    float* post_dbg;
};

// rows [rb, re): row i owns chunks i/64 .. nchunks-1; prefix over rows in closed form
__host__ __device__ __forceinline__ int64_t units_before_row(int64_t rb, int64_t i, int64_t nchunks) {
    // sum_{r=rb}^{i-1} (nchunks - r/64)
    auto tri = [](int64_t x) {   // sum_{r=0}^{x-1} r/64
        const int64_t q = x >> 6, r = x & 63;
        return 64 * (q * (q - 1) / 2) + q * r;
    };
    return (i - rb) * nchunks - (tri(i) - tri(rb));
}

__global__ void __launch_bounds__(256)
k_pair_generic(const GenArgs a) {
    extern __shared__ float zs[];                       // [waves][hE][64]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t unit = (int64_t)blockIdx.x * (blockDim.x >> 6) + wv;
    if (unit >= a.unit_base[a.nranges]) return;         // whole waves leave together: no barrier below
    int r = 0;
#pragma unroll
    for (int q = 1; q < MAXR; ++q)
        if (q < a.nranges && unit >= a.unit_base[q]) r = q;
    const int64_t ul = unit - a.unit_base[r], rb = a.row_begin[r], re = a.row_end[r];
    int64_t lo = rb, hi = re - 1;                       // largest row i with units_before_row(i) <= ul
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (units_before_row(rb, mid, a.nchunks) <= ul) lo = mid; else hi = mid - 1;
    }
    const int64_t i = lo;
    const int64_t chunk = (i >> 6) + (ul - units_before_row(rb, i, a.nchunks));
    const int64_t j = chunk * 64 + lane;
    const bool valid = j > i && j < a.n;
    const int hE = a.hE;
    float* z = zs + (size_t)wv * hE * 64;
    const float* hi_row = a.h + i * hE;
    const float* hj_row = a.h + (j < a.n ? j : a.n - 1) * hE;
    for (int k = 0; k < hE; ++k) z[k * 64 + lane] = hi_row[k] * hj_row[k];   // a lane reads back only what it wrote
    float sd[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}}, t0[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}}, t1[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
    const bool dbg = a.logits_dbg != nullptr;
    for (int o = 0; o < hE; ++o) {
        const float* w = a.W1 + (int64_t)o * hE;        // wave-uniform row
        float acc = a.b1[o];
        for (int k = 0; k < hE; ++k) acc = __builtin_fmaf(z[k * 64 + lane], w[k], acc);
        const float hid = relu63(acc);                            // oracle relu63: against weights scaled by 2^63
        const float dw = (a.W2[hE + o] - a.W2[o]) * RELU_UP;
        const int hs = (o >> 2) & 1, par = o & 1;                 // wave-uniform
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
                if (hs == x && par == y) {
                    sd[x][y] = __builtin_fmaf(hid, dw, sd[x][y]);
                    if (dbg) {
                        t0[x][y] = __builtin_fmaf(hid, a.W2[o] * RELU_UP, t0[x][y]);
                        t1[x][y] = __builtin_fmaf(hid, a.W2[hE + o] * RELU_UP, t1[x][y]);
                    }
                }
    }
    const float D = ((sd[0][0] + sd[0][1]) + (sd[1][0] + sd[1][1])) + (a.b2[1] - a.b2[0]);
    const int64_t wi = j >> 5;
    uint32_t etw = 0;
    if (wi < a.wpr) etw = a.bits_t[i * a.wpr + wi];
    const bool e = (etw >> (j & 31)) & 1u;
    float q0, q1;
    softmax2_diff(D, q0, q1);
    const float a0 = q0 * (e ? a.tab[1] : a.tab[0]);
    const float c0 = q1 * (e ? a.tab[5] : a.tab[4]);
    const float a1 = q0 * (e ? a.tab[3] : a.tab[2]);
    const float c1 = q1 * (e ? a.tab[7] : a.tab[6]);
    const float p0 = a0 + c0, p1 = a1 + c1;
    const int64_t rank = i * (2 * a.n - i - 1) / 2 + (j - i - 1);
    float u;
    if (a.u) u = valid ? a.u[rank] : 1.0f;
    else u = u_pair(a.seed, a.step, (uint32_t)i, (uint32_t)j);
    const bool bit = valid && draw2(u, p0, p1);
    const unsigned long long m = __ballot(bit);
    if (lane == 0) {
        const int64_t w0 = 2 * chunk;
        if (w0 < a.wpr) a.out_upper[i * a.wpr + w0] = (uint32_t)m;
        if (w0 + 1 < a.wpr) a.out_upper[i * a.wpr + w0 + 1] = (uint32_t)(m >> 32);
    }
    if (valid) {
        if (dbg) {
            a.logits_dbg[2 * rank] = ((t0[0][0] + t0[0][1]) + (t0[1][0] + t0[1][1])) + a.b2[0];
            a.logits_dbg[2 * rank + 1] = ((t1[0][0] + t1[0][1]) + (t1[1][0] + t1[1][1])) + a.b2[1];
        }
        if (a.post_dbg) { a.post_dbg[2 * rank] = p0; a.post_dbg[2 * rank + 1] = p1; }
    }
}

}  // namespace gm

using namespace gm;

// called by gm_pair_score_sample_ranges for hidden_E != 128 (same contract; `workspace` is not used)
int gm_pair_generic_launch(int64_t n, int64_t hE, const float* h, const float* W1, const float* b1, const float* W2,
                           const float* b2, const uint32_t* bits_t, int64_t wpr, const float* tab, int nranges,
                           const int64_t* row_ranges, const float* u, uint64_t seed, uint32_t step, uint32_t* out_upper,
                           float* logits_dbg, float* post_dbg, gm_stream_t stream) {
    GM_REQUIRE(hE >= 1 && hE <= GEN_MAX_HE, "hidden_E=%lld: supported are 128 (matrix-core kernel) and 1..%d (vector kernel)",
               (long long)hE, GEN_MAX_HE);
    GM_REQUIRE(n > 1 && n < (1ll << 31), "n out of range");
    GM_REQUIRE(wpr >= (n + 31) / 32 && wpr % 4 == 0, "bad wpr");
    GM_REQUIRE(h && W1 && b1 && W2 && b2 && bits_t && tab && out_upper, "null pointer");
    GM_REQUIRE(nranges >= 1 && nranges <= MAXR && row_ranges, "between 1 and %d row ranges", MAXR);
    GenArgs a;
    a.n = n; a.wpr = wpr; a.hE = (int)hE; a.h = h; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.bits_t = bits_t;
    for (int k = 0; k < 8; ++k) a.tab[k] = tab[k];
    a.u = u; a.seed = seed; a.step = step; a.out_upper = out_upper; a.logits_dbg = logits_dbg; a.post_dbg = post_dbg;
    a.nchunks = (n + 63) / 64;
    a.nranges = 0;
    a.unit_base[0] = 0;
    int64_t prev_end = 0;
    for (int r = 0; r < MAXR; ++r) { a.row_begin[r] = a.row_end[r] = 0; a.unit_base[r + 1] = 0; }
    for (int r = 0; r < nranges; ++r) {
        const int64_t rb = row_ranges[2 * r], re = row_ranges[2 * r + 1];
        GM_REQUIRE(0 <= rb && rb <= re && re <= n, "bad row range [%lld,%lld)", (long long)rb, (long long)re);
        GM_REQUIRE(rb >= prev_end, "row ranges must be ascending and disjoint");
        prev_end = re;
        if (rb == re) continue;
        const int q = a.nranges++;
        a.row_begin[q] = rb; a.row_end[q] = re;
        a.unit_base[q + 1] = a.unit_base[q] + units_before_row(rb, re, a.nchunks);
    }
    if (a.nranges == 0) return GM_OK;
    for (int q = a.nranges; q < MAXR; ++q) a.unit_base[q + 1] = a.unit_base[a.nranges];
    const int64_t units = a.unit_base[a.nranges];
    const size_t per_wave = (size_t)hE * 64 * sizeof(float);             // <= 128 KB at hidden_E = 512
    int waves = (int)(64 * 1024 / per_wave);
    waves = waves < 1 ? 1 : (waves > 4 ? 4 : waves);
    const int64_t blocks = (units + waves - 1) / waves;
    GM_REQUIRE(blocks > 0 && blocks < (1ll << 31), "grid out of range");
    const size_t lds = per_wave * waves;
    if (lds > 64 * 1024)
        GM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair_generic), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_pair_generic, dim3((unsigned)blocks), dim3(64 * waves), lds, as_stream(stream), a);
    GM_LAUNCH_CHECK();
    return GM_OK;
}
