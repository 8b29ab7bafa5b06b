// gm_pair_bf16.hip -- OPT-IN reduced-precision variant of the all-pairs structure denoiser (gm_pair.hip is the exact f32 one
// and the default everywhere).  Same loop (model/diffusion.py:444-470, model/gnn.py:246-248), same tile queue, same epilogue
// (logit difference -> softmax -> table posterior -> draw -> bitmap); only the 128x128 product hid = W1 z runs on the bf16
// matrix cores (v_mfma_f32_32x32x16_bf16, f32 accumulate, 16x the f32 MFMA rate):
//   NPROD = 3 ("bf16x3"): W1 and z = h_i * h_j (formed in f32) are each split into a bf16 head and a bf16 tail,
//                         x = hi + lo with hi = bf16(x), lo = bf16(x - hi), and  W z ~ Whi zhi + Whi zlo + Wlo zhi
//                         (the dropped Wlo zlo term is 2^-16 relative): logits within the f32 tolerance of the reference.
//   NPROD = 1 ("bf16")  : heads only, plain bf16 inputs (2^-8 relative per product).
// The caller chooses it per call (gm_pair_score_sample_bf16); draws are NOT bit-identical to the f32 kernel's: a draw can
// differ where u*(p0+p1) lies within the logit error of p1 (tests count such draws and bound them).
// Operand maps (MI355X guide): lane l = (r = l & 31, half = l >> 5) holds A[row r][k = 8 half + j] and B[k = 8 half + j][col r],
// j = 0..7; the accumulator layout equals the f32 32x32 form, so the epilogue is gm_pair.hip's.
// LDS: W1 head and tail as bf16 rows of 128 + 8 elements (272 B: the 16-byte fragment reads of 16 consecutive rows touch all
// 64 banks once), the tile's 64 + 64 h rows in f32 (row stride 132 floats), ~139 KB: one 512-thread workgroup per CU, the two
// waves of a SIMD overlap one's MFMAs with the other's VALU (z products, splits, epilogue) by hardware arbitration.
#include "gm_pair_common.h"

namespace gm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int LDWB = HE + 8;      // bf16 elements per W1 row in LDS
constexpr int LDHF = HE + 4;      // floats per h row in LDS

__device__ __forceinline__ void split_bf16(float x, __bf16& hi, __bf16& lo) {
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}

// the 64 h_i rows and 64 h_j rows of a tile, f32, natural k order; rows past n are clamped (they only feed masked pairs)
__device__ __forceinline__ void stage_tile_plain(const float* __restrict__ h, int64_t n, int64_t i0, int64_t j0, float* hIs,
                                                 float* hJs, int tid) {
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int q = tid + k * PAIR_THREADS;
        const int which = q >> 11;
        const int r = (q >> 5) & (TILE - 1), c4 = q & 31;
        int64_t node = (which ? j0 : i0) + r;
        node = node < n ? node : n - 1;
        v[k] = *reinterpret_cast<const float4*>(h + node * HE + 4 * c4);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int q = tid + k * PAIR_THREADS;
        const int which = q >> 11;
        const int r = (q >> 5) & (TILE - 1), c4 = q & 31;
        *reinterpret_cast<float4*>((which ? hJs : hIs) + r * LDHF + 4 * c4) = v[k];
    }
}

template <int NPROD, bool DBG>
__global__ void __launch_bounds__(PAIR_THREADS, 2) k_pair_bf16(const PairArgs a) {
    __shared__ __attribute__((aligned(16))) __bf16 Whi[HE * LDWB];
    __shared__ __attribute__((aligned(16))) __bf16 Wlo[NPROD == 3 ? HE * LDWB : 8];
    __shared__ __attribute__((aligned(16))) float hIs[TILE * LDHF];
    __shared__ __attribute__((aligned(16))) float hJs[TILE * LDHF];
    __shared__ __attribute__((aligned(16))) float b1s[HE], dws[HE], W2s[2 * HE];
    __shared__ int tile_s[2];
    __shared__ __attribute__((aligned(16))) uint32_t rng_s[8 * 256];   // [wave][4 row-units][64 columns] Philox words
    __shared__ __attribute__((aligned(16))) float tab_s[8];            // posterior table, rows [E_t bit][4]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;

    // ---- the pair MLP, once per workgroup: W1 split into bf16 head / tail
    for (int q = tid; q < HE * (HE / 4); q += PAIR_THREADS) {
        const int o = q >> 5, c4 = q & 31;
        const float4 w = *reinterpret_cast<const float4*>(a.W1 + o * HE + 4 * c4);
        const float wv4[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __bf16 hi, lo;
            split_bf16(wv4[e], hi, lo);
            Whi[o * LDWB + 4 * c4 + e] = hi;
            if (NPROD == 3) Wlo[o * LDWB + 4 * c4 + e] = lo;
        }
    }
    if (tid < HE) b1s[tid] = a.b1[tid];
    if (tid < HE) dws[tid] = (a.W2[HE + tid] - a.W2[tid]) * RELU_UP;
    if (tid < 2 * HE) W2s[tid] = a.W2[tid] * RELU_UP;
    if (tid < 8) tab_s[tid] = a.tab[((tid & 3) == 0 ? 0 : (tid & 3) == 1 ? 4 : (tid & 3) == 2 ? 2 : 6) + (tid >> 2)];
    const float b20 = a.b2[0], b21 = a.b2[1];
    const float db = b21 - b20;
    if (tid == 0) tile_s[0] = (int)atomicAdd(a.counter, 1u);
    __syncthreads();
    int64_t t = __builtin_amdgcn_readfirstlane(tile_s[0]);   // wave-uniform: tile decoding and addresses on the scalar unit
    int seq = 0;

    while (t < a.nitems) {
        int64_t i0, j0, row_end, tile;
        int tj, ub, ue;
        decode_item(a, t, tile, ub, ue);
        decode_tile(a, tile, i0, j0, tj, row_end);
        stage_tile_plain(a.h, a.n, i0, j0, hIs, hJs, tid);
        if (tid == 0) tile_s[(seq + 1) & 1] = (int)atomicAdd(a.counter, 1u);     // the next item, fetched under this one
        __syncthreads();

        for (int u = ub; u < ue; ++u) {
            const int il = wv + 8 * u;
            const int64_t i = i0 + il;
            if (!(i < row_end && i < a.n)) continue;                             // wave-uniform
            floatx16 acc[2][4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = *reinterpret_cast<const float4*>(b1s + mt * 32 + 8 * g + 4 * half);
                    acc[0][mt][4 * g + 0] = bv.x; acc[0][mt][4 * g + 1] = bv.y;
                    acc[0][mt][4 * g + 2] = bv.z; acc[0][mt][4 * g + 3] = bv.w;
                    acc[1][mt][4 * g + 0] = bv.x; acc[1][mt][4 * g + 1] = bv.y;
                    acc[1][mt][4 * g + 2] = bv.z; acc[1][mt][4 * g + 3] = bv.w;
                }
            const __bf16* whi_row = Whi + l31 * LDWB + 8 * half;
            const __bf16* wlo_row = Wlo + l31 * LDWB + 8 * half;
            const float* hirow = hIs + il * LDHF + 8 * half;
            const float* hjrow = hJs + l31 * LDHF + 8 * half;
#pragma unroll 2
            for (int kb = 0; kb < HE / 16; ++kb) {                               // 16 k per MFMA; this lane: k = 16 kb + 8 half + j
                float hi8[8], ha8[8], hb8[8];
                *reinterpret_cast<float4*>(hi8) = *reinterpret_cast<const float4*>(hirow + 16 * kb);
                *reinterpret_cast<float4*>(hi8 + 4) = *reinterpret_cast<const float4*>(hirow + 16 * kb + 4);
                *reinterpret_cast<float4*>(ha8) = *reinterpret_cast<const float4*>(hjrow + 16 * kb);
                *reinterpret_cast<float4*>(ha8 + 4) = *reinterpret_cast<const float4*>(hjrow + 16 * kb + 4);
                *reinterpret_cast<float4*>(hb8) = *reinterpret_cast<const float4*>(hjrow + 32 * LDHF + 16 * kb);
                *reinterpret_cast<float4*>(hb8 + 4) = *reinterpret_cast<const float4*>(hjrow + 32 * LDHF + 16 * kb + 4);
                bf16x8 zh0, zl0, zh1, zl1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    __bf16 hh, ll;
                    split_bf16(hi8[j] * ha8[j], hh, ll);
                    zh0[j] = hh; zl0[j] = ll;
                    split_bf16(hi8[j] * hb8[j], hh, ll);
                    zh1[j] = hh; zl1[j] = ll;
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const bf16x8 wh = *reinterpret_cast<const bf16x8*>(whi_row + mt * 32 * LDWB + 16 * kb);
                    if (NPROD == 3) {        // small terms first
                        const bf16x8 wl = *reinterpret_cast<const bf16x8*>(wlo_row + mt * 32 * LDWB + 16 * kb);
                        acc[0][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, zh0, acc[0][mt], 0, 0, 0);
                        acc[1][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, zh1, acc[1][mt], 0, 0, 0);
                        acc[0][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, zl0, acc[0][mt], 0, 0, 0);
                        acc[1][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, zl1, acc[1][mt], 0, 0, 0);
                    }
                    acc[0][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, zh0, acc[0][mt], 0, 0, 0);
                    acc[1][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, zh1, acc[1][mt], 0, 0, 0);
                }
            }
            EpiCtx ec;
            ec.dws = dws; ec.W2s = W2s; ec.tab_s = tab_s; ec.rng_mine = rng_s + wv * 256; ec.db = db; ec.b20 = b20; ec.b21 = b21;
            pair_epilogue<DBG>(a, ec, acc, i, i0, j0, tj, u, ub, wv, lane);
        }
        __syncthreads();                       // every wave is done with this tile's h rows (and tile_s has the next item)
        ++seq;
        t = __builtin_amdgcn_readfirstlane(tile_s[seq & 1]);
    }
}

}  // namespace gm

using namespace gm;

extern "C" int gm_pair_score_sample_bf16(int64_t n, int64_t hE, const float* h, const float* W1, const float* b1, const float* W2,
                                         const float* b2, const uint32_t* bits_t, int64_t wpr, const float* tab, int nranges,
                                         const int64_t* row_ranges, const float* u, uint64_t seed, uint32_t step,
                                         uint32_t* out_upper, float* logits_dbg, float* post_dbg, void* workspace, int nprod,
                                         gm_stream_t stream) {
    GM_REQUIRE(nprod == 1 || nprod == 3, "nprod must be 1 (plain bf16) or 3 (bf16x3 split), got %d", nprod);
    PairArgs a;
    unsigned grid = 0;
    if (int rc = gm_build_pair_args(n, hE, h, W1, b1, W2, b2, bits_t, wpr, tab, nranges, row_ranges, u, seed, step, out_upper,
                                    logits_dbg, post_dbg, workspace, stream, a, grid))
        return rc;
    if (grid == 0) return GM_OK;
    hipStream_t s = as_stream(stream);
    const dim3 g(grid), b(PAIR_THREADS);
    const bool dbg = logits_dbg || post_dbg || u;    // rank-indexed arrays exist in the DBG instantiation only
    if (nprod == 3) {
        if (dbg) hipLaunchKernelGGL((k_pair_bf16<3, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((k_pair_bf16<3, false>), g, b, 0, s, a);
    } else {
        if (dbg) hipLaunchKernelGGL((k_pair_bf16<1, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((k_pair_bf16<1, false>), g, b, 0, s, a);
    }
    GM_LAUNCH_CHECK();
    return GM_OK;
}
