// gm_pair.hip -- the all-pairs structure denoiser: the hot loop of get_E_t (model/diffusion.py:444-470).
//
// For every unordered pair i<j of the rows this call owns:
//     z      = h_i * h_j                                   (model/gnn.py:246)
//     hid    = relu(W1 z + b1)      128x128                (model/gnn.py:227-231,248)
//     logit  = W2 hid + b2          2x128, evaluated as the difference d = logit[1] - logit[0] (all a 2-class softmax needs)
//     q      = softmax(logit) = softmax([0, d])            (model/diffusion.py:457)
//     p      = posterior(q | E_t[i,j])                     (model/diffusion.py:351-406,463) -- 8-entry table
//     E_s    = draw(p)                                     (model/diffusion.py:421)
// and the draw goes straight into an upper-triangle bitmap: no (P,2) tensor, no pair index list, no dense N x N.
//
// MFMA mapping (v_mfma_f32_32x32x2_f32, exact f32): D[o][pair] += W1[o][k] * z[k][pair]; A = W1 rows (o), B = z
// columns (32 consecutive j of one i).  With the pair on the MFMA column, one lane ends up holding 64 of the 128
// hidden units of ONE pair, so the second layer is an in-lane dot product plus a single lane^32 exchange.
// k order: step s feeds k=2s (lanes 0-31) and k=2s+1 (lanes 32-63); LDS rows are stored [even k | odd k] so a
// lane's next four k-steps are one ds_read_b128.  hid[o] = fmaf chain over ascending k starting from b1[o]:
// bit-identical to oracle gmo_pair_logits.
//
// Tiling: persistent workgroups (512 threads = 8 waves, 2 per SIMD) pull 64x64 tiles of the upper triangle from an
// atomic counter.  The two waves of a SIMD (w and w+4) take turns on the matrix pipe: one runs the 512 MFMAs of its
// next row (both 32-column strips) while the other runs the epilogue of the row it has just finished and fetches what
// its next row needs (b1 into the accumulators, the first fragment); the turn passes through a flag in LDS.  (Left to
// the hardware, the older wave wins every MFMA arbitration, the younger one starves and later runs alone.)  LDS
// fragments for k-block q+1 are read while the MFMAs of block q run; the k loop is fully unrolled, so every LDS offset
// is an immediate.  The alternation runs straight through tile changes: there is no workgroup barrier in the loop.
// Waves 0-3 finish a tile one row-unit before waves 4-7; in that time they fetch the next queue item and load its h
// rows into the (now dead) accumulator registers, and write them to LDS as soon as every wave has issued its last
// fragment read of the old tile -- while waves 4-7 are still in their last epilogue, which reads no h rows.
// What bounds the kernel (tools/micro/): on gfx950 a vector-ALU instruction of ANY wave of a SIMD costs
// that SIMD 4-7 cycles of matrix-pipe time (the two do not overlap), LDS reads cost none.  A row-unit is therefore
// 512 x 65 cycles of MFMA plus ~5 cycles for every VALU instruction issued for it, and the code keeps those few:
// packed z products, packed ReLU (clamp modifier) and packed second-layer FMAs, one Philox block per four row-units,
// tile decoding and addresses on the scalar unit (pair_epilogue in gm_pair_common.h).
// W1 (64 KB) stays in LDS for the whole kernel; per tile the 64 h_i rows and 64 h_j rows are staged (64 KB), i.e.
// 16 B of HBM/L2 traffic per pair against 33,408 FLOP: MFMA-bound (f32 matrix peak 157 TFLOP/s).
#include "gm_pair_common.h"

namespace gm {

// store a float4 of 4 consecutive k into the [even k | odd k] row image
__device__ __forceinline__ void put_split(float* row, int c4, float4 v) {
    *reinterpret_cast<float2*>(row + 2 * c4) = make_float2(v.x, v.z);
    *reinterpret_cast<float2*>(row + 64 + 2 * c4) = make_float2(v.y, v.w);
}

// Stage the 64 h_i rows and 64 h_j rows of a tile: 8 float4 per thread, all loads issued before the first LDS write.
// Rows past n are clamped to n-1, not zeroed: they only feed pairs that are masked (j >= n) or skipped (i >= n).
__device__ __forceinline__ void stage_tile(const float* __restrict__ h, int64_t n, int64_t i0, int64_t j0, float* hIs,
                                           float* hJs, int tid) {
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int q = tid + k * PAIR_THREADS;
        const int which = q >> 11;             // 0: i rows, 1: j rows
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
        put_split((which ? hJs : hIs) + r * LDR, c4, v[k]);
    }
}

// The same staging done by waves 0-3 alone (256 threads, 16 float4 each), split so that the global loads can be in
// flight while the other waves still read the old tile.
__device__ __forceinline__ void stage_load_half(const float* __restrict__ h, int64_t n, int64_t i0, int64_t j0, int gt,
                                                float4 (&v)[16]) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int q = gt + k * (PAIR_THREADS / 2);
        const int which = q >> 11;
        const int r = (q >> 5) & (TILE - 1), c4 = q & 31;
        int64_t node = (which ? j0 : i0) + r;
        node = node < n ? node : n - 1;
        v[k] = *reinterpret_cast<const float4*>(h + node * HE + 4 * c4);
    }
}
__device__ __forceinline__ void stage_store_half(float* hIs, float* hJs, int gt, const float4 (&v)[16]) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int q = gt + k * (PAIR_THREADS / 2);
        const int which = q >> 11;
        const int r = (q >> 5) & (TILE - 1), c4 = q & 31;
        put_split((which ? hJs : hIs) + r * LDR, c4, v[k]);
    }
}

struct Frag {
    float4 w[4];
    float4 hi, ha, hb;
    floatx2 z0[2], z1[2];   // h_i * h_j for the fragment's four k-steps, strip 0 / strip 1 (pairs: one v_pk_mul_f32 each)
};

// z for a fragment whose LDS reads have landed.  Done one half-iteration ahead of the MFMAs that consume it, so the
// multiplies (and the VALU->MFMA operand wait states) run in the shadow of the previous block's last MFMAs.
__device__ __forceinline__ void prep_frag(Frag& f) {
    // spelled out: left to itself hipcc splits these into eight v_mul_f32 inside the unrolled loop
    const floatx2 hi01 = {f.hi.x, f.hi.y}, hi23 = {f.hi.z, f.hi.w};
    const floatx2 ha01 = {f.ha.x, f.ha.y}, ha23 = {f.ha.z, f.ha.w}, hb01 = {f.hb.x, f.hb.y}, hb23 = {f.hb.z, f.hb.w};
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(f.z0[0]) : "v"(hi01), "v"(ha01));
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(f.z0[1]) : "v"(hi23), "v"(ha23));
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(f.z1[0]) : "v"(hi01), "v"(hb01));
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(f.z1[1]) : "v"(hi23), "v"(hb23));
}

__device__ __forceinline__ void load_frag(Frag& f, const float* wrow, const float* hirow, const float* hjrow, int q) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) f.w[mt] = *reinterpret_cast<const float4*>(wrow + mt * 32 * LDR + 4 * q);
    f.hi = *reinterpret_cast<const float4*>(hirow + 4 * q);
    f.ha = *reinterpret_cast<const float4*>(hjrow + 4 * q);
    f.hb = *reinterpret_cast<const float4*>(hjrow + 32 * LDR + 4 * q);
}

// 4 k-steps (k = 8q .. 8q+7) on both strips: 32 MFMAs
__device__ __forceinline__ void mma_frag(floatx16 (&acc)[2][4], const Frag& f) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float z0 = f.z0[e >> 1][e & 1], z1 = f.z1[e >> 1][e & 1];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const float wv_ = e == 0 ? f.w[mt].x : e == 1 ? f.w[mt].y : e == 2 ? f.w[mt].z : f.w[mt].w;
            acc[0][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv_, z0, acc[0][mt], 0, 0, 0);
            acc[1][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv_, z1, acc[1][mt], 0, 0, 0);
        }
    }
}

// DBG = true is the instantiation that also writes both logits of every pair (logits_dbg, for comparison with the
// reference's LinkPredictor output); the draws and posteriors come from the same logit-difference path in both.
template <bool DBG>
__global__ void __launch_bounds__(PAIR_THREADS, 2) k_pair_score_sample(const PairArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[(HE + 2 * TILE) * LDR + HE + 3 * HE + 32 + HE + 8 * 256 + 8];
    float* W1s = lds;
    float* hIs = W1s + HE * LDR;
    float* hJs = hIs + TILE * LDR;
    float* b1s = hJs + TILE * LDR;
    float* dws = b1s + HE;                         // W2[1] - W2[0]: the second layer is evaluated as a logit difference
    float* W2s = dws + HE;                         // both rows of W2 (read by the DBG instantiation only)
    int* tile_s = reinterpret_cast<int*>(W2s + 2 * HE);   // tile_s[0..1]: ring of queue items, [2]: items published, [3]: abort
    int* stg_s = tile_s + 4;                       // stg_s[w], w < 4: tiles wave w has staged
    int* turn_s = tile_s + 8;                      // turn_s[w]: MFMA row-units wave w has issued so far (hand-off flags)
    float* b1s2 = W2s + 2 * HE + 32;               // a second copy of b1: strip 1's accumulators are read from it directly
    uint32_t* rng_s = reinterpret_cast<uint32_t*>(b1s2 + HE);   // [wave][4 row-units][64 columns] Philox words
    float* tab_s = reinterpret_cast<float*>(rng_s + 8 * 256);   // posterior table as two rows [E_t bit][4]: one ds_read_b128 per pair

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wv >> 2;                       // waves w and w+4 share a SIMD: group 0 / group 1
    const int l31 = lane & 31, half = lane >> 5;

    // ---- stage the pair MLP once per workgroup
    for (int q = tid; q < HE * (HE / 4); q += PAIR_THREADS) {
        const int o = q >> 5, c4 = q & 31;
        put_split(W1s + o * LDR, c4, *reinterpret_cast<const float4*>(a.W1 + o * HE + 4 * c4));
    }
    if (tid < HE) { b1s[tid] = a.b1[tid]; b1s2[tid] = a.b1[tid]; }
    if (tid < HE) dws[tid] = (a.W2[HE + tid] - a.W2[tid]) * RELU_UP;
    if (tid < 2 * HE) W2s[tid] = a.W2[tid] * RELU_UP;
    const float b20 = a.b2[0], b21 = a.b2[1];
    const float db = b21 - b20;

    // a launch that finds the stall word set (an earlier launch lost a hand-shake and the host has not dealt with it) does nothing
    if (tid == 0) { tile_s[0] = stall_word_set(a.stall) ? QUEUE_END : (int)atomicAdd(a.counter, 1u); tile_s[2] = 0; tile_s[3] = 0; }
    if (tid < 8) tab_s[tid] = a.tab[((tid & 3) == 0 ? 0 : (tid & 3) == 1 ? 4 : (tid & 3) == 2 ? 2 : 6) + (tid >> 2)];
    if (tid < 8) turn_s[tid] = 0;
    if (tid < 4) stg_s[tid] = 0;
    __syncthreads();
    // the queue item is wave-uniform: in a scalar register, the tile decoding and every address derived from it run on the
    // scalar unit (vector-ALU instructions cost matrix-pipe time)
    int64_t t = __builtin_amdgcn_readfirstlane(tile_s[0]);
    int seq = 0;                                   // queue items this workgroup has finished (same in every wave)
    int units_done = 0;                            // row-units this wave has multiplied (same count in every wave)
    lds_flag* abort_s = as_lds_flag(tile_s + 3);   // raised by the first wave of this workgroup whose wait gave up
    if (t < a.nitems) {                            // first item: staged by everybody, behind a barrier
        int64_t i0, j0, row_end, tile;
        int tj, ub, ue;
        decode_item(a, t, tile, ub, ue);
        decode_tile(a, tile, i0, j0, tj, row_end);
        stage_tile(a.h, a.n, i0, j0, hIs, hJs, tid);
    }
    __syncthreads();
    lds_flag* my_turn = as_lds_flag(turn_s + wv);
    lds_flag* partner_turn = as_lds_flag(turn_s + (wv ^ 4));

    while (t < a.nitems) {
        int64_t i0, j0, row_end, tile;
        int tj, ub, ue;
        decode_item(a, t, tile, ub, ue);
        decode_tile(a, tile, i0, j0, tj, row_end);

        // the h rows of item number `seq` are in LDS once waves 0-3 have all staged `seq` tiles
        if (!wait_all_ge(as_lds_flag(stg_s), 4, seq, lane, abort_s, a)) break;

        floatx16 acc[2][4];
        // 8 row-units per wave and tile.  The two waves of a SIMD (w, w+4) alternate on the matrix pipe: group 0 runs
        // unit u, then group 1 runs unit u while group 0 does the VALU epilogue of its unit, and so on.  The turn is
        // handed over through turn_s[] in LDS (no data passes between the waves, only the right to issue MFMAs).
        for (int u = ub; u < ue; ++u) {
            const int il = wv + 8 * u;
            const int64_t i = i0 + il;
            const bool active = i < row_end && i < a.n;      // wave-uniform
            const float* wrow = W1s + l31 * LDR + 64 * half;
            const float* hirow = hIs + il * LDR + 64 * half;
            const float* hjrow = hJs + l31 * LDR + 64 * half;
            Frag f0, f1;
            // Everything the first MFMA needs is fetched BEFORE the wave asks for its turn -- b1 into the accumulators, the
            // first fragment read and multiplied -- under the partner's MFMA run, so that the matrix pipe changes hands
            // without waiting for LDS round trips (the stamps showed ~2.6k idle cycles per hand-over otherwise).
            if (active) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        // every vector-ALU instruction on this SIMD costs matrix-pipe time: the second
                        // strip reads its own LDS copy of b1 straight into its accumulators instead of 64 v_mov
                        const float4 bv = *reinterpret_cast<const float4*>(b1s + mt * 32 + 8 * g + 4 * half);
                        const float4 bw = *reinterpret_cast<const float4*>(b1s2 + mt * 32 + 8 * g + 4 * half);
                        acc[0][mt][4 * g + 0] = bv.x; acc[0][mt][4 * g + 1] = bv.y;
                        acc[0][mt][4 * g + 2] = bv.z; acc[0][mt][4 * g + 3] = bv.w;
                        acc[1][mt][4 * g + 0] = bw.x; acc[1][mt][4 * g + 1] = bw.y;
                        acc[1][mt][4 * g + 2] = bw.z; acc[1][mt][4 * g + 3] = bw.w;
                    }
                load_frag(f0, wrow, hirow, hjrow, 0);
                prep_frag(f0);
            }
            __builtin_amdgcn_sched_barrier(0);
            {   // wait for the partner: group 0 needs its partner's unit (units_done-1), group 1 its partner's unit units_done
                // bounded, and giving up raises nothing: the order is a speed matter only, results never depend on it
                LdsWait w{partner_turn, 1, units_done + grp, lane, abort_s, &a};
                gm_pair_protocol::bounded_wait(w, gm_pair_protocol::TURN_SPINS, false);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (active) {
                // the multiplying wave outranks its partner at the SIMD's issue arbiter (no priorities: 22.8 instead of 21.9 ms;
                // epilogue wave high, or one group only: no difference -- DESIGN.md 4.1)
                __builtin_amdgcn_s_setprio(1);
                // fragments for k-block q+1 are read from LDS while the 32 MFMAs of block q run; fully unrolled: every
                // LDS offset is an immediate, no address arithmetic between the MFMAs
#pragma unroll
                for (int q = 0; q < HE / 8; q += 2) {
                    load_frag(f1, wrow, hirow, hjrow, q + 1);
                    __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of the MFMAs it overlaps
                    mma_frag(acc, f0);
                    __builtin_amdgcn_sched_barrier(0);
                    prep_frag(f1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (q + 2 < HE / 8) load_frag(f0, wrow, hirow, hjrow, q + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_frag(acc, f1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (q + 2 < HE / 8) prep_frag(f0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_s_setprio(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            ++units_done;
            if (lane == 0) *my_turn = units_done;
            __builtin_amdgcn_sched_barrier(0);
            if (active) {
                EpiCtx ec;
                ec.dws = dws; ec.W2s = W2s; ec.tab_s = tab_s; ec.rng_mine = rng_s + wv * 256; ec.db = db; ec.b20 = b20; ec.b21 = b21;
                pair_epilogue<DBG>(a, ec, acc, i, i0, j0, tj, u, ub, wv, lane);
            }
        }
        // ---- tile hand-over, no barrier.  Ordering rests on the LDS flags (LDS serves a wave's requests in order):
        //   * a wave publishes turn_s after the last fragment read of its unit was issued;
        //   * waves 0-3 overwrite the h rows only after all 8 waves have published the last unit of this item;
        //   * nobody reads the new rows before all of waves 0-3 have published stg_s (the wait at the loop head).
        // The epilogue touches no h rows, so waves 4-7 finish theirs under the staging and the next item's first unit.
        ++seq;
        if (wv == 0) {
            if (lane == 0) {
                const int next = (int)atomicAdd(a.counter, 1u);
                tile_s[seq & 1] = stall_word_set(a.stall) ? QUEUE_END : next;   // slot last read before item seq-1 began
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                *as_lds_flag(tile_s + 2) = seq;
            }
        }
        if (!wait_all_ge(as_lds_flag(tile_s + 2), 1, seq, lane, abort_s, a)) break;
        t = __builtin_amdgcn_readfirstlane(as_lds_flag(tile_s)[seq & 1]);
        if (grp == 0 && t < a.nitems) {
            int64_t ni0, nj0, nre, ntile;
            int ntj_, nub, nue;
            decode_item(a, t, ntile, nub, nue);
            decode_tile(a, ntile, ni0, nj0, ntj_, nre);
            float4 v[16];
            stage_load_half(a.h, a.n, ni0, nj0, tid, v);
            if (!wait_all_ge(as_lds_flag(turn_s), 8, units_done, lane, abort_s, a)) break;
            stage_store_half(hIs, hJs, tid, v);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) *as_lds_flag(stg_s + wv) = seq;
        }
    }
}

}  // namespace gm

using namespace gm;

extern "C" int gm_pair_score_sample_ranges(int64_t n, int64_t hE, const float* h, const float* W1, const float* b1,
                                           const float* W2, const float* b2, const uint32_t* bits_t, int64_t wpr,
                                           const float* tab, int nranges, const int64_t* row_ranges, const float* u,
                                           uint64_t seed, uint32_t step, uint32_t* out_upper, float* logits_dbg,
                                           float* post_dbg, void* workspace, gm_stream_t stream) {
    if (hE != HE)   // any other width the reference accepts (model/gnn.py:204-231): the vector kernel of gm_pair_generic.hip
        return gm_pair_generic_launch(n, hE, h, W1, b1, W2, b2, bits_t, wpr, tab, nranges, row_ranges, u, seed, step, out_upper,
                                      logits_dbg, post_dbg, stream);
    PairArgs a;
    unsigned grid = 0;
    if (int rc = gm_build_pair_args(n, hE, h, W1, b1, W2, b2, bits_t, wpr, tab, nranges, row_ranges, u, seed, step, out_upper,
                                    logits_dbg, post_dbg, workspace, stream, a, grid))
        return rc;
    if (grid == 0) return GM_OK;
    hipStream_t s = as_stream(stream);
    const bool dbg = logits_dbg || post_dbg || u;
    if (dbg) hipLaunchKernelGGL(k_pair_score_sample<true>, dim3(grid), dim3(PAIR_THREADS), 0, s, a);
    else hipLaunchKernelGGL(k_pair_score_sample<false>, dim3(grid), dim3(PAIR_THREADS), 0, s, a);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_pair_score_sample(int64_t n, int64_t hE, const float* h, const float* W1, const float* b1, const float* W2,
                                    const float* b2, const uint32_t* bits_t, int64_t wpr, const float* tab, int64_t row_begin,
                                    int64_t row_end, const float* u, uint64_t seed, uint32_t step, uint32_t* out_upper,
                                    float* logits_dbg, float* post_dbg, void* workspace, gm_stream_t stream) {
    const int64_t range[2] = {row_begin, row_end};
    return gm_pair_score_sample_ranges(n, hE, h, W1, b1, W2, b2, bits_t, wpr, tab, 1, range, u, seed, step, out_upper,
                                       logits_dbg, post_dbg, workspace, stream);
}
