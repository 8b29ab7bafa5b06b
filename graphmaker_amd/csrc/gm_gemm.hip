// gm_gemm.hip -- nn.Linear (+ReLU) for the denoiser towers (model/gnn.py:27-38,116-126,138-142) on the f32 MFMA.
//
//   C[M,N] = act( A[M,K] @ W[N,K]^T + bias[N] )            fp32 in, fp32 accumulate, exact fp32
//
// v_mfma_f32_16x16x4_f32 is a k-ordered fmaf chain (MI355X guide, "FP32-input MFMA"), so every output element is
//   acc = 0;  for k = 0..K-1: acc = fmaf(A[m][k], W[n][k], acc);  out = acc + bias[n]
// exactly what oracle/gm_oracle.c gmo_linear computes: no split-K, K walked in ascending order.
//
// Why the 16x16x4 form and not 32x32x2 (same peak rate): it advances a chain by 4 k per 32-cycle issue instead of 2 k
// per 64 cycles, so the serial K walk of one output is 3.2x shorter, its 4-register accumulators let 4 workgroups
// (16 waves) share a CU, and 64x64 block tiles give 4x as many workgroups -- which is what the row-sharded towers
// (N/G rows per rank) and the narrow layers (128 / 512 outputs) need.  Measured on the tower shapes (tools/gemm_ab.py):
// 13752x512x1534 80 -> 96 TFLOP/s, 1719x1744x1744 62 -> 85, 1719x128x1744 8 -> 15, 13752x1744x1744 106 -> 121.
//
// LDS tiles are plain row-major, double-buffered, one barrier per k-tile; row stride KT + 4 floats makes the operand
// read "row lane&15, k = 4s + lane>>4" touch all 64 banks once (16 distinct multiples of 4, plus lane>>4).
// Global -> register -> LDS staging with the next tile's loads in flight under the MFMAs.
// Roofline: MFMA-bound (157 TFLOP/s f32 matrix peak).
#include "gm_common.h"

namespace gm {

// 4 consecutive k of one operand row -> registers.  GUARD: zeros beyond K (ragged last tile); zero products leave the
// fmaf chain unchanged (an accumulator that starts at +0 never becomes -0).
template <int VEC, bool GUARD>
__device__ __forceinline__ float4 load_k4(const float* __restrict__ p, int64_t k, int64_t K) {
    float4 v;
    if (!GUARD) {
        if (VEC == 4) {
            v = *reinterpret_cast<const float4*>(p);
        } else if (VEC == 2) {
            const float2 a = *reinterpret_cast<const float2*>(p), b = *reinterpret_cast<const float2*>(p + 2);
            v = make_float4(a.x, a.y, b.x, b.y);
        } else {
            v = make_float4(p[0], p[1], p[2], p[3]);
        }
    } else {
        v.x = k + 0 < K ? p[0] : 0.0f;
        v.y = k + 1 < K ? p[1] : 0.0f;
        v.z = k + 2 < K ? p[2] : 0.0f;
        v.w = k + 3 < K ? p[3] : 0.0f;
    }
    return v;
}

typedef float floatx4 __attribute__((ext_vector_type(4)));

// TM x TN 16x16 MFMA tiles per wave, WGM x WGN waves: block tile (16 TM WGM) x (16 TN WGN); KT-wide k-tiles (row stride
// KT + 4 keeps the banks distinct for KT = 16 and 32); OCC = workgroups per CU the register/LDS budget is set for.
template <int VEC, int TM, int TN, int KT, int OCC, int WGM = 2, int WGN = 2>
__global__ void __launch_bounds__(64 * WGM * WGN, OCC)
k_linear(int64_t M, int64_t N, int64_t K, const float* __restrict__ A, int64_t lda, const float* __restrict__ W,
           int64_t ldw, const float* __restrict__ bias, float* __restrict__ C, int64_t ldc, int relu) {
    constexpr int SBM = 16 * TM * WGM, SBN = 16 * TN * WGN, LD = KT + 4, NT = 64 * WGM * WGN;
    constexpr int QPR = KT / 4;                                // float4 per tile row
    constexpr int NA = SBM * QPR / NT, NB = SBN * QPR / NT;    // float4 per thread and tile
    constexpr int RSTEP = NT / QPR;                            // rows between a thread's consecutive slots
    static_assert(NA >= 1 && NB >= 1, "tile too small for the staging threads");
    __shared__ __attribute__((aligned(16))) float As[2][SBM * LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][SBN * LD];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int i16 = lane & 15, kg = lane >> 4;
    const int wm = wv / WGN, wn = wv % WGN;
    const int64_t gx = (N + SBN - 1) / SBN, gy = (M + SBM - 1) / SBM, total = gx * gy;
    // workgroups are dealt round-robin to the 8 XCDs: give each XCD a contiguous run of tiles (n fastest) so that the A
    // row-tile a run shares stays in that XCD's L2
    const int64_t per = (total + 7) / 8;
    const int64_t logical = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per || logical >= total) return;
    const int64_t m0 = (logical / gx) * SBM, n0 = (logical % gx) * SBN;

    const int srow = tid / QPR, skq = (tid % QPR) * 4;         // float4 slot f = tid + NT*c: row f/QPR, k-quad f%QPR
    const float* pa[NA];
    const float* pw[NB];
#pragma unroll
    for (int c = 0; c < NA; ++c) {
        const int64_t r = m0 + srow + RSTEP * c;
        pa[c] = A + (r < M ? r : M - 1) * lda + skq;          // clamped rows only feed outputs that are never stored
    }
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        const int64_t r = n0 + srow + RSTEP * c;
        pw[c] = W + (r < N ? r : N - 1) * ldw + skq;
    }
    // Two register sets: while tile t is multiplied, tile t+1 sits in one set (loads issued an iteration ago) and tile
    // t+2 is being loaded into the other.  A load therefore has a whole iteration plus a multiply phase to land before
    // its s_waitcnt, and a slow load no longer holds the whole workgroup at the barrier (with one set the barrier made
    // every wave wait for the slowest wave's loads: 13% on the 1744-wide layers; small grids were latency-bound).
    struct Regs {
        float4 a[NA], b[NB];
    };
    Regs r0, r1;
    auto fetch_fast = [&](Regs& r, int64_t k0) {
#pragma unroll
        for (int c = 0; c < NA; ++c) r.a[c] = load_k4<VEC, false>(pa[c] + k0, 0, 0);
#pragma unroll
        for (int c = 0; c < NB; ++c) r.b[c] = load_k4<VEC, false>(pw[c] + k0, 0, 0);
    };
    auto fetch_any = [&](Regs& r, int64_t k0) {       // prologue / last iterations only: full or ragged tile
        if (k0 + KT <= K) {
            fetch_fast(r, k0);
        } else {
#pragma unroll
            for (int c = 0; c < NA; ++c) r.a[c] = load_k4<VEC, true>(pa[c] + k0, k0 + skq, K);
#pragma unroll
            for (int c = 0; c < NB; ++c) r.b[c] = load_k4<VEC, true>(pw[c] + k0, k0 + skq, K);
        }
    };
    auto stash = [&](const Regs& r, int buf) {
#pragma unroll
        for (int c = 0; c < NA; ++c) *reinterpret_cast<float4*>(As[buf] + (srow + RSTEP * c) * LD + skq) = r.a[c];
#pragma unroll
        for (int c = 0; c < NB; ++c) *reinterpret_cast<float4*>(Bs[buf] + (srow + RSTEP * c) * LD + skq) = r.b[c];
    };

    floatx4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.0f;

    const int aoff = (wm * 16 * TM + i16) * LD + kg;
    const int boff = (wn * 16 * TN + i16) * LD + kg;
    auto multiply = [&](int buf) {     // one k-tile: KT/4 MFMA k-steps in ascending k
        const float* as = As[buf] + aoff;
        const float* bs = Bs[buf] + boff;
        __builtin_amdgcn_s_setprio(1);     // a wave in its MFMA run outranks the waves issuing loads and LDS stores (+0.6%)
#pragma unroll
        for (int st = 0; st < KT / 4; ++st) {
            float av[TM], bv[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) av[a] = as[16 * a * LD + 4 * st];
#pragma unroll
            for (int b = 0; b < TN; ++b) bv[b] = bs[16 * b * LD + 4 * st];
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    const int64_t nfull = K / KT;                       // tiles loaded without guards
    const int64_t T = (K + KT - 1) / KT;                // all tiles (the last one may be ragged)
    // tile `it` lives in LDS buffer it&1; `cur` holds tile it+1 (in flight or landed), `nxt` receives tile it+2
    auto step_fast = [&](int64_t it, Regs& cur, Regs& nxt) {
        fetch_fast(nxt, (it + 2) * KT);
        __builtin_amdgcn_sched_barrier(0);              // loads above the MFMAs, their first use below them
        // tile it+1 goes to LDS BEFORE this tile's MFMAs (its buffer was last read before the previous barrier): the wave
        // reaches the barrier straight out of its MFMA run instead of via a load wait and four LDS stores (+2% over the shapes)
        stash(cur, (int)((it + 1) & 1));
        __builtin_amdgcn_sched_barrier(0);
        multiply((int)(it & 1));
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    auto step_any = [&](int64_t it, Regs& cur, Regs& nxt) {
        if (it + 2 < T) fetch_any(nxt, (it + 2) * KT);
        if (it + 1 < T) stash(cur, (int)((it + 1) & 1));
        multiply((int)(it & 1));
        if (it + 1 < T) __syncthreads();
    };
    fetch_any(r0, 0);
    stash(r0, 0);
    __syncthreads();
    if (T > 1) fetch_any(r0, KT);
    int64_t it = 0;
    for (; it + 3 < nfull; it += 2) {                   // both steps prefetch full tiles (it+2, it+3 < nfull)
        step_fast(it, r0, r1);
        step_fast(it + 1, r1, r0);
    }
    for (; it < T; it += 2) {                           // the last few tiles, ragged one included
        step_any(it, r0, r1);
        if (it + 1 < T) step_any(it + 1, r1, r0);
    }

    // epilogue: D tile layout col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int64_t n = n0 + wn * 16 * TN + b * 16 + i16;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t m = m0 + wm * 16 * TM + a * 16 + 4 * kg + r;
                if (m < M) {
                    float v = acc[a][b][r];
                    if (bias) v = v + bv;
                    if (relu && !(v > 0.0f)) v = 0.0f;
                    C[m * ldc + n] = v;
                }
            }
        }
    }
}

// ---- a layer with at most 4 outputs over very many rows (the pair MLP's 128 -> 2 layer on a 2,097,152-pair training
// batch): HBM-bound on reading A once, which a 128-wide MFMA tile does at 1.6 TB/s.  One lane per row keeps the
// ascending-k fmaf chain of the header (same bits as k_linear); a wave stages 64 rows x 32 k through LDS so that global
// reads stay 128-byte row pieces, W is read from LDS as a broadcast.
constexpr int SK_KC = 32, SK_LD = SK_KC + 4, SK_MAXK = 1024;
template <int NOUT>
__global__ void __launch_bounds__(256)
k_linear_skinny(int64_t M, int64_t K, const float* __restrict__ A, int64_t lda, const float* __restrict__ W, int64_t ldw,
                const float* __restrict__ bias, float* __restrict__ C, int64_t ldc, int relu) {
    __shared__ __attribute__((aligned(16))) float tile[4][64 * SK_LD];
    __shared__ __attribute__((aligned(16))) float Ws[NOUT * SK_MAXK];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    for (int e = tid; e < NOUT * (int)K; e += 256) Ws[e] = W[(int64_t)(e / (int)K) * ldw + e % (int)K];
    __syncthreads();
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + wv) * 64;
    if (r0 >= M) return;
    float* t = tile[wv];
    float acc[NOUT];
#pragma unroll
    for (int c = 0; c < NOUT; ++c) acc[c] = 0.0f;
    const int lr = lane >> 3, lq = lane & 7;                    // staging: 8 rows x 8 float4 per pass
    for (int kc = 0; kc < (int)K; kc += SK_KC) {
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t m = r0 + lr + 8 * i;
            v[i] = m < M ? *reinterpret_cast<const float4*>(A + m * lda + kc + 4 * lq) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(t + (lr + 8 * i) * SK_LD + 4 * lq) = v[i];
        // a wave's LDS writes and reads are served in order: no barrier needed between the passes of one wave
#pragma unroll
        for (int q = 0; q < SK_KC / 4; ++q) {
            const float4 a = *reinterpret_cast<const float4*>(t + lane * SK_LD + 4 * q);
#pragma unroll
            for (int c = 0; c < NOUT; ++c) {
                const float4 w = *reinterpret_cast<const float4*>(Ws + c * (int)K + kc + 4 * q);
                acc[c] = __builtin_fmaf(a.x, w.x, acc[c]);
                acc[c] = __builtin_fmaf(a.y, w.y, acc[c]);
                acc[c] = __builtin_fmaf(a.z, w.z, acc[c]);
                acc[c] = __builtin_fmaf(a.w, w.w, acc[c]);
            }
        }
    }
    const int64_t m = r0 + lane;
    if (m < M) {
#pragma unroll
        for (int c = 0; c < NOUT; ++c) {
            float v = acc[c];
            if (bias) v = v + bias[c];
            if (relu && !(v > 0.0f)) v = 0.0f;
            C[m * ldc + c] = v;
        }
    }
}

}  // namespace gm

using namespace gm;

template <int NOUT>
static int launch_skinny(int64_t M, int64_t K, const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                         float* C, int64_t ldc, int relu, hipStream_t s) {
    hipLaunchKernelGGL((k_linear_skinny<NOUT>), dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, M, K, A, lda, W, ldw, bias, C,
                       ldc, relu);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

template <int TM, int TN, int KT, int OCC, int WGM = 2, int WGN = 2>
static int launch_linear(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W, int64_t ldw,
                           const float* bias, float* C, int64_t ldc, int relu, hipStream_t s) {
    constexpr int SBM = 16 * TM * WGM, SBN = 16 * TN * WGN;
    const int64_t total_tiles = ((N + SBN - 1) / SBN) * ((M + SBM - 1) / SBM);
    GM_REQUIRE(total_tiles < (1ll << 31) - 8, "too many tiles");
    dim3 grid((unsigned)(((total_tiles + 7) / 8) * 8)), block(64 * WGM * WGN);
    const bool v4 = lda % 4 == 0 && ldw % 4 == 0 && aligned(A, 16) && aligned(W, 16);
    const bool v2 = lda % 2 == 0 && ldw % 2 == 0 && aligned(A, 8) && aligned(W, 8);
    if (v4) hipLaunchKernelGGL((k_linear<4, TM, TN, KT, OCC, WGM, WGN>), grid, block, 0, s, M, N, K, A, lda, W, ldw, bias, C, ldc, relu);
    else if (v2) hipLaunchKernelGGL((k_linear<2, TM, TN, KT, OCC, WGM, WGN>), grid, block, 0, s, M, N, K, A, lda, W, ldw, bias, C, ldc, relu);
    else hipLaunchKernelGGL((k_linear<1, TM, TN, KT, OCC, WGM, WGN>), grid, block, 0, s, M, N, K, A, lda, W, ldw, bias, C, ldc, relu);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

constexpr int64_t GM_GEMM_BIG_TILES = 400;     // at least this many 128x128 tiles: the 128x128 / 8-wave configuration
constexpr int64_t GM_GEMM_SMALL_TILES = 384;   // at most this many 64x64 tiles: the 32x64 configuration

extern "C" int gm_linear(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W, int64_t ldw,
                         const float* bias, float* C, int64_t ldc, int relu, gm_stream_t stream) {
    GM_REQUIRE(M > 0 && N > 0 && K > 0 && A && W && C, "bad arguments");
    GM_REQUIRE(lda >= K && ldw >= K && ldc >= N, "leading dimension too small");
    hipStream_t s = as_stream(stream);
    if (N <= 4 && M >= 65536 && K % SK_KC == 0 && K <= SK_MAXK && lda % 4 == 0 && aligned(A, 16) && (M + 255) / 256 < (1ll << 31)) {
        switch ((int)N) {      // the pair MLP's output layer on a training batch: see k_linear_skinny
            case 1: return launch_skinny<1>(M, K, A, lda, W, ldw, bias, C, ldc, relu, s);
            case 2: return launch_skinny<2>(M, K, A, lda, W, ldw, bias, C, ldc, relu, s);
            case 3: return launch_skinny<3>(M, K, A, lda, W, ldw, bias, C, ldc, relu, s);
            default: return launch_skinny<4>(M, K, A, lda, W, ldw, bias, C, ldc, relu, s);
        }
    }
    const int64_t tiles128 = ((N + 127) / 128) * ((M + 127) / 128);
    // wide layers with many rows: 128x256 block tile, 4x4 waves of 32x64 (a third of the operand traffic per FLOP of the 64x64 tile);
    // everything else: 64x64 block tile, 2x2 waves of 32x32, four workgroups per CU
    // (16 waves, 4x4 of 32x64, one workgroup per CU: 13752x1744x1744 113 -> 124 TFLOP/s, the 512-wide layers +5% over the
    // 8-wave 128x128 tile with two workgroups per CU that was here before; 256x128 measures the same)
    if (tiles128 >= GM_GEMM_BIG_TILES) {
        // ... unless it wastes clearly more of the chip than the 8-wave 128x128 tile (two workgroups per CU) would:
        // 2708 x 2866 (Cora's attribute head) is 264 tiles of 128x256 -- one full round and 8 stragglers -- but 506 of 128x128;
        // the training batch's 2097152 x 128 pair layer would compute 128 columns of padding per tile
        static int cus = 0;
        if (cus == 0) {
            int dev = 0, v = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
            else cus = 256;
        }
        // useful outputs / outputs computed over whole rounds of the chip (tile padding and the partly empty last round).
        // The 8-wave tile runs two workgroups per CU that share the CU's matrix pipe, so its rounds are counted per CU
        // like the 16-wave tile's (a CU left with one workgroup finishes it in about half the time): 6876 x 1744 is 756 tiles
        // of 128x128 = 2.95 per CU (0.357 ms) but 378 of 128x256 = 1.48 per CU (0.438 ms).  Per tile the 16-wave kernel is
        // ~8% faster (124 vs 113 TFLOP/s at 13752x1744x1744), hence the 0.92.
        const int64_t tiles256 = ((N + 255) / 256) * ((M + 127) / 128);
        const double useful = (double)M * (double)N;
        const double eff16 = useful / ((double)(((tiles256 + cus - 1) / cus) * cus) * 128.0 * 256.0);
        const double eff8 = 0.92 * useful / ((double)(((tiles128 + cus - 1) / cus) * cus) * 128.0 * 128.0);
        if (eff16 >= eff8) return launch_linear<2, 4, 32, 1, 4, 4>(M, N, K, A, lda, W, ldw, bias, C, ldc, relu, s);
        return launch_linear<2, 4, 32, 2, 4, 2>(M, N, K, A, lda, W, ldw, bias, C, ldc, relu, s);
    }
    // few 64x64 tiles (a rank's row block, Cora-sized graphs, the 128- and 64-wide layers): the serial K walk of one
    // workgroup is the whole run time, so halve the rows per wave and double the workgroups (32x64 block tile)
    const int64_t tiles64 = ((N + 63) / 64) * ((M + 63) / 64);
    if (tiles64 <= GM_GEMM_SMALL_TILES || N <= 128) return launch_linear<1, 2, 32, 5>(M, N, K, A, lda, W, ldw, bias, C, ldc, relu, s);
    return launch_linear<2, 2, 32, 4>(M, N, K, A, lda, W, ldw, bias, C, ldc, relu, s);
}
