// gm_gemm.hip -- nn.Linear (+ReLU) for the denoiser towers (model/gnn.py:27-38,116-126,138-142) on the f32 MFMA.
//
//   C[M,N] = act( A[M,K] @ W[N,K]^T + bias[N] )            fp32 in, fp32 accumulate, exact fp32
//
// v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain (MI355X guide, "FP32-input MFMA"), so every output element is
//   acc = 0;  for k = 0..K-1: acc = fmaf(A[m][k], W[n][k], acc);  out = acc + bias[n]
// exactly what oracle/gm_oracle.c gmo_linear computes: no split-K, K walked in ascending order.
//
// Tiling: 128x128 block tile, BK = 16, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles
// (64 accumulator registers).  Operand tiles sit in LDS k-major (As[k][m], Bs[k][n]) so that the MFMA operand
// read "lane l takes element (l&31) of k-row (l>>5)" is one conflict-free ds_read_b32 per fragment.
// Global -> register -> LDS staging with the next tile's loads issued before the current tile's MFMAs.
// Roofline: MFMA-bound (157 TFLOP/s f32 matrix peak); 32 FLOP per operand byte at this tile size.
#include "gm_common.h"

namespace gm {

typedef float floatx16 __attribute__((ext_vector_type(16)));

#ifndef GM_GEMM_BK
#define GM_GEMM_BK 16
#endif
#ifndef GM_GEMM_SMALL_TILES
#define GM_GEMM_SMALL_TILES 768
#endif
#define GM_GEMM_XCD 1   // XCD-aware tile order (measured +2-3% on the 1744-wide layers)
constexpr int BN = 128, BK = GM_GEMM_BK;
constexpr int KPT = BK / 2;   // k values staged per thread and operand (row = tid/2, k-half = tid%2)
#ifndef GM_GEMM_LDT
#define GM_GEMM_LDT 136
#endif
constexpr int LDT = GM_GEMM_LDT;  // padded k-row length (floats)

// Each thread stages 8 consecutive k of one tile row: row = tid/2, k-half = tid%2.
template <int VEC>
__device__ __forceinline__ void load_row8(const float* __restrict__ base, int64_t ld, int64_t row, int64_t nrows,
                                          int64_t k0, int64_t K, float (&r)[8]) {
    if (row >= nrows) {
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] = 0.0f;
        return;
    }
    const float* p = base + row * ld + k0;
    if (VEC == 4) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (k0 + 4 * q + 4 <= K) {
                float4 v = *reinterpret_cast<const float4*>(p + 4 * q);
                r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) r[4 * q + e] = (k0 + 4 * q + e < K) ? p[4 * q + e] : 0.0f;
            }
        }
    } else if (VEC == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (k0 + 2 * q + 2 <= K) {
                float2 v = *reinterpret_cast<const float2*>(p + 2 * q);
                r[2 * q] = v.x; r[2 * q + 1] = v.y;
            } else {
                r[2 * q] = (k0 + 2 * q < K) ? p[2 * q] : 0.0f;
                r[2 * q + 1] = 0.0f;
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] = (k0 + q < K) ? p[q] : 0.0f;
    }
}

// unguarded twin of load_row8 for tiles that lie fully inside K (rows are clamped by the caller): straight-line
// vector loads, so all of a tile's loads are in flight together instead of one branch + wait per vector.
template <int VEC>
__device__ __forceinline__ void load_row8_fast(const float* __restrict__ p, float (&r)[8]) {
    if (VEC == 4) {
        const float4 v0 = *reinterpret_cast<const float4*>(p), v1 = *reinterpret_cast<const float4*>(p + 4);
        r[0] = v0.x; r[1] = v0.y; r[2] = v0.z; r[3] = v0.w; r[4] = v1.x; r[5] = v1.y; r[6] = v1.z; r[7] = v1.w;
    } else if (VEC == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float2 v = *reinterpret_cast<const float2*>(p + 2 * q);
            r[2 * q] = v.x; r[2 * q + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] = p[q];
    }
}

// MT = 32-row MFMA tiles per wave along M: the block tile is (64*MT) x 128.  MT = 1 doubles the number of workgroups for
// layers whose 128-row tiling would leave CUs idle (N = 128 / 512 outputs); the k order per output is the same.
template <int VEC, int MT>
__global__ void __launch_bounds__(256, 2)
k_linear(int64_t M, int64_t N, int64_t K, const float* __restrict__ A, int64_t lda, const float* __restrict__ W,
         int64_t ldw, const float* __restrict__ bias, float* __restrict__ C, int64_t ldc, int relu) {
    __shared__ float As[BK * LDT];
    __shared__ float Bs[BK * LDT];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;  // wave position in the 2x2 grid
    constexpr int BM = 64 * MT;
#ifdef GM_GEMM_XCD
    // workgroups are dealt round-robin to the 8 XCDs: give each XCD a contiguous run of tiles (n fastest) so that the A
    // row-tile a run shares stays in that XCD's L2
    const int64_t gx = (N + BN - 1) / BN, gy = (M + BM - 1) / BM, total = gx * gy;
    const int64_t per = (total + 7) / 8;
    const int64_t logical = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per || logical >= total) return;
    const int64_t m0 = (logical / gx) * BM, n0 = (logical % gx) * BN;
#else
    const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
#endif

    const int srow = tid >> 1, skh = (tid & 1) * KPT;
    // rows past the edge are clamped, not zeroed: they only feed output rows/columns that are never stored
    const bool stage_a = srow < BM;   // MT = 1: only half of the threads stage A rows
    const float* pa = A + (m0 + srow < M ? m0 + srow : M - 1) * lda + skh;
    const float* pw = W + (n0 + srow < N ? n0 + srow : N - 1) * ldw + skh;
    float ra[KPT / 8][8], rb[KPT / 8][8];
#pragma unroll
    for (int c = 0; c < KPT / 8; ++c) {
        if (stage_a) load_row8<VEC>(A, lda, m0 + srow, M, skh + 8 * c, K, ra[c]);
        load_row8<VEC>(W, ldw, n0 + srow, N, skh + 8 * c, K, rb[c]);
    }

    floatx16 acc[MT][2];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    for (int64_t k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();  // previous tile's fragment reads are done
#pragma unroll
        for (int c = 0; c < KPT / 8; ++c)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (stage_a) As[(skh + 8 * c + q) * LDT + srow] = ra[c][q];
                Bs[(skh + 8 * c + q) * LDT + srow] = rb[c][q];
            }
        __syncthreads();
        if (k0 + 2 * BK <= K) {  // next tile (fully inside K) in flight while this one is multiplied
#pragma unroll
            for (int c = 0; c < KPT / 8; ++c) {
                if (stage_a) load_row8_fast<VEC>(pa + k0 + BK + 8 * c, ra[c]);
                load_row8_fast<VEC>(pw + k0 + BK + 8 * c, rb[c]);
            }
        } else if (k0 + BK < K) {  // ragged last tile: guarded loads, zeros beyond K
#pragma unroll
            for (int c = 0; c < KPT / 8; ++c) {
                if (stage_a) load_row8<VEC>(A, lda, m0 + srow, M, k0 + BK + skh + 8 * c, K, ra[c]);
                load_row8<VEC>(W, ldw, n0 + srow, N, k0 + BK + skh + 8 * c, K, rb[c]);
            }
        }
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const int krow = (2 * ks + half) * LDT;
            float b0 = Bs[krow + wn * 64 + l31];
            float b1 = Bs[krow + wn * 64 + 32 + l31];
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const float av = As[krow + wm * 32 * MT + 32 * a + l31];
                acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[a][0], 0, 0, 0);
                acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[a][1], 0, 0, 0);
            }
        }
    }

    // epilogue: D tile layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int64_t n = n0 + wn * 64 + b * 32 + l31;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm * 32 * MT + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
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

}  // namespace gm

using namespace gm;

template <int MT>
static int launch_linear(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W, int64_t ldw,
                         const float* bias, float* C, int64_t ldc, int relu, hipStream_t s) {
    constexpr int BM = 64 * MT;
#ifdef GM_GEMM_XCD
    const int64_t total_tiles = ((N + BN - 1) / BN) * ((M + BM - 1) / BM);
    dim3 grid((unsigned)(((total_tiles + 7) / 8) * 8)), block(256);
#else
    dim3 grid((unsigned)((N + BN - 1) / BN), (unsigned)((M + BM - 1) / BM)), block(256);
    GM_REQUIRE(grid.y < 65536u * 16u, "M too large");
#endif
    const bool v4 = lda % 4 == 0 && ldw % 4 == 0 && aligned(A, 16) && aligned(W, 16);
    const bool v2 = lda % 2 == 0 && ldw % 2 == 0 && aligned(A, 8) && aligned(W, 8);
    if (v4) hipLaunchKernelGGL((k_linear<4, MT>), grid, block, 0, s, M, N, K, A, lda, W, ldw, bias, C, ldc, relu);
    else if (v2) hipLaunchKernelGGL((k_linear<2, MT>), grid, block, 0, s, M, N, K, A, lda, W, ldw, bias, C, ldc, relu);
    else hipLaunchKernelGGL((k_linear<1, MT>), grid, block, 0, s, M, N, K, A, lda, W, ldw, bias, C, ldc, relu);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_linear(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W, int64_t ldw,
                         const float* bias, float* C, int64_t ldc, int relu, gm_stream_t stream) {
    GM_REQUIRE(M > 0 && N > 0 && K > 0 && A && W && C, "bad arguments");
    GM_REQUIRE(lda >= K && ldw >= K && ldc >= N, "leading dimension too small");
    hipStream_t s = as_stream(stream);
    // fewer than ~3 workgroups per CU at the 128-row tiling: halve the tile height (same arithmetic per output)
    const int64_t tiles128 = ((N + BN - 1) / BN) * ((M + 127) / 128);
    if (tiles128 < GM_GEMM_SMALL_TILES) return launch_linear<1>(M, N, K, A, lda, W, ldw, bias, C, ldc, relu, s);
    return launch_linear<2>(M, N, K, A, lda, W, ldw, bias, C, ldc, relu, s);
}
