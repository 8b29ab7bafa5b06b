// gm_train.hip -- kernels of the training step log_p_t (model/diffusion.py:606-654, :1035-1094; train_sync.py:94-120):
// the backward halves of the denoiser's operators.  torch.autograd is only the tape (graphmaker_amd/autograd.py);
// every gradient is computed here.
//
//   gm_gemm              C[m][n] = sum_k a(m,k) b(n,k) with either operand stored reduction-major, deterministic split-K:
//                          dX = dY W          (a = dY rows,      b = W  reduction-major)     nn.Linear backward
//                          dW = dY^T X        (a = dY red-major, b = X  reduction-major)
//   gm_colsum            db = column sums of dY
//   gm_layernorm_fwd/bwd nn.LayerNorm (model/gnn.py:30,36) out of place with saved mean / rstd; the backward also applies
//                        the mask of the ReLU in front of it and returns d gamma / d beta
//   gm_spmm_mean_bwd     (D^-1 A)^T G : out[j] = sum_{i in N(j)} G[i] / deg(i)   (A symmetric, model/gnn.py:60-61)
//   gm_embedding_bwd     d emb_Y.weight (model/gnn.py:155)
//   gm_hadamard_gather_bwd   d h of z = h[src] * h[dst] (model/gnn.py:246) by node-grouped segments, no atomics
//   gm_softmax_ce2       F.cross_entropy over 2 classes, mean reduction (LossX / LossE, model/diffusion.py:120-141,474-515)
//
// All reductions run in a fixed order (ascending k inside a split, splits / blocks summed in ascending order), so a
// training step is reproducible run to run.  MFMA f32 (v_mfma_f32_32x32x2_f32) for the GEMMs; the rest is HBM-bound.
#include "gm_common.h"

namespace gm {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int TBM = 128, TBN = 128, TBK = 16;
constexpr int TLDT = TBM + 4;

// ---- operand staging: a 128 (outputs) x 16 (reduction) tile -> 8 registers per thread
// RED = false: memory is [out][red] (reduction contiguous): thread -> out row tid/2, 8 reduction values
// RED = true : memory is [red][out] (outputs contiguous):   thread -> reduction row tid/16, 8 consecutive outputs
template <bool RED>
__device__ __forceinline__ void stage_load(const float* __restrict__ P, int64_t ld, int64_t o0, int64_t O, int64_t k0,
                                           int64_t kend, bool vec, int tid, float (&r)[8]) {
    if (!RED) {
        const int64_t o = o0 + (tid >> 1), kk = k0 + (tid & 1) * 8;
        if (o >= O || kk >= kend) {
#pragma unroll
            for (int q = 0; q < 8; ++q) r[q] = 0.0f;
            return;
        }
        const float* p = P + o * ld + kk;
        if (vec && kk + 8 <= kend) {
            const float4 v0 = *reinterpret_cast<const float4*>(p), v1 = *reinterpret_cast<const float4*>(p + 4);
            r[0] = v0.x; r[1] = v0.y; r[2] = v0.z; r[3] = v0.w; r[4] = v1.x; r[5] = v1.y; r[6] = v1.z; r[7] = v1.w;
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) r[q] = (kk + q < kend) ? p[q] : 0.0f;
        }
    } else {
        const int64_t kk = k0 + (tid >> 4), o = o0 + (tid & 15) * 8;
        if (kk >= kend || o >= O) {
#pragma unroll
            for (int q = 0; q < 8; ++q) r[q] = 0.0f;
            return;
        }
        const float* p = P + kk * ld + o;
        if (vec && o + 8 <= O) {
            const float4 v0 = *reinterpret_cast<const float4*>(p), v1 = *reinterpret_cast<const float4*>(p + 4);
            r[0] = v0.x; r[1] = v0.y; r[2] = v0.z; r[3] = v0.w; r[4] = v1.x; r[5] = v1.y; r[6] = v1.z; r[7] = v1.w;
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) r[q] = (o + q < O) ? p[q] : 0.0f;
        }
    }
}
template <bool RED>
__device__ __forceinline__ void stage_store(float* __restrict__ S, int tid, const float (&r)[8]) {
    if (!RED) {
        const int row = tid >> 1, kh = (tid & 1) * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) S[(kh + q) * TLDT + row] = r[q];
    } else {
        float* d = S + (tid >> 4) * TLDT + (tid & 15) * 8;
        *reinterpret_cast<float4*>(d) = make_float4(r[0], r[1], r[2], r[3]);
        *reinterpret_cast<float4*>(d + 4) = make_float4(r[4], r[5], r[6], r[7]);
    }
}

// C tile epilogue shared by the direct and the split-K reduce path
__device__ __forceinline__ float finish(float v, const float* bias, const float* mask, int64_t ldmask, int relu, int64_t m,
                                        int64_t n) {
    if (bias) v = v + bias[n];
    if (mask && !(mask[m * ldmask + n] > 0.0f)) v = 0.0f;
    if (relu && !(v > 0.0f)) v = 0.0f;
    return v;
}

template <bool A_RED, bool B_RED>
__global__ void __launch_bounds__(256, 2)
k_gemm_g(int64_t M, int64_t N, int64_t K, const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
         const float* __restrict__ bias, const float* __restrict__ mask, int64_t ldmask, int relu, float* __restrict__ C,
         int64_t ldc, int64_t kchunk, int64_t split_stride, int vecA, int vecB) {
    __shared__ __attribute__((aligned(16))) float As[TBK * TLDT];
    __shared__ __attribute__((aligned(16))) float Bs[TBK * TLDT];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    const int64_t m0 = (int64_t)blockIdx.y * TBM, n0 = (int64_t)blockIdx.x * TBN;
    const int64_t kb = (int64_t)blockIdx.z * kchunk;
    const int64_t ke = kb + kchunk < K ? kb + kchunk : K;
    const bool partial = split_stride != 0;
    float* Cz = C + (int64_t)blockIdx.z * split_stride;

    float ra[8], rb[8];
    stage_load<A_RED>(A, lda, m0, M, kb, ke, vecA != 0, tid, ra);
    stage_load<B_RED>(B, ldb, n0, N, kb, ke, vecB != 0, tid, rb);

    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    for (int64_t k0 = kb; k0 < ke; k0 += TBK) {
        __syncthreads();
        stage_store<A_RED>(As, tid, ra);
        stage_store<B_RED>(Bs, tid, rb);
        __syncthreads();
        if (k0 + TBK < ke) {
            stage_load<A_RED>(A, lda, m0, M, k0 + TBK, ke, vecA != 0, tid, ra);
            stage_load<B_RED>(B, ldb, n0, N, k0 + TBK, ke, vecB != 0, tid, rb);
        }
#pragma unroll
        for (int ks = 0; ks < TBK / 2; ++ks) {
            const int krow = (2 * ks + half) * TLDT;
            const float a0 = As[krow + wm * 64 + l31];
            const float a1 = As[krow + wm * 64 + 32 + l31];
            const float b0 = Bs[krow + wn * 64 + l31];
            const float b1 = Bs[krow + wn * 64 + 32 + l31];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }

#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int64_t n = n0 + wn * 64 + b * 32 + l31;
        if (n >= N) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < M) {
                    if (partial) Cz[m * N + n] = acc[a][b][r];
                    else C[m * ldc + n] = finish(acc[a][b][r], bias, mask, ldmask, relu, m, n);
                }
            }
        }
    }
}

// ---- C[m][n] = sum_k A[k][m] B[k][n] with at most 4 rows m, at most 128 columns n and a very long k (dW2 of the pair MLP:
// 2 x 128 over the 2,097,152 pairs of a training batch).  HBM-bound on reading B once; a 128x128 MFMA tile spends its time
// on 126 rows of padding.  A block owns a k range; thread (n, row lane) walks it in ascending k (stride 2, four loads in
// flight), the two lanes are combined in order, and the per-block sums go through the split-K reduction above in
// ascending block order (fixed order, like every other sum here).
template <int MR>
__global__ void __launch_bounds__(256)
k_gemm_tt_skinny(int64_t N, int64_t K, const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                 float* __restrict__ part, int64_t kchunk, int64_t split_stride) {
    __shared__ float red[MR][128];
    const int n = threadIdx.x & 127, rl = threadIdx.x >> 7;
    const int64_t kb = (int64_t)blockIdx.x * kchunk, ke = kb + kchunk < K ? kb + kchunk : K;
    float acc[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[m] = 0.0f;
    if (n < N) {
        int64_t k = kb + rl;
        for (; k + 6 < ke; k += 8) {
            float b[4], a[4][MR];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                b[q] = B[(k + 2 * q) * ldb + n];
#pragma unroll
                for (int m = 0; m < MR; ++m) a[q][m] = A[(k + 2 * q) * lda + m];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int m = 0; m < MR; ++m) acc[m] = __builtin_fmaf(a[q][m], b[q], acc[m]);
        }
        for (; k < ke; k += 2) {
            const float b = B[k * ldb + n];
#pragma unroll
            for (int m = 0; m < MR; ++m) acc[m] = __builtin_fmaf(A[k * lda + m], b, acc[m]);
        }
    }
    if (rl == 1)
#pragma unroll
        for (int m = 0; m < MR; ++m) red[m][n] = acc[m];
    __syncthreads();
    if (rl == 0 && n < N)
#pragma unroll
        for (int m = 0; m < MR; ++m) part[(int64_t)blockIdx.x * split_stride + m * N + n] = acc[m] + red[m][n];
}

// C[m][n] = finish( sum_{s ascending} part[s][m][n] )
__global__ void k_splitk_reduce(int S, int64_t M, int64_t N, const float* __restrict__ part, const float* __restrict__ bias,
                                const float* __restrict__ mask, int64_t ldmask, int relu, float* __restrict__ C, int64_t ldc) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= M * N) return;
    float acc = part[g];
    for (int s = 1; s < S; ++s) acc = acc + part[(int64_t)s * M * N + g];
    const int64_t m = g / N, n = g % N;
    C[m * ldc + n] = finish(acc, bias, mask, ldmask, relu, m, n);
}

// the same sum for many partials: 16 outputs x 16 partial lanes per block; lane l adds partials l, l+16, ... in ascending
// order, then the 16 lane sums are added in ascending lane order (fixed order, independent of the launch)
__global__ void __launch_bounds__(256)
k_splitk_reduce_wide(int S, int64_t M, int64_t N, const float* __restrict__ part, const float* __restrict__ bias,
                     const float* __restrict__ mask, int64_t ldmask, int relu, float* __restrict__ C, int64_t ldc) {
    __shared__ float red[16][17];
    const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int64_t g = (int64_t)blockIdx.x * 16 + o;
    float acc = 0.0f;
    if (g < M * N)
        for (int s = sl; s < S; s += 16) acc = acc + part[(int64_t)s * M * N + g];
    red[sl][o] = acc;
    __syncthreads();
    if (sl == 0 && g < M * N) {
        float v = red[0][o];
#pragma unroll
        for (int l = 1; l < 16; ++l) v = v + red[l][o];
        const int64_t m = g / N, n = g % N;
        C[m * ldc + n] = finish(v, bias, mask, ldmask, relu, m, n);
    }
}
static int launch_reduce(int S, int64_t M, int64_t N, const float* part, const float* bias, const float* mask, int64_t ldmask,
                         int relu, float* C, int64_t ldc, hipStream_t s);

// ---- column sums: part[blk][n] = sum over the block's rows (ascending, 4 row lanes combined in order)
constexpr int CS_ROWS = 64;   // rows per block ...
// ... up to 65536 rows; beyond (the pair batch: 2,097,152 rows) a block takes M/1024 rows, so that the second stage still
// adds about a thousand partials per column, not 32768 in a handful of threads (db1 of the pair MLP: 0.29 + 0.80 ms)
static inline int64_t colsum_rows(int64_t M) { return M <= 65536 ? CS_ROWS : ((M + 1023) / 1024 + 3) / 4 * 4; }
__global__ void __launch_bounds__(256)
k_colsum_partial(int64_t M, int64_t N, const float* __restrict__ X, int64_t ldx, float* __restrict__ part, int64_t rows) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * 64 + c;
    const int64_t r0 = (int64_t)blockIdx.y * rows;
    float acc = 0.0f;
    if (n < N)
        for (int64_t m = r0 + rl; m < r0 + rows && m < M; m += 4) acc = acc + X[m * ldx + n];
    red[rl][c] = acc;
    __syncthreads();
    if (rl == 0 && n < N) part[(int64_t)blockIdx.y * N + n] = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
}

// ---- hidden gradient of Linear -> ReLU -> Linear when the second layer has at most 4 outputs (the pair MLP: 128 -> 2):
//      g1[m][k] = y1[m][k] > 0 ? sum_c g[m][c] W2[c][k] : 0   (fmaf chain over c from 0: the bits gm_gemm(0,1) + mask gives)
//      db1[k]   = sum_m g1[m][k]
// one pass over y1 and g1 (1 GB each at 2M pairs) instead of a 128x128 MFMA tile on a K = 2 product (0.83 ms) followed by a
// column-sum pass that reads g1 again (0.28 ms).  Thread = (float4 column group, row lane); a block owns `rows` rows; column
// sums: ascending rows per lane, lanes in order, blocks in order (k_splitk_reduce).
template <int NC>
__global__ void __launch_bounds__(256)
k_mlp2_hidden_grad(int64_t M, int H4, int shift, const float* __restrict__ g, const float* __restrict__ W2, int64_t ldw,
                   const float* __restrict__ y1, float* __restrict__ g1, float* __restrict__ part, int64_t rows) {
    __shared__ float4 red[256];
    const int c4 = threadIdx.x & (H4 - 1), rl = threadIdx.x >> shift, nrl = 256 >> shift;
    float4 w[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) w[c] = *reinterpret_cast<const float4*>(W2 + c * ldw + 4 * c4);
    const int64_t r0 = (int64_t)blockIdx.x * rows, r1 = r0 + rows < M ? r0 + rows : M;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t m = r0 + rl; m < r1; m += nrl) {
        const float4 y = *reinterpret_cast<const float4*>(y1 + (m * H4 + c4) * 4);
        float gm_[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) gm_[c] = g[m * NC + c];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            v.x = __builtin_fmaf(gm_[c], w[c].x, v.x);
            v.y = __builtin_fmaf(gm_[c], w[c].y, v.y);
            v.z = __builtin_fmaf(gm_[c], w[c].z, v.z);
            v.w = __builtin_fmaf(gm_[c], w[c].w, v.w);
        }
        if (!(y.x > 0.0f)) v.x = 0.0f;
        if (!(y.y > 0.0f)) v.y = 0.0f;
        if (!(y.z > 0.0f)) v.z = 0.0f;
        if (!(y.w > 0.0f)) v.w = 0.0f;
        *reinterpret_cast<float4*>(g1 + (m * H4 + c4) * 4) = v;
        sum.x = sum.x + v.x; sum.y = sum.y + v.y; sum.z = sum.z + v.z; sum.w = sum.w + v.w;
    }
    red[threadIdx.x] = sum;
    __syncthreads();
    if (rl == 0 && part) {
        float4 t = red[c4];
        for (int l = 1; l < nrl; ++l) {
            const float4 a = red[(l << shift) + c4];
            t.x = t.x + a.x; t.y = t.y + a.y; t.z = t.z + a.z; t.w = t.w + a.w;
        }
        *reinterpret_cast<float4*>(part + ((int64_t)blockIdx.x * H4 + c4) * 4) = t;
    }
}

// ---- LayerNorm forward with saved statistics (same arithmetic as k_layernorm in gm_nodeops.hip)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off);
    return v;
}
__global__ void __launch_bounds__(256)
k_layernorm_fwd(int64_t M, int64_t N, const float* __restrict__ X, int64_t ldx, const float* __restrict__ gamma,
                const float* __restrict__ beta, float eps, float* __restrict__ Y, int64_t ldy, float* __restrict__ mean_out,
                float* __restrict__ rstd_out) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* x = X + m * ldx;
    float s = 0.0f;
    for (int64_t c = lane; c < N; c += 64) s = s + x[c];
    const float mean = wave_sum(s) / (float)N;
    float q = 0.0f;
    for (int64_t c = lane; c < N; c += 64) {
        float t = x[c] - mean;
        q = q + t * t;
    }
    const float var = wave_sum(q) / (float)N;
    const float rstd = 1.0f / __builtin_sqrtf(var + eps);
    for (int64_t c = lane; c < N; c += 64) {
        float t = (x[c] - mean) * rstd;
        float g = t * gamma[c];
        Y[m * ldy + c] = g + beta[c];
    }
    if (lane == 0) {
        mean_out[m] = mean;
        rstd_out[m] = rstd;
    }
}

// backward: one wave per row, LN_ROWS rows per block; per-lane column partials of d gamma / d beta are combined over the
// block's 4 waves in LDS and written as part[blk][0][c] (d gamma), part[blk][1][c] (d beta).  N <= 64*LN_MAXQ.
constexpr int LN_ROWS = 32, LN_MAXQ = 16;
__global__ void __launch_bounds__(256)
k_layernorm_bwd(int64_t M, int64_t N, const float* __restrict__ dY, int64_t lddy, const float* __restrict__ X, int64_t ldx,
                const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                int relu_mask, float* __restrict__ dX, int64_t lddx, float* __restrict__ part) {
    extern __shared__ float red[];   // [4][2][N]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float dg[LN_MAXQ], db[LN_MAXQ];
#pragma unroll
    for (int q = 0; q < LN_MAXQ; ++q) dg[q] = db[q] = 0.0f;
    const int64_t r0 = (int64_t)blockIdx.x * LN_ROWS;
    for (int64_t m = r0 + wv; m < r0 + LN_ROWS && m < M; m += 4) {
        const float mu = mean[m], rs = rstd[m];
        const float* x = X + m * ldx;
        const float* dy = dY + m * lddy;
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int q = 0; q < LN_MAXQ; ++q) {
            const int64_t c = lane + 64 * q;
            if (c < N) {
                const float xh = (x[c] - mu) * rs;
                const float g = dy[c] * gamma[c];
                s1 = s1 + g;
                s2 = s2 + g * xh;
                dg[q] = dg[q] + dy[c] * xh;
                db[q] = db[q] + dy[c];
            }
        }
        s1 = wave_sum(s1) / (float)N;
        s2 = wave_sum(s2) / (float)N;
#pragma unroll
        for (int q = 0; q < LN_MAXQ; ++q) {
            const int64_t c = lane + 64 * q;
            if (c < N) {
                const float xv = x[c];
                const float xh = (xv - mu) * rs;
                const float g = dy[c] * gamma[c];
                float v = rs * ((g - s1) - xh * s2);
                if (relu_mask && !(xv > 0.0f)) v = 0.0f;
                dX[m * lddx + c] = v;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < LN_MAXQ; ++q) {
        const int64_t c = lane + 64 * q;
        if (c < N) {
            red[(wv * 2 + 0) * N + c] = dg[q];
            red[(wv * 2 + 1) * N + c] = db[q];
        }
    }
    __syncthreads();
    for (int64_t c = threadIdx.x; c < 2 * N; c += 256) {
        const float v = ((red[0 * 2 * N + c] + red[1 * 2 * N + c]) + red[2 * 2 * N + c]) + red[3 * 2 * N + c];
        part[(int64_t)blockIdx.x * 2 * N + c] = v;
    }
}

// ---- d emb.weight[c][col] = sum_{v : idx[v] == c} G[v][col]: blocks of EB_ROWS rows write partial sums (rows ascending per
// wave, 4 waves combined in order), the chunks are then added in ascending order
constexpr int EB_ROWS = 512;
__global__ void __launch_bounds__(256)
k_embedding_bwd(int64_t n, int C, int w, const int32_t* __restrict__ idx, const float* __restrict__ G, int64_t ldg,
                float* __restrict__ part) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = blockIdx.x;
    const int col = blockIdx.y * 64 + lane;
    const int64_t r0 = (int64_t)blockIdx.z * EB_ROWS;
    float acc = 0.0f;
    for (int64_t v = r0 + wv; v < r0 + EB_ROWS && v < n; v += 4)
        if (idx[v] == c && col < w) acc = acc + G[v * ldg + col];
    red[wv][lane] = acc;
    __syncthreads();
    if (wv == 0 && col < w)
        part[((int64_t)blockIdx.z * C + c) * w + col] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

// ---- d h[i] = sum over the batch entries that touch node i (grouped by node, ascending position in cat[src, dst]) of
//      dZ[b] * h[other end].  order[k] indexes cat[src, dst]: b = order[k] mod B, role = order[k] >= B.
__global__ void __launch_bounds__(256)
k_hadamard_gather_bwd(int64_t n, int64_t B, int d, const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                      const int64_t* __restrict__ segptr, const int64_t* __restrict__ order, const float* __restrict__ h,
                      int64_t ldh, const float* __restrict__ dZ, int64_t lddz, float* __restrict__ dh, int64_t lddh) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t b0 = segptr[i], b1 = segptr[i + 1];
    // A node of the pair batch has some 300 entries; walked one at a time, each is a chain of three dependent loads (order ->
    // src/dst -> the two rows) and the wave waits for every link (1.27 ms at 2M pairs).  The lanes resolve 64 entries at once,
    // then the rows of 8 entries are in flight together; the fmaf chain itself stays in ascending k.
    for (int c0 = 0; c0 < d; c0 += 128) {
        float acc0 = 0.0f, acc1 = 0.0f;
        const int ca = c0 + lane, cb = c0 + 64 + lane;
        for (int64_t k0 = b0; k0 < b1; k0 += 64) {
            const int cnt = b1 - k0 < 64 ? (int)(b1 - k0) : 64;
            int my_b = 0, my_o = 0;
            if (lane < cnt) {
                const int64_t it = order[k0 + lane];
                const int64_t b = it >= B ? it - B : it;
                my_b = (int)b;
                my_o = (int)(it >= B ? src[b] : dst[b]);
            }
            for (int q0 = 0; q0 < cnt; q0 += 8) {
                float z0[8], z1[8], g0[8], g1[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const bool in = q0 + q < cnt;
                    const int64_t b = __shfl(my_b, (q0 + q) & 63), o = __shfl(my_o, (q0 + q) & 63);
                    z0[q] = in && ca < d ? dZ[b * lddz + ca] : 0.0f;
                    g0[q] = in && ca < d ? h[o * ldh + ca] : 0.0f;
                    z1[q] = in && cb < d ? dZ[b * lddz + cb] : 0.0f;
                    g1[q] = in && cb < d ? h[o * ldh + cb] : 0.0f;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q0 + q < cnt) {
                        acc0 = __builtin_fmaf(z0[q], g0[q], acc0);
                        acc1 = __builtin_fmaf(z1[q], g1[q], acc1);
                    }
            }
        }
        if (ca < d) dh[i * lddh + ca] = acc0;
        if (cb < d) dh[i * lddh + cb] = acc1;
    }
}

// ---- 2-class cross entropy, mean over `total` items.  dlogits = (softmax - onehot) / total.  Fixed grid of CE_BLOCKS
// blocks with grid-stride loops; block partial sums are combined in ascending block order by k_splitk_reduce.
constexpr int CE_BLOCKS = 1024;
__global__ void __launch_bounds__(256)
k_softmax_ce2(int64_t total, const float* __restrict__ logits, const uint8_t* __restrict__ target, float inv_total,
              float* __restrict__ part, float* __restrict__ dlogits) {
    __shared__ float red[4];
    float acc = 0.0f;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)CE_BLOCKS * 256) {
        const float2 l = *reinterpret_cast<const float2*>(logits + 2 * g);
        const int y = target[g] ? 1 : 0;
        const float m = l.x > l.y ? l.x : l.y;
        const float e0 = expf_nonpos(l.x - m), e1 = expf_nonpos(l.y - m);
        const float s = e0 + e1;
        const float ly = y ? l.y : l.x;
        acc = acc + ((m + __logf(s)) - ly);
        if (dlogits) {
            const float p0 = e0 / s, p1 = e1 / s;
            *reinterpret_cast<float2*>(dlogits + 2 * g) =
                make_float2((p0 - (y ? 0.0f : 1.0f)) * inv_total, (p1 - (y ? 1.0f : 0.0f)) * inv_total);
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (((red[0] + red[1]) + red[2]) + red[3]) * inv_total;
}

// ---- the same loss from the logit DIFFERENCE d = l1 - l0 (attribute head folded by gm_head_diff_prepare):
//      CE = log(1 + exp(-|d|)) + max(-s, 0), s = target ? d : -d;   dCE/dd = (sigmoid(d) - target) / total.
// Same fixed grid and summation order as k_softmax_ce2.
__global__ void __launch_bounds__(256)
k_ce2_diff(int64_t total, const float* __restrict__ diff, const uint8_t* __restrict__ target, float inv_total,
           float* __restrict__ part, float* __restrict__ ddiff) {
    __shared__ float red[4];
    float acc = 0.0f;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)CE_BLOCKS * 256) {
        const float d = diff[g];
        const bool y = target[g] != 0, pos = d > 0.0f;
        const float e = expf_nonpos(pos ? -d : d);                 // exp(-|d|): the smaller class against the larger
        const float s = 1.0f + e;
        const float sgn = y ? d : -d;
        acc = acc + (__logf(s) + (sgn < 0.0f ? -sgn : 0.0f));
        if (ddiff) {
            const float p1 = pos ? 1.0f / s : e / s;               // sigmoid(d)
            ddiff[g] = (p1 - (y ? 1.0f : 0.0f)) * inv_total;
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (((red[0] + red[1]) + red[2]) + red[3]) * inv_total;
}

// gradient of the folded head back to mlp_out.2: dW[2f+1] = dWd[f], dW[2f] = -dWd[f] (exact), the same for the bias
__global__ void k_head_diff_bwd(int64_t F, int64_t Hc, const float* __restrict__ dWd, const float* __restrict__ dbd,
                                float* __restrict__ dW, float* __restrict__ db) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= F * Hc) return;
    const int64_t f = g / Hc, k = g - f * Hc;
    const float v = dWd[g];
    dW[(2 * f + 1) * Hc + k] = v;
    dW[2 * f * Hc + k] = -v;
    if (g < F && db) { db[2 * g + 1] = dbd[g]; db[2 * g] = -dbd[g]; }
}

// ---- elementwise pieces of the tape
__global__ void k_relu_bwd(int64_t total, const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) out[i] = y[i] > 0.0f ? g[i] : 0.0f;
}
// nn.Dropout (model/gnn.py:31,37): keep with probability 1-p, scale by 1/(1-p); the mask is a pure function of
// (seed, call, element index) so the backward regenerates it instead of storing it.  Philox stream 7.
__global__ void k_dropout(int64_t total, const float* __restrict__ x, float p, float scale, uint64_t seed, uint32_t call,
                          float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float u = u_idx64(seed, call, 7u, (uint64_t)i);
    out[i] = u < p ? 0.0f : x[i] * scale;
}
__global__ void k_embedding_fwd(int64_t n, int w, const int32_t* __restrict__ idx, const float* __restrict__ table, int64_t ldt,
                                float* __restrict__ dst, int64_t ldd) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n * w) return;
    dst[(g / w) * ldd + (g % w)] = table[(int64_t)idx[g / w] * ldt + (g % w)];
}

// ---- validation: denoising-match KL term (BaseModel.denoise_match_Z, model/diffusion.py:258-303) for 2 classes.
// Item g uses table g % ntab (ntab = F for attributes laid out (v, f); 1 for edges).  Qt / Qbs: ntab x 4 floats Q[r*2+c].
//   left[k]  = Q_t[k][z_t]                  (Z_t_one_hot @ Q_t^T)
//   true[k]  = left[k] * Qbar_s[z_0][k]     / sum_k (0 -> 1)
//   pred[k]  = left[k] * sum_i p[i] Qbar_s[i][k] / sum_k (0 -> 1)
//   kl[k]    = true[k] * (log true[k] - log pred[k])   (0 where true[k] == 0), clamped at 0; result = mean over items and k
__global__ void __launch_bounds__(256)
k_denoise_match2(int64_t total, const float* __restrict__ logits, const uint8_t* __restrict__ z_t,
                 const uint8_t* __restrict__ z_0, const float* __restrict__ Qt, const float* __restrict__ Qbs, int64_t ntab,
                 float inv_count, float* __restrict__ part) {
    __shared__ float red[4];
    float acc = 0.0f;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)CE_BLOCKS * 256) {
        const float2 l = *reinterpret_cast<const float2*>(logits + 2 * g);
        float p0, p1;
        softmax2(l.x, l.y, p0, p1);
        const float* qt = Qt + 4 * (g % ntab);
        const float* qs = Qbs + 4 * (g % ntab);
        const int e = z_t[g] ? 1 : 0, r = z_0[g] ? 1 : 0;
        const float left0 = qt[0 + e], left1 = qt[2 + e];
        float t0 = left0 * qs[2 * r], t1 = left1 * qs[2 * r + 1];
        float dt = t0 + t1;
        if (dt == 0.0f) dt = 1.0f;
        t0 = t0 / dt;
        t1 = t1 / dt;
        float q0 = left0 * (p0 * qs[0] + p1 * qs[2]), q1 = left1 * (p0 * qs[1] + p1 * qs[3]);
        float dq = q0 + q1;
        if (dq == 0.0f) dq = 1.0f;
        q0 = q0 / dq;
        q1 = q1 / dq;
        float k0 = t0 > 0.0f ? t0 * (logf(t0) - logf(q0)) : 0.0f;
        float k1 = t1 > 0.0f ? t1 * (logf(t1) - logf(q1)) : 0.0f;
        if (!(k0 > 0.0f)) k0 = 0.0f;
        if (!(k1 > 0.0f)) k1 = 0.0f;
        acc = acc + (k0 + k1);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (((red[0] + red[1]) + red[2]) + red[3]) * inv_count;
}

// out[b] = bit (i[b], j[b]) of the bitmap  (E_t[batch_src, batch_dst], model/diffusion.py:765)
__global__ void k_bitmap_lookup(int64_t B, const int64_t* __restrict__ i, const int64_t* __restrict__ j,
                                const uint32_t* __restrict__ bits, int64_t wpr, uint8_t* __restrict__ out) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    out[b] = (bits[i[b] * wpr + (j[b] >> 5)] >> (j[b] & 31)) & 1u;
}

static int launch_reduce(int S, int64_t M, int64_t N, const float* part, const float* bias, const float* mask, int64_t ldmask,
                         int relu, float* C, int64_t ldc, hipStream_t s) {
    if (S <= 8)
        hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, s, S, M, N, part, bias, mask,
                           ldmask, relu, C, ldc);
    else
        hipLaunchKernelGGL(k_splitk_reduce_wide, dim3((unsigned)((M * N + 15) / 16)), dim3(256), 0, s, S, M, N, part, bias, mask,
                           ldmask, relu, C, ldc);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

static inline bool tt_skinny(int a_red, int b_red, int64_t M, int64_t N, int64_t K) {
    return a_red && b_red && M <= 4 && N <= 128 && K >= 65536;
}
static inline int pick_splits(int64_t M, int64_t N, int64_t K) {
    if (M <= 4 && N <= 128 && K >= 65536) return 2048;      // k_gemm_tt_skinny: one block per k range (also covers the
                                                            // workspace of the general kernel, which asks for fewer)
    const int64_t tiles = ((M + TBM - 1) / TBM) * ((N + TBN - 1) / TBN);
    int64_t smax = K / 256;
    if (smax > 1024) smax = 1024;
    if (tiles >= 512 || smax < 2) return 1;
    // the grid is tiles x splits workgroups on 512 slots (256 CUs, two workgroups each): take the split count that fills
    // whole rounds best, each further split charged for its share of the reduction pass.  1744 x 1744 over 13752 rows is 196
    // tiles: 2 splits (392 workgroups, 0.77 of a round) took 1.04 ms, 5 splits (980 = 0.96 of two rounds) take less.
    const double charge = 0.01 * (tiles < 196 ? (double)tiles / 196.0 : 1.0);
    int64_t best = 1;
    double best_score = -1.0;
    const int64_t stop = smax < 2 * (512 / tiles + 1) ? smax : 2 * (512 / tiles + 1);
    for (int64_t c = 1; c <= stop; ++c) {
        const int64_t w = tiles * c;
        const double fill = (double)w / (double)(((w + 511) / 512) * 512);
        const double score = fill - charge * (double)c;
        if (score > best_score) { best_score = score; best = c; }
    }
    return (int)best;
}

}  // namespace gm

using namespace gm;

extern "C" int64_t gm_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    const int s = pick_splits(M, N, K);
    return s > 1 ? (int64_t)s * M * N * 4 : 0;
}

extern "C" int gm_gemm(int a_red, int b_red, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                       int64_t ldb, const float* bias, const float* mask, int64_t ldmask, int relu, float* C, int64_t ldc,
                       void* workspace, int64_t workspace_bytes, gm_stream_t stream) {
    GM_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C, "bad arguments");
    GM_REQUIRE(lda >= (a_red ? M : K) && ldb >= (b_red ? N : K) && ldc >= N, "leading dimension too small");
    GM_REQUIRE(!mask || ldmask >= N, "mask leading dimension too small");
    hipStream_t s = as_stream(stream);
    int splits = pick_splits(M, N, K);
    int64_t kchunk = K;
    if (tt_skinny(a_red, b_red, M, N, K)) {
        kchunk = ((K + splits - 1) / splits + 1) / 2 * 2;      // even: the two row lanes of a block keep their parity
        splits = (int)((K + kchunk - 1) / kchunk);
        GM_REQUIRE(workspace && workspace_bytes >= (int64_t)splits * M * N * 4, "gm_gemm needs %lld workspace bytes (gm_gemm_workspace_bytes)",
                   (long long)((int64_t)splits * M * N * 4));
        float* part = static_cast<float*>(workspace);
#define GM_LAUNCH_SK(MR) hipLaunchKernelGGL((k_gemm_tt_skinny<MR>), dim3((unsigned)splits), dim3(256), 0, s, N, K, A, lda, B, ldb, part, kchunk, M * N)
        switch ((int)M) {
            case 1: GM_LAUNCH_SK(1); break;
            case 2: GM_LAUNCH_SK(2); break;
            case 3: GM_LAUNCH_SK(3); break;
            default: GM_LAUNCH_SK(4); break;
        }
#undef GM_LAUNCH_SK
        GM_LAUNCH_CHECK();
        return launch_reduce(splits, M, N, part, bias, mask, ldmask, relu, C, ldc, s);
    }
    if (splits > 1) {
        kchunk = ((K + splits - 1) / splits + TBK - 1) / TBK * TBK;
        splits = (int)((K + kchunk - 1) / kchunk);
    }
    if (splits > 1)
        GM_REQUIRE(workspace && workspace_bytes >= (int64_t)splits * M * N * 4 && aligned(workspace, 16),
                   "gm_gemm needs %lld workspace bytes (gm_gemm_workspace_bytes)", (long long)((int64_t)splits * M * N * 4));
    dim3 grid((unsigned)((N + TBN - 1) / TBN), (unsigned)((M + TBM - 1) / TBM), (unsigned)splits), block(256);
    GM_REQUIRE(grid.y < 65536u, "M too large");
    const int vecA = lda % 4 == 0 && aligned(A, 16), vecB = ldb % 4 == 0 && aligned(B, 16);
    float* out = splits > 1 ? static_cast<float*>(workspace) : C;
    const int64_t stride = splits > 1 ? M * N : 0;
#define GM_LAUNCH_G(AR, BR)                                                                                              \
    hipLaunchKernelGGL((k_gemm_g<AR, BR>), grid, block, 0, s, M, N, K, A, lda, B, ldb, bias, mask, ldmask, relu, out, ldc, \
                       kchunk, stride, vecA, vecB)
    if (a_red && b_red) GM_LAUNCH_G(true, true);
    else if (a_red) GM_LAUNCH_G(true, false);
    else if (b_red) GM_LAUNCH_G(false, true);
    else GM_LAUNCH_G(false, false);
#undef GM_LAUNCH_G
    GM_LAUNCH_CHECK();
    if (splits > 1) {
        return launch_reduce(splits, M, N, static_cast<const float*>(workspace), bias, mask, ldmask, relu, C, ldc, s);
    }
    return GM_OK;
}

extern "C" int64_t gm_colsum_workspace_bytes(int64_t M, int64_t N) {
    const int64_t rows = colsum_rows(M);
    return ((M + rows - 1) / rows) * N * 4;
}

extern "C" int gm_colsum(int64_t M, int64_t N, const float* X, int64_t ldx, float* out, void* workspace, int64_t workspace_bytes,
                         gm_stream_t stream) {
    GM_REQUIRE(M > 0 && N > 0 && X && out && ldx >= N, "bad arguments");
    const int64_t rows = colsum_rows(M);
    const int64_t nblk = (M + rows - 1) / rows;
    GM_REQUIRE(workspace && workspace_bytes >= nblk * N * 4, "gm_colsum needs %lld workspace bytes", (long long)(nblk * N * 4));
    GM_REQUIRE(nblk < 65536, "M too large");
    hipStream_t s = as_stream(stream);
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(k_colsum_partial, dim3((unsigned)((N + 63) / 64), (unsigned)nblk), dim3(256), 0, s, M, N, X, ldx, part, rows);
    GM_LAUNCH_CHECK();
    return launch_reduce((int)nblk, 1, N, part, nullptr, nullptr, 0, 0, out, N, s);
}

static inline bool hidden_grad_ok(int64_t H, int64_t N) {
    const int64_t h4 = H / 4;
    return H % 4 == 0 && h4 >= 1 && h4 <= 256 && (h4 & (h4 - 1)) == 0 && N >= 1 && N <= 4;
}
extern "C" int64_t gm_mlp2_hidden_grad_workspace_bytes(int64_t M, int64_t H, int64_t N) {
    if (!hidden_grad_ok(H, N) || M <= 0) return -1;
    const int64_t rows = colsum_rows(M);
    return ((M + rows - 1) / rows) * H * 4;
}

extern "C" int gm_mlp2_hidden_grad(int64_t M, int64_t H, int64_t N, const float* g, const float* W2, int64_t ldw, const float* y1,
                                   float* g1, float* db1, void* workspace, int64_t workspace_bytes, gm_stream_t stream) {
    GM_REQUIRE(M > 0 && g && W2 && y1 && g1 && ldw >= H, "bad arguments");
    GM_REQUIRE(hidden_grad_ok(H, N), "hidden width %lld must be 4 x a power of two up to 1024 and the output width %lld at most 4",
               (long long)H, (long long)N);
    GM_REQUIRE(ldw % 4 == 0 && aligned(W2, 16) && aligned(y1, 16) && aligned(g1, 16), "W2, y1, g1 must be 16-byte aligned");
    const int64_t rows = colsum_rows(M), nblk = (M + rows - 1) / rows;
    GM_REQUIRE(!db1 || (workspace && workspace_bytes >= nblk * H * 4 && aligned(workspace, 16)),
               "gm_mlp2_hidden_grad needs %lld workspace bytes", (long long)(nblk * H * 4));
    GM_REQUIRE(nblk < (1ll << 31), "M too large");
    hipStream_t s = as_stream(stream);
    const int H4 = (int)(H / 4);
    int shift = 0;
    while ((1 << shift) < H4) ++shift;
    float* part = db1 ? static_cast<float*>(workspace) : nullptr;
#define GM_LAUNCH_HG(NC) hipLaunchKernelGGL((k_mlp2_hidden_grad<NC>), dim3((unsigned)nblk), dim3(256), 0, s, M, H4, shift, g, W2, ldw, y1, g1, part, rows)
    switch ((int)N) {
        case 1: GM_LAUNCH_HG(1); break;
        case 2: GM_LAUNCH_HG(2); break;
        case 3: GM_LAUNCH_HG(3); break;
        default: GM_LAUNCH_HG(4); break;
    }
#undef GM_LAUNCH_HG
    GM_LAUNCH_CHECK();
    if (db1) return launch_reduce((int)nblk, 1, H, part, nullptr, nullptr, 0, 0, db1, H, s);
    return GM_OK;
}

extern "C" int gm_layernorm_fwd(int64_t M, int64_t N, const float* X, int64_t ldx, const float* gamma, const float* beta,
                                float eps, float* Y, int64_t ldy, float* mean, float* rstd, gm_stream_t stream) {
    GM_REQUIRE(M > 0 && N > 0 && X && gamma && beta && Y && mean && rstd && ldx >= N && ldy >= N, "bad arguments");
    hipLaunchKernelGGL(k_layernorm_fwd, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, as_stream(stream), M, N, X, ldx, gamma,
                       beta, eps, Y, ldy, mean, rstd);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int64_t gm_layernorm_bwd_workspace_bytes(int64_t M, int64_t N) { return ((M + LN_ROWS - 1) / LN_ROWS) * 2 * N * 4; }

extern "C" int gm_layernorm_bwd(int64_t M, int64_t N, const float* dY, int64_t lddy, const float* X, int64_t ldx,
                                const float* mean, const float* rstd, const float* gamma, int relu_mask, float* dX,
                                int64_t lddx, float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes,
                                gm_stream_t stream) {
    GM_REQUIRE(M > 0 && N > 0 && dY && X && mean && rstd && gamma && dX && dgamma && dbeta, "bad arguments");
    GM_REQUIRE(lddy >= N && ldx >= N && lddx >= N, "leading dimension too small");
    GM_REQUIRE(N <= 64 * LN_MAXQ, "LayerNorm width %lld exceeds %d", (long long)N, 64 * LN_MAXQ);
    GM_REQUIRE(dbeta == dgamma + N, "dgamma and dbeta must be the two halves of one (2, N) buffer");
    const int64_t nblk = (M + LN_ROWS - 1) / LN_ROWS;
    GM_REQUIRE(workspace && workspace_bytes >= nblk * 2 * N * 4, "gm_layernorm_bwd needs %lld workspace bytes",
               (long long)(nblk * 2 * N * 4));
    hipStream_t s = as_stream(stream);
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(k_layernorm_bwd, dim3((unsigned)nblk), dim3(256), (size_t)(8 * N * 4), s, M, N, dY, lddy, X, ldx, mean,
                       rstd, gamma, relu_mask, dX, lddx, part);
    GM_LAUNCH_CHECK();
    return launch_reduce((int)nblk, 1, 2 * N, part, nullptr, nullptr, 0, 0, dgamma, 2 * N, s);
}

extern "C" int64_t gm_embedding_bwd_workspace_bytes(int64_t n, int C, int w) {
    return ((n + EB_ROWS - 1) / EB_ROWS) * (int64_t)C * w * 4;
}

extern "C" int gm_embedding_bwd(int64_t n, int C, int w, const int32_t* idx, const float* G, int64_t ldg, float* dW, int64_t ldw,
                                void* workspace, int64_t workspace_bytes, gm_stream_t stream) {
    GM_REQUIRE(n > 0 && C > 0 && w > 0 && idx && G && dW && ldg >= w && ldw >= w, "bad arguments");
    const int64_t nchunk = (n + EB_ROWS - 1) / EB_ROWS;
    GM_REQUIRE(workspace && workspace_bytes >= nchunk * C * w * 4, "gm_embedding_bwd needs %lld workspace bytes",
               (long long)(nchunk * C * w * 4));
    GM_REQUIRE(nchunk < 65536, "n too large");
    hipStream_t s = as_stream(stream);
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(k_embedding_bwd, dim3((unsigned)C, (unsigned)((w + 63) / 64), (unsigned)nchunk), dim3(256), 0, s, n, C, w,
                       idx, G, ldg, part);
    GM_LAUNCH_CHECK();
    return launch_reduce((int)nchunk, C, w, part, nullptr, nullptr, 0, 0, dW, ldw, s);
}

extern "C" int gm_hadamard_gather_bwd(int64_t n, int64_t B, int64_t d, const int64_t* src, const int64_t* dst,
                                      const int64_t* segptr, const int64_t* order, const float* h, int64_t ldh, const float* dZ,
                                      int64_t lddz, float* dh, int64_t lddh, gm_stream_t stream) {
    GM_REQUIRE(n > 0 && B > 0 && d > 0 && src && dst && segptr && order && h && dZ && dh, "bad arguments");
    GM_REQUIRE(ldh >= d && lddz >= d && lddh >= d, "leading dimension too small");
    GM_REQUIRE(B < (1ll << 31) && n < (1ll << 31), "batch or node count too large");
    hipLaunchKernelGGL(k_hadamard_gather_bwd, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, as_stream(stream), n, B, (int)d, src,
                       dst, segptr, order, h, ldh, dZ, lddz, dh, lddh);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int64_t gm_softmax_ce2_workspace_bytes(void) { return CE_BLOCKS * 4; }

extern "C" int gm_softmax_ce2(int64_t total, const float* logits, const uint8_t* target, float* loss, float* dlogits,
                              void* workspace, int64_t workspace_bytes, gm_stream_t stream) {
    GM_REQUIRE(total > 0 && logits && target && loss && aligned(logits, 8) && (!dlogits || aligned(dlogits, 8)), "bad arguments");
    GM_REQUIRE(workspace && workspace_bytes >= CE_BLOCKS * 4, "gm_softmax_ce2 needs %d workspace bytes", CE_BLOCKS * 4);
    hipStream_t s = as_stream(stream);
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(k_softmax_ce2, dim3(CE_BLOCKS), dim3(256), 0, s, total, logits, target, 1.0f / (float)total, part, dlogits);
    GM_LAUNCH_CHECK();
    return launch_reduce(CE_BLOCKS, 1, 1, part, nullptr, nullptr, 0, 0, loss, 1, s);
}

extern "C" int gm_ce2_diff(int64_t total, const float* diff, const uint8_t* target, float* loss, float* ddiff, void* workspace,
                           int64_t workspace_bytes, gm_stream_t stream) {
    GM_REQUIRE(total > 0 && diff && target && loss, "bad arguments");
    GM_REQUIRE(workspace && workspace_bytes >= CE_BLOCKS * 4, "gm_ce2_diff needs %d workspace bytes", CE_BLOCKS * 4);
    hipStream_t s = as_stream(stream);
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(k_ce2_diff, dim3(CE_BLOCKS), dim3(256), 0, s, total, diff, target, 1.0f / (float)total, part, ddiff);
    GM_LAUNCH_CHECK();
    return launch_reduce(CE_BLOCKS, 1, 1, part, nullptr, nullptr, 0, 0, loss, 1, s);
}

extern "C" int gm_head_diff_bwd(int64_t F, int64_t Hc, const float* dWd, const float* dbd, float* dW, float* db,
                                gm_stream_t stream) {
    GM_REQUIRE(F > 0 && Hc > 0 && dWd && dW && (!db || dbd), "bad arguments");
    GM_REQUIRE(F * Hc < (1ll << 39), "F*Hc too large");
    hipLaunchKernelGGL(k_head_diff_bwd, dim3((unsigned)((F * Hc + 255) / 256)), dim3(256), 0, as_stream(stream), F, Hc, dWd, dbd, dW,
                       db);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_relu_bwd(int64_t total, const float* g, const float* y, float* out, gm_stream_t stream) {
    GM_REQUIRE(total > 0 && g && y && out, "bad arguments");
    hipLaunchKernelGGL(k_relu_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), total, g, y, out);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_dropout(int64_t total, const float* x, float p, uint64_t seed, uint32_t call, float* out, gm_stream_t stream) {
    GM_REQUIRE(total > 0 && x && out && p >= 0.0f && p < 1.0f, "bad arguments (0 <= p < 1)");
    hipLaunchKernelGGL(k_dropout, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), total, x, p,
                       1.0f / (1.0f - p), seed, call, out);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_embedding_fwd(int64_t n, int C, int w, const int32_t* idx, const float* weight, int64_t ldw, float* out,
                                int64_t ldo, gm_stream_t stream) {
    GM_REQUIRE(n > 0 && C > 0 && w > 0 && idx && weight && out && ldw >= w && ldo >= w, "bad arguments");
    hipLaunchKernelGGL(k_embedding_fwd, dim3((unsigned)((n * w + 255) / 256)), dim3(256), 0, as_stream(stream), n, w, idx, weight,
                       ldw, out, ldo);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_denoise_match2(int64_t total, const float* logits, const uint8_t* z_t, const uint8_t* z_0, const float* Q_t,
                                 const float* Q_bar_s, int64_t ntab, float* out, void* workspace, int64_t workspace_bytes,
                                 gm_stream_t stream) {
    GM_REQUIRE(total > 0 && logits && z_t && z_0 && Q_t && Q_bar_s && ntab > 0 && out && aligned(logits, 8), "bad arguments");
    GM_REQUIRE(workspace && workspace_bytes >= CE_BLOCKS * 4, "gm_denoise_match2 needs %d workspace bytes", CE_BLOCKS * 4);
    hipStream_t s = as_stream(stream);
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(k_denoise_match2, dim3(CE_BLOCKS), dim3(256), 0, s, total, logits, z_t, z_0, Q_t, Q_bar_s, ntab,
                       1.0f / (2.0f * (float)total), part);
    GM_LAUNCH_CHECK();
    return launch_reduce(CE_BLOCKS, 1, 1, part, nullptr, nullptr, 0, 0, out, 1, s);
}

extern "C" int gm_bitmap_lookup(int64_t n, int64_t wpr, const uint32_t* bits, int64_t B, const int64_t* i, const int64_t* j,
                                uint8_t* out, gm_stream_t stream) {
    GM_REQUIRE(n > 0 && wpr >= (n + 31) / 32 && bits && B > 0 && i && j && out, "bad arguments");
    hipLaunchKernelGGL(k_bitmap_lookup, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, as_stream(stream), B, i, j, bits, wpr, out);
    GM_LAUNCH_CHECK();
    return GM_OK;
}
