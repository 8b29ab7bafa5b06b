/*
 * oracle/gm_oracle.c -- CPU restatement of GraphMaker's reverse-diffusion denoising step.
 *
 * TEST INFRASTRUCTURE ONLY (see gm_oracle.h).  Build: oracle/Makefile -> oracle/libgm_oracle.so
 * Compile with -ffp-contract=off: every fused multiply-add below is an explicit fmaf(), every other
 * a*b+c is two roundings, so the arithmetic is exactly what the comments say.
 *
 * Floating-point orders are chosen so that the HIP library (f32 MFMA = k-ordered fmaf chains,
 * wave-level butterfly reductions) can reproduce the oracle bit for bit; against the reference
 * (torch fp32, BLAS orderings unknown) results agree within fp32 tolerance (tests/test_oracle_golden.py).
 */
#include "gm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define GMO_VERSION 0x00010000u

uint32_t gmo_version(void) { return GMO_VERSION; }

void gmo_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int gmo_get_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ RNG */
/* Philox4x32-10 (Salmon et al., SC'11), the generator torch's device RNG also uses.  The reference draws
 * with torch.multinomial (model/diffusion.py:202,228,421,833); shared-RNG parity replaces those draws by
 * thresholding uniforms, and this is where the uniforms come from. */
void gmo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

float gmo_u01(uint32_t word) { return (float)(word >> 8) * 5.9604644775390625e-08f; /* 2^-24 */ }

static inline float philox_u(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, int w) {
    uint32_t ctr[4] = {c0, c1, c2, c3};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t out[4];
    gmo_philox4x32_10(ctr, key, out);
    return gmo_u01(out[w]);
}
static inline float u_Y(uint64_t seed, int64_t v) {
    return philox_u(seed, (uint32_t)(v >> 2), 0u, 0u, GMO_STREAM_Y, (int)(v & 3));
}
static inline float u_idx64(uint64_t seed, uint32_t step, uint32_t stream, uint64_t idx) {
    uint64_t q = idx >> 2;
    return philox_u(seed, (uint32_t)q, (uint32_t)(q >> 32), step, stream, (int)(idx & 3));
}
static inline float u_E(uint64_t seed, uint32_t step, int64_t i, int64_t j) {
    return philox_u(seed, (uint32_t)(j >> 2), (uint32_t)i, step, GMO_STREAM_E, (int)(j & 3));
}
static inline int64_t pair_rank(int64_t n, int64_t i, int64_t j) { /* torch.triu_indices(n,n,offset=1) order */
    return i * (2 * n - i - 1) / 2 + (j - i - 1);
}

void gmo_fill_uniform_Y(uint64_t seed, int64_t n, float* u) {
    for (int64_t v = 0; v < n; ++v) u[v] = u_Y(seed, v);
}
void gmo_fill_uniform_X(uint64_t seed, uint32_t step, int64_t n, int64_t F, float* u) {
#pragma omp parallel for schedule(static)
    for (int64_t idx = 0; idx < n * F; ++idx) u[idx] = u_idx64(seed, step, GMO_STREAM_X, (uint64_t)idx);
}
void gmo_fill_uniform_E_prior(uint64_t seed, int64_t n, float* u) {
#pragma omp parallel for schedule(static)
    for (int64_t idx = 0; idx < n * n; ++idx) u[idx] = u_idx64(seed, 0u, GMO_STREAM_E_PRIOR, (uint64_t)idx);
}
void gmo_fill_uniform_E(uint64_t seed, uint32_t step, int64_t n, float* u) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = i + 1; j < n; ++j) u[pair_rank(n, i, j)] = u_E(seed, step, i, j);
}

/* ------------------------------------------------------------------ schedule / transitions */
/* model/diffusion.py:96-118 NoiseSchedule.__init__ : cosine schedule in float64, then fp32 log/cumsum/exp */
void gmo_noise_schedule(int T, float* betas, float* alphas, float* alpha_bars) {
    int num_steps = T + 2;
    double* ab = (double*)malloc(sizeof(double) * (size_t)num_steps);
    double step = (double)num_steps / (double)(num_steps - 1); /* np.linspace(0, num_steps, num_steps) */
    const double s = 0.008;
    for (int k = 0; k < num_steps; ++k) {
        double t = (k == num_steps - 1) ? (double)num_steps : (double)k * step;
        double c = cos(0.5 * M_PI * ((t / num_steps) + s) / (1 + s));
        ab[k] = c * c;
    }
    double ab0 = ab[0];
    for (int k = 0; k < num_steps; ++k) ab[k] /= ab0;
    float logsum = 0.0f;
    for (int k = 0; k <= T; ++k) {
        double a = ab[k + 1] / ab[k];
        float b = (float)(1.0 - a);
        betas[k] = b;
        float bc = b < 0.0f ? 0.0f : (b > 0.9999f ? 0.9999f : b);
        alphas[k] = 1.0f - bc;
        logsum = logsum + logf(alphas[k]);
        alpha_bars[k] = expf(logsum);
    }
    free(ab);
}

/* model/diffusion.py:52-84 get_Q_bar_E / get_Q_bar_X :  Q = a*I + (1-a)*m  with rows of m = marginal */
void gmo_transition(float a, float m0, float m1, float Q[4]) {
    float na = 1.0f - a;
    float m[2] = {m0, m1};
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 2; ++c) {
            float eye = (r == c) ? 1.0f : 0.0f;
            float t0 = a * eye;
            float t1 = na * m[c];
            Q[r * 2 + c] = t0 + t1;
        }
}

/* model/diffusion.py:351-406 BaseModel.posterior for a one-hot Z_t = e:
 *   out[i][k] = (Q_t[k][e] * Qbar_s[i][k]) / Qbar_t[i][e]   (denominator 0 -> 1, :397)
 *   prob[k]   = prior[0]*out[0][k] + prior[1]*out[1][k]
 * The table holds out[i][k] for both e:  tab[(i*2+k)*2+e]. */
void gmo_posterior_table(const float Qt[4], const float Qbs[4], const float Qbt[4], float tab[8]) {
    for (int i = 0; i < 2; ++i)
        for (int k = 0; k < 2; ++k)
            for (int e = 0; e < 2; ++e) {
                float num = Qt[k * 2 + e] * Qbs[i * 2 + k];
                float den = Qbt[i * 2 + e];
                if (den == 0.0f) den = 1.0f;
                tab[(i * 2 + k) * 2 + e] = num / den;
            }
}

void gmo_posterior_tables_X(int64_t F, const float* X_marginal, float alpha_t, float abar_s, float abar_t,
                            float* tab) {
    for (int64_t f = 0; f < F; ++f) {
        float Qt[4], Qbs[4], Qbt[4];
        gmo_transition(alpha_t, X_marginal[2 * f], X_marginal[2 * f + 1], Qt);
        gmo_transition(abar_s, X_marginal[2 * f], X_marginal[2 * f + 1], Qbs);
        gmo_transition(abar_t, X_marginal[2 * f], X_marginal[2 * f + 1], Qbt);
        gmo_posterior_table(Qt, Qbs, Qbt, tab + 8 * f);
    }
}

void gmo_posterior_apply(const float tab[8], int z_t, float prior0, float prior1, float* p0, float* p1) {
    float a0 = prior0 * tab[(0 * 2 + 0) * 2 + z_t];
    float b0 = prior1 * tab[(1 * 2 + 0) * 2 + z_t];
    float a1 = prior0 * tab[(0 * 2 + 1) * 2 + z_t];
    float b1 = prior1 * tab[(1 * 2 + 1) * 2 + z_t];
    *p0 = a0 + b0;
    *p1 = a1 + b1;
}

static inline int draw2(float u, float p0, float p1) {
    float s = p0 + p1;
    float us = u * s;
    return us < p1 ? 1 : 0;
}

/* exp for x <= 0, shared bit for bit with the HIP library (Cephes-style expf; ~1 ulp). */
static inline float gm_expf_nonpos(float x) {
    if (x < -80.0f) x = -80.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r) + 1.0f;
    return ldexpf(y, (int)n);
}

/* softmax over 2 classes (model/diffusion.py:457, :879) */
void gmo_softmax2(float l0, float l1, float* p0, float* p1) {
    float m = l0 > l1 ? l0 : l1;
    float e0 = gm_expf_nonpos(l0 - m);
    float e1 = gm_expf_nonpos(l1 - m);
    float s = e0 + e1;
    *p0 = e0 / s;
    *p1 = e1 / s;
}

/* softmax over the two edge classes from the logit DIFFERENCE d = l1 - l0 (model/diffusion.py:457 for 2 classes):
 * softmax([l0, l1]) = softmax([0, d]); the larger class has exp(0) = 1, so one exp: e = exp(-|d|), r = 1 / (1 + e),
 * big = r, small = e * r.  The pair kernel computes d directly (one fmaf chain on W2[1] - W2[0]), not two logits. */
void gmo_softmax2_diff(float d, float* p0, float* p1) {
    float e = gm_expf_nonpos(d > 0.0f ? -d : d);
    float s = 1.0f + e;
    float r = 1.0f / s;
    float small = e * r;
    if (d > 0.0f) { *p0 = small; *p1 = r; }
    else { *p0 = r; *p1 = small; }
}

/* ------------------------------------------------------------------ priors */
/* model/diffusion.py:831-833  Y_0 ~ Cat(Y_marginal) */
void gmo_prior_Y(int64_t n, int C, const float* Y_marginal, const float* u_or_null, uint64_t seed, int32_t* y) {
    float sum = 0.0f;
    for (int c = 0; c < C; ++c) sum = sum + Y_marginal[c];
    for (int64_t v = 0; v < n; ++v) {
        float u = u_or_null ? u_or_null[v] : u_Y(seed, v);
        float us = u * sum;
        float cum = 0.0f;
        int pick = C - 1;
        for (int c = 0; c < C; ++c) {
            cum = cum + Y_marginal[c];
            if (us < cum) { pick = c; break; }
        }
        y[v] = pick;
    }
}

/* model/diffusion.py:842-844 + 214-237  X_T[v][f] ~ Bernoulli(X_marginal[f]) ; draw index f*n+v */
void gmo_prior_X(int64_t n, int64_t F, const float* X_marginal, const float* u_or_null, uint64_t seed,
                 uint8_t* x) {
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < n; ++v)
        for (int64_t f = 0; f < F; ++f) {
            int64_t idx = f * n + v;
            float u = u_or_null ? u_or_null[idx] : u_idx64(seed, 0u, GMO_STREAM_X, (uint64_t)idx);
            x[v * F + f] = (uint8_t)draw2(u, X_marginal[2 * f], X_marginal[2 * f + 1]);
        }
}

/* model/diffusion.py:836-839 + 188-212  N^2 draws, upper triangle INCLUDING the diagonal mirrored down */
void gmo_prior_E(int64_t n, int64_t wpr, float p0, float p1, const float* u_or_null, uint64_t seed,
                 uint32_t* bits) {
    memset(bits, 0, sizeof(uint32_t) * (size_t)(n * wpr));
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = i; j < n; ++j) {
            int64_t idx = i * n + j;
            float u = u_or_null ? u_or_null[idx] : u_idx64(seed, 0u, GMO_STREAM_E_PRIOR, (uint64_t)idx);
            if (draw2(u, p0, p1)) {
                bits[i * wpr + (j >> 5)] |= 1u << (j & 31);
                bits[j * wpr + (i >> 5)] |= 1u << (i & 31);
            }
        }
}

/* ------------------------------------------------------------------ forward noising q(G^t | G) */
/* model/diffusion.py:591-594 (+ sample_E :188-212): prob_E = E_one_hot @ Q_bar_t, one draw per (i, j >= i), the
 * diagonal included, mirrored to the lower triangle.  Q: Q_bar_t[r*2+c].  Philox stream 5, idx = i*n+j. */
void gmo_noise_E(int64_t n, int64_t wpr, const uint32_t* bits0, const float Q[4], const float* u_or_null,
                 uint64_t seed, uint32_t step, uint32_t* bits_t) {
    memset(bits_t, 0, sizeof(uint32_t) * (size_t)(n * wpr));
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = i; j < n; ++j) {
            int64_t idx = i * n + j;
            float u = u_or_null ? u_or_null[idx] : u_idx64(seed, step, GMO_STREAM_E_NOISE, (uint64_t)idx);
            int e0 = (bits0[i * wpr + (j >> 5)] >> (j & 31)) & 1u;
            if (draw2(u, Q[2 * e0], Q[2 * e0 + 1])) {
                bits_t[i * wpr + (j >> 5)] |= 1u << (j & 31);
                bits_t[j * wpr + (i >> 5)] |= 1u << (i & 31);
            }
        }
}
/* model/diffusion.py:596-600 (+ sample_X :214-237): prob_X = bmm(X_one_hot_3d, Q_bar_t_X).  QX: F x 4.
 * Philox stream 6, idx = f*n+v. */
void gmo_noise_X(int64_t n, int64_t F, const uint8_t* x0, const float* QX, const float* u_or_null, uint64_t seed,
                 uint32_t step, uint8_t* x_t) {
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < n; ++v)
        for (int64_t f = 0; f < F; ++f) {
            int64_t idx = f * n + v;
            float u = u_or_null ? u_or_null[idx] : u_idx64(seed, step, GMO_STREAM_X_NOISE, (uint64_t)idx);
            int r = x0[v * F + f] ? 1 : 0;
            x_t[v * F + f] = (uint8_t)draw2(u, QX[4 * f + 2 * r], QX[4 * f + 2 * r + 1]);
        }
}

/* ------------------------------------------------------------------ adjacency */
void gmo_bitmap_from_dense_i64(int64_t n, int64_t wpr, const int64_t* dense, uint32_t* bits) {
    memset(bits, 0, sizeof(uint32_t) * (size_t)(n * wpr));
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j)
            if (dense[i * n + j] != 0) bits[i * wpr + (j >> 5)] |= 1u << (j & 31);
}
void gmo_dense_i64_from_bitmap(int64_t n, int64_t wpr, const uint32_t* bits, int64_t* dense) {
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j) dense[i * n + j] = (bits[i * wpr + (j >> 5)] >> (j & 31)) & 1u;
}
/* model/diffusion.py:422-424  E[dst,src] = E[src,dst] = draw : mirror the strict upper triangle */
void gmo_symmetrize_upper(int64_t n, int64_t wpr, const uint32_t* upper, uint32_t* sym) {
    memset(sym, 0, sizeof(uint32_t) * (size_t)(n * wpr));
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = i + 1; j < n; ++j)
            if ((upper[i * wpr + (j >> 5)] >> (j & 31)) & 1u) {
                sym[i * wpr + (j >> 5)] |= 1u << (j & 31);
                sym[j * wpr + (i >> 5)] |= 1u << (i & 31);
            }
}
/* model/diffusion.py:252-256 get_adj : E_t.nonzero() is row-major sorted -> CSR with ascending columns;
 * values are implicit: 1/deg[row]. */
int64_t gmo_csr_from_bitmap(int64_t n, int64_t wpr, const uint32_t* bits, int32_t* rowptr, int32_t* colidx) {
    int64_t nnz = 0;
    for (int64_t i = 0; i < n; ++i) {
        rowptr[i] = (int32_t)nnz;
        for (int64_t w = 0; w < wpr; ++w) {
            uint32_t word = bits[i * wpr + w];
            while (word) {
                int b = __builtin_ctz(word);
                word &= word - 1;
                if (colidx) colidx[nnz] = (int32_t)(w * 32 + b);
                ++nnz;
            }
        }
    }
    rowptr[n] = (int32_t)nnz;
    return nnz;
}
int64_t gmo_upper_edges_from_bitmap(int64_t n, int64_t wpr, const uint32_t* bits, int32_t* ei, int32_t* ej) {
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = i + 1; j < n; ++j)
            if ((bits[i * wpr + (j >> 5)] >> (j & 31)) & 1u) {
                if (ei) { ei[m] = (int32_t)i; ej[m] = (int32_t)j; }
                ++m;
            }
    return m;
}
void gmo_bitmap_from_upper_edges(int64_t n, int64_t wpr, int64_t m, const int32_t* ei, const int32_t* ej,
                                 uint32_t* bits) {
    memset(bits, 0, sizeof(uint32_t) * (size_t)(n * wpr));
    for (int64_t e = 0; e < m; ++e) {
        int64_t i = ei[e], j = ej[e];
        bits[i * wpr + (j >> 5)] |= 1u << (j & 31);
        bits[j * wpr + (i >> 5)] |= 1u << (i & 31);
    }
}

/* ------------------------------------------------------------------ building blocks */
/* model/gnn.py:60-61  A @ H with A = D^-1 A (diffusion.py:255-256): out[i] = sum_j (1/deg_i) * H[j].
 * Order: a row's neighbours (ascending j) are cut into segments of GMO_SPMM_SEG = 64; a segment is one fmaf chain
 * acc = fmaf(H[j], 1/deg_i, acc) from 0 over its neighbours in order; the row is the segment sums added in ascending segment
 * order ((s0 + s1) + s2 ...).  Rows of up to 64 neighbours are therefore a single chain; the cut is what lets the HIP kernel
 * spread a hub row (amazon_computer's largest degree is ~3000) over many lanes and still produce these bits. */
#define GMO_SPMM_SEG 64
void gmo_spmm_mean(int64_t n, const int32_t* rowptr, const int32_t* colidx, const float* H, int64_t ldh,
                   int64_t d, float* out, int64_t ldo) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < n; ++i) {
        float* o = out + i * ldo;
        for (int64_t c = 0; c < d; ++c) o[c] = 0.0f;
        int32_t b = rowptr[i], e = rowptr[i + 1];
        if (e == b) continue;
        float inv = 1.0f / (float)(e - b);
        if (e - b <= GMO_SPMM_SEG) {
            for (int32_t p = b; p < e; ++p) {
                const float* hrow = H + (int64_t)colidx[p] * ldh;
#pragma omp simd
                for (int64_t c = 0; c < d; ++c) o[c] = fmaf(hrow[c], inv, o[c]);
            }
            continue;
        }
        float* seg = (float*)malloc(sizeof(float) * (size_t)d);
        for (int32_t s0 = b; s0 < e; s0 += GMO_SPMM_SEG) {
            int32_t s1 = s0 + GMO_SPMM_SEG < e ? s0 + GMO_SPMM_SEG : e;
            for (int64_t c = 0; c < d; ++c) seg[c] = 0.0f;
            for (int32_t p = s0; p < s1; ++p) {
                const float* hrow = H + (int64_t)colidx[p] * ldh;
#pragma omp simd
                for (int64_t c = 0; c < d; ++c) seg[c] = fmaf(hrow[c], inv, seg[c]);
            }
            if (s0 == b) for (int64_t c = 0; c < d; ++c) o[c] = seg[c];
            else for (int64_t c = 0; c < d; ++c) o[c] = o[c] + seg[c];
        }
        free(seg);
    }
}

/* nn.Linear (+ReLU): C[m][n] = (sum_k fmaf chain, k ascending, from 0) + bias[n]. */
void gmo_linear(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W, int64_t ldw,
                const float* bias, float* C, int64_t ldc, int relu) {
    float* Wt = (float*)malloc(sizeof(float) * (size_t)(K * N));
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < K; ++k)
        for (int64_t nn = 0; nn < N; ++nn) Wt[k * N + nn] = W[nn * ldw + k];
    enum { RB = 8, NB = 256 };
    int64_t nrb = (M + RB - 1) / RB;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t rb = 0; rb < nrb; ++rb) {
        int64_t m0 = rb * RB;
        int64_t mr = M - m0 < RB ? M - m0 : RB;
        float acc[RB][NB];
        for (int64_t n0 = 0; n0 < N; n0 += NB) {
            int64_t nb = N - n0 < NB ? N - n0 : NB;
            for (int64_t r = 0; r < mr; ++r)
                for (int64_t c = 0; c < nb; ++c) acc[r][c] = 0.0f;
            for (int64_t k = 0; k < K; ++k) {
                const float* wrow = Wt + k * N + n0;
                for (int64_t r = 0; r < mr; ++r) {
                    float a = A[(m0 + r) * lda + k];
                    float* ar = acc[r];
#pragma omp simd
                    for (int64_t c = 0; c < nb; ++c) ar[c] = fmaf(a, wrow[c], ar[c]);
                }
            }
            for (int64_t r = 0; r < mr; ++r)
                for (int64_t c = 0; c < nb; ++c) {
                    float v = acc[r][c];
                    if (bias) v = v + bias[n0 + c];
                    if (relu && !(v > 0.0f)) v = 0.0f;
                    C[(m0 + r) * ldc + n0 + c] = v;
                }
        }
    }
    free(Wt);
}

/* 64-lane strided partial sums + xor butterfly: the order a 64-wide wavefront reduction produces. */
static float wave_sum64(const float* x, int64_t N) {
    float part[64];
    for (int l = 0; l < 64; ++l) {
        float s = 0.0f;
        for (int64_t c = l; c < N; c += 64) s = s + x[c];
        part[l] = s;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        float nxt[64];
        for (int l = 0; l < 64; ++l) nxt[l] = part[l] + part[l ^ off];
        memcpy(part, nxt, sizeof(part));
    }
    return part[0];
}

/* nn.LayerNorm(N), eps=1e-5, biased variance (model/gnn.py:30,36) */
void gmo_layernorm(int64_t M, int64_t N, float* X, int64_t ldx, const float* gamma, const float* beta, float eps) {
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        float* x = X + m * ldx;
        float tmp[4096];
        float* d = N <= 4096 ? tmp : (float*)malloc(sizeof(float) * (size_t)N);
        float mean = wave_sum64(x, N) / (float)N;
        for (int64_t c = 0; c < N; ++c) { float t = x[c] - mean; d[c] = t * t; }
        float var = wave_sum64(d, N) / (float)N;
        float rstd = 1.0f / sqrtf(var + eps);
        for (int64_t c = 0; c < N; ++c) {
            float t = (x[c] - mean) * rstd;
            float g = t * gamma[c];
            x[c] = g + beta[c];
        }
        if (d != tmp) free(d);
    }
}

/* ------------------------------------------------------------------ towers */
/* model/gnn.py:146-176 GNNTower.forward and :440-463 MLPTower.forward (dropout is identity in eval). */
void gmo_tower_forward(const gmo_tower_dims* d, const float* const* P, int64_t n, float t_float,
                       const float* X_one_hot, const int32_t* Y, const int32_t* rowptr, const int32_t* colidx,
                       float* out, float* hcat_dbg) {
    const int ht = d->hidden_t, hX = d->hidden_X, hY = d->hidden_Y, L = d->num_layers;
    const int Hc = gmo_tower_hidden_cat(d);
    const int offX = 0, offY = (L + 1) * hX, offT = (L + 1) * (hX + hY);
    float* hcat = hcat_dbg ? hcat_dbg : (float*)malloc(sizeof(float) * (size_t)(n * Hc));

    /* mlp_in_t (:116-120,153) */
    float tin[1] = {t_float};
    float* ht0 = (float*)malloc(sizeof(float) * (size_t)ht);
    float* ht1 = (float*)malloc(sizeof(float) * (size_t)ht);
    gmo_linear(1, ht, 1, tin, 1, P[0], 1, P[1], ht0, ht, 1);
    gmo_linear(1, ht, ht, ht0, ht, P[2], ht, P[3], ht1, ht, 1);
    for (int64_t v = 0; v < n; ++v) memcpy(hcat + v * Hc + offT, ht1, sizeof(float) * (size_t)ht);

    /* mlp_in_X (:121-126,154) */
    float* tmpX = (float*)malloc(sizeof(float) * (size_t)(n * hX));
    gmo_onehot_linear(n, d->in_X / 2, hX, X_one_hot, P[4], P[5], tmpX, hX, 1);
    gmo_linear(n, hX, hX, tmpX, hX, P[6], hX, P[7], hcat + offX, Hc, 1);
    free(tmpX);

    /* emb_Y (:127,155) */
    for (int64_t v = 0; v < n; ++v)
        memcpy(hcat + v * Hc + offY, P[8] + (int64_t)Y[v] * hY, sizeof(float) * (size_t)hY);

    const int ka = hX + hY + ht;
    float* agg = (float*)malloc(sizeof(float) * (size_t)(n * ka));
    for (int l = 0; l < L; ++l) {
        const float* const* Q = P + 9 + 8 * l;
        const float* hXl = hcat + offX + l * hX;
        const float* hYl = hcat + offY + l * hY;
        if (!d->is_mlp) {
            /* GNNLayer.forward (:60-68): A@[hX|hY], A@hY (= its last hY columns), concat h_t */
            gmo_spmm_mean(n, rowptr, colidx, hXl, Hc, hX, agg, ka);
            gmo_spmm_mean(n, rowptr, colidx, hYl, Hc, hY, agg + hX, ka);
        } else {
            /* MLPLayer.forward (:387-392): no aggregation */
            for (int64_t v = 0; v < n; ++v) {
                memcpy(agg + v * ka, hXl + v * Hc, sizeof(float) * (size_t)hX);
                memcpy(agg + v * ka + hX, hYl + v * Hc, sizeof(float) * (size_t)hY);
            }
        }
        for (int64_t v = 0; v < n; ++v) memcpy(agg + v * ka + hX + hY, ht1, sizeof(float) * (size_t)ht);
        float* hXn = hcat + offX + (l + 1) * hX;
        float* hYn = hcat + offY + (l + 1) * hY;
        gmo_linear(n, hX, ka, agg, ka, Q[0], ka, Q[1], hXn, Hc, 1);
        gmo_layernorm(n, hX, hXn, Hc, Q[2], Q[3], 1e-5f);
        gmo_linear(n, hY, hY, agg + hX, ka, Q[4], hY, Q[5], hYn, Hc, 1);
        gmo_layernorm(n, hY, hYn, Hc, Q[6], Q[7], 1e-5f);
    }
    free(agg);

    /* mlp_out (:138-142,165-176) */
    const float* const* O = P + 9 + 8 * L;
    float* mid = (float*)malloc(sizeof(float) * (size_t)(n * Hc));
    gmo_linear(n, Hc, Hc, hcat, Hc, O[0], Hc, O[1], mid, Hc, 1);
    gmo_linear(n, d->out_size, Hc, mid, Hc, O[2], Hc, O[3], out, d->out_size, 0);
    free(mid);
    free(ht0);
    free(ht1);
    if (!hcat_dbg) free(hcat);
}

/* ------------------------------------------------------------------ first tower layer on one-hot attributes */
/* relu(X_one_hot W^T + b) of mlp_in_X.0 (model/gnn.py:121-126,154) for a matrix that is exactly one-hot with two classes per
 * attribute (X_t_one_hot[v][2f + x_vf] = 1, model/diffusion.py:232-235):
 *     X W^T + b = B + sum_{f: x_vf = 1} D[:, f],   D[o][f] = W[o][2f+1] - W[o][2f],   B[o] = (sum_f W[o][2f]) + b[o].
 * Orders: B[o] = (plain adds over ascending f, from 0) + b[o];  out = (plain adds of D[o][f] over ascending set f, from 0) + B[o].
 * A chain fmaf(x_f, D[o][f], acc) over ALL f with x_f in {0,1} gives the same bits (a zero product leaves acc unchanged), which
 * is how the dense form of the HIP library (an MFMA GEMM with K = F instead of 2F) and the sparse form (a walk over the set
 * attributes) agree with each other and with this function. */
void gmo_onehot_linear(int64_t n, int64_t F, int64_t H, const float* X_one_hot, const float* W, const float* b,
                       float* out, int64_t ldo, int relu) {
    float* Dt = (float*)malloc(sizeof(float) * (size_t)(F * H));
    float* B = (float*)malloc(sizeof(float) * (size_t)H);
    for (int64_t o = 0; o < H; ++o) {
        float acc = 0.0f;
        for (int64_t f = 0; f < F; ++f) {
            acc = acc + W[o * 2 * F + 2 * f];
            Dt[f * H + o] = W[o * 2 * F + 2 * f + 1] - W[o * 2 * F + 2 * f];
        }
        B[o] = acc + (b ? b[o] : 0.0f);
    }
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t v = 0; v < n; ++v) {
        float* o_ = out + v * ldo;
        for (int64_t o = 0; o < H; ++o) o_[o] = 0.0f;
        for (int64_t f = 0; f < F; ++f) {
            if (X_one_hot[v * 2 * F + 2 * f + 1] > X_one_hot[v * 2 * F + 2 * f]) {
                const float* d = Dt + f * H;
#pragma omp simd
                for (int64_t o = 0; o < H; ++o) o_[o] = o_[o] + d[o];
            }
        }
        for (int64_t o = 0; o < H; ++o) {
            float r = o_[o] + B[o];
            if (relu && !(r > 0.0f)) r = 0.0f;
            o_[o] = r;
        }
    }
    free(Dt);
    free(B);
}

/* ------------------------------------------------------------------ structure denoiser */
/* model/gnn.py:246-248  logit = W2 relu(W1 (h_i * h_j) + b1) + b2.
 * Orders: hidden[o] = fmaf chain over k ascending starting from b1[o];
 *         logit[c]  = (S0 + S1) + b2[c],  S_h = C[h][0] + C[h][1],  C[h][p] = fmaf chain over ascending o with
 *         ((o>>2)&1)==h and (o&1)==p  (h: the split a 32x32 MFMA accumulator tile gives the two half-waves; p: even / odd
 *         hidden units sit in adjacent accumulator registers, so one packed FMA advances both chains). */
/* ReLU of the pair MLP's hidden layer, in the form the GPU evaluates it: one packed multiply with the hardware clamp to
 * [0,1] -- relu(x) * w == clamp01(x * 2^-63) * (w * 2^63).  Both scalings are exact, so this equals the plain ReLU bit for
 * bit for x <= 0 and for 2^-63 <= x <= 2^63; a smaller positive x (|x| < 1.1e-19) comes back rounded to a multiple of
 * 2^-86 (its product with a weight is far below the resolution of any logit it is added to), a larger x saturates at
 * 2^63 (9.2e18: no model has such activations), NaN gives 0. */
static inline float relu63(float x) {
    float c = x * 0x1p-63f;
    return !(c > 0.0f) ? 0.0f : (c > 1.0f ? 1.0f : c);
}

void gmo_pair_logits(int64_t hE, const float* hi, const float* hj, const float* W1, const float* b1,
                     const float* W2, const float* b2, float logit[2]) {
    float z[1024], hid[1024];
    for (int64_t k = 0; k < hE; ++k) z[k] = hi[k] * hj[k];
    for (int64_t o = 0; o < hE; ++o) {
        float acc = b1[o];
        const float* w = W1 + o * hE;
        for (int64_t k = 0; k < hE; ++k) acc = fmaf(z[k], w[k], acc);
        hid[o] = relu63(acc);
    }
    for (int c = 0; c < 2; ++c) {
        float s[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
        for (int64_t o = 0; o < hE; ++o) {
            int hsel = (int)((o >> 2) & 1), par = (int)(o & 1);
            s[hsel][par] = fmaf(hid[o], W2[c * hE + o] * 0x1p63f, s[hsel][par]);
        }
        logit[c] = ((s[0][0] + s[0][1]) + (s[1][0] + s[1][1])) + b2[c];
    }
}

void gmo_pair_score_sample(int64_t n, int64_t hE, const float* h, const float* W1, const float* b1,
                           const float* W2, const float* b2, const uint32_t* bits_t, int64_t wpr,
                           const float tab[8], int64_t row_begin, int64_t row_end, const float* u_or_null,
                           uint64_t seed, uint32_t step, uint32_t* out_upper, float* logits_dbg,
                           float* post_dbg) {
    /* transposed W1 so the o-loop vectorises while each o keeps its k-ascending chain */
    float* W1t = (float*)malloc(sizeof(float) * (size_t)(hE * hE));
    for (int64_t o = 0; o < hE; ++o)
        for (int64_t k = 0; k < hE; ++k) W1t[k * hE + o] = W1[o * hE + k];
    float dw[1024];
    for (int64_t o = 0; o < hE; ++o) dw[o] = (W2[hE + o] - W2[o]) * 0x1p63f;     /* scaled: see relu63 */
    const float db = b2[1] - b2[0];
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t i = row_begin; i < row_end; ++i) {
        uint32_t* orow = out_upper + i * wpr;
        for (int64_t w = 0; w < wpr; ++w) orow[w] = 0u;
        const float* hi = h + i * hE;
        enum { JB = 4, OB = 16 };
        float z[JB][1024], hid[JB][1024];
        for (int64_t jb0 = i + 1; jb0 < n; jb0 += JB) {
            const int nj = (int)(n - jb0 < JB ? n - jb0 : JB);
            for (int b = 0; b < JB; ++b) {
                const float* hj = h + (b < nj ? jb0 + b : jb0) * hE;
                for (int64_t k = 0; k < hE; ++k) z[b][k] = hi[k] * hj[k];
            }
            /* register-blocked: 4 pairs x 16 hidden units stay in registers across the k loop; every (pair, o)
             * still sees hid = b1[o], then fmaf over ascending k */
            for (int64_t o0 = 0; o0 < hE; o0 += OB) {
                if (o0 + OB <= hE) {
                    float acc[JB][OB];
                    for (int b = 0; b < JB; ++b)
                        for (int oo = 0; oo < OB; ++oo) acc[b][oo] = b1[o0 + oo];
                    for (int64_t k = 0; k < hE; ++k) {
                        const float* wr = W1t + k * hE + o0;
                        for (int b = 0; b < JB; ++b) {
                            const float zk = z[b][k];
#pragma omp simd
                            for (int oo = 0; oo < OB; ++oo) acc[b][oo] = fmaf(zk, wr[oo], acc[b][oo]);
                        }
                    }
                    for (int b = 0; b < JB; ++b)
                        for (int oo = 0; oo < OB; ++oo) hid[b][o0 + oo] = acc[b][oo];
                } else {
                    for (int b = 0; b < JB; ++b)
                        for (int64_t o = o0; o < hE; ++o) {
                            float acc = b1[o];
                            for (int64_t k = 0; k < hE; ++k) acc = fmaf(z[b][k], W1t[k * hE + o], acc);
                            hid[b][o] = acc;
                        }
                }
            }
            for (int b = 0; b < nj; ++b) {
                const int64_t j = jb0 + b;
                /* logit difference d = l1 - l0 on dw = W2[1] - W2[0]: (S0 + S1) + (b2[1] - b2[0]), S_h = C[h][0] + C[h][1],
                 * C[h][p] = fmaf chain over ascending o with ((o>>2)&1)==h, (o&1)==p (see gmo_pair_logits) */
                float s[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
                for (int64_t o = 0; o < hE; ++o) {
                    float r = relu63(hid[b][o]);
                    int hsel = (int)((o >> 2) & 1), par = (int)(o & 1);
                    s[hsel][par] = fmaf(r, dw[o], s[hsel][par]);
                }
                const float d = ((s[0][0] + s[0][1]) + (s[1][0] + s[1][1])) + db;
                float q0, q1, p0, p1;
                gmo_softmax2_diff(d, &q0, &q1);                                  /* diffusion.py:457 */
                int e_t = (int)((bits_t[i * wpr + (j >> 5)] >> (j & 31)) & 1u);  /* diffusion.py:460-462 */
                gmo_posterior_apply(tab, e_t, q0, q1, &p0, &p1);                 /* diffusion.py:463-464 */
                int64_t r = pair_rank(n, i, j);
                float u = u_or_null ? u_or_null[r] : u_E(seed, step, i, j);
                if (draw2(u, p0, p1)) orow[j >> 5] |= 1u << (j & 31);            /* diffusion.py:421 */
                if (logits_dbg) {      /* both logits (model/gnn.py:248), for comparison with the reference only */
                    for (int c = 0; c < 2; ++c) {
                        float t[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
                        for (int64_t o = 0; o < hE; ++o) {
                            float rl = relu63(hid[b][o]);
                            int hsel = (int)((o >> 2) & 1), par = (int)(o & 1);
                            t[hsel][par] = fmaf(rl, W2[c * hE + o] * 0x1p63f, t[hsel][par]);
                        }
                        logits_dbg[2 * r + c] = ((t[0][0] + t[0][1]) + (t[1][0] + t[1][1])) + b2[c];
                    }
                }
                if (post_dbg) { post_dbg[2 * r] = p0; post_dbg[2 * r + 1] = p1; }
            }
        }
    }
    free(W1t);
}

/* ------------------------------------------------------------------ attributes */
/* model/diffusion.py:879-895: softmax over the class dim of (n,F,2) logits, per-attribute posterior,
 * draw with index f*n+v (reshape of (F,n,2), :228-230). */
void gmo_attr_posterior_sample(int64_t n, int64_t F, const float* logits, const uint8_t* x_t, const float* tab,
                               const float* u_or_null, uint64_t seed, uint32_t step, uint8_t* x_s,
                               float* post_dbg) {
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < n; ++v)
        for (int64_t f = 0; f < F; ++f) {
            const float* l = logits + (v * F + f) * 2;
            float q0, q1, p0, p1;
            gmo_softmax2(l[0], l[1], &q0, &q1);
            gmo_posterior_apply(tab + 8 * f, (int)x_t[v * F + f], q0, q1, &p0, &p1);
            int64_t idx = f * n + v;
            float u = u_or_null ? u_or_null[idx] : u_idx64(seed, step, GMO_STREAM_X, (uint64_t)idx);
            x_s[v * F + f] = (uint8_t)draw2(u, p0, p1);
            if (post_dbg) { post_dbg[(v * F + f) * 2] = p0; post_dbg[(v * F + f) * 2 + 1] = p1; }
        }
}

/* The same step from the logit DIFFERENCE d[v][f] = l1 - l0: softmax([l0, l1]) = softmax([0, d]) (gmo_softmax2_diff, as
 * the structure denoiser does for the two edge classes).  The product's sampler runs the node tower with the half-width
 * head of gmo_head_diff and feeds this function; gmo_sample_sync / gmo_sample_async below do the same. */
void gmo_attr_posterior_sample_diff(int64_t n, int64_t F, const float* diff, const uint8_t* x_t, const float* tab,
                                    const float* u_or_null, uint64_t seed, uint32_t step, uint8_t* x_s,
                                    float* post_dbg) {
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < n; ++v)
        for (int64_t f = 0; f < F; ++f) {
            float q0, q1, p0, p1;
            gmo_softmax2_diff(diff[v * F + f], &q0, &q1);
            gmo_posterior_apply(tab + 8 * f, (int)x_t[v * F + f], q0, q1, &p0, &p1);
            int64_t idx = f * n + v;
            float u = u_or_null ? u_or_null[idx] : u_idx64(seed, step, GMO_STREAM_X, (uint64_t)idx);
            x_s[v * F + f] = (uint8_t)draw2(u, p0, p1);
            if (post_dbg) { post_dbg[(v * F + f) * 2] = p0; post_dbg[(v * F + f) * 2 + 1] = p1; }
        }
}

/* mlp_out.2 (model/gnn.py:139-143; W (2F,Hc) row-major, b (2F)) folded to one output per attribute:
 * Wd[f] = W[2f+1] - W[2f], bd[f] = b[2f+1] - b[2f], so that x.Wd[f] + bd[f] = l1 - l0 up to f32 rounding. */
void gmo_head_diff(int64_t F, int64_t Hc, const float* W, const float* b, float* Wd, float* bd) {
    for (int64_t f = 0; f < F; ++f) {
        for (int64_t k = 0; k < Hc; ++k) Wd[f * Hc + k] = W[(2 * f + 1) * Hc + k] - W[2 * f * Hc + k];
        bd[f] = b[2 * f + 1] - b[2 * f];
    }
}

/* a node tower with its head folded by gmo_head_diff: dims/params copies whose last Linear is (F,Hc) */
typedef struct { gmo_tower_dims d; const float** p; float* Wd; float* bd; } diff_tower;
static diff_tower diff_tower_make(const gmo_tower_dims* dX, const float* const* pX) {
    diff_tower t;
    int np = 9 + 8 * dX->num_layers + 4;
    int64_t F = dX->out_size / 2, Hc = gmo_tower_hidden_cat(dX);
    t.d = *dX;
    t.d.out_size = (int)F;
    t.p = (const float**)malloc(sizeof(float*) * (size_t)np);
    memcpy(t.p, pX, sizeof(float*) * (size_t)np);
    t.Wd = (float*)malloc(sizeof(float) * (size_t)(F * Hc));
    t.bd = (float*)malloc(sizeof(float) * (size_t)F);
    gmo_head_diff(F, Hc, pX[np - 2], pX[np - 1], t.Wd, t.bd);
    t.p[np - 2] = t.Wd;
    t.p[np - 1] = t.bd;
    return t;
}
static void diff_tower_free(diff_tower* t) { free(t->p); free(t->Wd); free(t->bd); }

void gmo_onehot_from_x(int64_t n, int64_t F, const uint8_t* x, float* oh) {
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < n; ++v)
        for (int64_t f = 0; f < F; ++f) {
            int b = x[v * F + f] != 0;
            oh[(v * F + f) * 2] = b ? 0.0f : 1.0f;
            oh[(v * F + f) * 2 + 1] = b ? 1.0f : 0.0f;
        }
}

/* ------------------------------------------------------------------ trajectories */
static void e_tables(const float* E_marginal, const float* alphas, const float* abars, int t, float tab[8]) {
    float Qt[4], Qbs[4], Qbt[4];
    gmo_transition(alphas[t], E_marginal[0], E_marginal[1], Qt);       /* diffusion.py:856 */
    gmo_transition(abars[t - 1], E_marginal[0], E_marginal[1], Qbs);   /* :857 */
    gmo_transition(abars[t], E_marginal[0], E_marginal[1], Qbt);       /* :858 */
    gmo_posterior_table(Qt, Qbs, Qbt, tab);
}

/* one structure step: get_E_t (diffusion.py:428-472) */
static void e_step(int64_t n, int64_t F, const gmo_tower_dims* dE, const float* const* pE,
                   const float* const* pPair, float t_float, const float* X_oh, const int32_t* y,
                   const uint32_t* bits_t, int64_t wpr, const float tab[8], uint64_t seed, uint32_t step,
                   int as_reference, int32_t* rowptr, int32_t** colidx_io, uint32_t* upper, uint32_t* bits_s) {
    (void)F;
    int64_t hE = dE->out_size;
    float* h = (float*)malloc(sizeof(float) * (size_t)(n * hE));
    int64_t nnz = gmo_csr_from_bitmap(n, wpr, bits_t, rowptr, NULL);
    free(*colidx_io);
    *colidx_io = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
    int32_t* colidx = *colidx_io;
    gmo_csr_from_bitmap(n, wpr, bits_t, rowptr, colidx);
    gmo_tower_forward(dE, pE, n, t_float, X_oh, y, rowptr, colidx, h, NULL);
    if (as_reference) {
        /* the reference re-runs the whole encoder for every 32768-pair batch (diffusion.py:444-455) */
        int64_t P = n * (n - 1) / 2, B = 32768;
        for (int64_t b = B; b < P; b += B) gmo_tower_forward(dE, pE, n, t_float, X_oh, y, rowptr, colidx, h, NULL);
    }
    gmo_pair_score_sample(n, hE, h, pPair[0], pPair[1], pPair[2], pPair[3], bits_t, wpr, tab, 0, n, NULL,
                          seed, step, upper, NULL, NULL);
    gmo_symmetrize_upper(n, wpr, upper, bits_s);
    free(h);
}

void gmo_sample_sync(int T, int64_t n, int64_t F, int C, const float* X_marginal, const float* Y_marginal,
                     const float* E_marginal, const gmo_tower_dims* dX, const float* const* pX,
                     const gmo_tower_dims* dE, const float* const* pE, const float* const* pPair,
                     uint64_t seed, int as_reference, uint8_t* x, int32_t* y, uint32_t* bits, int64_t wpr,
                     const float* sched) {
    float* betas = (float*)malloc(sizeof(float) * (size_t)(T + 1) * 3);
    float *alphas = betas + (T + 1), *abars = betas + 2 * (T + 1);
    gmo_noise_schedule(T, betas, alphas, abars);
    if (sched) { /* caller-supplied [alphas | alpha_bars]: the product's tables come from torch's log/exp */
        memcpy(alphas, sched, sizeof(float) * (size_t)(T + 1));
        memcpy(abars, sched + (T + 1), sizeof(float) * (size_t)(T + 1));
    }
    gmo_prior_Y(n, C, Y_marginal, NULL, seed, y);
    gmo_prior_E(n, wpr, E_marginal[0], E_marginal[1], NULL, seed, bits);
    gmo_prior_X(n, F, X_marginal, NULL, seed, x);

    diff_tower tX = diff_tower_make(dX, pX);   /* the sampler's node tower: logit differences (see gmo_head_diff) */
    float* X_oh = (float*)malloc(sizeof(float) * (size_t)(n * 2 * F));
    float* logits = (float*)malloc(sizeof(float) * (size_t)(n * F));
    float* tabX = (float*)malloc(sizeof(float) * (size_t)(8 * F));
    uint32_t* upper = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n * wpr));
    uint32_t* bits_s = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n * wpr));
    uint8_t* x_s = (uint8_t*)malloc((size_t)(n * F));
    int32_t* rowptr = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
    int32_t* colidx = NULL;

    for (int s = T - 1; s >= 0; --s) {                                  /* diffusion.py:847 */
        int t = s + 1;
        float t_float = (float)((double)t / (double)T);                 /* :860 torch.tensor([t / T]) */
        float tabE[8];
        e_tables(E_marginal, alphas, abars, t, tabE);
        gmo_onehot_from_x(n, F, x, X_oh);
        e_step(n, F, dE, pE, pPair, t_float, X_oh, y, bits, wpr, tabE, seed, (uint32_t)t, as_reference, rowptr,
               &colidx, upper, bits_s);                                  /* :862-872 */
        /* pred_X uses A_t from E_t, not E_s (:875-878); rowptr/colidx still describe E_t */
        gmo_tower_forward(&tX.d, tX.p, n, t_float, X_oh, y, rowptr, colidx, logits, NULL);
        gmo_posterior_tables_X(F, X_marginal, alphas[t], abars[t - 1], abars[t], tabX);   /* :889-891 */
        gmo_attr_posterior_sample_diff(n, F, logits, x, tabX, NULL, seed, (uint32_t)t, x_s, NULL);
        memcpy(x, x_s, (size_t)(n * F));
        memcpy(bits, bits_s, sizeof(uint32_t) * (size_t)(n * wpr));     /* :896 */
    }
    free(betas); free(X_oh); free(logits); free(tabX); free(upper); free(bits_s); free(x_s);
    free(rowptr); free(colidx);
    diff_tower_free(&tX);
}

void gmo_sample_async(int T_X, int T_E, int64_t n, int64_t F, int C, const float* X_marginal,
                      const float* Y_marginal, const float* E_marginal, const gmo_tower_dims* dX,
                      const float* const* pX, const gmo_tower_dims* dE, const float* const* pE,
                      const float* const* pPair, uint64_t seed, int as_reference, uint8_t* x, int32_t* y,
                      uint32_t* bits, int64_t wpr, const float* schedX, const float* schedE) {
    int Tm = T_X > T_E ? T_X : T_E;
    float* sx = (float*)malloc(sizeof(float) * (size_t)(Tm + 1) * 6);
    float *aX = sx + (Tm + 1), *abX = sx + 2 * (Tm + 1);
    float *bE = sx + 3 * (Tm + 1), *aE = sx + 4 * (Tm + 1), *abE = sx + 5 * (Tm + 1);
    gmo_noise_schedule(T_X, sx, aX, abX);                               /* diffusion.py:950 */
    gmo_noise_schedule(T_E, bE, aE, abE);                               /* :951 */
    if (schedX) {
        memcpy(aX, schedX, sizeof(float) * (size_t)(T_X + 1));
        memcpy(abX, schedX + (T_X + 1), sizeof(float) * (size_t)(T_X + 1));
    }
    if (schedE) {
        memcpy(aE, schedE, sizeof(float) * (size_t)(T_E + 1));
        memcpy(abE, schedE + (T_E + 1), sizeof(float) * (size_t)(T_E + 1));
    }
    gmo_prior_Y(n, C, Y_marginal, NULL, seed, y);                       /* :1319-1321 */
    gmo_prior_X(n, F, X_marginal, NULL, seed, x);                       /* :1325-1327 */

    diff_tower tX = diff_tower_make(dX, pX);
    float* X_oh = (float*)malloc(sizeof(float) * (size_t)(n * 2 * F));
    float* logits = (float*)malloc(sizeof(float) * (size_t)(n * F));
    float* tabX = (float*)malloc(sizeof(float) * (size_t)(8 * F));
    uint8_t* x_s = (uint8_t*)malloc((size_t)(n * F));
    for (int s = T_X - 1; s >= 0; --s) {                                /* :1330-1358 */
        int t = s + 1;
        float t_float = (float)((double)t / (double)T_X);
        gmo_onehot_from_x(n, F, x, X_oh);
        gmo_tower_forward(&tX.d, tX.p, n, t_float, X_oh, y, NULL, NULL, logits, NULL);
        gmo_posterior_tables_X(F, X_marginal, aX[t], abX[t - 1], abX[t], tabX);
        gmo_attr_posterior_sample_diff(n, F, logits, x, tabX, NULL, seed, (uint32_t)t, x_s, NULL);
        memcpy(x, x_s, (size_t)(n * F));
    }
    gmo_prior_E(n, wpr, E_marginal[0], E_marginal[1], NULL, seed, bits); /* :1362-1365 */
    gmo_onehot_from_x(n, F, x, X_oh);
    uint32_t* upper = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n * wpr));
    uint32_t* bits_s = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n * wpr));
    int32_t* rowptr = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
    int32_t* colidx = NULL;
    for (int s = T_E - 1; s >= 0; --s) {                                /* :1379-1401 */
        int t = s + 1;
        float t_float = (float)((double)t / (double)T_E);
        float tabE[8];
        e_tables(E_marginal, aE, abE, t, tabE);
        e_step(n, F, dE, pE, pPair, t_float, X_oh, y, bits, wpr, tabE, seed, (uint32_t)t, as_reference, rowptr,
               &colidx, upper, bits_s);
        memcpy(bits, bits_s, sizeof(uint32_t) * (size_t)(n * wpr));
    }
    free(sx); free(X_oh); free(logits); free(tabX); free(x_s); free(upper); free(bits_s);
    free(rowptr); free(colidx);
    diff_tower_free(&tX);
}
