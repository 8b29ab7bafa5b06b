/*
 * oracle/gm_oracle.h -- CPU restatement of GraphMaker's reverse-diffusion denoising step.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under graphmaker_amd/ may include, link, load or call this.
 * It is used by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg, always as the
 * checker or the reported CPU baseline, never as the product path.
 *
 * Every function cites the reference lines it restates (paths under /root/reference).
 * Parity pin: the reference has no tests or golden vectors of its own, so the
 * oracle is pinned by fixtures generated HERE by importing the reference's model/gnn.py and
 * model/diffusion.py (tools/gen_golden.py -> tests/golden/ npz files).  The semantics of dgl.sparse (absent
 * in this image; dgl==1.1.0+cu116 per README.md:25-27) are restated from its documented COO behaviour:
 * "parity unpinned at the DGL boundary" (get_adj / A @ H only).
 *
 * Conventions shared with the HIP library (include/graphmaker_hip.h):
 *   - adjacency state is a symmetric bitmap: row i, word w holds nodes j = 32w .. 32w+31, bit (j & 31);
 *     wpr = words per row (>= ceil(n/32)); bits beyond n are zero.
 *   - attributes are uint8 x[v*F + f] in {0,1}; one-hot float layout is oh[v*2F + 2f + c]
 *     (model/diffusion.py:232-235).
 *   - uniforms: u in [0,1) with 24 bits, u = (word >> 8) * 2^-24, words from Philox4x32-10,
 *     key = (seed lo, seed hi), counters per stream documented at gmo_fill_uniform_*.
 *   - 2-class draw:  x = ( u * (p0 + p1) < p1 )   in fp32, no fused multiply-add.
 *   - categorical draw (labels): x = first c with u * sum < cumsum_c, else C-1.
 */
#ifndef GM_ORACLE_H
#define GM_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { GMO_STREAM_Y = 1, GMO_STREAM_X = 2, GMO_STREAM_E_PRIOR = 3, GMO_STREAM_E = 4, GMO_STREAM_E_NOISE = 5,
       GMO_STREAM_X_NOISE = 6 };

typedef struct {
    int in_X;        /* num_attrs_X * num_classes_X = 2F      (model/gnn.py:112) */
    int num_classes_Y;
    int hidden_t, hidden_X, hidden_Y;
    int num_layers;  /* num_gnn_layers / num_mlp_layers */
    int out_size;    /* hidden_E (edge tower) or 2F (attribute towers) */
    int is_mlp;      /* 0: GNNTower (model/gnn.py:72-176), 1: MLPTower (model/gnn.py:396-463) */
} gmo_tower_dims;

/* Tower parameters: array of pointers in state_dict order:
 *   0 mlp_in_t.0.weight (ht,1)   1 mlp_in_t.0.bias   2 mlp_in_t.2.weight (ht,ht)   3 mlp_in_t.2.bias
 *   4 mlp_in_X.0.weight (hX,in_X) 5 .bias            6 mlp_in_X.2.weight (hX,hX)   7 .bias
 *   8 emb_Y.weight (C,hY)
 *   9+8l+0 update_X.0.weight (hX, hX+hY+ht)  +1 bias  +2 update_X.2.weight (LN gamma) +3 LN beta
 *   9+8l+4 update_Y.0.weight (hY,hY)         +5 bias  +6 update_Y.2.weight (LN gamma) +7 LN beta
 *   9+8L+0 mlp_out.0.weight (Hc,Hc) +1 bias  +2 mlp_out.2.weight (out,Hc) +3 bias
 */
static inline int gmo_tower_num_params(int num_layers) { return 9 + 8 * num_layers + 4; }
static inline int gmo_tower_hidden_cat(const gmo_tower_dims* d) {
    return (d->num_layers + 1) * (d->hidden_X + d->hidden_Y) + d->hidden_t;   /* model/gnn.py:137 */
}

uint32_t gmo_version(void);
void gmo_set_threads(int n);
int gmo_get_threads(void);

/* ---- RNG ---- */
void gmo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
float gmo_u01(uint32_t word);
/* labels: u[v], counter (v>>2, 0, 0, STREAM_Y), word v&3 */
void gmo_fill_uniform_Y(uint64_t seed, int64_t n, float* u);
/* attributes: u[f*n+v], idx=f*n+v, counter (lo(idx>>2), hi(idx>>2), step, STREAM_X), word idx&3 */
void gmo_fill_uniform_X(uint64_t seed, uint32_t step, int64_t n, int64_t F, float* u);
/* E prior: u[i*n+j] for all n*n, idx=i*n+j, counter (lo(idx>>2), hi(idx>>2), 0, STREAM_E_PRIOR), word idx&3 */
void gmo_fill_uniform_E_prior(uint64_t seed, int64_t n, float* u);
/* E step: u[rank(i,j)] for i<j row-major (torch.triu_indices offset=1), counter (j>>2, i, step, STREAM_E), word j&3 */
void gmo_fill_uniform_E(uint64_t seed, uint32_t step, int64_t n, float* u);

/* ---- schedule / transitions (model/diffusion.py:14-118) ---- */
void gmo_noise_schedule(int T, float* betas, float* alphas, float* alpha_bars); /* T+1 entries each */
void gmo_transition(float a, float m0, float m1, float Q[4]);                 /* Q[r*2+c] */
void gmo_posterior_table(const float Qt[4], const float Qbs[4], const float Qbt[4], float tab[8]);
void gmo_posterior_tables_X(int64_t F, const float* X_marginal, float alpha_t, float abar_s,
                            float abar_t, float* tab /* F*8 */);
void gmo_posterior_apply(const float tab[8], int z_t, float prior0, float prior1, float* p0, float* p1);

/* ---- priors (model/diffusion.py:188-237, 831-844) ---- */
void gmo_prior_Y(int64_t n, int C, const float* Y_marginal, const float* u_or_null, uint64_t seed,
                 int32_t* y);
void gmo_prior_X(int64_t n, int64_t F, const float* X_marginal, const float* u_or_null, uint64_t seed,
                 uint8_t* x);
void gmo_prior_E(int64_t n, int64_t wpr, float p0, float p1, const float* u_or_null, uint64_t seed,
                 uint32_t* bits /* symmetric, diagonal kept */);

/* ---- forward noising (model/diffusion.py:559-604, 962-1033) ---- */
void gmo_noise_E(int64_t n, int64_t wpr, const uint32_t* bits0, const float Q_bar_t[4], const float* u_or_null,
                 uint64_t seed, uint32_t step, uint32_t* bits_t /* symmetric, diagonal draws kept */);
void gmo_noise_X(int64_t n, int64_t F, const uint8_t* x0, const float* Q_bar_t_X /* F*4 */, const float* u_or_null,
                 uint64_t seed, uint32_t step, uint8_t* x_t);

/* ---- adjacency (model/diffusion.py:239-256, 408-426) ---- */
void gmo_bitmap_from_dense_i64(int64_t n, int64_t wpr, const int64_t* dense, uint32_t* bits);
void gmo_dense_i64_from_bitmap(int64_t n, int64_t wpr, const uint32_t* bits, int64_t* dense);
void gmo_symmetrize_upper(int64_t n, int64_t wpr, const uint32_t* upper, uint32_t* sym);
int64_t gmo_csr_from_bitmap(int64_t n, int64_t wpr, const uint32_t* bits, int32_t* rowptr,
                            int32_t* colidx /* may be NULL: count only */);
int64_t gmo_upper_edges_from_bitmap(int64_t n, int64_t wpr, const uint32_t* bits, int32_t* ei,
                                    int32_t* ej /* may be NULL */);
void gmo_bitmap_from_upper_edges(int64_t n, int64_t wpr, int64_t m, const int32_t* ei,
                                 const int32_t* ej, uint32_t* bits);

/* ---- dense / sparse building blocks (model/gnn.py:27-38,60-68) ---- */
void gmo_spmm_mean(int64_t n, const int32_t* rowptr, const int32_t* colidx, const float* H,
                   int64_t ldh, int64_t d, float* out, int64_t ldo);
void gmo_linear(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W,
                int64_t ldw, const float* bias, float* C, int64_t ldc, int relu);
void gmo_layernorm(int64_t M, int64_t N, float* X, int64_t ldx, const float* gamma,
                   const float* beta, float eps);

/* ---- towers ---- */
/* out: (n, out_size).  hcat_dbg (n, Hc) optional.  rowptr/colidx ignored for is_mlp. */
void gmo_tower_forward(const gmo_tower_dims* d, const float* const* params, int64_t n, float t_float,
                       const float* X_one_hot, const int32_t* Y, const int32_t* rowptr,
                       const int32_t* colidx, float* out, float* hcat_dbg);

/* ---- structure denoiser: pair MLP + posterior + draw (model/gnn.py:246-248, diffusion.py:444-470) ---- */
void gmo_pair_logits(int64_t hE, const float* hi, const float* hj, const float* W1, const float* b1,
                     const float* W2, const float* b2, float logit[2]);
void gmo_softmax2(float l0, float l1, float* p0, float* p1);
void gmo_softmax2_diff(float d, float* p0, float* p1);
/* Rows [row_begin,row_end) of the upper triangle.  out_upper: bitmap (n x wpr), only words of those rows
 * are written (bit j of row i for j>i).  logits_dbg/post_dbg: (P,2) indexed by rank, optional. */
void gmo_pair_score_sample(int64_t n, int64_t hE, const float* h, const float* W1, const float* b1,
                           const float* W2, const float* b2, const uint32_t* bits_t, int64_t wpr,
                           const float tab[8], int64_t row_begin, int64_t row_end,
                           const float* u_or_null, uint64_t seed, uint32_t step, uint32_t* out_upper,
                           float* logits_dbg, float* post_dbg);

/* ---- attribute posterior + draw (model/diffusion.py:875-895, 214-237) ---- */
void gmo_attr_posterior_sample(int64_t n, int64_t F, const float* logits /* (n,F,2) */,
                               const uint8_t* x_t, const float* tab /* F*8 */, const float* u_or_null,
                               uint64_t seed, uint32_t step, uint8_t* x_s, float* post_dbg /* (n,F,2) */);
/* the same from the logit difference diff (n,F) = l1 - l0, and the head folding that produces it (see gm_oracle.c) */
void gmo_attr_posterior_sample_diff(int64_t n, int64_t F, const float* diff /* (n,F) */, const uint8_t* x_t,
                                    const float* tab, const float* u_or_null, uint64_t seed, uint32_t step,
                                    uint8_t* x_s, float* post_dbg /* (n,F,2) */);
void gmo_head_diff(int64_t F, int64_t Hc, const float* W, const float* b, float* Wd, float* bd);
void gmo_onehot_from_x(int64_t n, int64_t F, const uint8_t* x, float* oh);
/* first tower layer on exactly-one-hot attributes, evaluated as base + corrections (see gm_oracle.c) */
void gmo_onehot_linear(int64_t n, int64_t F, int64_t H, const float* X_one_hot, const float* W, const float* b,
                       float* out, int64_t ldo, int relu);

/* ---- whole trajectories (model/diffusion.py:797-905, 1295-1410) ----
 * Philox mode only (seed).  Outputs: x (n*F uint8), y (n int32), bits (n*wpr symmetric, zero diagonal).
 * as_reference != 0 recomputes the edge encoder once per 32768-pair batch like model/diffusion.py:444-455
 * does (cost structure of the reference; same results). */
void gmo_sample_sync(int T, int64_t n, int64_t F, int C, const float* X_marginal, const float* Y_marginal,
                     const float* E_marginal, const gmo_tower_dims* dX, const float* const* pX,
                     const gmo_tower_dims* dE, const float* const* pE, const float* const* pPair,
                     uint64_t seed, int as_reference, uint8_t* x, int32_t* y, uint32_t* bits, int64_t wpr,
                     const float* sched /* optional [alphas | alpha_bars], T+1 each */);
void gmo_sample_async(int T_X, int T_E, int64_t n, int64_t F, int C, const float* X_marginal,
                      const float* Y_marginal, const float* E_marginal, const gmo_tower_dims* dX,
                      const float* const* pX, const gmo_tower_dims* dE, const float* const* pE,
                      const float* const* pPair, uint64_t seed, int as_reference, uint8_t* x, int32_t* y,
                      uint32_t* bits, int64_t wpr, const float* schedX, const float* schedE);

#ifdef __cplusplus
}
#endif
#endif
