/*
 * include/graphmaker_hip.h -- C ABI of libgraphmaker_hip.so, the MI355X (gfx950) implementation of
 * GraphMaker's reverse-diffusion denoising step.
 *
 * The reference (Graph-COM/GraphMaker) has no FFI of its own: its boundary is the Python API of
 * model/diffusion.py + model/gnn.py.  Each entry point below names the reference lines it replaces; the
 * Python host (graphmaker_amd/) keeps the reference's class/method signatures and calls these through ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host"; the caller (torch-ROCm) owns all
 *     memory, including outputs and workspaces; the library never allocates or frees device memory.
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it (no host sync).
 *   - return value: 0 = ok, negative = GM_E_*; text via gm_last_error() (thread-local).
 *   - adjacency state: symmetric bitmap, row i / word w holds nodes 32w..32w+31 (bit j&31), `wpr` 32-bit
 *     words per row (multiple of 4, >= ceil(n/32)); padding bits are zero.
 *   - attributes: uint8 x[v*F+f] in {0,1}; one-hot float layout oh[v*2F + 2f + c] (model/diffusion.py:232-235).
 *   - draws: u in [0,1) = (philox word >> 8) * 2^-24; 2-class draw x = (u*(p0+p1) < p1).  With `u` == NULL
 *     the kernels generate u themselves: Philox4x32-10, key = (seed lo, seed hi), counters
 *        labels      (v>>2, 0, 0, 1)            word v&3
 *        attributes  (lo(q), hi(q), step, 2)    word idx&3,  idx = f*n+v, q = idx>>2
 *        E prior     (lo(q), hi(q), 0, 3)       word idx&3,  idx = i*n+j, q = idx>>2
 *        E step      (j>>2, i, step, 4)         word j&3
 *     With `u` != NULL it is a device array of host-supplied uniforms indexed: labels u[v]; attributes
 *     u[f*n+v]; E prior u[i*n+j]; E step u[rank(i,j)], rank = row-major index of (i<j) as produced by
 *     torch.triu_indices(n, n, offset=1) (model/diffusion.py:818-819).
 */
#ifndef GRAPHMAKER_HIP_H
#define GRAPHMAKER_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GM_OK 0
#define GM_E_INVALID (-1)    /* bad argument (shape, alignment, null) */
#define GM_E_HIP (-2)        /* a HIP runtime call failed */
#define GM_E_UNSUPPORTED (-3)

typedef void* gm_stream_t;

typedef struct {
    int in_X;           /* num_attrs_X * num_classes_X = 2F                (model/gnn.py:112) */
    int num_classes_Y;
    int hidden_t, hidden_X, hidden_Y;
    int num_layers;     /* num_gnn_layers / num_mlp_layers */
    int out_size;       /* hidden_E (edge tower) or 2F (attribute towers) */
    int is_mlp;         /* 0: GNNTower model/gnn.py:72-176; 1: MLPTower model/gnn.py:396-463 */
} gm_tower_dims;

int gm_version(void);
const char* gm_last_error(void);
/* host: number of compute units of the current device (grid sizing is internal; exposed for reports) */
int gm_device_cus(int* cus);

/* ---- adjacency: replaces the dense (N,N) int64 E_t of model/diffusion.py:422-424 and get_adj :239-256 ---- */
int gm_bitmap_from_dense_i64(int64_t n, int64_t wpr, const int64_t* dense, uint32_t* bits, gm_stream_t stream);
int gm_dense_i64_from_bitmap(int64_t n, int64_t wpr, const uint32_t* bits, int64_t* dense, gm_stream_t stream);
/* sym = upper (bits j>=i of each row) mirrored below the diagonal; model/diffusion.py:208-210, 423-424 */
int gm_symmetrize_upper(int64_t n, int64_t wpr, const uint32_t* upper, uint32_t* sym, gm_stream_t stream);
/* get_adj (model/diffusion.py:252-256): E_t.nonzero() row-major -> CSR with ascending columns; the values
 * 1/deg[row] stay implicit.  upper_only != 0 keeps only j>i (sorted upper edge list; `rows` then receives the
 * row index of every entry, may be NULL).  Two calls because colidx is sized by rowptr[n]:
 *   gm_csr_rowptr  : rowptr[0..n]           (workspace: none)
 *   gm_csr_fill    : colidx[0..rowptr[n])   (caller read rowptr[n] to size it, or passes capacity >= nnz) */
int gm_csr_rowptr(int64_t n, int64_t wpr, const uint32_t* bits, int upper_only, int32_t* rowptr, gm_stream_t stream);
/* gm_csr_rowptr for a caller-owned colidx buffer of `capacity` entries: the row pointers are clamped to the capacity and
 * *overflow (device int32, caller-zeroed once) is OR-ed with 1 when the graph has more entries.  Followed by gm_csr_fill
 * with the same capacity it builds the adjacency of get_adj (model/diffusion.py:239-256) with no host read of the edge
 * count, so consecutive reverse steps can be enqueued without a device-to-host synchronisation; the caller reads the
 * flag once per sample and discards the sample if it is set. */
int gm_csr_rowptr_capped(int64_t n, int64_t wpr, const uint32_t* bits, int upper_only, int64_t capacity,
                         int32_t* rowptr, int32_t* overflow, gm_stream_t stream);
int gm_csr_fill(int64_t n, int64_t wpr, const uint32_t* bits, int upper_only, const int32_t* rowptr,
                int64_t capacity, int32_t* colidx, int32_t* rows, gm_stream_t stream);
/* edge list (i<j) -> symmetric bitmap (multi-GPU gather path; model/diffusion.py:423-424) */
int gm_bitmap_from_upper_edges(int64_t n, int64_t wpr, int64_t m, const int32_t* ei, const int32_t* ej,
                               uint32_t* bits, gm_stream_t stream);

/* Multi-GPU exchange of the sampled edges ("the sampled edge lists are gathered back": E_t[dst,src] = E_t[src,dst] = draw,
 * model/diffusion.py:421-424, done across ranks).
 * gm_upper_edges_ranges: the edges (i<j) of up to 4 row ranges (HOST array [b0,e0,b1,e1,...], a rank's pair chunks) of the
 *   upper-triangle bitmap the pair kernel wrote, as a compact list sorted by (i, j): ei[k], ej[k] for k < count[0] (device
 *   int32, clamped to `capacity`); rowptr_ws: scratch of R+1 entries, R = total number of rows in the ranges.  At most
 *   `capacity` edges are written; *overflow (device, sticky) is OR-ed with 1 when there were more, and count[1] receives a
 *   copy of *overflow, so that the word travels with the segment (`count` points at TWO int32).  The rows of `upper` must
 *   hold zeros in the words the pair kernel does not write (allocate it zeroed once; the kernel rewrites its words on every
 *   call).
 * gm_bitmap_from_edge_segments: the symmetric bitmap of E_s from every rank's list after ONE equal-sized all-gather of
 *   segments [ei (cap) | ej (cap) | count | overflowed | padding] of `stride` int32 each (stride >= 2*cap + 2).  A
 *   segment that arrives with its overflow word set (its rank sampled more edges than `cap`; the list is truncated) sets
 *   *overflow (device, sticky, may be NULL) on THIS rank too: every rank then rejects the sample, not only the one that
 *   overflowed. */
int gm_upper_edges_ranges(int64_t n, int64_t wpr, const uint32_t* upper, int nranges, const int64_t* row_ranges,
                          int64_t capacity, int32_t* rowptr_ws, int32_t* ei, int32_t* ej, int32_t* count, int32_t* overflow,
                          gm_stream_t stream);
int gm_bitmap_from_edge_segments(int64_t n, int64_t wpr, int nseg, int64_t cap, int64_t stride, const int32_t* edges,
                                 uint32_t* bits, int32_t* overflow, gm_stream_t stream);

/* ---- message passing: A @ H with A = D^-1 A (model/gnn.py:60-61, model/diffusion.py:255-256) ---- */
int gm_spmm_mean(int64_t n, const int32_t* rowptr, const int32_t* colidx, const float* H, int64_t ldh, int64_t d,
                 float* out, int64_t ldo, gm_stream_t stream);

/* ---- nn.Linear (+ReLU) and nn.LayerNorm as used by model/gnn.py:27-38,116-126,138-142,227-231 ----
 * C[M,N] (ldc) = act(A[M,K] (lda) @ W[N,K]^T (ldw) + bias[N]);  f32 MFMA, k-ascending accumulation.  The tile
 * configuration follows the shape (the bits do not: every output is the same ascending-k fmaf chain); N <= 4 over >= 65536
 * rows (the pair MLP's output layer on a training batch) runs a row-per-lane kernel bound by reading A once. */
int gm_linear(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W, int64_t ldw,
              const float* bias, float* C, int64_t ldc, int relu, gm_stream_t stream);
int gm_layernorm(int64_t M, int64_t N, float* X, int64_t ldx, const float* gamma, const float* beta, float eps,
                 gm_stream_t stream);

/* ---- towers: GNNTower.forward model/gnn.py:146-176, MLPTower.forward model/gnn.py:440-463 ----
 * params: HOST array of DEVICE pointers in state_dict order:
 *   mlp_in_t.{0,2}.{weight,bias}, mlp_in_X.{0,2}.{weight,bias}, emb_Y.weight,
 *   per layer: update_X.0.{weight,bias}, update_X.2.{weight,bias}, update_Y.0.{weight,bias}, update_Y.2.{weight,bias},
 *   mlp_out.{0,2}.{weight,bias}                                   (9 + 8*num_layers + 4 pointers)
 * t_float: device pointer to one float (the reference passes a (1,) tensor, model/diffusion.py:860).
 * out: (n, out_size) row-major.  hcat_out: optional (n, hidden_cat) copy of the concatenated hidden state.
 * [out_row_begin, out_row_end): rows of `out` to produce.  The graph part of the tower always runs on all n nodes
 * (message passing needs every row); the row-independent output MLP (model/gnn.py:138-142), most of the FLOPs,
 * runs only on this range, which is how one tower is sharded over GPUs (pass 0, n for everything).
 * X_one_hot == NULL (here and in stage -1 of gm_tower_stage): the attributes are those of the previous call on this
 * workspace, so mlp_in_X(X) (model/gnn.py:154) is taken from the workspace instead of being recomputed -- the structure
 * steps of ModelAsync.sample condition on the final attributes (model/diffusion.py:1379-1401), only t and A_t change.
 * First layer (mlp_in_X.0, model/gnn.py:121-126,154).  X_t_one_hot is exactly one-hot with two classes per attribute
 * (model/diffusion.py:232-235), so X W^T + b = B + sum_{f: x_f = 1} D[:, f] with D[o][f] = W[o][2f+1] - W[o][2f] and
 * B[o] = (sum_f W[o][2f]) + b[o]: half the GEMM (K = F), or -- sparse_X != 0, for bag-of-words attributes with few ones -- a
 * walk over each node's set attributes; both give the same bits.  `prep` holds B, D and D^T: gm_tower_prep_floats() floats,
 * filled by gm_tower_prepare once per version of the weights.  The attributes come either as x_u8 (n, F) in {0,1} or as
 * X_one_hot (n, 2F) (then they are recovered from it into the workspace); both NULL = reuse, as described above. */
int gm_tower_workspace_bytes(const gm_tower_dims* dims, int64_t n, size_t* bytes);
int64_t gm_tower_prep_floats(const gm_tower_dims* dims);
int gm_tower_prepare(const gm_tower_dims* dims, const float* const* params, float* prep, gm_stream_t stream);
int gm_tower_forward(const gm_tower_dims* dims, const float* const* params, int64_t n, const float* t_float,
                     const float* X_one_hot, const uint8_t* x_u8, const float* prep, int sparse_X, const int32_t* Y,
                     const int32_t* rowptr, const int32_t* colidx, float* out, float* hcat_out, int64_t out_row_begin,
                     int64_t out_row_end, void* workspace, size_t workspace_bytes, gm_stream_t stream);
/* The first layer on its own (what the towers call; also used by the training tape, model/diffusion.py:606-654).
 * gm_onehot_prepare: prep (gm_onehot_prep_floats(F, H) floats) from W (H, 2F) row-major and b (H, may be NULL).
 * gm_onehot_linear: out (n, H; ldo) = act(B + sum of the set attributes' columns of D); dense_ws = (n, roundup(F,4)) floats
 * selects the GEMM form, NULL the sparse walk (needs H % 4 == 0, H <= 2048, F < 65536). */
int64_t gm_onehot_prep_floats(int64_t F, int64_t H);
int gm_onehot_prepare(int64_t F, int64_t H, const float* W, const float* b, float* prep, gm_stream_t stream);
int gm_onehot_linear(int64_t n, int64_t F, int64_t H, const uint8_t* x, const float* prep, float* dense_ws, float* out,
                     int64_t ldo, int relu, gm_stream_t stream);

/* One stage of a tower for node rows [row_begin,row_end) -- the unit of the multi-GPU decomposition, in which every
 * stage is row-sharded and only the aggregation input crosses GPUs (same workspace for all stages of one forward):
 *   stage -1        : input projections mlp_in_t / mlp_in_X / emb_Y (model/gnn.py:153-155)      needs t_float, X_one_hot, Y
 *   stage l in [0,L): GNNLayer / MLPLayer l (model/gnn.py:40-70, :369-394)                        needs rowptr/colidx (GNN)
 *   stage L         : mlp_out (model/gnn.py:165-176) -> rows of `out`
 * layer_src: (n, hidden_X+hidden_Y) matrix [h_X^l | h_Y^l] of ALL nodes that stage l aggregates from (the all-gathered
 *            buffer); NULL = read the tower's own hidden state (valid when this process computed every row).
 * layer_dst: optional (n, hidden_X+hidden_Y) buffer; rows [row_begin,row_end) receive [h_X | h_Y] as written by this
 *            stage (stage -1 -> h^0, stage l -> h^(l+1)): the send side of the all-gather. */
int gm_tower_stage(const gm_tower_dims* dims, const float* const* params, int64_t n, int stage, int64_t row_begin,
                   int64_t row_end, const float* t_float, const float* X_one_hot, const uint8_t* x_u8, const float* prep,
                   int sparse_X, const int32_t* Y, const int32_t* rowptr, const int32_t* colidx, const float* layer_src,
                   float* layer_dst, float* out, void* workspace, size_t workspace_bytes, gm_stream_t stream);

/* z[b] = h[src[b]] * h[dst[b]] for an explicit pair list: LinkPredictor.forward on a caller-chosen batch
 * (model/gnn.py:246); followed by two gm_linear calls it gives the (B,2) logits of model/gnn.py:248. */
int gm_hadamard_gather(int64_t B, int64_t d, const int64_t* src, const int64_t* dst, const float* h, int64_t ldh,
                       float* out, int64_t ldo, gm_stream_t stream);

/* ---- structure denoiser: the loop of get_E_t (model/diffusion.py:444-470) over ALL pairs i<j of rows
 * [row_begin,row_end): pair MLP (model/gnn.py:246-248) -> softmax (:457) -> posterior (:351-406, as the table
 * `tab`) -> draw (:421).  Nothing P-sized is stored: the result is the upper-triangle bitmap `out_upper`
 * (rows [row_begin,row_end), words w >= (row>>5) are written).
 * hE = 128 (every shipped configuration) runs the f32 matrix-core kernel (gm_pair.hip); any other 1 <= hE <= 512, which the
 * reference also accepts (model/gnn.py:204-231), runs a vector-ALU kernel with the same arithmetic orders and the same
 * bits as the oracle (gm_pair_generic.hip; it does not use `workspace`).  The hidden ReLU is exact for activations that are
 * <= 0 or within [2^-63, 2^63] (it is evaluated as clamp01(x * 2^-63) against second-layer weights scaled by 2^63, which must
 * therefore stay below 2^64 in magnitude); smaller positive activations are rounded to multiples of 2^-86, larger ones saturate.  h and W1 16-byte aligned, bits_t and out_upper
 * 8-byte aligned with wpr a multiple of 4 (rows then stay aligned).  tab: HOST pointer to 8 floats tab[(i*2+k)*2+e] = Q_t[k][e]*Qbar_s[i][k]/Qbar_t[i][e].
 * workspace: >= 256 bytes, zeroed by the caller when it is allocated.  Word 0 is the tile counter (cleared by every
 * call).  Word 16 is a sticky trouble flag: the kernel's waves hand tiles to each other through flags in LDS without
 * workgroup barriers (gm_pair_protocol.h).  A wave that waits for such a flag far beyond any legitimate delay (2^18 polls,
 * ~30 ms) sets the word; from then on no wave of its workgroup waits again, every workgroup stops taking tiles, and the
 * launch ends: the GPU is never hung, and the outputs of that call are void.  A launch that finds the word set does
 * nothing (it ends at once, outputs untouched).  The library never clears it: test it wherever the caller synchronises
 * anyway, and clear it before the workspace is used again (graphmaker_amd does both once per trajectory).
 * logits_dbg/post_dbg: optional (P,2) arrays indexed by rank. */
int gm_pair_score_sample(int64_t n, int64_t hE, const float* h, const float* W1, const float* b1, const float* W2,
                         const float* b2, const uint32_t* bits_t, int64_t wpr, const float* tab,
                         int64_t row_begin, int64_t row_end, const float* u, uint64_t seed, uint32_t step,
                         uint32_t* out_upper, float* logits_dbg, float* post_dbg, void* workspace,
                         gm_stream_t stream);
/* The same loop over up to 4 ascending, disjoint row ranges in ONE launch (row_ranges: HOST array
 * [begin0,end0,begin1,end1,...]): the per-GPU share of the pair matrix is two folded row chunks, and one launch with a
 * shared tile queue keeps every CU busy where two back-to-back launches each end with a partly idle chip. */
int gm_pair_score_sample_ranges(int64_t n, int64_t hE, const float* h, const float* W1, const float* b1,
                                const float* W2, const float* b2, const uint32_t* bits_t, int64_t wpr, const float* tab,
                                int nranges, const int64_t* row_ranges, const float* u, uint64_t seed, uint32_t step,
                                uint32_t* out_upper, float* logits_dbg, float* post_dbg, void* workspace,
                                gm_stream_t stream);

/* OPT-IN reduced precision (never the default): the same loop with the 128x128 product hid = W1 (h_i * h_j) on the bf16
 * matrix cores, f32 accumulation, everything else (z products, second layer, softmax, posterior, draw) in f32 as above.
 *   nprod = 3 ("bf16x3"): W1 and z are split into bf16 head + tail and three products are summed (error 2^-16 relative per
 *                         product): logits stay within the f32 tolerance of the reference (atol 2e-5);
 *   nprod = 1 ("bf16")  : heads only (2^-8 relative per product; logits atol ~2e-2).
 * Draws are not bit-identical to gm_pair_score_sample_ranges: a draw may differ where |u*(p0+p1) - p1| is within the posterior
 * error.  Same arguments and workspace as gm_pair_score_sample_ranges. */
int gm_pair_score_sample_bf16(int64_t n, int64_t hE, const float* h, const float* W1, const float* b1, const float* W2,
                              const float* b2, const uint32_t* bits_t, int64_t wpr, const float* tab, int nranges,
                              const int64_t* row_ranges, const float* u, uint64_t seed, uint32_t step, uint32_t* out_upper,
                              float* logits_dbg, float* post_dbg, void* workspace, int nprod, gm_stream_t stream);

/* ---- attribute posterior + draw: model/diffusion.py:879-895 + sample_X :214-237 ----
 * logits (n,F,2); tab device (F,8) per-attribute posterior tables; x_s (n,F) uint8; onehot_out optional (n,2F).
 * Only nodes [v_begin, v_end) are processed (all pointers address the full arrays): the per-GPU shard. */
int gm_attr_posterior_sample(int64_t n, int64_t F, const float* logits, const uint8_t* x_t, const float* tab,
                             const float* u, uint64_t seed, uint32_t step, int64_t v_begin, int64_t v_end,
                             uint8_t* x_s, float* onehot_out, float* post_dbg, gm_stream_t stream);
/* The same step from the logit DIFFERENCE diff[v][f] = l1 - l0 (n,F): softmax([l0,l1]) = softmax([0, l1-l0]), the form
 * the pair kernel uses for the edge classes.  The sampler (model/diffusion.py:875-895) only ever takes the softmax of the
 * node tower's output, so it runs the tower with the half-width head of gm_head_diff_prepare in place of mlp_out.2
 * (out_size = F instead of 2F: half of the tower's largest GEMM) and calls this entry.  diff differs from the difference
 * of the two rounded logits by f32 rounding only (same order as the GEMM's own); the oracle computes it the same way.
 * forward() keeps the two-logit head; the training loss uses the difference too (gm_ce2_diff below). */
int gm_attr_posterior_sample_diff(int64_t n, int64_t F, const float* diff, const uint8_t* x_t, const float* tab,
                                  const float* u, uint64_t seed, uint32_t step, int64_t v_begin, int64_t v_end,
                                  uint8_t* x_s, float* onehot_out, float* post_dbg, gm_stream_t stream);
/* Wd (F,Hc) = W[2f+1] - W[2f], bd (F) = b[2f+1] - b[2f] of mlp_out.2 (W (2F,Hc) row-major, model/gnn.py:139-143):
 * pass Wd / bd as the last two tower parameters with dims.out_size = F to get diff.  Recompute when W or b change. */
int gm_head_diff_prepare(int64_t F, int64_t Hc, const float* W, const float* b, float* Wd, float* bd, gm_stream_t stream);
int gm_onehot_from_x(int64_t n, int64_t F, const uint8_t* x, float* onehot, gm_stream_t stream);
int gm_x_from_onehot(int64_t n, int64_t F, const float* onehot, uint8_t* x, gm_stream_t stream);

/* ---- priors: model/diffusion.py:831-844 (sample_E :188-212 keeps the diagonal) ---- */
int gm_prior_Y(int64_t n, int C, const float* Y_marginal, const float* u, uint64_t seed, int32_t* y,
               gm_stream_t stream);
int gm_prior_X(int64_t n, int64_t F, const float* X_marginal, const float* u, uint64_t seed, uint8_t* x,
               float* onehot_out, gm_stream_t stream);
/* upper_ws: scratch bitmap (n*wpr words); bits: symmetric result incl. diagonal draws */
int gm_prior_E(int64_t n, int64_t wpr, float p0, float p1, const float* u, uint64_t seed, uint32_t* upper_ws,
               uint32_t* bits, gm_stream_t stream);
/* gm_prior_E for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  It has NO counterpart in the reference.  Same result as gm_prior_E bit for bit; the scratch
 * bitmap `upper_ws` is a 32 MiB array inside the library (one copy per GPU, held until the process ends) instead of an
 * argument, and the call allocates nothing.  Limit: n <= 16384 and n*wpr <= 8,388,608 words -- larger graphs get
 * GM_E_INVALID and must use gm_prior_E.  Every call on a GPU shares that GPU's array: two calls that overlap on one GPU
 * (different streams, different threads) race; calls on one stream are ordered.  The array keeps the previous call's
 * contents, which never matter: a call overwrites every word it later reads. */
int gm_prior_E_static(int64_t n, int64_t wpr, float p0, float p1, const float* u, uint64_t seed, uint32_t* bits,
                      gm_stream_t stream);

/* ---- forward noising q(G^t | G): model/diffusion.py:559-604 (Sync apply_noise), :962-1033 (Async apply_noise_X/_E) ----
 * Philox streams 5 (E, idx = i*n+j for j >= i, diagonal drawn and kept like sample_E :188-212) and 6 (X, idx = f*n+v);
 * `step` is the caller's draw counter (any value; two calls with the same seed and step repeat the draw).
 * Q_bar_t: host pointer, 4 floats Q[r*2+c] (get_Q_bar_E :52-67).  Q_bar_t_X: device pointer, F*4 floats (:69-84). */
int gm_noise_E(int64_t n, int64_t wpr, const uint32_t* bits0, const float* Q_bar_t, const float* u, uint64_t seed,
               uint32_t step, uint32_t* upper_ws, uint32_t* bits_t, gm_stream_t stream);
int gm_noise_X(int64_t n, int64_t F, const uint8_t* x0, const float* Q_bar_t_X, const float* u, uint64_t seed,
               uint32_t step, uint8_t* x_t, float* onehot_out, gm_stream_t stream);
/* the reference's dense training tensors -> library state: E_one_hot (N,N,2) float (data.py:73-76) -> bitmap,
 * X_one_hot_3d (F,N,2) float (data.py:65-71) -> uint8 x[N][F] */
int gm_bitmap_from_onehot_f32(int64_t n, int64_t wpr, const float* E_one_hot, uint32_t* bits, gm_stream_t stream);
int gm_x_from_onehot3d(int64_t n, int64_t F, const float* X_one_hot_3d, uint8_t* x, gm_stream_t stream);

/* ---- training step log_p_t (model/diffusion.py:606-654, :1035-1094; train_sync.py:94-120): operator backward halves.
 * torch.autograd in the reference; here every gradient is a kernel with a fixed reduction order (reproducible). */
/* C[m][n] = act( sum_k a(m,k) b(n,k) + bias[n] ), zeroed where mask[m][n] <= 0 (ReLU backward);
 * a(m,k) = a_red ? A[k*lda+m] : A[m*lda+k];  b(n,k) = b_red ? B[k*ldb+n] : B[n*ldb+k].
 * nn.Linear (model/gnn.py:27-38): forward (0,0) A=X B=W; dX = dY W -> (0,1) A=dY B=W; dW = dY^T X -> (1,1) A=dY B=X.
 * Small outputs with a long reduction are split over K into `workspace` and summed in ascending split order; the split
 * count depends on the shape only (how well tiles x splits fills the chip), so a result is reproducible bit for bit.
 * (1,1) with M <= 4, N <= 128 and K >= 65536 (the weight gradient of a layer with a few outputs over a pair batch) is a
 * streaming kernel instead of an MFMA tile. */
int64_t gm_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);
int gm_gemm(int a_red, int b_red, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
            int64_t ldb, const float* bias, const float* mask, int64_t ldmask, int relu, float* C, int64_t ldc,
            void* workspace, int64_t workspace_bytes, gm_stream_t stream);
/* Hidden gradient of Linear -> ReLU -> Linear (model/gnn.py:227-231, the pair MLP) when the second layer has N <= 4 outputs:
 * g1[m][k] = y1[m][k] > 0 ? sum_c g[m][c] W2[c][k] : 0 (the bits of gm_gemm(0,1,...) with mask = y1) and, if db1 is given,
 * db1[k] = sum_m g1[m][k], in one pass over the M x H arrays.  g (M,N) and y1, g1 (M,H) are dense row-major; H = 4 x a power
 * of two up to 1024.  The workspace (only needed for db1) is gm_mlp2_hidden_grad_workspace_bytes (-1: shape not supported). */
int64_t gm_mlp2_hidden_grad_workspace_bytes(int64_t M, int64_t H, int64_t N);
int gm_mlp2_hidden_grad(int64_t M, int64_t H, int64_t N, const float* g, const float* W2, int64_t ldw, const float* y1, float* g1,
                        float* db1, void* workspace, int64_t workspace_bytes, gm_stream_t stream);
/* out[n] = sum_m X[m][n]  (bias gradient) */
int64_t gm_colsum_workspace_bytes(int64_t M, int64_t N);
int gm_colsum(int64_t M, int64_t N, const float* X, int64_t ldx, float* out, void* workspace, int64_t workspace_bytes,
              gm_stream_t stream);
/* nn.LayerNorm (model/gnn.py:30,36) out of place, statistics saved for the backward */
int gm_layernorm_fwd(int64_t M, int64_t N, const float* X, int64_t ldx, const float* gamma, const float* beta,
                     float eps, float* Y, int64_t ldy, float* mean, float* rstd, gm_stream_t stream);
/* dX (zeroed where X <= 0 when relu_mask: the ReLU in front of the LayerNorm), dgamma, dbeta = dgamma + N */
int64_t gm_layernorm_bwd_workspace_bytes(int64_t M, int64_t N);
int gm_layernorm_bwd(int64_t M, int64_t N, const float* dY, int64_t lddy, const float* X, int64_t ldx,
                     const float* mean, const float* rstd, const float* gamma, int relu_mask, float* dX,
                     int64_t lddx, float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes,
                     gm_stream_t stream);
/* backward of gm_spmm_mean for a symmetric adjacency: out[j] = sum_{i in N(j)} G[i] / deg(i)  (model/gnn.py:60-61) */
int gm_spmm_mean_bwd(int64_t n, const int32_t* rowptr, const int32_t* colidx, const float* G, int64_t ldg, int64_t d,
                     float* out, int64_t ldo, gm_stream_t stream);
/* d emb_Y.weight (C, w) of the label embedding (model/gnn.py:155): dW[c] = sum_{v: idx[v]==c} G[v] */
int64_t gm_embedding_bwd_workspace_bytes(int64_t n, int C, int w);
int gm_embedding_bwd(int64_t n, int C, int w, const int32_t* idx, const float* G, int64_t ldg, float* dW, int64_t ldw,
                     void* workspace, int64_t workspace_bytes, gm_stream_t stream);
/* d h of z[b] = h[src[b]] * h[dst[b]] (model/gnn.py:246).  order (2B): stable sort permutation of cat[src, dst] by
 * node; segptr (n+1): start of each node's run in `order`.  No atomics: one wave per node walks its run in order. */
int gm_hadamard_gather_bwd(int64_t n, int64_t B, int64_t d, const int64_t* src, const int64_t* dst,
                           const int64_t* segptr, const int64_t* order, const float* h, int64_t ldh, const float* dZ,
                           int64_t lddz, float* dh, int64_t lddh, gm_stream_t stream);
/* out = g where y > 0 else 0 (nn.ReLU backward, model/gnn.py:29,35,118,124,140) */
/* The attribute loss from the logit difference (LossX, model/diffusion.py:474-515, for two classes): a training step may run the
 * node tower with the folded head of gm_head_diff_prepare (half of its output GEMM forward and backward), since
 * CE(softmax([l0,l1]), y) = log(1 + exp(-|d|)) + max(-s, 0) with d = l1 - l0, s = y ? d : -d.  loss (1 float) = mean over
 * `total` items, ddiff (optional) = d loss / d diff; workspace as gm_softmax_ce2.  gm_head_diff_bwd maps the gradient of the
 * folded head back to mlp_out.2: dW[2f+1] = dWd[f], dW[2f] = -dWd[f], likewise the bias (db / dbd may be NULL). */
int gm_ce2_diff(int64_t total, const float* diff, const uint8_t* target, float* loss, float* ddiff, void* workspace,
                int64_t workspace_bytes, gm_stream_t stream);
int gm_head_diff_bwd(int64_t F, int64_t Hc, const float* dWd, const float* dbd, float* dW, float* db, gm_stream_t stream);
int gm_relu_bwd(int64_t total, const float* g, const float* y, float* out, gm_stream_t stream);
/* nn.Dropout (model/gnn.py:31,37) in training mode: out = x/(1-p) where u >= p else 0, u = Philox stream 7 at element
 * index with counter word `call`; the same (seed, call) applied to the gradient is the backward */
int gm_dropout(int64_t total, const float* x, float p, uint64_t seed, uint32_t call, float* out, gm_stream_t stream);
/* nn.Embedding forward (emb_Y, model/gnn.py:155): out[v] = weight[idx[v]]; idx must lie in [0, C) */
int gm_embedding_fwd(int64_t n, int C, int w, const int32_t* idx, const float* weight, int64_t ldw, float* out,
                     int64_t ldo, gm_stream_t stream);
/* F.cross_entropy over 2 classes, mean over `total` items (LossX / LossE, model/diffusion.py:120-141, 474-515):
 * loss[0] = mean(logsumexp(l) - l[target]); dlogits (optional) = (softmax(l) - onehot(target)) / total */
int64_t gm_softmax_ce2_workspace_bytes(void);
int gm_softmax_ce2(int64_t total, const float* logits, const uint8_t* target, float* loss, float* dlogits,
                   void* workspace, int64_t workspace_bytes, gm_stream_t stream);

/* ---- validation (val_step, model/diffusion.py:709-795, :1196-1293) ----
 * denoising-match KL of BaseModel.denoise_match_Z (:258-303) for 2 classes: out[0] = mean over items and classes of
 * clamp(KL element, 0).  Item g uses transition tables g % ntab (ntab = F for attributes laid out (v,f), 1 for edges);
 * Q_t, Q_bar_s: device, ntab*4 floats Q[r*2+c].  workspace: gm_softmax_ce2_workspace_bytes(). */
int gm_denoise_match2(int64_t total, const float* logits, const uint8_t* z_t, const uint8_t* z_0, const float* Q_t,
                      const float* Q_bar_s, int64_t ntab, float* out, void* workspace, int64_t workspace_bytes,
                      gm_stream_t stream);
/* out[b] = E[i[b]][j[b]] read from the bitmap (E_t[batch_src, batch_dst], :765); ids must lie in [0, n) */
int gm_bitmap_lookup(int64_t n, int64_t wpr, const uint32_t* bits, int64_t B, const int64_t* i, const int64_t* j,
                     uint8_t* out, gm_stream_t stream);

/* ---- integer statistics of a generated graph (eval_utils.py:19-74, 506-596) on bitmap + CSR; no self loops ----
 * tri2[v] = sum_{u in N(v)} |N(v) & N(u)| = 2 * (triangles through v): get_triangle_count (:19-21) = sum(tri2) / 6,
 * nx.clustering (:585) = tri2[v] / (deg(v) * (deg(v) - 1)) */
int gm_triangles_per_node(int64_t n, int64_t wpr, const uint32_t* bits, const int32_t* rowptr, const int32_t* colidx,
                          int64_t* tri2, gm_stream_t stream);
/* out[v] = |{u in N(v): y[u] == y[v]}|  (linkx_homophily, eval_utils.py:52-59) */
int gm_same_class_degree(int64_t n, const int32_t* rowptr, const int32_t* colidx, const int32_t* y, int32_t* out,
                         gm_stream_t stream);
/* pattern of A @ A (k_order_g, eval_utils.py:521-529): out[i] = OR_{j in N(i)} bits[j] (the diagonal is set where deg > 0) */
int gm_bitmap_square(int64_t n, int64_t wpr, const uint32_t* bits, const int32_t* rowptr, const int32_t* colidx,
                     uint32_t* out, gm_stream_t stream);
/* out[c] = bitmap row of the nodes with label c (C rows of wpr words) */
int gm_class_bitmaps(int64_t n, int64_t wpr, int C, const int32_t* y, uint32_t* out, gm_stream_t stream);
/* deg[v] = popcount(bits[v]); same[v] = popcount(bits[v] & classrows[y[v]]) */
int gm_masked_degree(int64_t n, int64_t wpr, const uint32_t* bits, const uint32_t* classrows, const int32_t* y,
                     int32_t* deg, int32_t* same, gm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
