// gm_nodeops.hip -- HBM-bound per-node kernels of the denoiser towers:
//   * CSR mean aggregation  (model/gnn.py:60-61  A @ H with A = D^-1 A, model/diffusion.py:255-256)
//   * LayerNorm             (model/gnn.py:30,36)
//   * small data movers used by the tower driver (embedding gather, row broadcast, strided copy)
#include "gm_common.h"

namespace gm {

// ---------------------------------------------------------------- SpMM (mean aggregation)
// out[i] = sum_{j in N(i)} w * H[j], w = 1/deg(i) (TRANSPOSED: 1/deg(j), the backward for a symmetric adjacency:
// (D^-1 A)^T G, model/gnn.py:60-61 under autograd).
//
// Arithmetic (oracle gmo_spmm_mean): a row's neighbours, ascending, are cut into segments of SEG = 64; a segment is one
// fmaf chain from 0; the row is the segment sums added in ascending segment order.  The cut makes the unit of work a
// (row, segment) ITEM of at most 64 neighbours, whatever the degree distribution: a hub row of 3000 neighbours becomes 47
// items that different lane groups run side by side, instead of one lane group walking it alone.
//
// Layout of a launch: the dense operand is cut into column slabs of 8 float4 chunks (128 B per row).  A workgroup (256
// threads = 32 groups of 8 lanes, one chunk per lane) owns 32 consecutive rows of one slab.  Thread 0 lists the block's items
// in LDS; the groups pull items from an LDS counter; for an item the 8 lanes read its <= 64 column indices from global memory
// in one coalesced pass into the group's LDS slot, then walk them 8 at a time: eight 128-byte row pieces in flight per group,
// eight fmaf in order.  Single-segment rows store straight to the output; the segment sums of longer rows meet in LDS and
// are added in order after a barrier.
// Slabs and XCDs: workgroups are dealt round-robin to the 8 XCDs, and a slab (n rows x 128 B <= 2.5 MB up to 20480 rows)
// fits an XCD's 4 MB L2, so all gathers of a slab are issued from the XCD(s) that hold it after first touch -- the operand
// is fetched across the L2 boundary about once instead of once per XCD.  With r < 8 slabs in a (last) round, XCD x takes slab
// x % r and the rows are dealt over the XCDs that share a slab.  A wrong guess about placement costs speed only.
constexpr int SP_SEG = 64;          // neighbours per segment (oracle: GMO_SPMM_SEG)
constexpr int SP_RB = 32;           // rows per workgroup (<= 64: wave 0 lists them, a lane per row)
constexpr int SP_PARTS = 96;        // segment sums a workgroup can hold in LDS for its multi-segment rows (128-byte slabs)
constexpr int SP_PARTS_WIDE = 16;   // ... with 2 KB slabs (each sum is 16x larger)
constexpr int SP_SERIAL = 0xFFFF;   // item that walks all segments of its row itself (LDS slots exhausted)

struct SpmmSched {
    int64_t rowgroups;              // ceil(n / SP_RB)
    int full_rounds;                // rounds of 8 slabs
    int rem;                        // slabs in the last, partial round (0..7)
    int64_t blocks_full;            // workgroups per full round = 8 * rowgroups
    int64_t k_rem;                  // workgroups per XCD in the partial round
};

// TPR lanes per group and CH float4 chunks per lane: a slab is TPR * CH chunks wide.  <8, 1> is the 128-byte slab described
// above; <64, 2> (one wave per item, 2 KB of a row per gather) is for operands far larger than the L2s (n > 20480), whose
// gathers come from the Infinity Cache or HBM and want long contiguous pieces instead of XCD affinity.
template <bool TRANSPOSED, int TPR, int CH>
__global__ void __launch_bounds__(256)
k_spmm_seg(int64_t n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx, const float* __restrict__ H,
           const float* __restrict__ H2, int split, int64_t ldh, int nchunks, float* __restrict__ out, int64_t ldo,
           const SpmmSched sc) {
    constexpr int NG = 256 / TPR;                       // groups per workgroup
    constexpr int SW = TPR * CH;                        // chunks per slab
    constexpr int NB = TPR == 8 ? 8 : 4;                // neighbours in flight per lane
    constexpr int PARTS = SW <= 8 ? SP_PARTS : SP_PARTS_WIDE;
    constexpr int SP_ITEMS = SP_RB + PARTS;
    // column indices (and, transposed, neighbour weights): a slot of SP_SEG per group -- or, when the workgroup's rows have at
    // most SP_PRE neighbours in all (their colidx ranges are one contiguous piece), all of them, read in ONE coalesced pass
    // before the item loop instead of one dependent global read per item
    constexpr int SP_PRE = NG * SP_SEG;                 // 2048 for the 128-byte slabs; the 2 KB-slab variant keeps its 256
    __shared__ int idx_s[SP_PRE];
    __shared__ float w_s[TRANSPOSED ? SP_PRE : 1];
    __shared__ float4 part_s[PARTS][SW];
    __shared__ int row_b[SP_RB], row_e[SP_RB], row_slot[SP_RB], row_nseg[SP_RB];
    __shared__ unsigned short item_row[SP_ITEMS], item_seg[SP_ITEMS];
    __shared__ int nitems_s, next_s;

    // ---- which slab, which 32 rows
    int64_t bid = blockIdx.x;
    int slab;
    int64_t rg;
    const int64_t nfull = sc.blocks_full * sc.full_rounds;
    if (bid < nfull) {
        const int64_t round = bid / sc.blocks_full, r = bid - round * sc.blocks_full;
        slab = (int)(round * 8 + (r & 7));
        rg = r >> 3;
    } else {
        const int64_t r = bid - nfull;
        const int x = (int)(r & 7);
        const int64_t k = r >> 3;
        const int sl = x % sc.rem;
        const int cnt = (8 - sl + sc.rem - 1) / sc.rem;     // XCDs that share this slab
        slab = sc.full_rounds * 8 + sl;
        rg = k * cnt + x / sc.rem;
        if (rg >= sc.rowgroups) return;
    }
    const int64_t i0 = rg * SP_RB;
    const int tid = threadIdx.x, g = tid / TPR, l = tid % TPR;
    int cc[CH];                                           // this lane's float4 chunks
    bool on[CH];
    const float* src[CH];                                 // chunk cc of the operand: H below `split`, H2 from there on
#pragma unroll                                            // (the towers' [h_X | h_Y] are two column ranges of one buffer)
    for (int c = 0; c < CH; ++c) {
        cc[c] = slab * SW + l + TPR * c;
        on[c] = cc[c] < nchunks;
        src[c] = cc[c] < split ? H + 4 * cc[c] : H2 + 4 * (cc[c] - split);
    }

    // ---- the block's item list: lanes 0..31 of wave 0 take a row each, two in-wave prefix sums place its items and slots
    if (tid < 64) {
        const int r = tid;
        const int64_t i = i0 + r;
        int rb = 0, re = 0;
        const bool live = r < SP_RB && i < n;
        if (live) { rb = rowptr[i]; re = rowptr[i + 1]; }
        const int deg = re - rb;
        const int ns = deg <= SP_SEG ? 1 : (deg + SP_SEG - 1) / SP_SEG;
        const bool multi = live && ns > 1;
        int slot_inc = multi ? ns : 0;                    // inclusive prefix of the slots multi-segment rows ask for
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(slot_inc, off);
            if (tid >= off) slot_inc += t;
        }
        const int slot_base = slot_inc - (multi ? ns : 0);
        const bool fits = multi && slot_base + ns <= PARTS;   // else one group walks the whole row (SP_SERIAL)
        // the (full, 64-neighbour) items of split rows go to the front of the list, so that a hub never starts last
        // (a row whose sums found no slot is walked whole by one group: those first of all)
        const bool serial = multi && !fits;
        int serial_inc = serial ? 1 : 0, heavy_inc = fits ? ns : 0, light_inc = (live && !multi) ? 1 : 0;
        for (int off = 1; off < 64; off <<= 1) {
            const int t0 = __shfl_up(heavy_inc, off), t1 = __shfl_up(light_inc, off), t2 = __shfl_up(serial_inc, off);
            if (tid >= off) { heavy_inc += t0; light_inc += t1; serial_inc += t2; }
        }
        const int heavy_total = __shfl(heavy_inc, 63), light_total = __shfl(light_inc, 63), serial_total = __shfl(serial_inc, 63);
        // single-segment rows are listed by descending degree: the 8 groups of a wave run in lockstep, so a round of items
        // costs the wave its longest one, and neighbours in the list go to the same wave (a power-law graph mixes rows of
        // 1 and 60 neighbours; unsorted, the SpMM of such a graph took 1.45x the time of a uniform graph with the same nnz)
        const int key = (live && !multi) ? deg : -1;
        int light_rank = 0;
        for (int k = 0; k < SP_RB; ++k) {
            const int dk = __shfl(key, k);
            light_rank += (dk > key || (dk == key && k < r)) ? 1 : 0;
        }
        (void)light_inc;
        const int item_base = serial ? serial_inc - 1 : fits ? serial_total + heavy_inc - ns : serial_total + heavy_total + light_rank;
        const int item_inc = serial_total + heavy_total + light_total;
        if (r < SP_RB) {
            row_b[r] = rb;
            row_e[r] = re;
            row_slot[r] = fits ? slot_base : -1;
            row_nseg[r] = fits ? ns : 1;
            if (live) {
                if (fits) {
                    for (int sg = 0; sg < ns; ++sg) { item_row[item_base + sg] = (unsigned short)r; item_seg[item_base + sg] = (unsigned short)sg; }
                } else {
                    item_row[item_base] = (unsigned short)r;
                    item_seg[item_base] = (unsigned short)(multi ? SP_SERIAL : 0);
                }
            }
        }
        if (tid == 63) { nitems_s = item_inc; next_s = 0; }
    }
    __syncthreads();
    const int nitems = nitems_s;
    const int64_t ilast = i0 + SP_RB < n ? i0 + SP_RB : n;
    const int pre_base = row_b[0], pre_cnt = row_e[(int)(ilast - i0) - 1] - pre_base;
    const bool pre = pre_cnt <= SP_PRE;
    if (pre) {
        for (int e = tid; e < pre_cnt; e += 256) {
            const int j = colidx[pre_base + e];
            idx_s[e] = j;
            if (TRANSPOSED) w_s[e] = 1.0f / (float)(rowptr[j + 1] - rowptr[j]);
        }
        __syncthreads();
    }

    for (;;) {
        int it = 0;
        if (l == 0) it = atomicAdd(&next_s, 1);
        it = __shfl(it, 0, TPR);
        if (it >= nitems) break;
        const int r = item_row[it], sgi = item_seg[it];
        const int rb = row_b[r], re = row_e[r];
        const float inv_row = re > rb ? 1.0f / (float)(re - rb) : 0.0f;
        int p0 = rb, p1 = re;
        if (sgi != SP_SERIAL) {
            p0 = rb + sgi * SP_SEG;
            p1 = p0 + SP_SEG < re ? p0 + SP_SEG : re;
        }
        float4 total[CH];
        for (int ps = p0; ps < p1 || ps == p0; ps += SP_SEG) {          // one pass unless SP_SERIAL (an empty row: one pass of nothing)
            const int pe = ps + SP_SEG < p1 ? ps + SP_SEG : p1;
            const int cnt = pe - ps;
            // the segment's column indices: already in LDS, or one coalesced read by the group's lanes into its LDS slot
            const int ib = pre ? ps - pre_base : g * SP_SEG;
            if (!pre) {
#pragma unroll
                for (int q = 0; q < (SP_SEG + TPR - 1) / TPR; ++q) {
                    const int k = l + TPR * q;
                    if (k < cnt) {
                        const int j = colidx[ps + k];
                        idx_s[ib + k] = j;
                        if (TRANSPOSED) w_s[ib + k] = 1.0f / (float)(rowptr[j + 1] - rowptr[j]);
                    }
                }
            }
            float4 acc[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = 0; k < cnt; k += NB) {                         // LDS serves a wave's own writes and reads in order
                float4 v[NB][CH];
                float w[NB];
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    const bool in = k + q < cnt;
                    int j = in ? idx_s[ib + k + q] : 0;
                    w[q] = TRANSPOSED ? (in ? w_s[ib + k + q] : 0.0f) : inv_row;
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        v[q][c] = (in && on[c]) ? *reinterpret_cast<const float4*>(src[c] + (int64_t)j * ldh)
                                                : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int q = 0; q < NB; ++q)
                    if (k + q < cnt) {
#pragma unroll
                        for (int c = 0; c < CH; ++c) {
                            acc[c].x = __builtin_fmaf(v[q][c].x, w[q], acc[c].x);
                            acc[c].y = __builtin_fmaf(v[q][c].y, w[q], acc[c].y);
                            acc[c].z = __builtin_fmaf(v[q][c].z, w[q], acc[c].z);
                            acc[c].w = __builtin_fmaf(v[q][c].w, w[q], acc[c].w);
                        }
                    }
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (ps == p0) total[c] = acc[c];
                else {
                    total[c].x = total[c].x + acc[c].x; total[c].y = total[c].y + acc[c].y;
                    total[c].z = total[c].z + acc[c].z; total[c].w = total[c].w + acc[c].w;
                }
            }
            if (ps + SP_SEG >= p1) break;
        }
        const int slot = row_slot[r];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (slot < 0) {
                if (on[c]) *reinterpret_cast<float4*>(out + (i0 + r) * ldo + 4 * cc[c]) = total[c];
            } else {
                part_s[slot + sgi][l + TPR * c] = total[c];
            }
        }
    }
    __syncthreads();
    // rows whose segment sums met in LDS: added in ascending segment order
    for (int r = g; r < SP_RB; r += NG) {
        const int slot = row_slot[r];
        if (slot < 0) continue;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            float4 t = part_s[slot][l + TPR * c];
            for (int sg = 1; sg < row_nseg[r]; ++sg) {
                const float4 a = part_s[slot + sg][l + TPR * c];
                t.x = t.x + a.x; t.y = t.y + a.y; t.z = t.z + a.z; t.w = t.w + a.w;
            }
            if (on[c]) *reinterpret_cast<float4*>(out + (i0 + r) * ldo + 4 * cc[c]) = t;
        }
    }
}

// scalar fallback for widths / strides that are not 16-byte friendly: same segment rule, one thread per column
template <bool TRANSPOSED = false>
__global__ void k_spmm_mean_scalar(int64_t n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                   const float* __restrict__ H, int64_t ldh, int64_t d, float* __restrict__ out, int64_t ldo) {
    const int64_t i = blockIdx.x;
    const int b = rowptr[i], e = rowptr[i + 1];
    const float inv_row = e > b ? 1.0f / (float)(e - b) : 0.0f;
    for (int64_t c = threadIdx.x; c < d; c += blockDim.x) {
        float total = 0.0f;
        for (int s0 = b; s0 < e; s0 += SP_SEG) {
            const int s1 = s0 + SP_SEG < e ? s0 + SP_SEG : e;
            float acc = 0.0f;
            for (int p = s0; p < s1; ++p) {
                const int j = colidx[p];
                const float inv = TRANSPOSED ? 1.0f / (float)(rowptr[j + 1] - rowptr[j]) : inv_row;
                acc = __builtin_fmaf(H[(int64_t)j * ldh + c], inv, acc);
            }
            total = s0 == b ? acc : total + acc;
        }
        out[i * ldo + c] = total;
    }
}

// ---------------------------------------------------------------- LayerNorm
// One wave per row.  Sums are lane-strided partials followed by an xor butterfly (32,16,...,1): the order the
// oracle's wave_sum64 reproduces.  Three passes over a row that sits in L2; the op is a few MB.
__device__ __forceinline__ float wave_allsum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off);
    return v;
}
__global__ void __launch_bounds__(256)
k_layernorm(int64_t M, int64_t N, float* __restrict__ X, int64_t ldx, const float* __restrict__ gamma,
            const float* __restrict__ beta, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float* x = X + m * ldx;
    float s = 0.0f;
    for (int64_t c = lane; c < N; c += 64) s = s + x[c];
    const float mean = wave_allsum(s) / (float)N;
    float q = 0.0f;
    for (int64_t c = lane; c < N; c += 64) {
        float t = x[c] - mean;
        q = q + t * t;
    }
    const float var = wave_allsum(q) / (float)N;
    const float rstd = 1.0f / __builtin_sqrtf(var + eps);
    for (int64_t c = lane; c < N; c += 64) {
        float t = (x[c] - mean) * rstd;
        float g = t * gamma[c];
        x[c] = g + beta[c];
    }
}
// The same arithmetic in the same order with the row held in registers (N <= 64 NQ): one read and one write of the row instead
// of three reads and a write.
template <int NQ>
__global__ void __launch_bounds__(256)
k_layernorm_reg(int64_t M, int64_t N, float* __restrict__ X, int64_t ldx, const float* __restrict__ gamma,
                const float* __restrict__ beta, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float* x = X + m * ldx;
    float v[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) v[k] = lane + 64 * k < N ? x[lane + 64 * k] : 0.0f;
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < NQ; ++k)
        if (lane + 64 * k < N) s = s + v[k];
    const float mean = wave_allsum(s) / (float)N;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < NQ; ++k)
        if (lane + 64 * k < N) {
            float t = v[k] - mean;
            q = q + t * t;
        }
    const float var = wave_allsum(q) / (float)N;
    const float rstd = 1.0f / __builtin_sqrtf(var + eps);
#pragma unroll
    for (int k = 0; k < NQ; ++k)
        if (lane + 64 * k < N) {
            float t = (v[k] - mean) * rstd;
            float g = t * gamma[lane + 64 * k];
            x[lane + 64 * k] = g + beta[lane + 64 * k];
        }
}

// ---------------------------------------------------------------- movers
// dst[v][0..w) = table[idx[v]][0..w)      (emb_Y, model/gnn.py:155)
__global__ void k_gather_rows(int64_t n, int w, const int32_t* __restrict__ idx, const float* __restrict__ table,
                              int64_t ldt, float* __restrict__ dst, int64_t ldd) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n * w) return;
    const int64_t v = g / w;
    const int c = (int)(g % w);
    dst[v * ldd + c] = table[(int64_t)idx[v] * ldt + c];
}
// dst[v][0..w) = row[0..w)                (h_t.expand, model/gnn.py:64,165)
__global__ void k_broadcast_row(int64_t n, int w, const float* __restrict__ row, float* __restrict__ dst, int64_t ldd) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n * w) return;
    dst[(g / w) * ldd + (g % w)] = row[g % w];
}
// mlp_in_t (model/gnn.py:116-120,153) on the one time row, Linear(1,ht)+ReLU then Linear(ht,ht)+ReLU in gm_linear's
// arithmetic (fmaf chain over ascending k from 0, then + bias), recomputed by every block and written into the h_t
// columns of two row-major buffers: one launch instead of two one-row GEMMs and two broadcasts.
__global__ void __launch_bounds__(256)
k_time_embed(int ht, const float* __restrict__ t, const float* __restrict__ W0, const float* __restrict__ b0,
             const float* __restrict__ W1, const float* __restrict__ b1, int64_t rows, float* __restrict__ dst1, int64_t ld1,
             float* __restrict__ dst2, int64_t ld2) {
    __shared__ float a0[256], a1[256];
    const int tid = threadIdx.x;
    if (tid < ht) {
        float v = __builtin_fmaf(t[0], W0[tid], 0.0f) + b0[tid];
        if (!(v > 0.0f)) v = 0.0f;
        a0[tid] = v;
    }
    __syncthreads();
    if (tid < ht) {
        float acc = 0.0f;
        for (int k = 0; k < ht; ++k) acc = __builtin_fmaf(a0[k], W1[tid * ht + k], acc);
        float v = acc + b1[tid];
        if (!(v > 0.0f)) v = 0.0f;
        a1[tid] = v;
    }
    __syncthreads();
    const int64_t total = rows * ht;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + tid; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = g / ht;
        const int c = (int)(g - r * ht);
        dst1[r * ld1 + c] = a1[c];
        dst2[r * ld2 + c] = a1[c];
    }
}
__global__ void k_copy2d(int64_t n, int w, const float* __restrict__ src, int64_t lds_, float* __restrict__ dst, int64_t ldd) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n * w) return;
    dst[(g / w) * ldd + (g % w)] = src[(g / w) * lds_ + (g % w)];
}

// out[b][0..d) = h[src[b]][0..d) * h[dst[b]][0..d)   (model/gnn.py:246 for an explicit pair list)
__global__ void k_hadamard_gather(int64_t B, int d, const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                  const float* __restrict__ h, int64_t ldh, float* __restrict__ out, int64_t ldo) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= B * d) return;
    const int64_t b = g / d;
    const int c = (int)(g % d);
    out[b * ldo + c] = h[src[b] * ldh + c] * h[dst[b] * ldh + c];
}

// the same with one float4 per thread and 256 / (d/4) pairs per workgroup (d/4 a power of two up to 64: the hidden sizes the
// configs use): no 64-bit division per element, 16-byte accesses -- the z matrix of a 2M-pair training batch is 1 GB
__global__ void __launch_bounds__(256)
k_hadamard_gather_v4(int64_t B, int d4, int shift, const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                     const float* __restrict__ h, int64_t ldh, float* __restrict__ out, int64_t ldo) {
    const int64_t b = (int64_t)blockIdx.x * (256 >> shift) + (threadIdx.x >> shift);
    if (b >= B) return;
    const int c = 4 * (threadIdx.x & (d4 - 1));
    const float4 a = *reinterpret_cast<const float4*>(h + src[b] * ldh + c);
    const float4 e = *reinterpret_cast<const float4*>(h + dst[b] * ldh + c);
    *reinterpret_cast<float4*>(out + b * ldo + c) = make_float4(a.x * e.x, a.y * e.y, a.z * e.z, a.w * e.w);
}

int launch_gather_rows(int64_t n, int w, const int32_t* idx, const float* table, int64_t ldt, float* dst, int64_t ldd,
                       hipStream_t s) {
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((n * w + 255) / 256)), dim3(256), 0, s, n, w, idx, table, ldt, dst, ldd);
    GM_LAUNCH_CHECK();
    return GM_OK;
}
int launch_broadcast_row(int64_t n, int w, const float* row, float* dst, int64_t ldd, hipStream_t s) {
    hipLaunchKernelGGL(k_broadcast_row, dim3((unsigned)((n * w + 255) / 256)), dim3(256), 0, s, n, w, row, dst, ldd);
    GM_LAUNCH_CHECK();
    return GM_OK;
}
int launch_time_embed(int ht, const float* t, const float* W0, const float* b0, const float* W1, const float* b1, int64_t rows,
                      float* dst1, int64_t ld1, float* dst2, int64_t ld2, hipStream_t s) {
    GM_REQUIRE(ht > 0 && ht <= 256, "hidden_t out of range for the fused time embedding");
    int64_t blocks = (rows * ht + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_time_embed, dim3((unsigned)blocks), dim3(256), 0, s, ht, t, W0, b0, W1, b1, rows, dst1, ld1, dst2, ld2);
    GM_LAUNCH_CHECK();
    return GM_OK;
}
int launch_copy2d(int64_t n, int w, const float* src, int64_t lds_, float* dst, int64_t ldd, hipStream_t s) {
    hipLaunchKernelGGL(k_copy2d, dim3((unsigned)((n * w + 255) / 256)), dim3(256), 0, s, n, w, src, lds_, dst, ldd);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

}  // namespace gm

using namespace gm;

// n output rows (rowptr[0..n]); the dense operand H may have more rows (a rank aggregates only its node block)
// H2 / d2: optional second column range of the operand (same row stride), appended to the d columns of H in the output
template <bool TRANSPOSED>
static int launch_spmm(int64_t n, const int32_t* rowptr, const int32_t* colidx, const float* H, int64_t ldh, int64_t d,
                       float* out, int64_t ldo, hipStream_t s, int64_t hn = 0, const float* H2 = nullptr, int64_t d2 = 0) {
    if (hn < n) hn = n;
    const bool vec = (d % 4 == 0) && (d2 % 4 == 0) && (ldh % 4 == 0) && (ldo % 4 == 0) && aligned(H, 16) && aligned(out, 16) &&
                     (!H2 || aligned(H2, 16));
    if (!vec) {
        hipLaunchKernelGGL((k_spmm_mean_scalar<TRANSPOSED>), dim3((unsigned)n), dim3(128), 0, s, n, rowptr, colidx, H, ldh, d, out, ldo);
        GM_LAUNCH_CHECK();
        if (H2) {
            hipLaunchKernelGGL((k_spmm_mean_scalar<TRANSPOSED>), dim3((unsigned)n), dim3(128), 0, s, n, rowptr, colidx, H2, ldh, d2, out + d, ldo);
            GM_LAUNCH_CHECK();
        }
        return GM_OK;
    }
    const int split = (int)(d / 4);
    if (!H2) { H2 = H; d2 = 0; }
    d += d2;
    const int nchunks = (int)(d / 4);
    const bool wide = hn > 20480;                         // a 128-byte slab of all rows no longer fits an L2
    const int nslabs = wide ? (nchunks + 127) / 128 : (nchunks + 7) / 8;
    SpmmSched sc;
    sc.rowgroups = (n + SP_RB - 1) / SP_RB;
    sc.full_rounds = nslabs / 8;
    sc.rem = nslabs % 8;
    sc.blocks_full = 8 * sc.rowgroups;
    sc.k_rem = 0;
    if (sc.rem) {
        const int cnt_min = 8 / sc.rem;                    // the slab shared by the fewest XCDs
        sc.k_rem = (sc.rowgroups + cnt_min - 1) / cnt_min;
    }
    const int64_t blocks = sc.blocks_full * sc.full_rounds + 8 * sc.k_rem;
    GM_REQUIRE(blocks > 0 && blocks < (1ll << 31), "SpMM grid out of range");
    if (wide) hipLaunchKernelGGL((k_spmm_seg<TRANSPOSED, 64, 2>), dim3((unsigned)blocks), dim3(256), 0, s, n, rowptr, colidx, H, H2, split, ldh, nchunks, out, ldo, sc);
    else hipLaunchKernelGGL((k_spmm_seg<TRANSPOSED, 8, 1>), dim3((unsigned)blocks), dim3(256), 0, s, n, rowptr, colidx, H, H2, split, ldh, nchunks, out, ldo, sc);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

// tower driver: the aggregation of a rank's node block out of all hn rows of H
int gm::launch_spmm_mean_rows(int64_t n, int64_t hn, const int32_t* rowptr, const int32_t* colidx, const float* H, int64_t ldh,
                              int64_t d, float* out, int64_t ldo, hipStream_t s) {
    GM_REQUIRE(n > 0 && d > 0 && rowptr && H && out && ldh >= d && ldo >= d, "bad arguments");
    return launch_spmm<false>(n, rowptr, colidx, H, ldh, d, out, ldo, s, hn);
}

// tower driver, single GPU: A @ [h_X | h_Y] where the two blocks are separate column ranges of hcat (row stride ldh)
int gm::launch_spmm_mean_two(int64_t n, const int32_t* rowptr, const int32_t* colidx, const float* HX, int64_t dX, const float* HY,
                             int64_t dY, int64_t ldh, float* out, int64_t ldo, hipStream_t s) {
    GM_REQUIRE(n > 0 && dX > 0 && dY > 0 && rowptr && HX && HY && out && ldh >= dX && ldh >= dY && ldo >= dX + dY, "bad arguments");
    return launch_spmm<false>(n, rowptr, colidx, HX, ldh, dX, out, ldo, s, 0, HY, dY);
}

extern "C" int gm_spmm_mean(int64_t n, const int32_t* rowptr, const int32_t* colidx, const float* H, int64_t ldh, int64_t d,
                            float* out, int64_t ldo, gm_stream_t stream) {
    GM_REQUIRE(n > 0 && d > 0 && rowptr && H && out, "bad arguments");
    GM_REQUIRE(ldh >= d && ldo >= d, "leading dimensions smaller than d");
    return launch_spmm<false>(n, rowptr, colidx, H, ldh, d, out, ldo, as_stream(stream));
}

extern "C" int gm_spmm_mean_bwd(int64_t n, const int32_t* rowptr, const int32_t* colidx, const float* G, int64_t ldg, int64_t d,
                                float* out, int64_t ldo, gm_stream_t stream) {
    GM_REQUIRE(n > 0 && d > 0 && rowptr && G && out && ldg >= d && ldo >= d, "bad arguments");
    return launch_spmm<true>(n, rowptr, colidx, G, ldg, d, out, ldo, as_stream(stream));
}

extern "C" int gm_layernorm(int64_t M, int64_t N, float* X, int64_t ldx, const float* gamma, const float* beta, float eps,
                            gm_stream_t stream) {
    GM_REQUIRE(M > 0 && N > 0 && X && gamma && beta && ldx >= N, "bad arguments");
    const dim3 grid((unsigned)((M + 3) / 4)), block(256);
    hipStream_t s = as_stream(stream);
    if (N <= 64) hipLaunchKernelGGL(k_layernorm_reg<1>, grid, block, 0, s, M, N, X, ldx, gamma, beta, eps);
    else if (N <= 512) hipLaunchKernelGGL(k_layernorm_reg<8>, grid, block, 0, s, M, N, X, ldx, gamma, beta, eps);
    else if (N <= 1024) hipLaunchKernelGGL(k_layernorm_reg<16>, grid, block, 0, s, M, N, X, ldx, gamma, beta, eps);
    else hipLaunchKernelGGL(k_layernorm, grid, block, 0, s, M, N, X, ldx, gamma, beta, eps);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

extern "C" int gm_hadamard_gather(int64_t B, int64_t d, const int64_t* src, const int64_t* dst, const float* h, int64_t ldh,
                                  float* out, int64_t ldo, gm_stream_t stream) {
    GM_REQUIRE(B > 0 && d > 0 && src && dst && h && out && ldh >= d && ldo >= d, "bad arguments");
    const int64_t d4 = d / 4;
    if (d % 4 == 0 && d4 <= 64 && (d4 & (d4 - 1)) == 0 && ldh % 4 == 0 && ldo % 4 == 0 && aligned(h, 16) && aligned(out, 16)) {
        int shift = 0;
        while ((1 << shift) < d4) ++shift;
        const int64_t per = 256 >> shift;
        GM_REQUIRE((B + per - 1) / per < (1ll << 31), "batch too large");
        hipLaunchKernelGGL(k_hadamard_gather_v4, dim3((unsigned)((B + per - 1) / per)), dim3(256), 0, as_stream(stream), B, (int)d4,
                           shift, src, dst, h, ldh, out, ldo);
    } else {
        hipLaunchKernelGGL(k_hadamard_gather, dim3((unsigned)((B * d + 255) / 256)), dim3(256), 0, as_stream(stream), B, (int)d, src,
                           dst, h, ldh, out, ldo);
    }
    GM_LAUNCH_CHECK();
    return GM_OK;
}
