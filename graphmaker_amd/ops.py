"""Tensor-level wrappers of the C ABI (include/graphmaker_hip.h).  Each function is one library call on torch's
current stream; torch supplies device memory and nothing else."""
import ctypes as C
import weakref

import torch

from . import _lib as L
from ._lib import F32, I32, I64, U32, U64, ptr, require, stream_ptr, words_per_row


# Index arguments that come from the caller (labels, pair lists) are range-checked on the host before a kernel indexes with
# them: the reference's nn.Embedding / tensor indexing raise on a bad index, an unchecked kernel would read out of bounds.
# One fused min/max reduction + ONE host read per tensor OBJECT and version: the memo is keyed on the tensor object (weakly),
# never on its address (a fresh batch tensor that the caching allocator puts at the address of the previous one
# is a different object and is checked again).  A training loop that builds new index tensors every step therefore pays one
# device-to-host synchronisation per index argument and step; loops that cannot afford it validate their batches once
# up front and set VALIDATE_INDICES = False.
VALIDATE_INDICES = True
_validated = {}       # id(tensor) -> (weak reference to that tensor, {(what, hi): version}); an entry dies with its tensor


def check_index_range(t, hi, what):
    """Raise IndexError unless every entry of the integer tensor t lies in [0, hi)."""
    if not VALIDATE_INDICES or t.numel() == 0:
        return
    entry = _validated.get(id(t))
    if entry is not None and entry[0]() is not t:          # the id was recycled before the old tensor's callback ran
        entry = None
    if entry is not None and entry[1].get((what, int(hi))) == t._version:
        return
    lo_v, hi_v = (int(v) for v in torch.stack(torch.aminmax(t)).tolist())
    if lo_v < 0 or hi_v >= hi:
        raise IndexError(f"{what}: values span [{lo_v}, {hi_v}], valid range is [0, {hi})")
    if entry is None:
        key = id(t)
        entry = _validated[key] = (weakref.ref(t, lambda _r, k=key: _validated.pop(k, None)), {})
    entry[1][(what, int(hi))] = t._version


class Adjacency:
    """What get_adj (model/diffusion.py:239-256) returns here: the row-normalised adjacency D^-1 A in CSR form with
    implicit values 1/deg[row], plus the symmetric bitmap it was built from."""

    def __init__(self, n, bits, rowptr, colidx, nnz=None):
        self.n, self.bits, self.rowptr, self.colidx = n, bits, rowptr, colidx
        self._nnz = nnz

    @property
    def shape(self):
        return (self.n, self.n)

    @property
    def nnz(self):
        """Directed non-zeros.  An adjacency built into a CsrWorkspace has not read its edge count back yet: the first
        use of this property is a device-to-host read."""
        if self._nnz is None:
            self._nnz = int(self.rowptr[self.n].item())
        return self._nnz

    def __matmul__(self, H):
        """A @ H (model/gnn.py:60-61)."""
        return spmm_mean(self, H)


# ------------------------------------------------------------------ adjacency
def bitmap_from_dense(E):
    require(E, torch.int64, "E")
    n = E.shape[0]
    wpr = words_per_row(n)
    bits = torch.empty((n, wpr), dtype=torch.int32, device=E.device)
    L.call("gm_bitmap_from_dense_i64", I64(n), I64(wpr), ptr(E), ptr(bits), stream_ptr())
    return bits


def dense_from_bitmap(bits, n):
    require(bits, torch.int32, "bits")
    E = torch.empty((n, n), dtype=torch.int64, device=bits.device)
    L.call("gm_dense_i64_from_bitmap", I64(n), I64(bits.shape[1]), ptr(bits), ptr(E), stream_ptr())
    return E


def symmetrize_upper(upper, n, out=None):
    require(upper, torch.int32, "upper")
    sym = out if out is not None else torch.empty_like(upper)
    L.call("gm_symmetrize_upper", I64(n), I64(upper.shape[1]), ptr(upper), ptr(sym), stream_ptr())
    return sym


def csr_from_bitmap(bits, n, upper_only=False, want_rows=False):
    """(rowptr, colidx[, rows]) int32.  One host read of rowptr[n] sizes colidx."""
    require(bits, torch.int32, "bits")
    wpr = bits.shape[1]
    rowptr = torch.empty(n + 1, dtype=torch.int32, device=bits.device)
    L.call("gm_csr_rowptr", I64(n), I64(wpr), ptr(bits), I32(int(upper_only)), ptr(rowptr), stream_ptr())
    nnz = int(rowptr[n].item())
    colidx = torch.empty(max(nnz, 1), dtype=torch.int32, device=bits.device)
    rows = torch.empty(max(nnz, 1), dtype=torch.int32, device=bits.device) if want_rows else None
    L.call("gm_csr_fill", I64(n), I64(wpr), ptr(bits), I32(int(upper_only)), ptr(rowptr), I64(nnz), ptr(colidx),
           ptr(rows), stream_ptr())
    if want_rows:
        return rowptr, colidx[:nnz], rows[:nnz]
    return rowptr, colidx[:nnz]


class CsrWorkspace:
    """A colidx buffer of fixed capacity that the reverse steps of one model build their adjacency into, so that no step
    has to read the edge count back to size an allocation (gm_csr_rowptr_capped).  Up to 16384 nodes the capacity is n*n
    (no graph can overflow it); beyond, 2^28 entries (1 GiB), and `overflow` tells after the fact that a denser graph was
    truncated -- check() raises then.  One adjacency at a time lives in it: callers that keep several use the exact path."""

    def __init__(self, n, device):
        self.capacity = int(min(n * n, 1 << 28))
        self.colidx = torch.empty(self.capacity, dtype=torch.int32, device=device)
        self.overflow = torch.zeros(1, dtype=torch.int32, device=device)
        self.n, self.device = n, device

    def check(self):
        if int(self.overflow.item()):
            self.overflow.zero_()
            raise RuntimeError(f"a sampled graph has more than {self.capacity} directed edges: the step-local adjacency was "
                               "truncated and the sample is invalid (graphs this dense need the exact-size CSR path)")


def adjacency_from_bitmap(bits, n, workspace=None, rows=None):
    """get_adj on the library's bitmap state.  workspace: a CsrWorkspace -> no host synchronisation (the step path).
    rows=(r0, r1) (with a workspace): only those rows of the adjacency are built -- rowptr[r0..r1] counts from 0 at r0 and
    colidx holds their neighbours: what a rank of the multi-GPU step aggregates (its node block); other rows are undefined."""
    if rows is not None and workspace is not None:
        r0, r1 = rows
        require(bits, torch.int32, "bits")
        wpr = bits.shape[1]
        rowptr = torch.empty(n + 1, dtype=torch.int32, device=bits.device)
        if r1 > r0:
            L.call("gm_csr_rowptr_capped", I64(r1 - r0), I64(wpr), ptr(bits[r0:r1]), I32(0), I64(workspace.capacity),
                   ptr(rowptr[r0:r1 + 1]), ptr(workspace.overflow), stream_ptr())
            L.call("gm_csr_fill", I64(r1 - r0), I64(wpr), ptr(bits[r0:r1]), I32(0), ptr(rowptr[r0:r1 + 1]),
                   I64(workspace.capacity), ptr(workspace.colidx), None, stream_ptr())
        return Adjacency(n, bits, rowptr, workspace.colidx, nnz=-1)
    if workspace is None:
        rowptr, colidx = csr_from_bitmap(bits, n)
        return Adjacency(n, bits, rowptr, colidx, nnz=int(colidx.numel()))
    require(bits, torch.int32, "bits")
    wpr = bits.shape[1]
    rowptr = torch.empty(n + 1, dtype=torch.int32, device=bits.device)
    L.call("gm_csr_rowptr_capped", I64(n), I64(wpr), ptr(bits), I32(0), I64(workspace.capacity), ptr(rowptr),
           ptr(workspace.overflow), stream_ptr())
    L.call("gm_csr_fill", I64(n), I64(wpr), ptr(bits), I32(0), ptr(rowptr), I64(workspace.capacity), ptr(workspace.colidx),
           None, stream_ptr())
    return Adjacency(n, bits, rowptr, workspace.colidx)


def upper_edges_of_rows(upper, n, row_ranges, segment, cap, rowptr_ws, overflow):
    """gm_upper_edges_ranges into a segment [ei (cap) | ej (cap) | count | overflowed | pad] (int32, 2*cap + 16 entries): the
    sampled edges (i<j) of `row_ranges` of the upper-triangle bitmap, sorted; count and the overflow flag stay on the device
    (and travel inside the segment)."""
    require(upper, torch.int32, "upper")
    require(segment, torch.int32, "segment")
    flat = [int(v) for be in row_ranges for v in be]
    ranges_host = (C.c_int64 * len(flat))(*flat)
    L.call("gm_upper_edges_ranges", I64(n), I64(upper.shape[1]), ptr(upper), I32(len(flat) // 2), ranges_host, I64(cap),
           ptr(rowptr_ws), ptr(segment), ptr(segment[cap:]), ptr(segment[2 * cap:]), ptr(overflow), stream_ptr())
    return segment


def bitmap_from_edge_segments(n, segments, cap, out=None, overflow=None):
    """The symmetric bitmap of the edges held in `segments` (nseg, 2*cap + 16) int32 (upper_edges_of_rows layout).
    overflow: this rank's sticky (1,) int32 flag; set when ANY gathered segment was truncated on its rank."""
    require(segments, torch.int32, "segments")
    wpr = words_per_row(n)
    bits = out if out is not None else torch.empty((n, wpr), dtype=torch.int32, device=segments.device)
    L.call("gm_bitmap_from_edge_segments", I64(n), I64(wpr), I32(segments.shape[0]), I64(cap), I64(segments.shape[1]),
           ptr(segments), ptr(bits), ptr(overflow), stream_ptr())
    return bits


def bitmap_from_upper_edges(n, ei, ej):
    require(ei, torch.int32, "ei")
    require(ej, torch.int32, "ej")
    wpr = words_per_row(n)
    bits = torch.empty((n, wpr), dtype=torch.int32, device=ei.device)
    L.call("gm_bitmap_from_upper_edges", I64(n), I64(wpr), I64(ei.numel()), ptr(ei), ptr(ej), ptr(bits), stream_ptr())
    return bits


# ------------------------------------------------------------------ building blocks
def spmm_mean(adj, H):
    require(H, torch.float32, "H")
    n, d = H.shape
    out = torch.empty_like(H)
    L.call("gm_spmm_mean", I64(n), ptr(adj.rowptr), ptr(adj.colidx), ptr(H), I64(d), I64(d), ptr(out), I64(d),
           stream_ptr())
    return out


def linear(A, W, bias=None, relu=False):
    require(A, torch.float32, "A")
    require(W, torch.float32, "W")
    M, K = A.shape
    N = W.shape[0]
    if W.shape[1] != K:
        raise ValueError(f"linear: A is {tuple(A.shape)}, W is {tuple(W.shape)}")
    out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    L.call("gm_linear", I64(M), I64(N), I64(K), ptr(A), I64(K), ptr(W), I64(K), ptr(bias), ptr(out), I64(N),
           I32(int(relu)), stream_ptr())
    return out


def onehot_linear(x_u8, W, bias, relu=False, sparse=False):
    """relu?(X_one_hot @ W^T + b) for the one-hot matrix of the binary attributes x_u8 (n, F); W is (H, 2F)."""
    require(x_u8, torch.uint8, "x")
    require(W, torch.float32, "W")
    n, F = x_u8.shape
    H = W.shape[0]
    if W.shape[1] != 2 * F:
        raise ValueError(f"onehot_linear: x is {tuple(x_u8.shape)}, W is {tuple(W.shape)}")
    dev = x_u8.device
    prep = torch.empty(L.load().gm_onehot_prep_floats(I64(F), I64(H)), dtype=torch.float32, device=dev)
    L.call("gm_onehot_prepare", I64(F), I64(H), ptr(W), ptr(bias), ptr(prep), stream_ptr())
    out = torch.empty((n, H), dtype=torch.float32, device=dev)
    dense = None if sparse else torch.empty((n, (F + 3) // 4 * 4), dtype=torch.float32, device=dev)
    L.call("gm_onehot_linear", I64(n), I64(F), I64(H), ptr(x_u8), ptr(prep), ptr(dense), ptr(out), I64(H), I32(int(relu)),
           stream_ptr())
    return out


def layernorm_(X, gamma, beta, eps=1e-5):
    require(X, torch.float32, "X")
    M, N = X.shape
    L.call("gm_layernorm", I64(M), I64(N), ptr(X), I64(N), ptr(gamma), ptr(beta), F32(eps), stream_ptr())
    return X


def tower_workspace_bytes(dims, n):
    nbytes = C.c_size_t()
    L.call("gm_tower_workspace_bytes", C.byref(dims), I64(n), C.byref(nbytes))
    return nbytes.value


def tower_prepare(dims, param_ptrs, device):
    """The weight-only part of a tower's first layer (gm_tower_prepare): B, D and D^T of include/graphmaker_hip.h."""
    nfl = L.load().gm_tower_prep_floats(C.byref(dims))
    if nfl <= 0:
        raise ValueError("bad tower dims")
    prep = torch.empty(nfl, dtype=torch.float32, device=device)
    L.call("gm_tower_prepare", C.byref(dims), param_ptrs, ptr(prep), stream_ptr())
    return prep


def tower_forward(dims, param_ptrs, n, t_float, X_one_hot, Y32, adj, workspace, out=None, want_hcat=False, rows=None, *,
                  prep, x_u8=None, sparse_X=False):
    """X_one_hot=None and x_u8=None: reuse mlp_in_X(X) left in `workspace` by the previous call (same attributes, see the
    header).  x_u8 (n, F) uint8 takes precedence over X_one_hot (n, 2F)."""
    require(t_float, torch.float32, "t_float")
    require(Y32, torch.int32, "Y")
    require(prep, torch.float32, "prep")
    if x_u8 is not None:
        require(x_u8, torch.uint8, "x")
        if x_u8.shape != (n, dims.in_X // 2):
            raise ValueError(f"x is {tuple(x_u8.shape)}, expected {(n, dims.in_X // 2)}")
        X_one_hot = None
    if X_one_hot is not None:
        require(X_one_hot, torch.float32, "X_t_one_hot")
        if X_one_hot.shape != (n, dims.in_X):
            raise ValueError(f"X_t_one_hot is {tuple(X_one_hot.shape)}, expected {(n, dims.in_X)}")
    dev = Y32.device
    check_index_range(Y32, dims.num_classes_Y, "Y")
    if out is None:
        out = torch.empty((n, dims.out_size), dtype=torch.float32, device=dev)
    hcat = torch.empty((n, dims.hidden_cat), dtype=torch.float32, device=dev) if want_hcat else None
    rowptr = adj.rowptr if adj is not None else None
    colidx = adj.colidx if adj is not None else None
    L.call("gm_tower_forward", C.byref(dims), param_ptrs, I64(n), ptr(t_float), ptr(X_one_hot), ptr(x_u8), ptr(prep),
           I32(int(sparse_X)), ptr(Y32), ptr(rowptr), ptr(colidx), ptr(out), ptr(hcat), I64(rows[0] if rows else 0),
           I64(rows[1] if rows else n), ptr(workspace), C.c_size_t(workspace.numel()), stream_ptr())
    return (out, hcat) if want_hcat else out


def tower_stage(dims, param_ptrs, n, stage, rows, workspace, *, t_float=None, X_one_hot=None, Y32=None, adj=None,
                layer_src=None, layer_dst=None, out=None, prep=None, x_u8=None, sparse_X=False):
    """One row-sharded stage of a tower (gm_tower_stage): -1 input projections, 0..L-1 layers, L output MLP."""
    rowptr = adj.rowptr if adj is not None else None
    colidx = adj.colidx if adj is not None else None
    if stage == -1 and Y32 is not None:
        check_index_range(Y32, dims.num_classes_Y, "Y")
    if x_u8 is not None:
        X_one_hot = None
    L.call("gm_tower_stage", C.byref(dims), param_ptrs, I64(n), I32(stage), I64(rows[0]), I64(rows[1]), ptr(t_float),
           ptr(X_one_hot), ptr(x_u8), ptr(prep), I32(int(sparse_X)), ptr(Y32), ptr(rowptr), ptr(colidx), ptr(layer_src),
           ptr(layer_dst), ptr(out), ptr(workspace), C.c_size_t(workspace.numel()), stream_ptr())


def hadamard_gather(h, src, dst):
    require(h, torch.float32, "h")
    require(src, torch.int64, "src")
    require(dst, torch.int64, "dst")
    B, d = src.numel(), h.shape[1]
    check_index_range(src, h.shape[0], "src")
    check_index_range(dst, h.shape[0], "dst")
    out = torch.empty((B, d), dtype=torch.float32, device=h.device)
    L.call("gm_hadamard_gather", I64(B), I64(d), ptr(src), ptr(dst), ptr(h), I64(d), ptr(out), I64(d), stream_ptr())
    return out


# ------------------------------------------------------------------ structure denoiser
def pair_workspace(device):
    """Workspace of gm_pair_score_sample*: tile counter + the sticky trouble flag (include/graphmaker_hip.h)."""
    return torch.zeros(256, dtype=torch.uint8, device=device)


def check_pair_workspace(workspace):
    """Host read (synchronises): raise if a pair kernel that used this workspace reported a timed-out hand-shake."""
    word = workspace[64:68].view(torch.int32)
    if int(word.item()) != 0:
        word.zero_()          # a launch that finds the word set does nothing: clear it so the workspace can be used again
        raise RuntimeError("gm_pair_score_sample: a wave timed out waiting for an LDS hand-shake; the sampled "
                           "adjacency of this trajectory is void (see include/graphmaker_hip.h)")


PAIR_PRECISIONS = {"f32": 0, "bf16x3": 3, "bf16": 1}


def pair_score_sample(h, W1, b1, W2, b2, bits_t, tab, *, seed=0, step=0, u=None, row_begin=0, row_end=None,
                      row_ranges=None, out_upper=None, workspace=None, want_debug=False, precision="f32"):
    """All pairs i<j of rows [row_begin,row_end) -- or of up to four ascending, disjoint `row_ranges` [(b0,e0),...] in
    one launch: returns the upper-triangle bitmap of E_s (and (P,2) logits / posteriors when want_debug).
    precision: "f32" (exact, the default) or the opt-in "bf16x3" / "bf16" matrix-core paths (include/graphmaker_hip.h)."""
    if precision not in PAIR_PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(PAIR_PRECISIONS)}, got {precision!r}")
    require(h, torch.float32, "h")
    require(bits_t, torch.int32, "bits_t")
    n, hE = h.shape
    wpr = bits_t.shape[1]
    row_end = n if row_end is None else row_end
    dev = h.device
    if out_upper is None:
        out_upper = torch.zeros((n, wpr), dtype=torch.int32, device=dev)
    own_workspace = workspace is None
    if own_workspace:
        workspace = pair_workspace(dev)
    P = n * (n - 1) // 2
    lg = torch.zeros((P, 2), dtype=torch.float32, device=dev) if want_debug else None
    po = torch.zeros((P, 2), dtype=torch.float32, device=dev) if want_debug else None
    tab_host = (C.c_float * 8)(*[float(x) for x in tab])
    if u is not None:
        require(u, torch.float32, "u")
        if u.numel() != P:
            raise ValueError(f"u has {u.numel()} entries, expected P={P}")
    if row_ranges is None:
        row_ranges = [(row_begin, row_end)]
    flat = [int(v) for be in row_ranges for v in be]
    ranges_host = (C.c_int64 * len(flat))(*flat)
    if precision == "f32":
        L.call("gm_pair_score_sample_ranges", I64(n), I64(hE), ptr(h), ptr(W1), ptr(b1), ptr(W2), ptr(b2), ptr(bits_t),
               I64(wpr), tab_host, I32(len(flat) // 2), ranges_host, ptr(u), U64(seed), U32(step), ptr(out_upper), ptr(lg),
               ptr(po), ptr(workspace), stream_ptr())
    else:
        L.call("gm_pair_score_sample_bf16", I64(n), I64(hE), ptr(h), ptr(W1), ptr(b1), ptr(W2), ptr(b2), ptr(bits_t),
               I64(wpr), tab_host, I32(len(flat) // 2), ranges_host, ptr(u), U64(seed), U32(step), ptr(out_upper), ptr(lg),
               ptr(po), ptr(workspace), I32(PAIR_PRECISIONS[precision]), stream_ptr())
    if own_workspace:
        check_pair_workspace(workspace)
    return (out_upper, lg, po) if want_debug else out_upper


# ------------------------------------------------------------------ attributes
def head_diff(W, b):
    """(Wd, bd) of gm_head_diff_prepare: the two-class head mlp_out.2 folded to one logit difference per attribute."""
    require(W, torch.float32, "W")
    require(b, torch.float32, "b")
    F2, Hc = W.shape
    if F2 % 2 or b.numel() != F2:
        raise ValueError(f"W is {tuple(W.shape)}, b has {b.numel()} entries: expected (2F, Hc) and 2F")
    Wd = torch.empty((F2 // 2, Hc), dtype=torch.float32, device=W.device)
    bd = torch.empty(F2 // 2, dtype=torch.float32, device=W.device)
    L.call("gm_head_diff_prepare", I64(F2 // 2), I64(Hc), ptr(W), ptr(b), ptr(Wd), ptr(bd), stream_ptr())
    return Wd, bd


def attr_posterior_sample(logits, x_t, tab, *, seed=0, step=0, u=None, want_onehot=True, want_debug=False, rows=None,
                          x_s=None, diff=False):
    """diff=False: logits (n, 2F) / (n, F, 2).  diff=True: logits is the (n, F) difference l1 - l0 of a tower run with the
    half-width head (gm_attr_posterior_sample_diff)."""
    require(logits, torch.float32, "logits")
    require(x_t, torch.uint8, "x_t")
    require(tab, torch.float32, "tab")
    n, F = x_t.shape
    if logits.numel() != (1 if diff else 2) * n * F:
        raise ValueError(f"logits has {logits.numel()} entries, expected {(1 if diff else 2) * n * F} (diff={diff})")
    dev = x_t.device
    x_s = torch.empty_like(x_t) if x_s is None else x_s
    oh = torch.empty((n, 2 * F), dtype=torch.float32, device=dev) if want_onehot else None
    po = torch.empty((n, F, 2), dtype=torch.float32, device=dev) if want_debug else None
    L.call("gm_attr_posterior_sample_diff" if diff else "gm_attr_posterior_sample", I64(n), I64(F), ptr(logits), ptr(x_t), ptr(tab), ptr(u), U64(seed), U32(step),
           I64(rows[0] if rows else 0), I64(rows[1] if rows else n), ptr(x_s), ptr(oh), ptr(po), stream_ptr())
    res = (x_s, oh)
    return res + (po,) if want_debug else res


def onehot_from_x(x):
    require(x, torch.uint8, "x")
    n, F = x.shape
    oh = torch.empty((n, 2 * F), dtype=torch.float32, device=x.device)
    L.call("gm_onehot_from_x", I64(n), I64(F), ptr(x), ptr(oh), stream_ptr())
    return oh


def x_from_onehot(oh):
    require(oh, torch.float32, "X_one_hot")
    n, F2 = oh.shape
    x = torch.empty((n, F2 // 2), dtype=torch.uint8, device=oh.device)
    L.call("gm_x_from_onehot", I64(n), I64(F2 // 2), ptr(oh), ptr(x), stream_ptr())
    return x


# ------------------------------------------------------------------ priors
def prior_Y(n, Y_marginal, *, seed=0, u=None):
    require(Y_marginal, torch.float32, "Y_marginal")
    y = torch.empty(n, dtype=torch.int32, device=Y_marginal.device)
    L.call("gm_prior_Y", I64(n), I32(Y_marginal.numel()), ptr(Y_marginal), ptr(u), U64(seed), ptr(y), stream_ptr())
    return y


def prior_X(n, X_marginal, *, seed=0, u=None):
    require(X_marginal, torch.float32, "X_marginal")
    F = X_marginal.shape[0]
    x = torch.empty((n, F), dtype=torch.uint8, device=X_marginal.device)
    oh = torch.empty((n, 2 * F), dtype=torch.float32, device=X_marginal.device)
    L.call("gm_prior_X", I64(n), I64(F), ptr(X_marginal), ptr(u), U64(seed), ptr(x), ptr(oh), stream_ptr())
    return x, oh


def prior_E(n, p0, p1, device, *, seed=0, u=None):
    wpr = words_per_row(n)
    upper = torch.empty((n, wpr), dtype=torch.int32, device=device)
    bits = torch.empty((n, wpr), dtype=torch.int32, device=device)
    L.call("gm_prior_E", I64(n), I64(wpr), F32(p0), F32(p1), ptr(u), U64(seed), ptr(upper), ptr(bits), stream_ptr())
    return bits


def prior_E_static(n, p0, p1, bits, *, seed=0, u=None):
    """gm_prior_E_static: the prior draw into a caller-owned bitmap with the library's own static scratch (no allocation
    anywhere in the call; n <= 16384)."""
    require(bits, torch.int32, "bits")
    if bits.shape != (n, words_per_row(n)):
        raise ValueError(f"bits is {tuple(bits.shape)}, expected {(n, words_per_row(n))}")
    L.call("gm_prior_E_static", I64(n), I64(bits.shape[1]), F32(p0), F32(p1), ptr(u), U64(seed), ptr(bits), stream_ptr())
    return bits


# ------------------------------------------------------------------ forward noising / training-side converters
def noise_E(n, bits0, Q_bar_t, *, seed=0, step=0, u=None):
    """q(E^t | E): symmetric bitmap of E_t (model/diffusion.py:591-594).  Q_bar_t: 4 host floats Q[r*2+c]."""
    require(bits0, torch.int32, "bits0")
    wpr = bits0.shape[1]
    upper = torch.empty_like(bits0)
    bits = torch.empty_like(bits0)
    Q = (C.c_float * 4)(*[float(q) for q in Q_bar_t])
    if u is not None:
        require(u, torch.float32, "u")
        if u.numel() != n * n:
            raise ValueError(f"u has {u.numel()} entries, expected n*n={n * n}")
    L.call("gm_noise_E", I64(n), I64(wpr), ptr(bits0), Q, ptr(u), U64(seed), U32(step), ptr(upper), ptr(bits), stream_ptr())
    return bits


def noise_X(x0, Q_bar_t_X, *, seed=0, step=0, u=None):
    """q(X^t | X): (x_t uint8 (N,F), one-hot float (N,2F)) (model/diffusion.py:596-600).  Q_bar_t_X: (F,4) device."""
    require(x0, torch.uint8, "x0")
    require(Q_bar_t_X, torch.float32, "Q_bar_t_X")
    n, F = x0.shape
    if Q_bar_t_X.numel() != 4 * F:
        raise ValueError(f"Q_bar_t_X has {Q_bar_t_X.numel()} entries, expected 4F={4 * F}")
    if u is not None:
        require(u, torch.float32, "u")
        if u.numel() != n * F:
            raise ValueError(f"u has {u.numel()} entries, expected n*F={n * F}")
    x_t = torch.empty_like(x0)
    oh = torch.empty((n, 2 * F), dtype=torch.float32, device=x0.device)
    L.call("gm_noise_X", I64(n), I64(F), ptr(x0), ptr(Q_bar_t_X), ptr(u), U64(seed), U32(step), ptr(x_t), ptr(oh), stream_ptr())
    return x_t, oh


def bitmap_from_onehot(E_one_hot):
    require(E_one_hot, torch.float32, "E_one_hot")
    n = E_one_hot.shape[0]
    if tuple(E_one_hot.shape) != (n, n, 2):
        raise ValueError(f"E_one_hot is {tuple(E_one_hot.shape)}, expected (N,N,2)")
    wpr = words_per_row(n)
    bits = torch.empty((n, wpr), dtype=torch.int32, device=E_one_hot.device)
    L.call("gm_bitmap_from_onehot_f32", I64(n), I64(wpr), ptr(E_one_hot), ptr(bits), stream_ptr())
    return bits


def x_from_onehot3d(X_one_hot_3d):
    require(X_one_hot_3d, torch.float32, "X_one_hot_3d")
    F, n, c = X_one_hot_3d.shape
    if c != 2:
        raise ValueError("binary attributes only")
    x = torch.empty((n, F), dtype=torch.uint8, device=X_one_hot_3d.device)
    L.call("gm_x_from_onehot3d", I64(n), I64(F), ptr(X_one_hot_3d), ptr(x), stream_ptr())
    return x


def bitmap_lookup(bits, n, i, j):
    require(bits, torch.int32, "bits")
    require(i, torch.int64, "i")
    require(j, torch.int64, "j")
    check_index_range(i, n, "i")
    check_index_range(j, n, "j")
    out = torch.empty(i.numel(), dtype=torch.uint8, device=bits.device)
    L.call("gm_bitmap_lookup", I64(n), I64(bits.shape[1]), ptr(bits), I64(i.numel()), ptr(i), ptr(j), ptr(out), stream_ptr())
    return out


_ce_ws = {}


def denoise_match2(logits, z_t, z_0, Q_t, Q_bar_s):
    """Denoising-match KL (BaseModel.denoise_match_Z, model/diffusion.py:258-303) -> (1,) device tensor."""
    logits = logits.contiguous()
    require(logits, torch.float32, "logits")
    require(z_t, torch.uint8, "z_t")
    require(z_0, torch.uint8, "z_0")
    total = z_t.numel()
    if logits.numel() != 2 * total or z_0.numel() != total:
        raise ValueError("denoise_match2: logits / z_t / z_0 sizes disagree")
    dev = logits.device
    if dev not in _ce_ws:
        _ce_ws[dev] = torch.empty(4096, dtype=torch.uint8, device=dev)
    ws = _ce_ws[dev]
    out = torch.empty(1, dtype=torch.float32, device=dev)
    L.call("gm_denoise_match2", I64(total), ptr(logits), ptr(z_t), ptr(z_0), ptr(Q_t), ptr(Q_bar_s), I64(Q_t.numel() // 4),
           ptr(out), ptr(ws), I64(ws.numel()), stream_ptr())
    return out
