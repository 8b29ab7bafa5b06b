"""ctypes binding of oracle/libgm_oracle.so (the CPU restatement; TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "libgm_oracle.so")

STREAM_Y, STREAM_X, STREAM_E_PRIOR, STREAM_E = 1, 2, 3, 4


def build(force=False):
    src = os.path.join(_ROOT, "oracle", "gm_oracle.c")
    if force or not os.path.exists(_SO) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_SO)):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")])
    return _SO


class TowerDims(C.Structure):
    _fields_ = [("in_X", C.c_int), ("num_classes_Y", C.c_int), ("hidden_t", C.c_int),
                ("hidden_X", C.c_int), ("hidden_Y", C.c_int), ("num_layers", C.c_int),
                ("out_size", C.c_int), ("is_mlp", C.c_int)]

    @property
    def hidden_cat(self):
        return (self.num_layers + 1) * (self.hidden_X + self.hidden_Y) + self.hidden_t


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.gmo_version.restype = C.c_uint32
        _lib.gmo_get_threads.restype = C.c_int
        _lib.gmo_u01.restype = C.c_float
        _lib.gmo_u01.argtypes = [C.c_uint32]
        _lib.gmo_csr_from_bitmap.restype = C.c_int64
        _lib.gmo_upper_edges_from_bitmap.restype = C.c_int64
    return _lib


def _p(a, ctype=None):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle wants contiguous arrays"
    return a.ctypes.data_as(C.c_void_p)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def words_per_row(n):
    return ((n + 31) // 32 + 3) // 4 * 4


def set_threads(n):
    lib().gmo_set_threads(C.c_int(n))


def get_threads():
    return lib().gmo_get_threads()


# ---------------------------------------------------------------- RNG
def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().gmo_philox4x32_10(c, k, o)
    return list(o)


def uniform_Y(seed, n):
    u = np.empty(n, np.float32)
    lib().gmo_fill_uniform_Y(C.c_uint64(seed), C.c_int64(n), _p(u))
    return u


def uniform_X(seed, step, n, F):
    u = np.empty(n * F, np.float32)
    lib().gmo_fill_uniform_X(C.c_uint64(seed), C.c_uint32(step), C.c_int64(n), C.c_int64(F), _p(u))
    return u


def uniform_E_prior(seed, n):
    u = np.empty(n * n, np.float32)
    lib().gmo_fill_uniform_E_prior(C.c_uint64(seed), C.c_int64(n), _p(u))
    return u


def uniform_E(seed, step, n):
    u = np.empty(n * (n - 1) // 2, np.float32)
    lib().gmo_fill_uniform_E(C.c_uint64(seed), C.c_uint32(step), C.c_int64(n), _p(u))
    return u


# ---------------------------------------------------------------- schedule
def noise_schedule(T):
    b = np.empty(T + 1, np.float32)
    a = np.empty(T + 1, np.float32)
    ab = np.empty(T + 1, np.float32)
    lib().gmo_noise_schedule(C.c_int(T), _p(b), _p(a), _p(ab))
    return b, a, ab


def transition(a, m0, m1):
    Q = np.empty(4, np.float32)
    lib().gmo_transition(C.c_float(a), C.c_float(m0), C.c_float(m1), _p(Q))
    return Q.reshape(2, 2)


def posterior_table(Qt, Qbs, Qbt):
    tab = np.empty(8, np.float32)
    lib().gmo_posterior_table(_p(f32(Qt)), _p(f32(Qbs)), _p(f32(Qbt)), _p(tab))
    return tab


def posterior_table_E(E_marginal, alpha_t, abar_s, abar_t):
    m = f32(E_marginal)
    return posterior_table(transition(alpha_t, m[0], m[1]), transition(abar_s, m[0], m[1]),
                           transition(abar_t, m[0], m[1]))


def posterior_tables_X(X_marginal, alpha_t, abar_s, abar_t):
    Xm = f32(X_marginal)
    F = Xm.shape[0]
    tab = np.empty((F, 8), np.float32)
    lib().gmo_posterior_tables_X(C.c_int64(F), _p(Xm), C.c_float(alpha_t), C.c_float(abar_s),
                                 C.c_float(abar_t), _p(tab))
    return tab


def posterior_apply(tab, z_t, q0, q1):
    p0, p1 = C.c_float(), C.c_float()
    lib().gmo_posterior_apply(_p(f32(tab)), C.c_int(int(z_t)), C.c_float(q0), C.c_float(q1),
                              C.byref(p0), C.byref(p1))
    return p0.value, p1.value


def softmax2(l0, l1):
    p0, p1 = C.c_float(), C.c_float()
    lib().gmo_softmax2(C.c_float(l0), C.c_float(l1), C.byref(p0), C.byref(p1))
    return p0.value, p1.value


# ---------------------------------------------------------------- priors
def prior_Y(n, Y_marginal, u=None, seed=0):
    Ym = f32(Y_marginal)
    y = np.empty(n, np.int32)
    lib().gmo_prior_Y(C.c_int64(n), C.c_int(len(Ym)), _p(Ym), _p(u), C.c_uint64(seed), _p(y))
    return y


def prior_X(n, X_marginal, u=None, seed=0):
    Xm = f32(X_marginal)
    F = Xm.shape[0]
    x = np.empty((n, F), np.uint8)
    lib().gmo_prior_X(C.c_int64(n), C.c_int64(F), _p(Xm), _p(u), C.c_uint64(seed), _p(x))
    return x


def prior_E(n, E_marginal, u=None, seed=0, wpr=None):
    wpr = wpr or words_per_row(n)
    m = f32(E_marginal)
    bits = np.empty((n, wpr), np.uint32)
    lib().gmo_prior_E(C.c_int64(n), C.c_int64(wpr), C.c_float(m[0]), C.c_float(m[1]), _p(u),
                      C.c_uint64(seed), _p(bits))
    return bits


# ---------------------------------------------------------------- forward noising
def noise_E(n, bits0, Q_bar_t, u=None, seed=0, step=0):
    bits0 = np.ascontiguousarray(bits0, dtype=np.uint32)
    Q = f32(np.asarray(Q_bar_t).reshape(4))
    bits = np.empty_like(bits0)
    lib().gmo_noise_E(C.c_int64(n), C.c_int64(bits0.shape[1]), _p(bits0), _p(Q), _p(u), C.c_uint64(seed),
                      C.c_uint32(step), _p(bits))
    return bits


def noise_X(x0, Q_bar_t_X, u=None, seed=0, step=0):
    x0 = np.ascontiguousarray(x0, dtype=np.uint8)
    n, F = x0.shape
    Q = f32(np.asarray(Q_bar_t_X).reshape(F, 4))
    x_t = np.empty_like(x0)
    lib().gmo_noise_X(C.c_int64(n), C.c_int64(F), _p(x0), _p(Q), _p(u), C.c_uint64(seed), C.c_uint32(step), _p(x_t))
    return x_t


# ---------------------------------------------------------------- adjacency
def bitmap_from_dense(dense, wpr=None):
    d = np.ascontiguousarray(dense, dtype=np.int64)
    n = d.shape[0]
    wpr = wpr or words_per_row(n)
    bits = np.empty((n, wpr), np.uint32)
    lib().gmo_bitmap_from_dense_i64(C.c_int64(n), C.c_int64(wpr), _p(d), _p(bits))
    return bits


def dense_from_bitmap(bits, n):
    bits = np.ascontiguousarray(bits, dtype=np.uint32)
    d = np.empty((n, n), np.int64)
    lib().gmo_dense_i64_from_bitmap(C.c_int64(n), C.c_int64(bits.shape[1]), _p(bits), _p(d))
    return d


def symmetrize_upper(upper, n):
    upper = np.ascontiguousarray(upper, dtype=np.uint32)
    sym = np.empty_like(upper)
    lib().gmo_symmetrize_upper(C.c_int64(n), C.c_int64(upper.shape[1]), _p(upper), _p(sym))
    return sym


def csr_from_bitmap(bits, n):
    bits = np.ascontiguousarray(bits, dtype=np.uint32)
    rowptr = np.empty(n + 1, np.int32)
    nnz = lib().gmo_csr_from_bitmap(C.c_int64(n), C.c_int64(bits.shape[1]), _p(bits), _p(rowptr), None)
    colidx = np.empty(max(int(nnz), 1), np.int32)
    lib().gmo_csr_from_bitmap(C.c_int64(n), C.c_int64(bits.shape[1]), _p(bits), _p(rowptr), _p(colidx))
    return rowptr, colidx[:nnz]


def upper_edges_from_bitmap(bits, n):
    bits = np.ascontiguousarray(bits, dtype=np.uint32)
    m = lib().gmo_upper_edges_from_bitmap(C.c_int64(n), C.c_int64(bits.shape[1]), _p(bits), None, None)
    ei = np.empty(max(int(m), 1), np.int32)
    ej = np.empty(max(int(m), 1), np.int32)
    lib().gmo_upper_edges_from_bitmap(C.c_int64(n), C.c_int64(bits.shape[1]), _p(bits), _p(ei), _p(ej))
    return ei[:m], ej[:m]


def bitmap_from_upper_edges(n, ei, ej, wpr=None):
    wpr = wpr or words_per_row(n)
    ei = np.ascontiguousarray(ei, np.int32)
    ej = np.ascontiguousarray(ej, np.int32)
    bits = np.empty((n, wpr), np.uint32)
    lib().gmo_bitmap_from_upper_edges(C.c_int64(n), C.c_int64(wpr), C.c_int64(len(ei)), _p(ei), _p(ej),
                                      _p(bits))
    return bits


# ---------------------------------------------------------------- building blocks
def spmm_mean(rowptr, colidx, H):
    H = f32(H)
    n, d = H.shape
    out = np.empty((n, d), np.float32)
    lib().gmo_spmm_mean(C.c_int64(n), _p(np.ascontiguousarray(rowptr, np.int32)),
                        _p(np.ascontiguousarray(colidx, np.int32)), _p(H), C.c_int64(d), C.c_int64(d),
                        _p(out), C.c_int64(d))
    return out


def linear(A, W, bias=None, relu=False):
    A, W = f32(A), f32(W)
    M, K = A.shape
    N = W.shape[0]
    b = f32(bias) if bias is not None else None
    out = np.empty((M, N), np.float32)
    lib().gmo_linear(C.c_int64(M), C.c_int64(N), C.c_int64(K), _p(A), C.c_int64(K), _p(W), C.c_int64(K),
                     _p(b), _p(out), C.c_int64(N), C.c_int(int(relu)))
    return out


def onehot_linear(X_one_hot, W, bias=None, relu=False):
    X, W = f32(X_one_hot), f32(W)
    n, F2 = X.shape
    H = W.shape[0]
    b = f32(bias) if bias is not None else None
    out = np.empty((n, H), np.float32)
    lib().gmo_onehot_linear(C.c_int64(n), C.c_int64(F2 // 2), C.c_int64(H), _p(X), _p(W), _p(b), _p(out), C.c_int64(H),
                            C.c_int(int(relu)))
    return out


def layernorm(X, gamma, beta, eps=1e-5):
    X = f32(X).copy()
    M, N = X.shape
    lib().gmo_layernorm(C.c_int64(M), C.c_int64(N), _p(X), C.c_int64(N), _p(f32(gamma)), _p(f32(beta)),
                        C.c_float(eps))
    return X


# ---------------------------------------------------------------- towers
class Tower:
    """Holds dims + parameter arrays in state_dict order (see gm_oracle.h)."""

    def __init__(self, dims, params):
        self.dims = dims
        self.params = [f32(p) for p in params]
        n = 9 + 8 * dims.num_layers + 4
        assert len(self.params) == n, (len(self.params), n)
        self._ptrs = (C.c_void_p * n)(*[p.ctypes.data for p in self.params])

    @staticmethod
    def from_state_dict(sd, is_mlp, prefix=""):
        """sd: mapping name -> array, reference key names."""
        g = lambda k: np.asarray(sd[prefix + k], dtype=np.float32)
        lay = "mlp_layers" if is_mlp else "gnn_layers"
        L = 0
        while f"{prefix}{lay}.{L}.update_X.0.weight" in sd:
            L += 1
        names = ["mlp_in_t.0.weight", "mlp_in_t.0.bias", "mlp_in_t.2.weight", "mlp_in_t.2.bias",
                 "mlp_in_X.0.weight", "mlp_in_X.0.bias", "mlp_in_X.2.weight", "mlp_in_X.2.bias",
                 "emb_Y.weight"]
        for l in range(L):
            for part in ("update_X", "update_Y"):
                names += [f"{lay}.{l}.{part}.0.weight", f"{lay}.{l}.{part}.0.bias",
                          f"{lay}.{l}.{part}.2.weight", f"{lay}.{l}.{part}.2.bias"]
        names += ["mlp_out.0.weight", "mlp_out.0.bias", "mlp_out.2.weight", "mlp_out.2.bias"]
        params = [g(k) for k in names]
        dims = TowerDims(in_X=params[4].shape[1], num_classes_Y=params[8].shape[0],
                         hidden_t=params[0].shape[0], hidden_X=params[4].shape[0],
                         hidden_Y=params[8].shape[1], num_layers=L,
                         out_size=params[-2].shape[0], is_mlp=int(is_mlp))
        return Tower(dims, params)

    def diff_head(self):
        """This tower with mlp_out.2 folded to the logit difference l1 - l0 per attribute (gmo_head_diff): the node tower
        as the sampler runs it (out_size F instead of 2F)."""
        if getattr(self, "_diff", None) is None:
            d = self.dims
            F, Hc = d.out_size // 2, d.hidden_cat
            Wd, bd = np.empty((F, Hc), np.float32), np.empty(F, np.float32)
            lib().gmo_head_diff(C.c_int64(F), C.c_int64(Hc), _p(self.params[-2]), _p(self.params[-1]), _p(Wd), _p(bd))
            dims = TowerDims(in_X=d.in_X, num_classes_Y=d.num_classes_Y, hidden_t=d.hidden_t, hidden_X=d.hidden_X,
                             hidden_Y=d.hidden_Y, num_layers=d.num_layers, out_size=F, is_mlp=d.is_mlp)
            self._diff = Tower(dims, self.params[:-2] + [Wd, bd])
        return self._diff

    def forward(self, t_float, X_one_hot, Y, rowptr=None, colidx=None, want_hcat=False, diff_head=False):
        if diff_head:
            return self.diff_head().forward(t_float, X_one_hot, Y, rowptr, colidx, want_hcat)
        X = f32(X_one_hot)
        n = X.shape[0]
        Y = np.ascontiguousarray(Y, np.int32)
        out = np.empty((n, self.dims.out_size), np.float32)
        hcat = np.empty((n, self.dims.hidden_cat), np.float32) if want_hcat else None
        rp = np.ascontiguousarray(rowptr, np.int32) if rowptr is not None else None
        ci = np.ascontiguousarray(colidx, np.int32) if colidx is not None else None
        lib().gmo_tower_forward(C.byref(self.dims), self._ptrs, C.c_int64(n), C.c_float(t_float), _p(X),
                                _p(Y), _p(rp), _p(ci), _p(out), _p(hcat))
        return (out, hcat) if want_hcat else out


class PairMLP:
    def __init__(self, W1, b1, W2, b2):
        self.params = [f32(W1), f32(b1), f32(W2), f32(b2)]
        self._ptrs = (C.c_void_p * 4)(*[p.ctypes.data for p in self.params])
        self.hE = self.params[0].shape[0]


def pair_score_sample(n, h, pair, bits_t, tab, u=None, seed=0, step=0, row_begin=0, row_end=None,
                      want_debug=False):
    h = f32(h)
    bits_t = np.ascontiguousarray(bits_t, np.uint32)
    wpr = bits_t.shape[1]
    row_end = n if row_end is None else row_end
    out = np.zeros((n, wpr), np.uint32)
    P = n * (n - 1) // 2
    lg = np.zeros((P, 2), np.float32) if want_debug else None
    po = np.zeros((P, 2), np.float32) if want_debug else None
    W1, b1, W2, b2 = pair.params
    lib().gmo_pair_score_sample(C.c_int64(n), C.c_int64(pair.hE), _p(h), _p(W1), _p(b1), _p(W2), _p(b2),
                                _p(bits_t), C.c_int64(wpr), _p(f32(tab)), C.c_int64(row_begin),
                                C.c_int64(row_end), _p(u), C.c_uint64(seed), C.c_uint32(step), _p(out),
                                _p(lg), _p(po))
    return (out, lg, po) if want_debug else out


def attr_posterior_sample(logits, x_t, tab, u=None, seed=0, step=0, want_debug=False, diff=False):
    """diff: `logits` is the (n, F) difference l1 - l0 of a tower run with diff_head (gmo_attr_posterior_sample_diff)."""
    x_t = np.ascontiguousarray(x_t, np.uint8)
    n, F = x_t.shape
    logits = f32(logits).reshape((n, F) if diff else (n, F, 2))
    x_s = np.empty((n, F), np.uint8)
    po = np.empty((n, F, 2), np.float32) if want_debug else None
    fn = lib().gmo_attr_posterior_sample_diff if diff else lib().gmo_attr_posterior_sample
    fn(C.c_int64(n), C.c_int64(F), _p(logits), _p(x_t), _p(f32(tab)), _p(u), C.c_uint64(seed), C.c_uint32(step), _p(x_s),
       _p(po))
    return (x_s, po) if want_debug else x_s


def onehot_from_x(x):
    x = np.ascontiguousarray(x, np.uint8)
    n, F = x.shape
    oh = np.empty((n, 2 * F), np.float32)
    lib().gmo_onehot_from_x(C.c_int64(n), C.c_int64(F), _p(x), _p(oh))
    return oh


def sample_sync(T, n, X_marginal, Y_marginal, E_marginal, towerX, towerE, pair, seed, as_reference=False, sched=None):
    Xm, Ym, Em = f32(X_marginal), f32(Y_marginal), f32(E_marginal)
    F = Xm.shape[0]
    wpr = words_per_row(n)
    x = np.empty((n, F), np.uint8)
    y = np.empty(n, np.int32)
    bits = np.empty((n, wpr), np.uint32)
    lib().gmo_sample_sync(C.c_int(T), C.c_int64(n), C.c_int64(F), C.c_int(len(Ym)), _p(Xm), _p(Ym), _p(Em),
                          C.byref(towerX.dims), towerX._ptrs, C.byref(towerE.dims), towerE._ptrs,
                          pair._ptrs, C.c_uint64(seed), C.c_int(int(as_reference)), _p(x), _p(y), _p(bits),
                          C.c_int64(wpr), _p(_sched(sched)))
    return x, y, bits


def _sched(s):
    """(alphas, alpha_bars) -> one contiguous float32 array, or None."""
    if s is None:
        return None
    return np.ascontiguousarray(np.concatenate([f32(s[0]), f32(s[1])]), np.float32)


def sample_async(T_X, T_E, n, X_marginal, Y_marginal, E_marginal, towerX, towerE, pair, seed,
                 as_reference=False, schedX=None, schedE=None):
    Xm, Ym, Em = f32(X_marginal), f32(Y_marginal), f32(E_marginal)
    F = Xm.shape[0]
    wpr = words_per_row(n)
    x = np.empty((n, F), np.uint8)
    y = np.empty(n, np.int32)
    bits = np.empty((n, wpr), np.uint32)
    lib().gmo_sample_async(C.c_int(T_X), C.c_int(T_E), C.c_int64(n), C.c_int64(F), C.c_int(len(Ym)), _p(Xm),
                           _p(Ym), _p(Em), C.byref(towerX.dims), towerX._ptrs, C.byref(towerE.dims),
                           towerE._ptrs, pair._ptrs, C.c_uint64(seed), C.c_int(int(as_reference)), _p(x),
                           _p(y), _p(bits), C.c_int64(wpr), _p(_sched(schedX)), _p(_sched(schedE)))
    return x, y, bits
