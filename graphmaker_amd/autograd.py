"""The tape of the training step (reference: log_p_t, model/diffusion.py:606-654, :1035-1094, driven by
train_sync.py:94-120 / train_async.py).  torch.autograd only records the operator graph; every forward and every
gradient below is one or a few calls into libgraphmaker_hip.so (gm_train.hip, gm_gemm.hip, gm_nodeops.hip).  All
reductions run in a fixed order, so a step is reproducible run to run.

Operators (each a torch.autograd.Function over contiguous float32 device tensors):
    linear(x, W, b, relu)            nn.Linear (+ReLU)                         model/gnn.py:116-126,138-142,226-230
    update_block(x, W, b, g, beta)   Linear -> ReLU -> LayerNorm               model/gnn.py:27-38
    spmm_mean(adj, H)                A @ H with A = D^-1 A                     model/gnn.py:60-61
    embedding(weight, idx)           nn.Embedding                              model/gnn.py:155
    hadamard_gather(h, src, dst)     h[src] * h[dst]                           model/gnn.py:246
    softmax_ce2(logits, target)      F.cross_entropy, 2 classes, mean          model/diffusion.py:139,513
    dropout(x, p, seed, call)        nn.Dropout in training mode               model/gnn.py:31,37
"""
import ctypes

import torch

from . import _lib as L
from . import ops
from ._lib import F32, I32, I64, U32, U64, ptr, require, stream_ptr

_scratch_bufs = {}


def _scratch(nbytes, device, slot=0):
    """Device scratch for split-K / block partials.  One buffer per (device, slot), grown on demand; kernels that use
    it run back to back on the current stream, so sharing is safe."""
    key = (device, slot)
    buf = _scratch_bufs.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _scratch_bufs[key] = buf
    return buf


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def C_void(t, offset_elems):
    """Pointer to element `offset_elems` of a float32 tensor."""
    return ctypes.c_void_p(t.data_ptr() + 4 * offset_elems)


def gemm(a_red, b_red, M, N, K, A, lda, B, ldb, *, bias=None, mask=None, relu=False, out=None):
    """gm_gemm: C[m][n] = act(sum_k a(m,k) b(n,k) + bias[n]) (zero where mask <= 0)."""
    dev = A.device
    C_ = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=dev)
    need = L.load().gm_gemm_workspace_bytes(I64(M), I64(N), I64(K))
    ws = _scratch(need, dev) if need else None
    L.call("gm_gemm", I32(int(a_red)), I32(int(b_red)), I64(M), I64(N), I64(K), ptr(A), I64(lda), ptr(B), I64(ldb), ptr(bias),
           ptr(mask), I64(N if mask is not None else 0), I32(int(relu)), ptr(C_), I64(N), ptr(ws), I64(ws.numel() if ws is not None else 0),
           stream_ptr())
    return C_


def colsum(X):
    M, N = X.shape
    out = torch.empty(N, dtype=torch.float32, device=X.device)
    need = L.load().gm_colsum_workspace_bytes(I64(M), I64(N))
    ws = _scratch(need, X.device)
    L.call("gm_colsum", I64(M), I64(N), ptr(X), I64(N), ptr(out), ptr(ws), I64(ws.numel()), stream_ptr())
    return out


def _dy_w(g, W, M, K, N, mask=None):
    """dY W (M x K, reduced over W's N rows), zero where mask <= 0.  Wide layers over many rows go through the forward
    kernel with a transposed copy of W: gm_linear's 16-wave tile runs 13752 x 1744 x 1744 at 112 TFLOP/s where gm_gemm(0,1)
    reaches 96 (1744 x 1534: 112 vs 78), the copy costs 0.03 ms, and both walk the reduced dimension in ascending order
    from 0 -- the same bits wherever gm_gemm would not split the reduction, as on the tower shapes (tools/dx_ab.py)."""
    if M >= 4096 and N >= 1024 and K >= 1024:
        Wt = W.t().contiguous()
        out = torch.empty((M, K), dtype=torch.float32, device=g.device)
        L.call("gm_linear", I64(M), I64(K), I64(N), ptr(g), I64(N), ptr(Wt), I64(N), None, ptr(out), I64(K), I32(0), stream_ptr())
        if mask is not None:
            L.call("gm_relu_bwd", I64(out.numel()), ptr(out), ptr(mask), ptr(out), stream_ptr())
        return out
    return gemm(False, True, M, K, N, g, N, W, K, mask=mask)


def _linear_backward(x, W, g, needs):
    """(dx, dW, db) of y = x W^T + b for the gradient g of y (already masked by any ReLU)."""
    M, K = x.shape
    N = W.shape[0]
    dx = _dy_w(g, W, M, K, N) if needs[0] else None                               # dY W
    dW = gemm(True, True, N, K, M, g, N, x, K) if needs[1] else None              # dY^T X
    db = colsum(g) if needs[2] else None
    return dx, dW, db


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, relu):
        x, W = _c(x), _c(W)
        require(x, torch.float32, "x")
        require(W, torch.float32, "W")
        M, K = x.shape
        N = W.shape[0]
        if W.shape[1] != K:
            raise ValueError(f"linear: x is {tuple(x.shape)}, W is {tuple(W.shape)}")
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        L.call("gm_linear", I64(M), I64(N), I64(K), ptr(x), I64(K), ptr(W), I64(K), ptr(b), ptr(y), I64(N), I32(int(relu)),
               stream_ptr())
        ctx.relu = relu
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, W, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, W, y = ctx.saved_tensors
        g = _c(g)
        if ctx.relu:
            gm = torch.empty_like(g)
            L.call("gm_relu_bwd", I64(g.numel()), ptr(g), ptr(y), ptr(gm), stream_ptr())
            g = gm
        needs = (ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2])
        dx, dW, db = _linear_backward(x, W, g, needs)
        return dx, dW, db, None


class _Mlp2(torch.autograd.Function):
    """Linear -> ReLU -> Linear [-> ReLU] (mlp_in_X, mlp_out and the pair MLP, model/gnn.py:116-126,138-142,227-231) as one
    tape node, so that the backward applies the inner ReLU mask in the epilogue of the GEMM that produces the hidden
    gradient instead of in a pass of its own (the pair MLP's hidden gradient is batch x 128 floats: 1 GB per step)."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, relu_out, onehot_sparse=None):
        x, W1, W2 = _c(x), _c(W1), _c(W2)
        require(x, torch.float32, "x")
        M, K = x.shape
        H, N = W1.shape[0], W2.shape[0]
        if W1.shape[1] != K or W2.shape[1] != H:
            raise ValueError(f"mlp2: x is {tuple(x.shape)}, W1 is {tuple(W1.shape)}, W2 is {tuple(W2.shape)}")
        if onehot_sparse is not None:
            # x is the one-hot attribute matrix (mlp_in_X): the first layer in the arithmetic of the fused tower call
            # (gm_onehot.hip), so that the tape's forward and gm_tower_forward give the same bits; the backward is unchanged
            y1 = ops.onehot_linear(ops.x_from_onehot(x), W1, b1, relu=True, sparse=bool(onehot_sparse))
        else:
            y1 = torch.empty((M, H), dtype=torch.float32, device=x.device)
            L.call("gm_linear", I64(M), I64(H), I64(K), ptr(x), I64(K), ptr(W1), I64(K), ptr(b1), ptr(y1), I64(H), I32(1),
                   stream_ptr())
        y2 = torch.empty((M, N), dtype=torch.float32, device=x.device)
        L.call("gm_linear", I64(M), I64(N), I64(H), ptr(y1), I64(H), ptr(W2), I64(H), ptr(b2), ptr(y2), I64(N), I32(int(relu_out)),
               stream_ptr())
        ctx.relu_out = relu_out
        ctx.has_bias = (b1 is not None, b2 is not None)
        ctx.save_for_backward(x, W1, W2, y1, y2 if relu_out else None)
        return y2

    @staticmethod
    def backward(ctx, g):
        x, W1, W2, y1, y2 = ctx.saved_tensors
        g = _c(g)
        if ctx.relu_out:
            gm = torch.empty_like(g)
            L.call("gm_relu_bwd", I64(g.numel()), ptr(g), ptr(y2), ptr(gm), stream_ptr())
            g = gm
        M, K = x.shape
        H, N = W1.shape[0], W2.shape[0]
        need = ctx.needs_input_grad
        dW2 = gemm(True, True, N, H, M, g, N, y1, H) if need[3] else None
        db2 = colsum(g) if ctx.has_bias[1] and need[4] else None
        want_db1 = ctx.has_bias[0] and need[2]
        ws_bytes = L.load().gm_mlp2_hidden_grad_workspace_bytes(I64(M), I64(H), I64(N))
        if ws_bytes >= 0:      # few outputs (the pair MLP): hidden gradient and its column sums in one pass over the batch
            g1 = torch.empty((M, H), dtype=torch.float32, device=g.device)
            db1 = torch.empty(H, dtype=torch.float32, device=g.device) if want_db1 else None
            ws = _scratch(ws_bytes, g.device) if want_db1 else None
            L.call("gm_mlp2_hidden_grad", I64(M), I64(H), I64(N), ptr(g), ptr(W2), I64(H), ptr(y1), ptr(g1), ptr(db1), ptr(ws),
                   I64(ws.numel() if ws is not None else 0), stream_ptr())
            dx, dW1, _ = _linear_backward(x, W1, g1, (need[0], need[1], False))
        else:
            g1 = _dy_w(g, W2, M, H, N, mask=y1)                            # (dY W2) where the hidden unit was active
            dx, dW1, db1 = _linear_backward(x, W1, g1, (need[0], need[1], want_db1))
        return dx, dW1, db1, dW2, db2, None, None


class _UpdateBlock(torch.autograd.Function):
    """Linear -> ReLU -> LayerNorm of GNNLayer / MLPLayer (model/gnn.py:27-38)."""

    @staticmethod
    def forward(ctx, x, W, b, gamma, beta, eps):
        x, W = _c(x), _c(W)
        require(x, torch.float32, "x")
        M, K = x.shape
        N = W.shape[0]
        dev = x.device
        a = torch.empty((M, N), dtype=torch.float32, device=dev)
        L.call("gm_linear", I64(M), I64(N), I64(K), ptr(x), I64(K), ptr(W), I64(K), ptr(b), ptr(a), I64(N), I32(1), stream_ptr())
        y = torch.empty_like(a)
        mean = torch.empty(M, dtype=torch.float32, device=dev)
        rstd = torch.empty(M, dtype=torch.float32, device=dev)
        L.call("gm_layernorm_fwd", I64(M), I64(N), ptr(a), I64(N), ptr(gamma), ptr(beta), F32(eps), ptr(y), I64(N), ptr(mean),
               ptr(rstd), stream_ptr())
        ctx.save_for_backward(x, W, a, mean, rstd, gamma)
        return y

    @staticmethod
    def backward(ctx, g):
        x, W, a, mean, rstd, gamma = ctx.saved_tensors
        g = _c(g)
        M, N = a.shape
        dev = a.device
        da = torch.empty_like(a)
        dgb = torch.empty((2, N), dtype=torch.float32, device=dev)
        need = L.load().gm_layernorm_bwd_workspace_bytes(I64(M), I64(N))
        ws = _scratch(need, dev)
        L.call("gm_layernorm_bwd", I64(M), I64(N), ptr(g), I64(N), ptr(a), I64(N), ptr(mean), ptr(rstd), ptr(gamma), I32(1),
               ptr(da), I64(N), ptr(dgb), C_void(dgb, N), ptr(ws), I64(ws.numel()), stream_ptr())
        dx, dW, db = _linear_backward(x, W, da, (ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]))
        return dx, dW, db, dgb[0], dgb[1], None


class _SpmmMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, H, adj):
        H = _c(H)
        require(H, torch.float32, "H")
        n, d = H.shape
        out = torch.empty_like(H)
        L.call("gm_spmm_mean", I64(n), ptr(adj.rowptr), ptr(adj.colidx), ptr(H), I64(d), I64(d), ptr(out), I64(d), stream_ptr())
        ctx.adj = adj
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        n, d = g.shape
        adj = ctx.adj
        out = torch.empty_like(g)
        L.call("gm_spmm_mean_bwd", I64(n), ptr(adj.rowptr), ptr(adj.colidx), ptr(g), I64(d), I64(d), ptr(out), I64(d),
               stream_ptr())
        return out, None


class _Embedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, idx32):
        weight = _c(weight)
        require(weight, torch.float32, "emb.weight")
        require(idx32, torch.int32, "idx")
        n = idx32.numel()
        Cn, w = weight.shape
        ops.check_index_range(idx32, Cn, "Y")
        out = torch.empty((n, w), dtype=torch.float32, device=weight.device)
        L.call("gm_embedding_fwd", I64(n), I32(Cn), I32(w), ptr(idx32), ptr(weight), I64(w), ptr(out), I64(w), stream_ptr())
        ctx.save_for_backward(idx32)
        ctx.shape = (Cn, w)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx32,) = ctx.saved_tensors
        g = _c(g)
        Cn, w = ctx.shape
        dW = torch.empty((Cn, w), dtype=torch.float32, device=g.device)
        ws = _scratch(L.load().gm_embedding_bwd_workspace_bytes(I64(idx32.numel()), I32(Cn), I32(w)), g.device)
        L.call("gm_embedding_bwd", I64(idx32.numel()), I32(Cn), I32(w), ptr(idx32), ptr(g), I64(w), ptr(dW), I64(w), ptr(ws),
               I64(ws.numel()), stream_ptr())
        return dW, None


class _HadamardGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, src, dst):
        h = _c(h)
        require(h, torch.float32, "h")
        src, dst = _c(src), _c(dst)
        require(src, torch.int64, "src")
        require(dst, torch.int64, "dst")
        B, d = src.numel(), h.shape[1]
        ops.check_index_range(src, h.shape[0], "src")
        ops.check_index_range(dst, h.shape[0], "dst")
        out = torch.empty((B, d), dtype=torch.float32, device=h.device)
        L.call("gm_hadamard_gather", I64(B), I64(d), ptr(src), ptr(dst), ptr(h), I64(d), ptr(out), I64(d), stream_ptr())
        ctx.save_for_backward(h, src, dst)
        return out

    @staticmethod
    def backward(ctx, g):
        h, src, dst = ctx.saved_tensors
        g = _c(g)
        n, d = h.shape
        B = src.numel()
        # group the 2B (pair, role) entries by node: index plumbing (a stable sort of the ids), then one kernel
        keys = torch.cat([src, dst])
        # node ids fit 16 (or 32) bits: the radix sort then makes 2 (4) passes over the 2B keys instead of 8
        narrow = keys.to(torch.int16) if n <= 32767 else keys.to(torch.int32) if n <= 2**31 - 1 else keys
        order = torch.argsort(narrow, stable=True)
        counts = torch.bincount(keys, minlength=n)
        segptr = torch.zeros(n + 1, dtype=torch.int64, device=h.device)
        segptr[1:] = torch.cumsum(counts, 0)
        dh = torch.empty_like(h)
        L.call("gm_hadamard_gather_bwd", I64(n), I64(B), I64(d), ptr(src), ptr(dst), ptr(segptr), ptr(order), ptr(h), I64(d),
               ptr(g), I64(d), ptr(dh), I64(d), stream_ptr())
        return dh, None, None


class _SoftmaxCE2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target_u8):
        logits = _c(logits)
        require(logits, torch.float32, "logits")
        require(target_u8, torch.uint8, "target")
        total = target_u8.numel()
        if logits.numel() != 2 * total:
            raise ValueError(f"softmax_ce2: {logits.numel()} logits for {total} targets")
        dev = logits.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        dl = torch.empty_like(logits) if ctx.needs_input_grad[0] else None
        ws = _scratch(L.load().gm_softmax_ce2_workspace_bytes(), dev, slot=1)
        L.call("gm_softmax_ce2", I64(total), ptr(logits), ptr(target_u8), ptr(loss), ptr(dl), ptr(ws), I64(ws.numel()),
               stream_ptr())
        ctx.save_for_backward(dl)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None


class _CE2Diff(torch.autograd.Function):
    """Two-class cross entropy from the logit difference d = l1 - l0 (gm_ce2_diff)."""

    @staticmethod
    def forward(ctx, diff, target_u8):
        diff = _c(diff)
        require(diff, torch.float32, "diff")
        require(target_u8, torch.uint8, "target")
        total = target_u8.numel()
        if diff.numel() != total:
            raise ValueError(f"ce2_diff: {diff.numel()} differences for {total} targets")
        dev = diff.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        dd = torch.empty_like(diff) if ctx.needs_input_grad[0] else None
        ws = _scratch(L.load().gm_softmax_ce2_workspace_bytes(), dev, slot=1)
        L.call("gm_ce2_diff", I64(total), ptr(diff), ptr(target_u8), ptr(loss), ptr(dd), ptr(ws), I64(ws.numel()), stream_ptr())
        ctx.save_for_backward(dd)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dd,) = ctx.saved_tensors
        return dd * g, None


class _HeadDiff(torch.autograd.Function):
    """(Wd, bd) = (W[1::2] - W[0::2], b[1::2] - b[0::2]) of a two-class head (gm_head_diff_prepare); the gradient goes back
    as dW[1::2] = dWd, dW[0::2] = -dWd (gm_head_diff_bwd)."""

    @staticmethod
    def forward(ctx, W, b):
        W, b = _c(W), _c(b)
        Wd, bd = ops.head_diff(W, b)
        ctx.shape = tuple(W.shape)
        return Wd, bd

    @staticmethod
    def backward(ctx, gWd, gbd):
        F2, Hc = ctx.shape
        if gWd is None:
            gWd = torch.zeros((F2 // 2, Hc), dtype=torch.float32, device=gbd.device)
        gWd = _c(gWd)
        gbd = _c(gbd) if gbd is not None else None
        dW = torch.empty((F2, Hc), dtype=torch.float32, device=gWd.device)
        db = torch.empty(F2, dtype=torch.float32, device=gWd.device) if gbd is not None else None
        L.call("gm_head_diff_bwd", I64(F2 // 2), I64(Hc), ptr(gWd), ptr(gbd), ptr(dW), ptr(db), stream_ptr())
        return dW, db


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, call):
        x = _c(x)
        out = torch.empty_like(x)
        L.call("gm_dropout", I64(x.numel()), ptr(x), F32(p), U64(seed), U32(call), ptr(out), stream_ptr())
        ctx.args = (p, seed, call)
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        p, seed, call = ctx.args
        out = torch.empty_like(g)
        L.call("gm_dropout", I64(g.numel()), ptr(g), F32(p), U64(seed), U32(call), ptr(out), stream_ptr())
        return out, None, None, None


def linear(x, W, b=None, relu=False):
    return _Linear.apply(x, W, b, relu)


def mlp2(x, W1, b1, W2, b2, relu_out=False, onehot_sparse=None):
    """onehot_sparse: None = a general input; False / True = x is the towers' one-hot attribute matrix and the first layer
    runs in the dense / sparse one-hot form (include/graphmaker_hip.h, gm_onehot_linear)."""
    return _Mlp2.apply(x, W1, b1, W2, b2, relu_out, onehot_sparse)


def update_block(x, W, b, gamma, beta, eps=1e-5):
    return _UpdateBlock.apply(x, W, b, gamma, beta, eps)


def spmm_mean(adj, H):
    return _SpmmMean.apply(H, adj)


def embedding(weight, idx32):
    return _Embedding.apply(weight, idx32)


def hadamard_gather(h, src, dst):
    return _HadamardGather.apply(h, src, dst)


def softmax_ce2(logits, target_u8):
    return _SoftmaxCE2.apply(logits, target_u8)


def ce2_diff(diff, target_u8):
    return _CE2Diff.apply(diff, target_u8)


def head_diff(W, b):
    return _HeadDiff.apply(W, b)


class DropoutState:
    """Seed and call counter of the dropout masks of one model (a fresh mask per application, reproducible per seed)."""

    def __init__(self, seed=0):
        self.seed, self.calls = int(seed) & 0xFFFFFFFFFFFFFFFF, 0

    def next(self):
        self.calls = (self.calls + 1) & 0xFFFFFFFF
        return self.seed, self.calls


def dropout(x, p, state):
    if p <= 0.0:
        return x
    seed, call = state.next()
    return _Dropout.apply(x, float(p), seed, call)
