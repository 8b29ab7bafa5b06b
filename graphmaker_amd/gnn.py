"""Denoiser networks with the reference's module tree, constructor arguments, forward signatures and state_dict
keys (reference: model/gnn.py), executed by libgraphmaker_hip.so.

The nn.Module objects here only HOLD parameters (so `load_state_dict` of a reference Sync/Async checkpoint works
key for key, sample.py:64-65,78-79); every forward is a call into the HIP library.

Two execution modes with the same arithmetic: without autograd (sampling, validation) a tower is ONE fused library
call (gm_tower_forward); with autograd enabled and trainable parameters (log_p_t, train_sync.py:98-109) the same
operators are issued one by one through graphmaker_amd.autograd, whose backward halves are HIP kernels too.
"""
import ctypes as C
import weakref

import torch
import torch.nn as nn

from . import autograd as ag
from . import ops
from ._lib import TowerDims

__all__ = ["GNN", "LinkPredictor", "GNNAsymm", "GNNTower", "MLPTower"]


def _update_block(in_f, out_f, dropout):
    # Linear / ReLU / LayerNorm / Dropout, indices 0..3 as in model/gnn.py:27-38
    return nn.Sequential(nn.Linear(in_f, out_f), nn.ReLU(), nn.LayerNorm(out_f), nn.Dropout(dropout))


class _Layer(nn.Module):
    """Parameter holder for GNNLayer (model/gnn.py:6-70) / MLPLayer (model/gnn.py:336-394)."""

    def __init__(self, hidden_X, hidden_Y, hidden_t, dropout):
        super().__init__()
        self.update_X = _update_block(hidden_X + hidden_Y + hidden_t, hidden_X, dropout)
        self.update_Y = _update_block(hidden_Y, hidden_Y, dropout)


class _TowerBase(nn.Module):
    _layers_attr = None
    _is_mlp = False

    def _labels(self, Y_real):
        """Y as the contiguous int32 tensor the library reads.  An int64 label tensor (the reference's dtype) is converted ONCE
        per tensor object and version and the converted tensor is kept: the same object comes back on every call, so the
        library's index-range check (ops.check_index_range, memoised per tensor object) costs its host read once, not once
        per tower call of a training loop."""
        if Y_real.dtype == torch.int32 and Y_real.is_contiguous():
            return Y_real
        memo = getattr(self, "_label_memo", None)
        if memo is not None and memo[0]() is Y_real and memo[1] == Y_real._version:
            return memo[2]
        Y32 = Y_real.to(torch.int32).contiguous()
        self._label_memo = (weakref.ref(Y_real), Y_real._version, Y32)
        return Y32

    def _dropout(self):
        """This tower's dropout stream, re-keyed whenever torch.manual_seed() has changed the global seed."""
        base = torch.initial_seed()
        st = getattr(self, "_dropout_state", None)
        if st is None or getattr(self, "_dropout_base", None) != base:
            st = self._dropout_state = ag.DropoutState(base + 0x9E3779B97F4A7C15 * self._tower_id)
            self._dropout_base = base
        return st

    def _build(self, num_attrs_X, num_classes_X, num_classes_Y, hidden_t, hidden_X, hidden_Y, out_size, num_layers,
               dropout):
        in_X = num_attrs_X * num_classes_X
        self.num_attrs_X = num_attrs_X
        self.num_classes_X = num_classes_X
        self.mlp_in_t = nn.Sequential(nn.Linear(1, hidden_t), nn.ReLU(), nn.Linear(hidden_t, hidden_t), nn.ReLU())
        self.mlp_in_X = nn.Sequential(nn.Linear(in_X, hidden_X), nn.ReLU(), nn.Linear(hidden_X, hidden_X), nn.ReLU())
        self.emb_Y = nn.Embedding(num_classes_Y, hidden_Y)
        setattr(self, self._layers_attr,
                nn.ModuleList([_Layer(hidden_X, hidden_Y, hidden_t, dropout) for _ in range(num_layers)]))
        hidden_cat = (num_layers + 1) * (hidden_X + hidden_Y) + hidden_t      # model/gnn.py:137
        self.mlp_out = nn.Sequential(nn.Linear(hidden_cat, hidden_cat), nn.ReLU(), nn.Linear(hidden_cat, out_size))
        self._dims = TowerDims(in_X=in_X, num_classes_Y=num_classes_Y, hidden_t=hidden_t, hidden_X=hidden_X,
                               hidden_Y=hidden_Y, num_layers=num_layers, out_size=out_size, is_mlp=int(self._is_mlp))
        self._ptr_key = None
        self._ptr_arr = None
        self._ws = None
        self._prep_buf, self._prep_key = None, None
        # True: the first layer walks each node's set attributes instead of running the GEMM (same bits; pays when the
        # attributes are sparse).  The diffusion models set it from the attribute marginals.
        self._sparse_X = False
        # dropout masks are Philox(seed, call, element): the containers (GNN / GNNAsymm) give each of their towers its own
        # key, otherwise pred_X and pred_E.gnn_encoder (same hidden sizes in every reference config) would draw identical
        # masks step after step
        self._tower_id = 0

    # ---- C-ABI plumbing
    def _ordered_params(self):
        ps = [self.mlp_in_t[0].weight, self.mlp_in_t[0].bias, self.mlp_in_t[2].weight, self.mlp_in_t[2].bias,
              self.mlp_in_X[0].weight, self.mlp_in_X[0].bias, self.mlp_in_X[2].weight, self.mlp_in_X[2].bias,
              self.emb_Y.weight]
        for layer in getattr(self, self._layers_attr):
            ps += [layer.update_X[0].weight, layer.update_X[0].bias, layer.update_X[2].weight, layer.update_X[2].bias,
                   layer.update_Y[0].weight, layer.update_Y[0].bias, layer.update_Y[2].weight, layer.update_Y[2].bias]
        ps += [self.mlp_out[0].weight, self.mlp_out[0].bias, self.mlp_out[2].weight, self.mlp_out[2].bias]
        return ps

    def _param_ptrs(self):
        ps = self._ordered_params()
        key = tuple(p.data_ptr() for p in ps)
        if key != self._ptr_key:
            for p in ps:
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise TypeError("tower parameters must be contiguous float32")
            self._ptr_arr = (C.c_void_p * len(ps))(*key)
            self._ptr_key = key
        return self._ptr_arr

    def _head(self, diff):
        """(dims, parameter pointers) of the fused calls.  diff: the sampler's half-width head -- mlp_out.2 folded to the
        logit difference l1 - l0 per attribute (gm_head_diff_prepare; out_size F instead of 2F), recomputed when mlp_out.2
        has been updated.  forward() and training keep the two-logit head."""
        ptrs = self._param_ptrs()
        if not diff:
            return self._dims, ptrs
        if self.num_classes_X != 2 or self._dims.out_size != 2 * self.num_attrs_X:
            raise ValueError("the logit-difference head needs a two-class attribute head (out_size = 2 * num_attrs_X)")
        w, b = self.mlp_out[2].weight, self.mlp_out[2].bias
        key = (w.data_ptr(), w._version, b.data_ptr(), b._version)
        if key != getattr(self, "_diff_key", None):
            self._diff_wb = ops.head_diff(w.detach(), b.detach())
            d = self._dims
            self._diff_dims = TowerDims(in_X=d.in_X, num_classes_Y=d.num_classes_Y, hidden_t=d.hidden_t, hidden_X=d.hidden_X,
                                        hidden_Y=d.hidden_Y, num_layers=d.num_layers, out_size=d.out_size // 2,
                                        is_mlp=d.is_mlp)
            self._diff_key = key
        if getattr(self, "_diff_ptr_key", None) != (self._ptr_key, self._diff_key):
            arr = (C.c_void_p * len(ptrs))(*ptrs)
            arr[len(ptrs) - 2], arr[len(ptrs) - 1] = self._diff_wb[0].data_ptr(), self._diff_wb[1].data_ptr()
            self._diff_ptrs = arr
            self._diff_ptr_key = (self._ptr_key, self._diff_key)
        return self._diff_dims, self._diff_ptrs

    def _prep(self):
        """B, D, D^T of the first layer (include/graphmaker_hip.h), recomputed when mlp_in_X.0 has been updated."""
        w, b = self.mlp_in_X[0].weight, self.mlp_in_X[0].bias
        key = (w.data_ptr(), w._version, b.data_ptr(), b._version)
        if key != self._prep_key:
            self._prep_buf = ops.tower_prepare(self._dims, self._param_ptrs(), w.device)
            self._prep_key = key
            self._hx0_rows = None                  # h_X^0 kept in the workspace came from the old weights
        return self._prep_buf

    def _workspace(self, n, device):
        need = ops.tower_workspace_bytes(self._dims, n)
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
            self._hx0_rows = None                  # a fresh workspace holds no mlp_in_X(X) to reuse
        return self._ws

    def _on_tape(self):
        return torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    def _run_tape(self, t_float, X_t_one_hot, Y_real, A_t, diff_head=False):
        """The tower operator by operator with autograd recording (model/gnn.py:146-176 / :440-463); returns (n, out).
        diff_head: the last Linear folded to the logit difference per attribute, (n, F) -- what a two-class cross entropy
        needs (half of the head's GEMMs forward and backward); the gradient reaches mlp_out.2 through autograd.head_diff."""
        def lin(m, x, relu):
            return ag.linear(x, m.weight, m.bias, relu)

        n = X_t_one_hot.shape[0]
        Y32 = self._labels(Y_real)
        t = t_float.reshape(1, 1).to(torch.float32)
        h_t = lin(self.mlp_in_t[2], lin(self.mlp_in_t[0], t, True), True)                     # (1, hidden_t)
        def mlp2(seq, x, relu_out):
            return ag.mlp2(x, seq[0].weight, seq[0].bias, seq[2].weight, seq[2].bias, relu_out)

        h_X = ag.mlp2(X_t_one_hot, self.mlp_in_X[0].weight, self.mlp_in_X[0].bias, self.mlp_in_X[2].weight,
                      self.mlp_in_X[2].bias, True, onehot_sparse=self._sparse_X)      # first layer as in the fused call
        h_Y = ag.embedding(self.emb_Y.weight, Y32)
        h_t_exp = h_t.expand(n, -1)
        hX_list, hY_list = [h_X], [h_Y]
        hX_w = h_X.shape[1]
        dropout_state = self._dropout()
        for layer in getattr(self, self._layers_attr):
            if self._is_mlp:                                                                   # model/gnn.py:386-392
                in_X, in_Y = torch.cat([h_X, h_Y, h_t_exp], dim=1), h_Y
            else:                                                                              # model/gnn.py:60-68
                agg = ag.spmm_mean(A_t, torch.cat([h_X, h_Y], dim=1))
                in_X, in_Y = torch.cat([agg, h_t_exp], dim=1), agg[:, hX_w:]                   # A @ h_Y = columns of A @ [h_X|h_Y]
            ux, uy = layer.update_X, layer.update_Y
            h_X = ag.update_block(in_X, ux[0].weight, ux[0].bias, ux[2].weight, ux[2].bias, ux[2].eps)
            h_Y = ag.update_block(in_Y, uy[0].weight, uy[0].bias, uy[2].weight, uy[2].bias, uy[2].eps)
            if self.training:
                h_X = ag.dropout(h_X, ux[3].p, dropout_state)
                h_Y = ag.dropout(h_Y, uy[3].p, dropout_state)
            hX_list.append(h_X)
            hY_list.append(h_Y)
        h_cat = torch.cat(hX_list + hY_list + [h_t_exp], dim=1)
        if diff_head:
            if self.num_classes_X != 2 or self._dims.out_size != 2 * self.num_attrs_X:
                raise ValueError("the logit-difference head needs a two-class attribute head (out_size = 2 * num_attrs_X)")
            mo = self.mlp_out
            Wd, bd = ag.head_diff(mo[2].weight, mo[2].bias)
            return ag.mlp2(h_cat, mo[0].weight, mo[0].bias, Wd, bd, False)
        return mlp2(self.mlp_out, h_cat, False)

    def _run(self, t_float, X_t_one_hot, Y_real, A_t, want_hcat=False, rows=None, out=None, reuse_X=False, x_u8=None,
             diff_head=False):
        """reuse_X: the caller guarantees that X_t_one_hot, the parameters and the node count are those of the previous
        fused call of THIS tower, so mlp_in_X(X) is taken from the tower's workspace (ModelAsync's structure steps).
        diff_head: return the (n, F) logit differences of _head(True) instead of the (n, 2F) logits (sampling only)."""
        if self._on_tape() and not want_hcat and rows is None and out is None:
            return self._run_tape(t_float, X_t_one_hot, Y_real, A_t, diff_head=diff_head)
        if self.training and any(m.p > 0 for m in self.modules() if isinstance(m, nn.Dropout)):
            raise NotImplementedError("the fused tower call has no dropout: use train mode with autograd enabled, or eval()")
        n = X_t_one_hot.shape[0]
        X = X_t_one_hot if X_t_one_hot.is_contiguous() else X_t_one_hot.contiguous()
        Y32 = self._labels(Y_real)
        t = t_float.reshape(1).to(torch.float32)
        ws = self._workspace(n, X.device)
        prep = self._prep()
        reuse = reuse_X and getattr(self, "_hx0_rows", None) == (n, None)
        self._hx0_rows = (n, None)
        dims, ptrs = self._head(diff_head)
        return ops.tower_forward(dims, ptrs, n, t, None if reuse else X, Y32.contiguous(), A_t, ws,
                                 want_hcat=want_hcat, rows=rows, out=out, prep=prep, x_u8=None if reuse else x_u8,
                                 sparse_X=self._sparse_X)


    def _run_sharded(self, t_float, X_t_one_hot, Y_real, A_t, plan, group, out, gather, reuse_X=False, x_u8=None,
                     diff_head=False):
        """Row-sharded forward for the one-process-per-GPU decomposition: this rank computes node rows
        plan.node_range() of every stage; before each message-passing layer the [h_X | h_Y] rows of all ranks are
        replicated by `gather` (parallel.all_gather_node_rows).  Writes rows node_range() of `out` (n, out_size)."""
        class _Now:
            def __init__(self, *a):
                gather(*a)

            def wait(self):
                pass
        for _ in self._sharded_stages(t_float, X_t_one_hot, Y_real, A_t, plan, group, out, _Now, reuse_X=reuse_X, x_u8=x_u8,
                                      diff_head=diff_head):
            pass
        return out

    def _sharded_stages(self, t_float, X_t_one_hot, Y_real, A_t, plan, group, out, gather_start, reuse_X=False, x_u8=None,
                        diff_head=False):
        """_run_sharded as a generator: it yields right after each all-gather has been STARTED
        (gather_start(plan, buffer, group) -> handle with .wait(), parallel.start_all_gather_node_rows) and waits for it
        when resumed, so the caller can enqueue other kernels (the other tower's stage) under the transfer."""
        n = X_t_one_hot.shape[0]
        X = X_t_one_hot if X_t_one_hot.is_contiguous() else X_t_one_hot.contiguous()
        Y32 = self._labels(Y_real)
        t = t_float.reshape(1).to(torch.float32)
        dev = X.device
        (d, ptrs), ws = self._head(diff_head), self._workspace(n, dev)
        prep = self._prep()
        rows = plan.node_range()
        if plan.replicate_graph_part and not self._is_mlp:
            # input projections + message passing on all n rows on every rank (no exchange), mlp_out on this rank's rows
            reuse = reuse_X and getattr(self, "_hx0_rows", None) == (n, None)
            self._hx0_rows = (n, None)
            ops.tower_forward(d, ptrs, n, t, None if reuse else X, Y32, A_t, ws, rows=rows, out=out, prep=prep,
                              x_u8=None if reuse else x_u8, sparse_X=self._sparse_X)
            return
        if reuse_X and getattr(self, "_hx0_rows", None) == (n, rows):
            X = x_u8 = None                        # mlp_in_X(X) of these rows is still in the workspace
        self._hx0_rows = (n, rows)
        need_gather = not self._is_mlp and d.num_layers > 0
        width = d.hidden_X + d.hidden_Y
        bufs = [None, None]
        if need_gather:   # persistent across steps: the collectives' stream keeps references, re-allocating churns
            key = (plan.world * plan.node_rows, width, dev)
            if getattr(self, "_gather_key", None) != key:
                self._gather_bufs = [torch.empty(key[:2], dtype=torch.float32, device=dev) for _ in range(2)]
                self._gather_key = key
            bufs = self._gather_bufs
        ops.tower_stage(d, ptrs, n, -1, rows, ws, t_float=t, X_one_hot=X, Y32=Y32, layer_dst=bufs[0], prep=prep, x_u8=x_u8,
                        sparse_X=self._sparse_X)
        for l in range(d.num_layers):
            src = None
            if need_gather:
                handle = gather_start(plan, bufs[l & 1], group)
                yield
                handle.wait()
                src = bufs[l & 1]
            dst = bufs[(l + 1) & 1] if need_gather and l + 1 < d.num_layers else None
            ops.tower_stage(d, ptrs, n, l, rows, ws, adj=A_t, layer_src=src, layer_dst=dst)
        ops.tower_stage(d, ptrs, n, d.num_layers, rows, ws, out=out)


class GNNTower(_TowerBase):
    """model/gnn.py:72-176.  Same constructor and forward(t_float, X_t_one_hot, Y_real, A_t)."""
    _layers_attr = "gnn_layers"

    def __init__(self, num_attrs_X, num_classes_X, num_classes_Y, hidden_t, hidden_X, hidden_Y, out_size,
                 num_gnn_layers, dropout, node_mode):
        super().__init__()
        self._build(num_attrs_X, num_classes_X, num_classes_Y, hidden_t, hidden_X, hidden_Y, out_size, num_gnn_layers,
                    dropout)
        self.node_mode = node_mode

    def forward(self, t_float, X_t_one_hot, Y_real, A_t):
        if not isinstance(A_t, ops.Adjacency):
            raise TypeError("A_t must come from graphmaker_amd get_adj()/ops.adjacency_from_bitmap()")
        out = self._run(t_float, X_t_one_hot, Y_real, A_t)
        if self.node_mode:
            return out.reshape(Y_real.size(0), self.num_attrs_X, -1)       # model/gnn.py:172
        return out


class MLPTower(_TowerBase):
    """model/gnn.py:396-463.  forward(t_float, X_t_one_hot, Y_real)."""
    _layers_attr = "mlp_layers"
    _is_mlp = True

    def __init__(self, num_attrs_X, num_classes_X, num_classes_Y, hidden_t, hidden_X, hidden_Y, num_mlp_layers,
                 dropout):
        super().__init__()
        self._build(num_attrs_X, num_classes_X, num_classes_Y, hidden_t, hidden_X, hidden_Y,
                    num_attrs_X * num_classes_X, num_mlp_layers, dropout)

    def forward(self, t_float, X_t_one_hot, Y_real):
        out = self._run(t_float, X_t_one_hot, Y_real, None)
        return out.reshape(Y_real.size(0), self.num_attrs_X, -1)           # model/gnn.py:461


class LinkPredictor(nn.Module):
    """model/gnn.py:178-250.  forward(t_float, X_t_one_hot, Y_real, A_t, src, dst) -> (B, 2) logits."""

    def __init__(self, num_attrs_X, num_classes_X, num_classes_Y, num_classes_E, hidden_t, hidden_X, hidden_Y, hidden_E,
                 num_gnn_layers, dropout):
        super().__init__()
        self.gnn_encoder = GNNTower(num_attrs_X, num_classes_X, num_classes_Y, hidden_t, hidden_X, hidden_Y, hidden_E,
                                    num_gnn_layers, dropout, node_mode=False)
        self.mlp_out = nn.Sequential(nn.Linear(hidden_E, hidden_E), nn.ReLU(), nn.Linear(hidden_E, num_classes_E))

    def pair_params(self):
        return (self.mlp_out[0].weight, self.mlp_out[0].bias, self.mlp_out[2].weight, self.mlp_out[2].bias)

    RELU63_LIMIT = 2.0 ** 63

    @torch.no_grad()
    def relu63_bounds(self):
        """(largest possible |hidden activation| of the pair MLP, largest |second-layer weight|), from the weights alone.
        The all-pairs kernel evaluates the hidden ReLU (model/gnn.py:229) as clamp01(x * 2^-63) against second-layer weights
        scaled by 2^63: exact for x <= 0 and 2^-63 <= x <= 2^63; an activation above 2^63 would saturate and
        a weight of 2^64 would overflow.  The bound follows the encoder: one-hot attributes select one first-layer column per
        attribute, LayerNorm outputs are at most |gamma| sqrt(d) + |beta|, t lies in [0, 1]; every Linear is bounded by
        |W| @ bound + |b|."""
        def lin(m, v):
            return m.weight.abs().float() @ v + (m.bias.abs().float() if m.bias is not None else 0.0)

        enc = self.gnn_encoder
        t0 = enc.mlp_in_t[0]
        h_t = lin(enc.mlp_in_t[2], t0.weight.abs().float()[:, 0] + t0.bias.abs().float())
        w0 = enc.mlp_in_X[0].weight.abs().float()
        first = w0.reshape(w0.shape[0], enc.num_attrs_X, enc.num_classes_X).amax(2).sum(1) + enc.mlp_in_X[0].bias.abs().float()
        hX, hY = [lin(enc.mlp_in_X[2], first)], [enc.emb_Y.weight.abs().float().amax(0)]
        for layer in getattr(enc, enc._layers_attr):
            for seq, acc in ((layer.update_X, hX), (layer.update_Y, hY)):
                ln = seq[2]
                acc.append(ln.weight.abs().float() * float(ln.normalized_shape[0]) ** 0.5 + ln.bias.abs().float())
        h = lin(enc.mlp_out[2], lin(enc.mlp_out[0], torch.cat(hX + hY + [h_t])))
        hid = lin(self.mlp_out[0], h * h)
        return float(hid.max()), float(self.mlp_out[2].weight.abs().max())

    def check_relu63(self):
        """Warn (once per weight version; one small device-to-host read) when the weights do not rule out a hidden activation
        outside the range in which the kernel's ReLU is exact.  Activations in (0, 2^-63) cannot be ruled out by a bound: they
        come back rounded to a multiple of 2^-86, an absolute error below 2^-86 |w| per hidden unit."""
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if key == getattr(self, "_relu63_key", None):
            return
        self._relu63_key = key
        hid, w2 = self.relu63_bounds()
        if not (hid < self.RELU63_LIMIT and 2.0 * w2 < 2.0 * self.RELU63_LIMIT):
            import warnings
            warnings.warn(f"pred_E: the weights allow a pair-MLP hidden activation of up to {hid:.3g} and hold a second-layer "
                          f"weight of {w2:.3g}; the all-pairs kernel's ReLU is exact only up to 2^63 = 9.2e18 (weights below "
                          "2^64).  Larger activations saturate there where nn.ReLU would not (INTEGRATION.md, 'Numerical "
                          "contract').", RuntimeWarning, stacklevel=3)

    def forward(self, t_float, X_t_one_hot, Y_real, A_t, src, dst):
        h = self.gnn_encoder(t_float, X_t_one_hot, Y_real, A_t)
        W1, b1, W2, b2 = self.pair_params()
        if torch.is_grad_enabled() and (h.requires_grad or W1.requires_grad):
            z = ag.hadamard_gather(h, src, dst)
            return ag.mlp2(z, W1, b1, W2, b2, False)
        z = ops.hadamard_gather(h, src.contiguous(), dst.contiguous())     # model/gnn.py:246
        return ops.linear(ops.linear(z, W1, b1, relu=True), W2, b2)        # model/gnn.py:248


class GNN(nn.Module):
    """model/gnn.py:252-334 (Sync denoiser: pred_X GNNTower + pred_E LinkPredictor)."""

    def __init__(self, num_attrs_X, num_classes_X, num_classes_Y, num_classes_E, gnn_X_config, gnn_E_config):
        super().__init__()
        self.pred_X = GNNTower(num_attrs_X, num_classes_X, num_classes_Y, out_size=num_attrs_X * num_classes_X,
                               node_mode=True, **gnn_X_config)
        self.pred_E = LinkPredictor(num_attrs_X, num_classes_X, num_classes_Y, num_classes_E, **gnn_E_config)
        self.pred_X._tower_id, self.pred_E.gnn_encoder._tower_id = 1, 2

    def forward(self, t_float, X_t_one_hot, Y, A_t, batch_src, batch_dst):
        logit_X = self.pred_X(t_float, X_t_one_hot, Y, A_t)
        logit_E = self.pred_E(t_float, X_t_one_hot, Y, A_t, batch_src, batch_dst)
        return logit_X, logit_E


class GNNAsymm(nn.Module):
    """model/gnn.py:465-550 (Async denoiser: pred_X MLPTower + pred_E LinkPredictor)."""

    def __init__(self, num_attrs_X, num_classes_X, num_classes_Y, num_classes_E, mlp_X_config, gnn_E_config):
        super().__init__()
        self.pred_X = MLPTower(num_attrs_X, num_classes_X, num_classes_Y, **mlp_X_config)
        self.pred_E = LinkPredictor(num_attrs_X, num_classes_X, num_classes_Y, num_classes_E, **gnn_E_config)
        self.pred_X._tower_id, self.pred_E.gnn_encoder._tower_id = 1, 2

    def forward(self, t_float_X, t_float_E, X_t_one_hot, Y, X_one_hot_2d, A_t, batch_src, batch_dst):
        logit_X = self.pred_X(t_float_X, X_t_one_hot, Y)
        logit_E = self.pred_E(t_float_E, X_one_hot_2d, Y, A_t, batch_src, batch_dst)
        return logit_X, logit_E
