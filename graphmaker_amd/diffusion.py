"""Reverse-diffusion sampler with the reference's classes and signatures (reference: model/diffusion.py), driven by
libgraphmaker_hip.so.

    ModelSync(T, X_marginal, Y_marginal, E_marginal, gnn_X_config, gnn_E_config, num_nodes)      (:536-543)
    ModelAsync(T_X, T_E, X_marginal, Y_marginal, E_marginal, mlp_X_config, gnn_E_config, num_nodes) (:928-936)
    .sample(batch_size=32768, num_workers=4) -> (X (F,N,2) float, Y one-hot (N,C) float, E (N,N) int64)  (:797-905, :1295-1410)
    .get_adj(E_t), .graph_encoder.pred_X / .pred_E                                                 (:239-256)

What differs inside: the edge encoder runs once per step instead of once per 32768-pair batch
(identical in eval mode); all pairs are scored by one tiled kernel with posterior + draw fused, E_t
lives as a bitmap + CSR instead of a dense int64 matrix; draws threshold counter-based Philox uniforms instead of
torch.multinomial (same distribution; host-supplied uniforms are accepted for parity runs).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import autograd as ag
from . import ops, parallel
from .gnn import GNN, GNNAsymm

__all__ = ["ModelSync", "ModelAsync", "NoiseSchedule", "MarginalTransition"]

_f32 = np.float32


class NoiseSchedule(nn.Module):
    """Cosine schedule, model/diffusion.py:96-118: float64 alpha-bars on T+2 points, then betas/alphas/alpha_bars in
    fp32 with torch's log/cumsum/exp (so the tables equal the reference's bit for bit on the same torch)."""

    def __init__(self, T, device, s=0.008):
        super().__init__()
        n = T + 2
        grid = np.linspace(0, n, n)
        ab = np.cos(0.5 * np.pi * ((grid / n) + s) / (1 + s)) ** 2
        ab = ab / ab[0]
        ratio = ab[1:] / ab[:-1]
        betas = torch.from_numpy(1 - ratio).float()
        alphas = 1 - torch.clamp(betas, min=0, max=0.9999)
        alpha_bars = torch.exp(torch.cumsum(torch.log(alphas), dim=0))
        # host copies drive the per-step tables (no device sync in the sampling loop)
        self.alphas_host = alphas.numpy().copy()
        self.alpha_bars_host = alpha_bars.numpy().copy()
        self.betas = nn.Parameter(betas.to(device), requires_grad=False)
        self.alphas = nn.Parameter(alphas.to(device), requires_grad=False)
        self.alpha_bars = nn.Parameter(alpha_bars.to(device), requires_grad=False)


def _transition(a, m):
    """Q = a*I + (1-a)*1 m^T in fp32 (model/diffusion.py:52-84).  m: (..., 2) -> Q: (..., 2, 2) with Q[r][c]."""
    a = _f32(a)
    m = np.asarray(m, dtype=_f32)
    eye = np.eye(2, dtype=_f32)
    t0 = a * eye                                       # (2,2)
    t1 = (_f32(1.0) - a) * m[..., None, :]             # (...,1,2) broadcast over rows
    return (t0 + t1).astype(_f32)


def posterior_table(Q_t, Q_bar_s, Q_bar_t):
    """tab[..., (i*2+k)*2+e] = Q_t[k][e] * Qbar_s[i][k] / Qbar_t[i][e]  (zero denominator -> 1): BaseModel.posterior
    (model/diffusion.py:351-406) evaluated for both one-hot values e of Z_t; the kernels then need
    p[k] = prior[0]*tab[0,k,e] + prior[1]*tab[1,k,e]."""
    Q_t, Q_bar_s, Q_bar_t = (np.asarray(q, dtype=_f32) for q in (Q_t, Q_bar_s, Q_bar_t))
    lead = Q_t.shape[:-2]
    tab = np.empty(lead + (8,), dtype=_f32)
    for i in range(2):
        for k in range(2):
            for e in range(2):
                num = Q_t[..., k, e] * Q_bar_s[..., i, k]
                den = np.where(Q_bar_t[..., i, e] == 0, _f32(1.0), Q_bar_t[..., i, e]).astype(_f32)
                tab[..., (i * 2 + k) * 2 + e] = (num / den).astype(_f32)
    return tab


class MarginalTransition(nn.Module):
    """model/diffusion.py:14-84.  get_Q_bar_E / get_Q_bar_X return device tensors like the reference; the sampler
    itself uses the host-side tables below."""

    def __init__(self, device, X_marginal, E_marginal, num_classes_E):
        super().__init__()
        F_, C_ = X_marginal.shape
        self.I_X = nn.Parameter(torch.eye(C_, device=device).unsqueeze(0).expand(F_, C_, C_).clone(), requires_grad=False)
        self.I_E = nn.Parameter(torch.eye(num_classes_E, device=device), requires_grad=False)
        self.m_X = nn.Parameter(X_marginal.unsqueeze(1).expand(F_, C_, -1).clone(), requires_grad=False)
        self.m_E = nn.Parameter(E_marginal.unsqueeze(0).expand(num_classes_E, -1).clone(), requires_grad=False)
        self.X_marginal_host = X_marginal.detach().cpu().numpy().astype(_f32)
        self.E_marginal_host = E_marginal.detach().cpu().numpy().astype(_f32)

    def get_Q_bar_E(self, alpha_bar_t):
        return alpha_bar_t * self.I_E + (1 - alpha_bar_t) * self.m_E

    def get_Q_bar_X(self, alpha_bar_t):
        return alpha_bar_t * self.I_X + (1 - alpha_bar_t) * self.m_X

    def table_E(self, alpha_t, abar_s, abar_t):
        m = self.E_marginal_host
        return posterior_table(_transition(alpha_t, m), _transition(abar_s, m), _transition(abar_t, m))

    def table_X(self, alpha_t, abar_s, abar_t):
        m = self.X_marginal_host                       # (F,2)
        return posterior_table(_transition(alpha_t, m), _transition(abar_s, m), _transition(abar_t, m))   # (F,8)


class BaseModel(nn.Module):
    """model/diffusion.py:143-472."""

    def __init__(self, T, X_marginal, Y_marginal, E_marginal, num_nodes):
        super().__init__()
        device = X_marginal.device
        self.num_classes_E = 2
        self.num_attrs_X, self.num_classes_X = X_marginal.shape
        if self.num_classes_X != 2:
            raise NotImplementedError("binary node attributes only (as every reference dataset, data.py:19-29)")
        self.num_classes_Y = len(Y_marginal)
        self.transition = MarginalTransition(device, X_marginal, E_marginal, self.num_classes_E)
        self.T = T
        self.noise_schedule = NoiseSchedule(T, device)
        self.num_nodes = num_nodes
        self.X_marginal = X_marginal
        self.Y_marginal = Y_marginal
        self.E_marginal = E_marginal
        self._sample_calls = 0
        # bag-of-words-like attributes (few ones per node): the towers' first layer walks the set attributes instead of
        # running its GEMM (same bits either way, gm_onehot.hip); break-even is near 9% ones at the reference's hidden sizes
        self._sparse_X = float(X_marginal[:, 1].float().mean()) < 0.06
        self._pair_ws = None
        # "f32" (exact, default) | "bf16x3" | "bf16": opt-in matrix-core precision of the pair MLP's hidden layer (ops.py)
        self.pair_precision = "f32"
        self._prof = None          # None, or dict phase -> [(start_event, end_event)] (bench.py / profiling)
        # edges per rank an exchange segment of the multi-GPU step holds; None: parallel.edge_segment_capacity (4x the edge
        # marginal's share).  The same value on every rank.
        self.edge_segment_capacity = None

    class _Phase:
        """Brackets a phase of the step with events on torch's current stream (the stream every library call is
        enqueued on) when profiling is on; free otherwise."""

        def __init__(self, model, name):
            only = model.__dict__.get("_prof_only")
            self.rec = model._prof if only is None or name in only else None
            self.name = name

        def __enter__(self):
            if self.rec is not None:
                self.e0 = torch.cuda.Event(enable_timing=True)
                self.e0.record()

        def __exit__(self, *exc):
            if self.rec is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                self.rec.setdefault(self.name, []).append((self.e0, e1))
            return False

    def phase(self, name):
        return BaseModel._Phase(self, name)

    def _buf(self, name, shape, dtype, device, zero=False):
        """Step-persistent scratch tensors (buffers that collectives touch are kept, not re-made).  zero: cleared when it is
        first made, never again."""
        cache = self.__dict__.setdefault("_bufs", {})
        key = (name, tuple(shape), dtype, device)
        if key not in cache:
            cache[key] = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=device)
        return cache[key]

    def _exchange_edges(self, plan, upper, group):
        """The sampled edges of this rank's pair chunks -> compact sorted list -> one all-gather -> the symmetric bitmap of
        E_s on every rank (model/diffusion.py:421-424 across ranks; parallel.py)."""
        n, dev = self.num_nodes, upper.device
        cap = self.edge_segment_capacity or parallel.edge_segment_capacity(plan, self.transition.E_marginal_host[1])
        seg = self._buf("edge_seg", (2 * cap + parallel.SEG_HEADER,), torch.int32, dev, zero=True)
        recv = self._buf("edge_recv", (plan.world, 2 * cap + parallel.SEG_HEADER), torch.int32, dev, zero=True)
        rowptr_ws = self._buf("edge_rowptr", (2 * plan.chunk_rows + 1,), torch.int32, dev)
        ovf = self._buf("edge_overflow", (1,), torch.int32, dev, zero=True)
        with self.phase("edges_pack"):
            ops.upper_edges_of_rows(upper, n, plan.pair_chunks(), seg, cap, rowptr_ws, ovf)
        with self.phase("allgather_E"):
            parallel.gather_edge_segments(plan, seg, recv, group)
        with self.phase("symmetrize"):
            return ops.bitmap_from_edge_segments(n, recv, cap, overflow=ovf)     # any rank's overflow reaches every rank

    def _const(self, key, host_array, device):
        """Small per-step constants (time embedding input, posterior tables) uploaded once per (key, device) instead of
        once per step: a pageable host-to-device copy stalls the launch stream of every step."""
        cache = self.__dict__.setdefault("_consts", {})
        k = (key, device)
        if k not in cache:
            cache[k] = torch.as_tensor(host_array).to(device)
        return cache[k]

    def _csr_ws(self, device):
        """The step-local adjacency workspace (ops.CsrWorkspace) of this model on `device`."""
        ws = self.__dict__.get("_csr_workspace")
        if ws is None or ws.device != device or ws.n != self.num_nodes:
            ws = self.__dict__["_csr_workspace"] = ops.CsrWorkspace(self.num_nodes, device)
        return ws

    def _side_stream(self, device):
        """The second stream of the Sync step (attribute branch beside the edge encoder), one per model and device."""
        st = self.__dict__.get("_side")
        if st is None or st.device != device:
            # normal priority.  A high-priority stream (the attribute branch is the longer of the two) made both branches end
            # together and took 0.1 ms off the step, but after a particular sequence of earlier work in the same process a
            # trajectory then never finished (GPU 100% busy, every kernel here is finite): withdrawn, see DESIGN.md 7b
            st = self.__dict__["_side"] = torch.cuda.Stream(device=device, priority=0)
        return st

    def enable_phase_timing(self, on=True, only=None):
        """only: names of the phases to bracket (every event record is a barrier in the stream, about 5-10 us of GPU idle
        time: a step with all of its phases bracketed runs measurably slower than the product's step)."""
        self._prof = {} if on else None
        self.__dict__["_prof_only"] = set(only) if (on and only is not None) else None

    def phase_times_ms(self):
        """{phase: [ms per occurrence]}; call after torch.cuda.synchronize()."""
        return {k: [a.elapsed_time(b) for a, b in v] for k, v in (self._prof or {}).items()}

    # ---- reference helper API
    def get_adj(self, E_t):
        """model/diffusion.py:239-256: dense symmetric (N,N) int64 -> row-normalised adjacency (CSR, values 1/deg)."""
        bits = ops.bitmap_from_dense(E_t.contiguous())
        return ops.adjacency_from_bitmap(bits, E_t.size(0))

    def posterior(self, Z_t, Q_t, Q_bar_s, Q_bar_t, prior):
        """model/diffusion.py:351-406 for callers that hold torch tensors (not used by sample())."""
        left = (Z_t @ torch.transpose(Q_t, -1, -2)).unsqueeze(dim=-2)
        numerator = left * Q_bar_s.unsqueeze(dim=-3)
        denominator = torch.transpose(Q_bar_t @ torch.transpose(Z_t, -1, -2), -1, -2).unsqueeze(-1).clone()
        denominator[denominator == 0.] = 1.
        return (prior.unsqueeze(-1) * (numerator / denominator)).sum(dim=-2)

    # ---- training / validation plumbing (model/diffusion.py:258-349, 559-604)
    def _real_state(self, X_one_hot_3d, E_one_hot):
        """The real graph in library layout, converted once per (tensor, shape): x0 uint8 (N,F), bits0 bitmap."""
        cache = self.__dict__.setdefault("_real_cache", {})
        out = []
        for name, t in (("X", X_one_hot_3d), ("E", E_one_hot)):
            if t is None:
                out.append(None)
                continue
            key = (name, t.data_ptr(), tuple(t.shape), t._version)
            if cache.get(name, (None, None))[0] != key:
                tc = t.contiguous().float()
                cache[name] = (key, ops.x_from_onehot3d(tc) if name == "X" else ops.bitmap_from_onehot(tc))
            out.append(cache[name][1])
        return out

    def _noise_X(self, x0, alpha_bar_t, seed, step, u=None):
        QX = torch.from_numpy(_transition(alpha_bar_t, self.transition.X_marginal_host).reshape(-1, 4)).to(x0.device)
        return ops.noise_X(x0, QX, seed=seed, step=step, u=u)

    def _noise_E(self, bits0, alpha_bar_t, seed, step, u=None):
        Q = _transition(alpha_bar_t, self.transition.E_marginal_host).reshape(4)
        return ops.noise_E(self.num_nodes, bits0, Q, seed=seed, step=step, u=u)

    def _draw_t(self, t, T, device):
        """t given (tensor / int) or drawn uniformly from 0..T like model/diffusion.py:582-587; returns (int, tensor)."""
        if t is None:     # drawn on the host (torch's CPU generator): the schedule tables are indexed on the host, and a device
            t = int(torch.randint(low=0, high=T + 1, size=(1,)).item())   # draw would have to be read back (a sync per step)
        t_int = int(t.item()) if torch.is_tensor(t) else int(t)
        if not 0 <= t_int <= T:
            raise ValueError(f"time step {t_int} outside 0..{T}")
        t_dev = t if torch.is_tensor(t) else torch.tensor([t_int], device=device)
        return t_int, t_dev

    @staticmethod
    def _edge_targets(batch_E_one_hot):
        """argmax of the (B,2) one-hot rows as uint8 (LossE, model/diffusion.py:138)."""
        return (batch_E_one_hot[:, 1] > batch_E_one_hot[:, 0]).to(torch.uint8).contiguous()

    @staticmethod
    def _kl_time_index(t, T):
        """The time index denoise_match_E/X actually use (model/diffusion.py:332, :681): they do not receive t but
        t_float = t / T (a float32 division, :587) and recover t = int(t_float.item() * T) in double precision, which
        truncates to t-1 for some (t, T) -- e.g. T=6 (cora / amazon_photo Async T_X), t=5 -> 4.  Reproduced as is: the
        validation terms select checkpoints (train_sync.py:150-181), so they have to equal the reference's."""
        return int(float(np.float32(t) / np.float32(T)) * T)

    def _kl_tables(self, ns, t, marginal_host, device, T=None):
        """(Q_t, Q_bar_s) as (ntab,4) device tensors for denoise_match (model/diffusion.py:332-341, 681-690).
        T: the schedule length, to recover the index the way the reference does (_kl_time_index)."""
        if T is not None:
            t = self._kl_time_index(t, T)
        Qt = _transition(ns.alphas_host[t], marginal_host).reshape(-1, 4)
        Qs = _transition(ns.alpha_bars_host[t - 1], marginal_host).reshape(-1, 4)
        return torch.from_numpy(Qt).to(device), torch.from_numpy(Qs).to(device)

    # ---- sampling plumbing
    def _next_seed(self, seed):
        if seed is None:
            seed = (torch.initial_seed() + 0x9E3779B97F4A7C15 * self._sample_calls) & 0xFFFFFFFFFFFFFFFF
        self._sample_calls += 1
        return int(seed) & 0xFFFFFFFFFFFFFFFF

    def _device(self):
        dev = self.X_marginal.device
        if dev.type != "cuda":
            raise RuntimeError("graphmaker_amd samples on a ROCm device only; move the marginals and the model to "
                               f"cuda first (got {dev}).  There is no CPU path.")
        return dev

    def _structure_step(self, pred_E, t_float, X_one_hot, Y32, bits_t, tab_E, seed, step, u=None, plan=None, group=None,
                        reuse_X=False, after_csr=None, before_pair=None, x_u8=None):
        """get_E_t (model/diffusion.py:428-472): returns (A_t, symmetric bitmap of E_s).
        plan: parallel.ShardPlan for the one-process-per-GPU decomposition (None = single GPU).
        reuse_X: X_one_hot is the tensor of the previous call (ModelAsync's structure steps): mlp_in_X(X) is not recomputed.
        after_csr(A_t) / before_pair(): hooks of the Sync step, which runs its attribute branch beside the edge encoder."""
        n = self.num_nodes
        with self.phase("csr"):      # built into the model's workspace: no host read of the edge count in a step
            A_t = ops.adjacency_from_bitmap(bits_t, n, workspace=self._csr_ws(bits_t.device),
                                            rows=None if plan is None else plan.adjacency_rows())
        if after_csr is not None:
            after_csr(A_t)
        enc = pred_E.gnn_encoder
        with self.phase("tower_E"):
            if plan is None:
                h = enc._run(t_float, X_one_hot, Y32, A_t, reuse_X=reuse_X, x_u8=x_u8)
            else:
                h_pad = self._buf("h_pad", (plan.world * plan.node_rows, enc._dims.out_size), torch.float32,
                                  X_one_hot.device)
                enc._run_sharded(t_float, X_one_hot, Y32, A_t, plan, group, h_pad[:n], parallel.all_gather_node_rows,
                                 reuse_X=reuse_X, x_u8=x_u8)
        if plan is not None:
            with self.phase("allgather_h"):
                h = parallel.all_gather_node_rows(plan, h_pad, group)[:n]
        if self._pair_ws is None or self._pair_ws.device != h.device:
            self._pair_ws = ops.pair_workspace(h.device)
        W1, b1, W2, b2 = pred_E.pair_params()
        pred_E.check_relu63()                 # once per weight version: the kernel's ReLU range (gnn.LinkPredictor.relu63_bounds)
        ranges = [(0, n)] if plan is None else plan.pair_chunks()
        # zeroed once: the pair kernel rewrites every word it owns on each call and the others stay zero
        upper = self._buf("upper", (n if plan is None else plan.padded_pair_rows, bits_t.shape[1]), torch.int32, h.device,
                          zero=True)
        if before_pair is not None:
            before_pair()
        with self.phase("pair"):      # a rank's two folded chunks go into ONE launch (shared tile queue)
            ops.pair_score_sample(h, W1, b1, W2, b2, bits_t, tab_E, seed=seed, step=step, u=u, row_ranges=ranges,
                                  out_upper=upper, workspace=self._pair_ws, precision=self.pair_precision)
        if plan is not None:
            return A_t, self._exchange_edges(plan, upper, group)
        with self.phase("symmetrize"):
            bits_s = ops.symmetrize_upper(upper[:n], n)
        return A_t, bits_s

    def _logits_fn(self, pred_X, t_float, x_onehot, Y32, A_t, plan, group, x_u8=None):
        """rows -> (n, F) attribute logit differences l1 - l0 valid on `rows` (all rows on a single GPU).  The sampler only
        takes softmax(logits) (model/diffusion.py:879), a function of the difference, so the node tower runs with its
        half-width head (gnn._TowerBase._head): half of its largest GEMM."""
        if plan is None:
            return lambda rows: pred_X._run(t_float, x_onehot, Y32, A_t, x_u8=x_u8, diff_head=True)
        n = self.num_nodes

        def sharded(rows):
            out = self._buf("logit_diff", (n, self.num_attrs_X), torch.float32, x_onehot.device)
            return pred_X._run_sharded(t_float, x_onehot, Y32, A_t, plan, group, out, parallel.all_gather_node_rows, x_u8=x_u8,
                                       diff_head=True)
        return sharded

    def _attribute_draw(self, logits_fn, x, tab_X, seed, step, u=None, plan=None, group=None):
        """softmax + posterior + sample_X (model/diffusion.py:879-895) on this rank's node rows, then replicate.
        logits_fn(rows) -> (n, F) logit differences valid on `rows` (_logits_fn)."""
        n = self.num_nodes
        dev = x.device
        rows = None if plan is None else plan.node_range()
        with self.phase("tower_X"):
            logits = logits_fn(rows)
        tabX_dev = self._const(("tab_X", step, tab_X.tobytes()), tab_X, dev)
        if plan is None:
            with self.phase("attr"):
                return ops.attr_posterior_sample(logits, x, tabX_dev, seed=seed, step=step, u=u, diff=True)
        x_pad = self._buf("x_pad", (plan.world * plan.node_rows, x.shape[1]), torch.uint8, dev)
        with self.phase("attr"):
            ops.attr_posterior_sample(logits, x, tabX_dev, seed=seed, step=step, u=u, rows=rows, x_s=x_pad[:n],
                                      want_onehot=False, diff=True)
        with self.phase("allgather_X"):
            x_s = parallel.all_gather_node_rows(plan, x_pad, group)[:n].clone()      # x_pad is reused by the next step
        with self.phase("attr"):
            return x_s, ops.onehot_from_x(x_s)

    def _finish(self, x_onehot, Y32, bits, dense):
        n, Fa = self.num_nodes, self.num_attrs_X
        if self.__dict__.get("_csr_workspace") is not None:
            self._csr_workspace.check()                                     # the one host read of a trajectory
        if self._pair_ws is not None:
            ops.check_pair_workspace(self._pair_ws)
        for key, buf in self.__dict__.get("_bufs", {}).items():
            if key[0] == "edge_overflow" and int(buf.item()):
                buf.zero_()
                raise RuntimeError("a rank sampled more edges than its exchange segment holds (parallel.edge_segment_capacity: "
                                   "4x the edge marginal's share; every rank raises this together): the sample is invalid; the "
                                   "model's E_marginal does not describe the graphs it generates -- set "
                                   "model.edge_segment_capacity on every rank")
        X = torch.transpose(x_onehot.reshape(n, Fa, -1), 0, 1)             # model/diffusion.py:899-901
        Y_one_hot = F.one_hot(Y32.long(), num_classes=self.num_classes_Y).float()
        if dense:
            return X, Y_one_hot, ops.dense_from_bitmap(bits, n)
        rowptr, colidx, rows = ops.csr_from_bitmap(bits, n, upper_only=True, want_rows=True)
        return X, Y_one_hot, torch.stack([rows, colidx]).long()


class ModelSync(BaseModel):
    """model/diffusion.py:517-905."""

    def __init__(self, T, X_marginal, Y_marginal, E_marginal, gnn_X_config, gnn_E_config, num_nodes):
        super().__init__(T=T, X_marginal=X_marginal, Y_marginal=Y_marginal, E_marginal=E_marginal, num_nodes=num_nodes)
        self.graph_encoder = GNN(num_attrs_X=self.num_attrs_X, num_classes_X=self.num_classes_X,
                                 num_classes_Y=self.num_classes_Y, num_classes_E=self.num_classes_E,
                                 gnn_X_config=gnn_X_config, gnn_E_config=gnn_E_config)
        self.graph_encoder.pred_X._sparse_X = self.graph_encoder.pred_E.gnn_encoder._sparse_X = self._sparse_X

    def step_tables(self, t):
        """Host tables of reverse step t (s = t-1): model/diffusion.py:852-858, 889-891."""
        ns = self.noise_schedule
        a_t, ab_s, ab_t = ns.alphas_host[t], ns.alpha_bars_host[t - 1], ns.alpha_bars_host[t]
        return self.transition.table_E(a_t, ab_s, ab_t), self.transition.table_X(a_t, ab_s, ab_t)

    # ---- training / validation (model/diffusion.py:559-795)
    def _apply_noise_state(self, X_one_hot_3d, E_one_hot, t=None, seed=None, uniforms=None):
        x0, bits0 = self._real_state(X_one_hot_3d, E_one_hot)
        t_int, t_dev = self._draw_t(t, self.T, X_one_hot_3d.device)
        seed = self._next_seed(seed)
        uni = uniforms or {}
        ab = self.noise_schedule.alpha_bars_host[t_int]
        bits_t = self._noise_E(bits0, ab, seed, t_int, u=uni.get("E"))
        x_t, x_t_onehot = self._noise_X(x0, ab, seed, t_int, u=uni.get("X"))
        return t_int, t_dev / self.T, x_t, x_t_onehot, bits_t

    def apply_noise(self, X_one_hot_3d, E_one_hot, t=None, *, seed=None, uniforms=None):
        """model/diffusion.py:559-604: returns (t_float (1,), X_t_one_hot (N,2F), E_t (N,N) int64)."""
        _, t_float, _, x_t_onehot, bits_t = self._apply_noise_state(X_one_hot_3d, E_one_hot, t, seed, uniforms)
        return t_float, x_t_onehot, ops.dense_from_bitmap(bits_t, self.num_nodes)

    def log_p_t(self, X_one_hot_3d, E_one_hot, Y, batch_src, batch_dst, batch_E_one_hot, t=None, *, seed=None,
                uniforms=None):
        """model/diffusion.py:606-654: corrupt G, run both denoisers, return (loss_X, loss_E) with autograd attached
        (train_sync.py:98-109)."""
        x0, _ = self._real_state(X_one_hot_3d, E_one_hot)
        _, t_float, _, x_t_onehot, bits_t = self._apply_noise_state(X_one_hot_3d, E_one_hot, t, seed, uniforms)
        A_t = ops.adjacency_from_bitmap(bits_t, self.num_nodes)
        # GNN.forward (model/gnn.py:330-334) in two calls: the two-class attribute loss is a function of l1 - l0 only, so the
        # node tower runs with its head folded to the difference (half of the head's GEMMs, forward and backward)
        ge = self.graph_encoder
        diff_X = ge.pred_X._run(t_float, x_t_onehot, Y, A_t, diff_head=True)
        logit_E = ge.pred_E(t_float, x_t_onehot, Y, A_t, batch_src, batch_dst)
        loss_X = ag.ce2_diff(diff_X, x0)                                             # LossX :474-515
        loss_E = ag.softmax_ce2(logit_E, self._edge_targets(batch_E_one_hot))        # LossE :120-141
        return loss_X, loss_E

    @torch.no_grad()
    def val_step(self, X_one_hot_3d, E_one_hot, Y, batch_src, batch_dst, batch_E_one_hot, *, seed=None):
        """model/diffusion.py:709-795: (denoise_match_E, denoise_match_X, log_p_0_E, log_p_0_X) as floats."""
        dev = E_one_hot.device
        x0, bits0 = self._real_state(X_one_hot_3d, E_one_hot)
        e0 = self._edge_targets(batch_E_one_hot)
        src, dst = batch_src.contiguous(), batch_dst.contiguous()
        ns, n = self.noise_schedule, self.num_nodes
        dm_X, dm_E = [], []
        for t_sample in range(1, self.T + 1):
            t_int, t_float, x_t, x_t_onehot, bits_t = self._apply_noise_state(X_one_hot_3d, E_one_hot, t_sample, seed)
            A_t = ops.adjacency_from_bitmap(bits_t, n)
            logit_X, logit_E = self.graph_encoder(t_float, x_t_onehot, Y, A_t, src, dst)
            e_t = ops.bitmap_lookup(bits_t, n, src, dst)                              # E_t[batch_src, batch_dst] :765
            QtE, QsE = self._kl_tables(ns, t_int, self.transition.E_marginal_host, dev, T=self.T)
            dm_E.append(ops.denoise_match2(logit_E, e_t, e0, QtE, QsE).item())
            QtX, QsX = self._kl_tables(ns, t_int, self.transition.X_marginal_host, dev, T=self.T)
            dm_X.append(ops.denoise_match2(logit_X, x_t, x0, QtX, QsX).item())
        denoise_match_E = float(np.mean(dm_E)) * self.T
        denoise_match_X = float(np.mean(dm_X)) * self.T
        loss_X, loss_E = self.log_p_t(X_one_hot_3d, E_one_hot, Y, src, dst, batch_E_one_hot, 0, seed=seed)
        return denoise_match_E, denoise_match_X, loss_E.item(), loss_X.item()

    @torch.no_grad()
    def denoise_step(self, t, x, x_onehot, Y32, bits_t, seed, uniforms=None, plan=None, group=None):
        """One reverse step t -> t-1 (the body of the loop at model/diffusion.py:847-896).
        State: x (N,F) uint8, x_onehot (N,2F) float, Y32 (N,) int32, bits_t symmetric bitmap.
        Returns (x_s, x_onehot_s, bits_s)."""
        dev = x.device
        n = self.num_nodes
        tab_E, tab_X = self.step_tables(t)
        t_float = self._const(("t", t, self.T), np.array([t / self.T], np.float32), dev)   # :860
        uE = uniforms.get(("E", t)) if uniforms else None
        uX = uniforms.get(("X", t)) if uniforms else None
        if plan is not None:
            return self._denoise_step_sharded(t, t_float, x, x_onehot, Y32, bits_t, tab_E, tab_X, seed, uE, uX, plan, group)
        # The attribute branch (node tower + draw) reads the same A_t / X_t / Y as the edge encoder (A_t of E_t, not E_s:
        # model/diffusion.py:875-878) and nothing of the structure branch, so it is issued on a second stream beside the edge
        # encoder: its memory-bound kernels (SpMM, LayerNorm, the draw) and the partly empty last rounds of the GEMM grids fill
        # with the other tower's work.  Both join before the pair kernel, which takes the whole chip.
        cur = torch.cuda.current_stream(dev)
        side = self._side_stream(dev)
        pred_X = self.graph_encoder.pred_X
        res = {}

        wall = self.phase("towers_wall")        # both towers + the attribute draw as they overlap: CSR done -> joined

        def attribute_branch(A_t):
            wall.__enter__()
            side.wait_stream(cur)                                               # A_t (and the step's inputs) are ready
            with torch.cuda.stream(side):
                res["x"], res["oh"] = self._attribute_draw(self._logits_fn(pred_X, t_float, x_onehot, Y32, A_t, None, None,
                                                                           x_u8=x), x, tab_X, seed, t, u=uX)

        def join():
            cur.wait_stream(side)
            wall.__exit__()
            for v in res.values():            # allocated on the side stream, consumed on this one from here on
                v.record_stream(cur)

        _, bits_s = self._structure_step(self.graph_encoder.pred_E, t_float, x_onehot, Y32, bits_t, tab_E, seed, t, u=uE,
                                         after_csr=attribute_branch, before_pair=join, x_u8=x)
        return res["x"], res["oh"], bits_s

    def _denoise_step_sharded(self, t, t_float, x, x_onehot, Y32, bits_t, tab_E, tab_X, seed, uE, uX, plan, group):
        """denoise_step on one rank of the one-process-per-GPU decomposition (parallel.py), scheduled so that the
        transfers hide under compute.  Both towers read the same A_t / X_t / Y (the attribute tower uses A_t of E_t,
        model/diffusion.py:875-878), so their row-sharded stages are issued alternately, each tower on its own stream:
        while the [h_X | h_Y] rows of one tower's stage are on the wire, the other tower's stage computes.  The attribute draw runs before the pair
        kernel and its all-gather travels under it.  Every rank issues the collectives in the same order.  Draws are
        keyed on global indices, so the result is bit-identical to the single-GPU step."""
        n, dev = self.num_nodes, x.device
        ge = self.graph_encoder
        enc, pred_X = ge.pred_E.gnn_encoder, ge.pred_X
        with self.phase("csr"):      # only the rows this rank aggregates (its node block, or all with a replicated graph part)
            A_t = ops.adjacency_from_bitmap(bits_t, n, workspace=self._csr_ws(dev), rows=plan.adjacency_rows())
        h_pad = self._buf("h_pad", (plan.world * plan.node_rows, enc._dims.out_size), torch.float32, dev)
        logits = self._buf("logit_diff", (n, self.num_attrs_X), torch.float32, dev)      # l1 - l0 per attribute (_logits_fn)
        x_pad = self._buf("x_pad", (plan.world * plan.node_rows, x.shape[1]), torch.uint8, dev)
        start = parallel.start_all_gather_node_rows
        # A rank's row block is small (1719 rows at 8 ranks: most GEMMs of a stage are a few hundred workgroups), so each
        # tower gets its own stream and the stages of the two fill the chip together, as on a single GPU.  The host still
        # issues the stages alternately, which fixes the order of the collectives on every rank; an all-gather waits for
        # the stream it was started from, and its tower's next stage waits for it on that stream.
        cur, side = torch.cuda.current_stream(dev), self._side_stream(dev)
        with self.phase("towers"):                                   # CSR done -> both towers and the attribute draw done
            side.wait_stream(cur)
            stages = [(enc._sharded_stages(t_float, x_onehot, Y32, A_t, plan, group, h_pad[:n], start, x_u8=x), cur),
                      (pred_X._sharded_stages(t_float, x_onehot, Y32, A_t, plan, group, logits, start, x_u8=x, diff_head=True),
                       side)]
            while stages:
                for item in list(stages):
                    try:
                        with torch.cuda.stream(item[1]):
                            next(item[0])
                    except StopIteration:
                        stages.remove(item)
            h_on_wire = start(plan, h_pad, group)                    # node embeddings of every rank (pair kernel input)
            with torch.cuda.stream(side):                            # ... while this rank draws its attribute rows
                with self.phase("attr"):
                    ops.attr_posterior_sample(logits, x, self._const(("tab_X", t, tab_X.tobytes()), tab_X, dev), seed=seed, step=t,
                                              u=uX, rows=plan.node_range(), x_s=x_pad[:n], want_onehot=False, diff=True)
                x_on_wire = start(plan, x_pad, group)                # travels under the pair kernel
            cur.wait_stream(side)                                    # the pair kernel takes the whole chip: join first
        with self.phase("allgather_h"):
            h_on_wire.wait()
        h = h_pad[:n]
        if self._pair_ws is None or self._pair_ws.device != dev:
            self._pair_ws = ops.pair_workspace(dev)
        W1, b1, W2, b2 = ge.pred_E.pair_params()
        ge.pred_E.check_relu63()
        upper = self._buf("upper", (plan.padded_pair_rows, bits_t.shape[1]), torch.int32, dev, zero=True)
        with self.phase("pair"):
            ops.pair_score_sample(h, W1, b1, W2, b2, bits_t, tab_E, seed=seed, step=t, u=uE,
                                  row_ranges=plan.pair_chunks(), out_upper=upper, workspace=self._pair_ws,
                                  precision=self.pair_precision)
        bits_s = self._exchange_edges(plan, upper, group)
        with self.phase("allgather_X"):
            x_on_wire.wait()
            x_s = x_pad[:n].clone()                                  # x_pad is reused by the next step
        with self.phase("attr"):
            oh_s = ops.onehot_from_x(x_s)
        return x_s, oh_s, bits_s

    @torch.no_grad()
    def sample(self, batch_size=32768, num_workers=4, *, seed=None, uniforms=None, dense=True, plan=None, group=None):
        """model/diffusion.py:797-905.  batch_size / num_workers are accepted and ignored (no pair batches exist).
        seed: Philox key (default derived from torch.initial_seed() and the call count).
        uniforms: optional dict of device float32 arrays with host-supplied draws, keys "Y", "E_prior", ("X", t)
        with t=0 for the prior, ("E", t); indexing as in include/graphmaker_hip.h.
        dense=False returns the (2, m) upper edge list instead of the (N,N) int64 matrix."""
        dev = self._device()
        n = self.num_nodes
        seed = self._next_seed(seed)
        uni = uniforms or {}
        Em = self.transition.E_marginal_host
        Y32 = ops.prior_Y(n, self.Y_marginal.float().contiguous(), seed=seed, u=uni.get("Y"))          # :831-833
        bits = ops.prior_E(n, float(Em[0]), float(Em[1]), dev, seed=seed, u=uni.get("E_prior"))        # :836-839
        x, x_onehot = ops.prior_X(n, self.X_marginal.float().contiguous(), seed=seed, u=uni.get(("X", 0)))  # :842-844
        for s in reversed(range(0, self.T)):                                                            # :847
            x, x_onehot, bits = self.denoise_step(s + 1, x, x_onehot, Y32, bits, seed, uniforms=uni, plan=plan,
                                                  group=group)
        return self._finish(x_onehot, Y32, bits, dense)


class ModelAsync(BaseModel):
    """model/diffusion.py:907-1410."""

    def __init__(self, T_X, T_E, X_marginal, Y_marginal, E_marginal, mlp_X_config, gnn_E_config, num_nodes):
        super().__init__(T=T_X, X_marginal=X_marginal, Y_marginal=Y_marginal, E_marginal=E_marginal, num_nodes=num_nodes)
        del self.T
        del self.noise_schedule
        self.T_X = T_X
        self.T_E = T_E
        device = X_marginal.device
        self.noise_schedule_X = NoiseSchedule(T_X, device)
        self.noise_schedule_E = NoiseSchedule(T_E, device)
        self.graph_encoder = GNNAsymm(num_attrs_X=self.num_attrs_X, num_classes_X=self.num_classes_X,
                                      num_classes_Y=self.num_classes_Y, num_classes_E=self.num_classes_E,
                                      mlp_X_config=mlp_X_config, gnn_E_config=gnn_E_config)
        self.graph_encoder.pred_X._sparse_X = self.graph_encoder.pred_E.gnn_encoder._sparse_X = self._sparse_X

    def table_X(self, t):
        ns = self.noise_schedule_X
        return self.transition.table_X(ns.alphas_host[t], ns.alpha_bars_host[t - 1], ns.alpha_bars_host[t])

    def table_E(self, t):
        ns = self.noise_schedule_E
        return self.transition.table_E(ns.alphas_host[t], ns.alpha_bars_host[t - 1], ns.alpha_bars_host[t])

    # ---- training / validation (model/diffusion.py:962-1293)
    def _apply_noise_X_state(self, X_one_hot_3d, t=None, seed=None, u=None):
        x0, _ = self._real_state(X_one_hot_3d, None)
        t_int, t_dev = self._draw_t(t, self.T_X, X_one_hot_3d.device)
        x_t, x_t_onehot = self._noise_X(x0, self.noise_schedule_X.alpha_bars_host[t_int], self._next_seed(seed), t_int, u=u)
        return t_int, t_dev / self.T_X, x_t, x_t_onehot

    def _apply_noise_E_state(self, E_one_hot, t=None, seed=None, u=None):
        _, bits0 = self._real_state(None, E_one_hot)
        t_int, t_dev = self._draw_t(t, self.T_E, E_one_hot.device)
        bits_t = self._noise_E(bits0, self.noise_schedule_E.alpha_bars_host[t_int], self._next_seed(seed), t_int, u=u)
        return t_int, t_dev / self.T_E, bits_t

    def apply_noise_X(self, X_one_hot_3d, t=None, *, seed=None, u=None):
        """model/diffusion.py:962-996: (t_float_X, X_t_one_hot)."""
        _, t_float, _, x_t_onehot = self._apply_noise_X_state(X_one_hot_3d, t, seed, u)
        return t_float, x_t_onehot

    def apply_noise_E(self, E_one_hot, t=None, *, seed=None, u=None):
        """model/diffusion.py:998-1033: (t_float_E, E_t (N,N) int64)."""
        _, t_float, bits_t = self._apply_noise_E_state(E_one_hot, t, seed, u)
        return t_float, ops.dense_from_bitmap(bits_t, self.num_nodes)

    def log_p_t(self, X_one_hot_3d, E_one_hot, Y, X_one_hot_2d, batch_src, batch_dst, batch_E_one_hot, t_X=None, t_E=None, *,
                seed=None, uniforms=None):
        """model/diffusion.py:1035-1094."""
        uni = uniforms or {}
        x0, _ = self._real_state(X_one_hot_3d, E_one_hot)
        _, t_float_X, _, x_t_onehot = self._apply_noise_X_state(X_one_hot_3d, t_X, seed, uni.get("X"))
        _, t_float_E, bits_t = self._apply_noise_E_state(E_one_hot, t_E, seed, uni.get("E"))
        A_t = ops.adjacency_from_bitmap(bits_t, self.num_nodes)
        ge = self.graph_encoder      # GNNAsymm.forward (model/gnn.py:545-550) in two calls, the attribute head folded as in ModelSync
        diff_X = ge.pred_X._run(t_float_X, x_t_onehot, Y, None, diff_head=True)
        logit_E = ge.pred_E(t_float_E, X_one_hot_2d, Y, A_t, batch_src, batch_dst)
        loss_X = ag.ce2_diff(diff_X, x0)
        loss_E = ag.softmax_ce2(logit_E, self._edge_targets(batch_E_one_hot))
        return loss_X, loss_E

    @torch.no_grad()
    def val_step(self, X_one_hot_3d, E_one_hot, Y, X_one_hot_2d, batch_src, batch_dst, batch_E_one_hot, *, seed=None):
        """model/diffusion.py:1196-1293."""
        dev = E_one_hot.device
        x0, bits0 = self._real_state(X_one_hot_3d, E_one_hot)
        e0 = self._edge_targets(batch_E_one_hot)
        src, dst = batch_src.contiguous(), batch_dst.contiguous()
        n = self.num_nodes
        dm_X, dm_E = [], []
        for t_sample in range(1, self.T_X + 1):
            t_int, t_float, x_t, x_t_onehot = self._apply_noise_X_state(X_one_hot_3d, t_sample, seed)
            logit_X = self.graph_encoder.pred_X(t_float, x_t_onehot, Y)
            QtX, QsX = self._kl_tables(self.noise_schedule_X, t_int, self.transition.X_marginal_host, dev, T=self.T_X)
            dm_X.append(ops.denoise_match2(logit_X, x_t, x0, QtX, QsX).item())
        denoise_match_X = float(np.mean(dm_X)) * self.T_X
        for t_sample in range(1, self.T_E + 1):
            t_int, t_float, bits_t = self._apply_noise_E_state(E_one_hot, t_sample, seed)
            A_t = ops.adjacency_from_bitmap(bits_t, n)
            logit_E = self.graph_encoder.pred_E(t_float, X_one_hot_2d, Y, A_t, src, dst)
            e_t = ops.bitmap_lookup(bits_t, n, src, dst)
            QtE, QsE = self._kl_tables(self.noise_schedule_E, t_int, self.transition.E_marginal_host, dev, T=self.T_E)
            dm_E.append(ops.denoise_match2(logit_E, e_t, e0, QtE, QsE).item())
        denoise_match_E = float(np.mean(dm_E)) * self.T_E
        loss_X, loss_E = self.log_p_t(X_one_hot_3d, E_one_hot, Y, X_one_hot_2d, src, dst, batch_E_one_hot, 0, 0, seed=seed)
        return denoise_match_E, denoise_match_X, loss_E.item(), loss_X.item()

    @torch.no_grad()
    def attribute_step(self, t, x, x_onehot, Y32, seed, uniforms=None, plan=None, group=None):
        """Body of the loop at model/diffusion.py:1330-1358."""
        dev = x.device
        t_float = self._const(("tX", t, self.T_X), np.array([t / self.T_X], np.float32), dev)
        pred_X = self.graph_encoder.pred_X
        uX = uniforms.get(("X", t)) if uniforms else None
        return self._attribute_draw(self._logits_fn(pred_X, t_float, x_onehot, Y32, None, plan, group, x_u8=x), x,
                                    self.table_X(t), seed, t, u=uX, plan=plan, group=group)

    @torch.no_grad()
    def structure_step(self, t, x_onehot, Y32, bits_t, seed, uniforms=None, plan=None, group=None, reuse_X=False):
        """Body of the loop at model/diffusion.py:1379-1401.  reuse_X: x_onehot is unchanged since the previous structure
        step of this model (the steps condition on the final attributes), so its input projection is kept."""
        dev = x_onehot.device
        t_float = self._const(("tE", t, self.T_E), np.array([t / self.T_E], np.float32), dev)
        uE = uniforms.get(("E", t)) if uniforms else None
        _, bits_s = self._structure_step(self.graph_encoder.pred_E, t_float, x_onehot, Y32, bits_t, self.table_E(t), seed,
                                         t, u=uE, plan=plan, group=group, reuse_X=reuse_X)
        return bits_s

    @torch.no_grad()
    def sample(self, batch_size=32768, num_workers=4, *, seed=None, uniforms=None, dense=True, plan=None, group=None):
        """model/diffusion.py:1295-1410: attributes first (T_X steps, no graph), then structure (T_E steps)."""
        dev = self._device()
        n = self.num_nodes
        seed = self._next_seed(seed)
        uni = uniforms or {}
        Em = self.transition.E_marginal_host
        Y32 = ops.prior_Y(n, self.Y_marginal.float().contiguous(), seed=seed, u=uni.get("Y"))          # :1319-1321
        x, x_onehot = ops.prior_X(n, self.X_marginal.float().contiguous(), seed=seed, u=uni.get(("X", 0)))  # :1325-1327
        for s_X in reversed(range(0, self.T_X)):                                                        # :1330
            x, x_onehot = self.attribute_step(s_X + 1, x, x_onehot, Y32, seed, uniforms=uni, plan=plan, group=group)
        bits = ops.prior_E(n, float(Em[0]), float(Em[1]), dev, seed=seed, u=uni.get("E_prior"))        # :1362-1365
        for s_E in reversed(range(0, self.T_E)):                                                        # :1379
            # every structure step conditions on the final attributes: their input projection is computed once (s_E = T_E-1)
            bits = self.structure_step(s_E + 1, x_onehot, Y32, bits, seed, uniforms=uni, plan=plan, group=group,
                                       reuse_X=s_E != self.T_E - 1)
        return self._finish(x_onehot, Y32, bits, dense)
