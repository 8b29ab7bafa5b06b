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
    """model/diffusion.py:143-472 (inference half)."""

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
        self._pair_ws = None
        self._prof = None          # None, or dict phase -> [(start_event, end_event)] (bench.py / profiling)

    class _Phase:
        """Brackets a phase of the step with events on torch's current stream (the stream every library call is
        enqueued on) when profiling is on; free otherwise."""

        def __init__(self, model, name):
            self.rec = model._prof
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

    def _buf(self, name, shape, dtype, device):
        """Step-persistent scratch tensors of the sharded path (buffers that collectives touch are kept, not re-made)."""
        cache = self.__dict__.setdefault("_bufs", {})
        key = (name, tuple(shape), dtype, device)
        if key not in cache:
            cache[key] = torch.empty(shape, dtype=dtype, device=device)
        return cache[key]

    def enable_phase_timing(self, on=True):
        self._prof = {} if on else None

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

    def _structure_step(self, pred_E, t_float, X_one_hot, Y32, bits_t, tab_E, seed, step, u=None, plan=None, group=None):
        """get_E_t (model/diffusion.py:428-472): returns (A_t, symmetric bitmap of E_s).
        plan: parallel.ShardPlan for the one-process-per-GPU decomposition (None = single GPU)."""
        n = self.num_nodes
        with self.phase("csr"):
            A_t = ops.adjacency_from_bitmap(bits_t, n)
        enc = pred_E.gnn_encoder
        with self.phase("tower_E"):
            if plan is None:
                h = enc(t_float, X_one_hot, Y32, A_t)
            else:
                h_pad = self._buf("h_pad", (plan.world * plan.node_rows, enc._dims.out_size), torch.float32,
                                  X_one_hot.device)
                enc._run_sharded(t_float, X_one_hot, Y32, A_t, plan, group, h_pad[:n], parallel.all_gather_node_rows)
        if plan is not None:
            with self.phase("allgather_h"):
                h = parallel.all_gather_node_rows(plan, h_pad, group)[:n]
        if self._pair_ws is None or self._pair_ws.device != h.device:
            self._pair_ws = torch.empty(256, dtype=torch.uint8, device=h.device)
        W1, b1, W2, b2 = pred_E.pair_params()
        ranges = [(0, n)] if plan is None else plan.pair_chunks()
        upper = torch.zeros((n, bits_t.shape[1]), dtype=torch.int32, device=h.device)
        with self.phase("pair"):
            for r0, r1 in ranges:
                if r1 > r0:
                    ops.pair_score_sample(h, W1, b1, W2, b2, bits_t, tab_E, seed=seed, step=step, u=u, row_begin=r0,
                                          row_end=r1, out_upper=upper, workspace=self._pair_ws)
        if plan is not None:
            with self.phase("allgather_E"):
                parallel.gather_upper_rows(plan, upper, group)
        with self.phase("symmetrize"):
            bits_s = ops.symmetrize_upper(upper, n)
        return A_t, bits_s

    def _logits_fn(self, pred_X, t_float, x_onehot, Y32, A_t, plan, group):
        """rows -> (n, 2F) attribute logits valid on `rows` (all rows on a single GPU)."""
        if plan is None:
            return lambda rows: pred_X._run(t_float, x_onehot, Y32, A_t)
        n = self.num_nodes

        def sharded(rows):
            out = self._buf("logits", (n, pred_X._dims.out_size), torch.float32, x_onehot.device)
            return pred_X._run_sharded(t_float, x_onehot, Y32, A_t, plan, group, out, parallel.all_gather_node_rows)
        return sharded

    def _attribute_draw(self, logits_fn, x, tab_X, seed, step, u=None, plan=None, group=None):
        """softmax + posterior + sample_X (model/diffusion.py:879-895) on this rank's node rows, then replicate.
        logits_fn(rows) -> (n, 2F) logits valid on `rows`."""
        n = self.num_nodes
        dev = x.device
        rows = None if plan is None else plan.node_range()
        with self.phase("tower_X"):
            logits = logits_fn(rows)
        tabX_dev = torch.from_numpy(tab_X).to(dev)
        if plan is None:
            with self.phase("attr"):
                return ops.attr_posterior_sample(logits, x, tabX_dev, seed=seed, step=step, u=u)
        x_pad = self._buf("x_pad", (plan.world * plan.node_rows, x.shape[1]), torch.uint8, dev)
        with self.phase("attr"):
            ops.attr_posterior_sample(logits, x, tabX_dev, seed=seed, step=step, u=u, rows=rows, x_s=x_pad[:n],
                                      want_onehot=False)
        with self.phase("allgather_X"):
            x_s = parallel.all_gather_node_rows(plan, x_pad, group)[:n].clone()      # x_pad is reused by the next step
        with self.phase("attr"):
            return x_s, ops.onehot_from_x(x_s)

    def _finish(self, x_onehot, Y32, bits, dense):
        n, Fa = self.num_nodes, self.num_attrs_X
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

    def step_tables(self, t):
        """Host tables of reverse step t (s = t-1): model/diffusion.py:852-858, 889-891."""
        ns = self.noise_schedule
        a_t, ab_s, ab_t = ns.alphas_host[t], ns.alpha_bars_host[t - 1], ns.alpha_bars_host[t]
        return self.transition.table_E(a_t, ab_s, ab_t), self.transition.table_X(a_t, ab_s, ab_t)

    @torch.no_grad()
    def denoise_step(self, t, x, x_onehot, Y32, bits_t, seed, uniforms=None, plan=None, group=None):
        """One reverse step t -> t-1 (the body of the loop at model/diffusion.py:847-896).
        State: x (N,F) uint8, x_onehot (N,2F) float, Y32 (N,) int32, bits_t symmetric bitmap.
        Returns (x_s, x_onehot_s, bits_s)."""
        dev = x.device
        n = self.num_nodes
        tab_E, tab_X = self.step_tables(t)
        t_float = torch.tensor([t / self.T], dtype=torch.float32).to(dev)      # :860
        uE = uniforms.get(("E", t)) if uniforms else None
        uX = uniforms.get(("X", t)) if uniforms else None
        A_t, bits_s = self._structure_step(self.graph_encoder.pred_E, t_float, x_onehot, Y32, bits_t, tab_E, seed, t,
                                           u=uE, plan=plan, group=group)
        pred_X = self.graph_encoder.pred_X                                      # A_t of E_t, not E_s (:875-878)
        x_s, oh_s = self._attribute_draw(self._logits_fn(pred_X, t_float, x_onehot, Y32, A_t, plan, group), x, tab_X,
                                         seed, t, u=uX, plan=plan, group=group)
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

    def table_X(self, t):
        ns = self.noise_schedule_X
        return self.transition.table_X(ns.alphas_host[t], ns.alpha_bars_host[t - 1], ns.alpha_bars_host[t])

    def table_E(self, t):
        ns = self.noise_schedule_E
        return self.transition.table_E(ns.alphas_host[t], ns.alpha_bars_host[t - 1], ns.alpha_bars_host[t])

    @torch.no_grad()
    def attribute_step(self, t, x, x_onehot, Y32, seed, uniforms=None, plan=None, group=None):
        """Body of the loop at model/diffusion.py:1330-1358."""
        dev = x.device
        t_float = torch.tensor([t / self.T_X], dtype=torch.float32).to(dev)
        pred_X = self.graph_encoder.pred_X
        uX = uniforms.get(("X", t)) if uniforms else None
        return self._attribute_draw(self._logits_fn(pred_X, t_float, x_onehot, Y32, None, plan, group), x, self.table_X(t),
                                    seed, t, u=uX, plan=plan, group=group)

    @torch.no_grad()
    def structure_step(self, t, x_onehot, Y32, bits_t, seed, uniforms=None, plan=None, group=None):
        """Body of the loop at model/diffusion.py:1379-1401."""
        dev = x_onehot.device
        t_float = torch.tensor([t / self.T_E], dtype=torch.float32).to(dev)
        uE = uniforms.get(("E", t)) if uniforms else None
        _, bits_s = self._structure_step(self.graph_encoder.pred_E, t_float, x_onehot, Y32, bits_t, self.table_E(t), seed,
                                         t, u=uE, plan=plan, group=group)
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
            bits = self.structure_step(s_E + 1, x_onehot, Y32, bits, seed, uniforms=uni, plan=plan, group=group)
        return self._finish(x_onehot, Y32, bits, dense)
