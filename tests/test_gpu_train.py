"""GPU tests of the training / validation path: forward noising, log_p_t with the HIP backward
kernels, val_step.  Checked against (a) fixtures produced by the reference's own apply_noise / log_p_t / autograd /
val_step (tools/gen_golden_train.py -> tests/golden/train_*.npz), (b) plain PyTorch fp32 CPU references of each
operator, (c) the CPU oracle for the Philox-keyed integer work.

Tolerances (fp32, different but fixed summation orders): losses and validation terms rtol 1e-5; gradients
|d - ref| <= 2e-6 + 2e-4 * |ref| per element and relative Frobenius error <= 2e-5 per tensor.  Integer results (noised
adjacency / attributes) are bit-exact.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_util as gu
import oracle_lib as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def _gm():
    import graphmaker_amd as gm
    return gm


def _ag():
    from graphmaker_amd import autograd as ag
    return ag


def _ops():
    from graphmaker_amd import ops
    return ops


def _close(a, b, atol, rtol, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b)
    bad = err > atol + rtol * np.abs(b)
    assert not bad.any(), f"{what}: {bad.sum()} of {bad.size} off, max err {err.max():.3e} (ref max {np.abs(b).max():.3e})"


# ------------------------------------------------------------------ operators vs plain PyTorch fp32
@pytest.mark.parametrize("M,N,K", [(70, 50, 33), (256, 128, 64), (300, 129, 1000), (5, 2, 40000), (130, 130, 7000)])
def test_gemm_variants(dev, M, N, K):
    ag = _ag()
    g = torch.Generator().manual_seed(M * 7 + N)
    X = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g)
    ref = X @ W.T
    scale = float(ref.abs().max())
    Xd, Wd = X.to(dev), W.to(dev)
    out00 = ag.gemm(False, False, M, N, K, Xd, K, Wd, K)
    out01 = ag.gemm(False, True, M, N, K, Xd, K, Wd.T.contiguous(), N)
    out10 = ag.gemm(True, False, M, N, K, Xd.T.contiguous(), M, Wd, K)
    out11 = ag.gemm(True, True, M, N, K, Xd.T.contiguous(), M, Wd.T.contiguous(), N)
    for name, o in (("00", out00), ("01", out01), ("10", out10), ("11", out11)):
        _close(o.cpu().numpy(), ref.numpy(), 2e-5 * scale, 1e-5, f"gemm {name} {M}x{N}x{K}")
    # all four layouts walk k in the same order with the same split -> identical bits
    assert torch.equal(out00, out11) and torch.equal(out01, out10) and torch.equal(out00, out01)
    # epilogue: bias + mask + relu
    b = torch.randn(N, generator=g)
    mask = torch.randn(M, N, generator=g)
    o = ag.gemm(False, False, M, N, K, Xd, K, Wd, K, bias=b.to(dev), mask=mask.to(dev), relu=True)
    ref2 = torch.relu(torch.where(mask > 0, ref + b, torch.zeros(()))).numpy()
    _close(o.cpu().numpy(), ref2, 2e-5 * scale, 1e-5, "gemm epilogue")


@pytest.mark.parametrize("M,N,K", [(2, 128, 70001), (4, 100, 65539), (1, 7, 131072), (3, 128, 65536)])
def test_gemm_skinny_weight_gradient(dev, M, N, K):
    """dW of a layer with <= 4 outputs over a pair batch (k_gemm_tt_skinny): C = A^T B with a very long reduced dimension,
    plus the epilogue, against float64."""
    ag = _ag()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(K, M, generator=g)
    B = torch.randn(K, N, generator=g)
    ref = (A.double().T @ B.double())
    tol = 3e-6 * float(K) ** 0.5 * 4
    out = ag.gemm(True, True, M, N, K, A.to(dev), M, B.to(dev), N)
    _close(out.cpu().numpy(), ref.numpy(), tol, 1e-5, f"skinny dW {M}x{N}x{K}")
    again = ag.gemm(True, True, M, N, K, A.to(dev), M, B.to(dev), N)
    assert torch.equal(out, again)                                  # fixed summation order
    b = torch.randn(N, generator=g)
    o = ag.gemm(True, True, M, N, K, A.to(dev), M, B.to(dev), N, bias=b.to(dev), relu=True)
    _close(o.cpu().numpy(), torch.relu(ref + b.double()).numpy(), tol, 1e-5, "skinny dW epilogue")


@pytest.mark.parametrize("M,H,N", [(1000, 128, 2), (70001, 128, 2), (333, 64, 4), (65, 4, 1), (130, 1024, 3)])
def test_mlp2_hidden_grad_fused(dev, M, H, N):
    """gm_mlp2_hidden_grad: the masked hidden gradient equals gm_gemm(0,1) + mask bit for bit, its column sums equal the
    float64 sums within rounding; unsupported shapes report -1."""
    ag = _ag()
    from graphmaker_amd import _lib as L
    g = torch.Generator().manual_seed(M + H + N)
    gy = torch.randn(M, N, generator=g).to(dev)
    W2 = torch.randn(N, H, generator=g).to(dev)
    y1 = torch.relu(torch.randn(M, H, generator=g)).to(dev)
    want = ag.gemm(False, True, M, H, N, gy, N, W2, H, mask=y1)
    nbytes = L.load().gm_mlp2_hidden_grad_workspace_bytes(L.I64(M), L.I64(H), L.I64(N))
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    g1 = torch.empty(M, H, device=dev)
    db1 = torch.empty(H, device=dev)
    L.call("gm_mlp2_hidden_grad", L.I64(M), L.I64(H), L.I64(N), L.ptr(gy), L.ptr(W2), L.I64(H), L.ptr(y1), L.ptr(g1), L.ptr(db1),
           L.ptr(ws), L.I64(nbytes), L.stream_ptr())
    assert torch.equal(g1, want)
    ref = want.double().sum(0).cpu().numpy()
    _close(db1.cpu().numpy(), ref, 1e-5 * float(M) ** 0.5 * 4, 1e-5, "db1")
    g1b = torch.empty(M, H, device=dev)                                     # without the bias gradient: no workspace
    L.call("gm_mlp2_hidden_grad", L.I64(M), L.I64(H), L.I64(N), L.ptr(gy), L.ptr(W2), L.I64(H), L.ptr(y1), L.ptr(g1b), None, None,
           L.I64(0), L.stream_ptr())
    assert torch.equal(g1b, want)
    assert L.load().gm_mlp2_hidden_grad_workspace_bytes(L.I64(M), L.I64(H), L.I64(5)) == -1
    assert L.load().gm_mlp2_hidden_grad_workspace_bytes(L.I64(M), L.I64(96), L.I64(2)) == -1


def test_wide_input_gradient_through_the_forward_kernel(dev):
    """autograd._dy_w: dY W of a wide layer over many rows runs gm_linear on a transposed copy of W; it must give the bits
    of gm_gemm(0,1) where that kernel does not split the reduction (>= 512 output tiles; both then walk the reduced dimension
    in ascending order from 0), with and without the ReLU mask."""
    ag = _ag()
    g = torch.Generator().manual_seed(11)
    M, K, N = 8200, 1030, 1100
    gy = torch.randn(M, N, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) / N ** 0.5).to(dev)
    mask = torch.randn(M, K, generator=g).to(dev)
    for m in (None, mask):
        want = ag.gemm(False, True, M, K, N, gy, N, W, K, mask=m)
        got = ag._dy_w(gy, W, M, K, N, mask=m)
        assert torch.equal(got, want)
    ref = gy.double() @ W.double()
    _close(ag._dy_w(gy, W, M, K, N).cpu().numpy(), ref.cpu().numpy(), 2e-5 * float(ref.abs().max()), 1e-5, "dY W")
    small = ag._dy_w(gy[:100].contiguous(), W, 100, K, N)                 # below the thresholds: the general kernel
    assert torch.equal(small, ag.gemm(False, True, 100, K, N, gy[:100].contiguous(), N, W, K))


def test_colsum_and_relu_bwd(dev):
    ag = _ag()
    from graphmaker_amd import _lib as L
    g = torch.Generator().manual_seed(3)
    X = torch.randn(1000, 77, generator=g)
    _close(ag.colsum(X.to(dev)).cpu().numpy(), X.sum(0).numpy(), 1e-4, 1e-5, "colsum")
    # more than 65536 rows (the pair batch): taller blocks, ragged last block, 2 and 130 columns
    for M, N in ((70001, 2), (200003, 130)):
        big = torch.randn(M, N, generator=g)
        _close(ag.colsum(big.to(dev)).cpu().numpy(), big.double().sum(0).float().numpy(), 2e-3, 1e-4, f"colsum {M}x{N}")
    y = torch.randn(1000, 77, generator=g)
    out = torch.empty_like(X, device=dev)
    Xd, yd = X.to(dev), y.to(dev)
    L.call("gm_relu_bwd", L.I64(X.numel()), L.ptr(Xd), L.ptr(yd), L.ptr(out), L.stream_ptr())
    assert torch.equal(out.cpu(), torch.where(y > 0, X, torch.zeros(())))


@pytest.mark.parametrize("M,N,K", [(67, 32, 56), (300, 512, 592), (130, 16, 16)])
def test_update_block_forward_backward(dev, M, N, K):
    """Linear -> ReLU -> LayerNorm (model/gnn.py:27-38) against torch autograd on the CPU."""
    ag = _ag()
    g = torch.Generator().manual_seed(N)
    x = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    gamma = 1 + 0.1 * torch.randn(N, generator=g)
    beta = 0.1 * torch.randn(N, generator=g)
    go = torch.randn(M, N, generator=g)
    cpu = [t.clone().requires_grad_(True) for t in (x, W, b, gamma, beta)]
    ref = F.layer_norm(torch.relu(F.linear(cpu[0], cpu[1], cpu[2])), (N,), cpu[3], cpu[4], 1e-5)
    ref.backward(go)
    gpu = [t.to(dev).requires_grad_(True) for t in (x, W, b, gamma, beta)]
    out = ag.update_block(*gpu, 1e-5)
    out.backward(go.to(dev))
    _close(out.detach().cpu().numpy(), ref.detach().numpy(), 2e-5, 1e-5, "update_block forward")
    for name, a, r in zip(("dx", "dW", "db", "dgamma", "dbeta"), gpu, cpu):
        sc = float(r.grad.abs().max())
        _close(a.grad.cpu().numpy(), r.grad.numpy(), 3e-5 * sc, 1e-4, f"update_block {name}")


def test_linear_relu_backward(dev):
    ag = _ag()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(200, 40, generator=g)
    W = torch.randn(24, 40, generator=g)
    b = torch.randn(24, generator=g)
    go = torch.randn(200, 24, generator=g)
    for relu in (False, True):
        cpu = [t.clone().requires_grad_(True) for t in (x, W, b)]
        r = F.linear(*cpu)
        r = torch.relu(r) if relu else r
        r.backward(go)
        gpu = [t.to(dev).requires_grad_(True) for t in (x, W, b)]
        o = ag.linear(gpu[0], gpu[1], gpu[2], relu)
        o.backward(go.to(dev))
        _close(o.detach().cpu().numpy(), r.detach().numpy(), 2e-5, 1e-5, "linear forward")
        for name, a, c in zip(("dx", "dW", "db"), gpu, cpu):
            _close(a.grad.cpu().numpy(), c.grad.numpy(), 3e-5 * float(c.grad.abs().max()), 1e-4, f"linear {name} relu={relu}")


def _random_sym_bits(n, density, seed, diag=False):
    rs = np.random.RandomState(seed)
    up = np.triu(rs.uniform(size=(n, n)) < density, k=0 if diag else 1)
    return (up | up.T).astype(np.int64)


def test_spmm_mean_backward(dev):
    ag, ops = _ag(), _ops()
    n, d = 150, 48
    E = _random_sym_bits(n, 0.05, 4, diag=True)
    E[7, :] = 0
    E[:, 7] = 0                                                            # an isolated node
    adj = ops.adjacency_from_bitmap(ops.bitmap_from_dense(torch.from_numpy(E).to(dev)), n)
    g = torch.Generator().manual_seed(1)
    H = torch.randn(n, d, generator=g)
    go = torch.randn(n, d, generator=g)
    deg = torch.from_numpy(E.sum(1)).float()
    A = torch.from_numpy(E).float() / torch.where(deg > 0, deg, torch.ones(()))[:, None]
    Hc = H.clone().requires_grad_(True)
    (A @ Hc).backward(go)
    Hd = H.to(dev).requires_grad_(True)
    out = ag.spmm_mean(adj, Hd)
    out.backward(go.to(dev))
    _close(out.detach().cpu().numpy(), (A @ H).numpy(), 1e-5, 1e-5, "spmm forward")
    _close(Hd.grad.cpu().numpy(), Hc.grad.numpy(), 1e-5, 1e-5, "spmm backward")


def test_embedding_hadamard_ce_backward(dev):
    ag = _ag()
    g = torch.Generator().manual_seed(2)
    n, C, w, d, B = 90, 5, 24, 128, 700
    idx = torch.randint(0, C, (n,), generator=g)
    Wt = torch.randn(C, w, generator=g)
    go = torch.randn(n, w, generator=g)
    Wc = Wt.clone().requires_grad_(True)
    F.embedding(idx, Wc).backward(go)
    Wd = Wt.to(dev).requires_grad_(True)
    o = ag.embedding(Wd, idx.to(torch.int32).to(dev))
    o.backward(go.to(dev))
    assert torch.equal(o.detach().cpu(), Wt[idx])
    _close(Wd.grad.cpu().numpy(), Wc.grad.numpy(), 1e-5, 1e-5, "embedding backward")

    # widths on the float4 path (d/4 a power of two) and off it; nodes with far more than 64 batch entries (the backward
    # resolves 64 entries at a time) and nodes with none
    for n_h, d_h, B_h in ((n, d, B), (5, 128, 700), (40, 200, 333), (7, 20, 100), (3, 64, 1000), (300, 256, 50)):
        h = torch.randn(n_h, d_h, generator=g)
        src = torch.randint(0, n_h, (B_h,), generator=g)
        dst = torch.randint(0, n_h, (B_h,), generator=g)
        gz = torch.randn(B_h, d_h, generator=g)
        hc = h.clone().requires_grad_(True)
        (hc[src] * hc[dst]).backward(gz)
        hd = h.to(dev).requires_grad_(True)
        z = ag.hadamard_gather(hd, src.to(dev), dst.to(dev))
        z.backward(gz.to(dev))
        assert torch.equal(z.detach().cpu(), h[src] * h[dst])
        scale = max(1.0, float(hc.grad.abs().max()))
        _close(hd.grad.cpu().numpy(), hc.grad.numpy(), 2e-5 * scale, 1e-5, f"hadamard backward n={n_h} d={d_h} B={B_h}")

    logits = torch.randn(5000, 2, generator=g) * 3
    tgt = torch.randint(0, 2, (5000,), generator=g)
    lc = logits.clone().requires_grad_(True)
    ref = F.cross_entropy(lc, tgt)
    ref.backward()
    ld = logits.to(dev).requires_grad_(True)
    loss = ag.softmax_ce2(ld, tgt.to(torch.uint8).to(dev))
    (loss * 1.0).backward()
    assert abs(loss.item() - ref.item()) <= 1e-6 * max(1.0, abs(ref.item()))
    _close(ld.grad.cpu().numpy(), lc.grad.numpy(), 1e-9, 1e-5, "cross entropy backward")


def test_ce2_from_logit_difference_and_folded_head_backward(dev):
    """gm_ce2_diff == two-class cross entropy of [l0, l1] as a function of d = l1 - l0 (value and gradient, against torch in
    float64); autograd.head_diff: forward = W[1::2] - W[0::2], backward = (+g, -g) interleaved, exactly."""
    ag = _ag()
    g = torch.Generator().manual_seed(21)
    logits = torch.randn(7001, 2, generator=g) * 4
    logits[:5] = torch.tensor([[0.0, 0.0], [90.0, -90.0], [-90.0, 90.0], [1e-8, 0.0], [-3.0, -3.0]])
    tgt = torch.randint(0, 2, (7001,), generator=g)
    lc = logits.double().clone().requires_grad_(True)
    ref = F.cross_entropy(lc, tgt)
    ref.backward()
    d = (logits[:, 1] - logits[:, 0]).to(dev).requires_grad_(True)
    loss = ag.ce2_diff(d, tgt.to(torch.uint8).to(dev))
    (loss * 2.0).backward()
    d64 = (logits[:, 1] - logits[:, 0]).double().requires_grad_(True)       # the f32 difference as the exact input
    ref_d = F.cross_entropy(torch.stack([torch.zeros_like(d64), d64], 1), tgt)
    ref_d.backward()
    assert abs(loss.item() - ref_d.item()) <= 2e-6 * abs(ref_d.item())
    _close(d.grad.cpu().numpy(), 2.0 * d64.grad.numpy(), 1e-10, 2e-5, "d CE / d diff")
    assert abs(ref.item() - ref_d.item()) < 1e-6                                  # the two forms agree (up to the f32 subtraction)

    Fh, Hc = 37, 53
    W = torch.randn(2 * Fh, Hc, generator=g).to(dev).requires_grad_(True)
    b = torch.randn(2 * Fh, generator=g).to(dev).requires_grad_(True)
    Wd, bd = ag.head_diff(W, b)
    assert torch.equal(Wd, W[1::2] - W[0::2]) and torch.equal(bd, b[1::2] - b[0::2])
    gW, gb = torch.randn(Fh, Hc, generator=g).to(dev), torch.randn(Fh, generator=g).to(dev)
    ((Wd * gW).sum() + (bd * gb).sum()).backward()
    assert torch.equal(W.grad[1::2], gW) and torch.equal(W.grad[0::2], -gW)
    assert torch.equal(b.grad[1::2], gb) and torch.equal(b.grad[0::2], -gb)


def test_dropout_mask_is_reproducible(dev):
    ag = _ag()
    x = torch.ones(200000, device=dev, requires_grad=True)
    st = ag.DropoutState(seed=5)
    y = ag.dropout(x, 0.1, st)
    y.sum().backward()
    kept = (y > 0).float().mean().item()
    assert abs(kept - 0.9) < 5e-3
    assert torch.equal(y.detach(), x.grad)                                  # same mask, same 1/(1-p) scale
    assert torch.allclose(y.detach()[y > 0], torch.tensor(1 / 0.9, device=dev))
    y2 = ag.dropout(x, 0.1, st)
    assert not torch.equal(y2.detach(), y.detach())                         # a fresh mask per application
    st2 = ag.DropoutState(seed=5)
    assert torch.equal(ag.dropout(x, 0.1, st2).detach(), y.detach())        # and the same one for the same (seed, call)


# ------------------------------------------------------------------ model-level helpers
def _model(cfg, dev, train=True):
    gm = _gm()
    Xm, Ym, Em = gu.marginals(cfg)
    sdX, sdE = gu.make_checkpoint(cfg)
    kw = dict(X_marginal=torch.from_numpy(Xm).to(dev), Y_marginal=torch.from_numpy(Ym).to(dev),
              E_marginal=torch.from_numpy(Em).to(dev), gnn_E_config=cfg["gnn_E"], num_nodes=cfg["N"])
    if cfg["variant"] == "Sync":
        model = gm.ModelSync(T=cfg["T"], gnn_X_config=cfg["gnn_X"], **kw)
    else:
        model = gm.ModelAsync(T_X=cfg["T_X"], T_E=cfg["T_E"], mlp_X_config=cfg["mlp_X"], **kw)
    model.graph_encoder.pred_X.load_state_dict({k: torch.from_numpy(v) for k, v in sdX.items()})
    model.graph_encoder.pred_E.load_state_dict({k: torch.from_numpy(v) for k, v in sdE.items()})
    model = model.to(dev)
    return model.train() if train else model.eval()


def _data(cfg, dev, B=1500):
    x0, Y, E0 = gu.real_graph(cfg)
    N = cfg["N"]
    X3 = F.one_hot(torch.from_numpy(x0.T.astype(np.int64)), 2).float().to(dev)
    X2 = torch.transpose(X3, 0, 1).reshape(N, -1).contiguous()
    E1h = F.one_hot(torch.from_numpy(E0), 2).float().to(dev)
    bdst, bsrc = (torch.from_numpy(a).to(dev) for a in gu.train_batch(cfg, B=B))
    return dict(x0=x0, Y=torch.from_numpy(Y).to(dev), E0=E0, X3=X3, X2=X2, E1h=E1h, bdst=bdst, bsrc=bsrc,
                bE=E1h[bdst, bsrc])


def _uni(cfg, dev, t_E, t_X=None):
    t_X = t_E if t_X is None else t_X
    return {"E": torch.from_numpy(gu.train_uniforms(cfg, "E", t_E)).to(dev),
            "X": torch.from_numpy(gu.train_uniforms(cfg, "X", t_X)).to(dev)}


def _nz(E):
    return np.argwhere(np.asarray(E) != 0).astype(np.int32)


def _load_train(name):
    z = np.load(gu.golden_path("train_" + name), allow_pickle=False)
    return {k: z[k] for k in z.files}


# ------------------------------------------------------------------ forward noising: bit-exact
def test_converters_roundtrip(dev):
    ops = _ops()
    cfg = gu.CASES["sync_small"]
    d = _data(cfg, dev)
    assert np.array_equal(ops.x_from_onehot3d(d["X3"]).cpu().numpy(), d["x0"])
    bits = ops.bitmap_from_onehot(d["E1h"])
    assert np.array_equal(ops.dense_from_bitmap(bits, cfg["N"]).cpu().numpy(), d["E0"])


@pytest.mark.parametrize("name", gu.TRAIN_CASES)
def test_apply_noise_matches_reference_under_shared_uniforms(dev, name):
    cfg = gu.CASES[name]
    gold = _load_train(name)
    model, d = _model(cfg, dev), _data(cfg, dev)
    N, Fa = cfg["N"], cfg["F"]
    if cfg["variant"] == "Sync":
        for t in range(cfg["T"] + 1):
            t_float, X_t, E_t = model.apply_noise(d["X3"], d["E1h"], torch.tensor([t], device=dev), uniforms=_uni(cfg, dev, t))
            assert abs(t_float.item() - t / cfg["T"]) < 1e-7
            assert np.array_equal(_nz(E_t.cpu().numpy()), gold[f"t{t}_E_t_nz"]), f"E_t differs at t={t}"
            assert np.array_equal(X_t.reshape(N, Fa, 2)[:, :, 1].cpu().numpy().astype(np.uint8), gold[f"t{t}_x_t"])
    else:
        for t in range(min(cfg["T_X"], cfg["T_E"]) + 1):
            u = _uni(cfg, dev, t)
            _, X_t = model.apply_noise_X(d["X3"], torch.tensor([t], device=dev), u=u["X"])
            _, E_t = model.apply_noise_E(d["E1h"], torch.tensor([t], device=dev), u=u["E"])
            assert np.array_equal(_nz(E_t.cpu().numpy()), gold[f"tE{t}_E_t_nz"])
            assert np.array_equal(X_t.reshape(N, Fa, 2)[:, :, 1].cpu().numpy().astype(np.uint8), gold[f"tX{t}_x_t"])


def test_noise_philox_matches_oracle(dev):
    ops = _ops()
    from graphmaker_amd.diffusion import _transition
    cfg = gu.CASES["sync_mid"]
    n, Fa = cfg["N"], cfg["F"]
    x0, _, E0 = gu.real_graph(cfg)
    Xm, _, Em = gu.marginals(cfg)
    Q = _transition(0.37, Em).reshape(4)
    QX = _transition(0.37, Xm).reshape(-1, 4)
    bits0 = ops.bitmap_from_dense(torch.from_numpy(E0).to(dev))
    got = ops.noise_E(n, bits0, Q, seed=77, step=2).cpu().numpy().view(np.uint32)
    want = orc.noise_E(n, bits0.cpu().numpy().view(np.uint32), Q, seed=77, step=2)
    assert np.array_equal(got, want)
    assert 0 < got.sum()
    x_t, oh = ops.noise_X(torch.from_numpy(x0).to(dev), torch.from_numpy(QX).to(dev), seed=77, step=2)
    assert np.array_equal(x_t.cpu().numpy(), orc.noise_X(x0, QX, seed=77, step=2))
    assert np.array_equal(oh.reshape(n, Fa, 2)[:, :, 1].cpu().numpy().astype(np.uint8), x_t.cpu().numpy())


# ------------------------------------------------------------------ log_p_t: losses and gradients vs the reference
def _log_p_t(model, cfg, d, t, dev):
    tt = torch.tensor([t], device=dev)
    if cfg["variant"] == "Sync":
        return model.log_p_t(d["X3"], d["E1h"], d["Y"], d["bsrc"], d["bdst"], d["bE"], tt, uniforms=_uni(cfg, dev, t))
    return model.log_p_t(d["X3"], d["E1h"], d["Y"], d["X2"], d["bsrc"], d["bdst"], d["bE"], tt, tt, uniforms=_uni(cfg, dev, t))


@pytest.mark.parametrize("name", gu.TRAIN_CASES)
def test_log_p_t_losses_and_gradients_match_reference(dev, name):
    cfg = gu.CASES[name]
    gold = _load_train(name)
    model, d = _model(cfg, dev), _data(cfg, dev)
    for t, (lX_ref, lE_ref) in enumerate(gold["losses"]):
        lX, lE = _log_p_t(model, cfg, d, t, dev)
        assert abs(lX.item() - lX_ref) <= 1e-5 * abs(lX_ref), (t, lX.item(), lX_ref)
        assert abs(lE.item() - lE_ref) <= 1e-5 * abs(lE_ref), (t, lE.item(), lE_ref)
    t = int(gold["grad_t"][0])
    model.zero_grad()
    lX, lE = _log_p_t(model, cfg, d, t, dev)
    (lX + lE).backward()
    checked = 0
    for pre, mod in (("X", model.graph_encoder.pred_X), ("E", model.graph_encoder.pred_E)):
        for k, p in mod.named_parameters():
            ref = gold[f"grad_{pre}.{k}"].astype(np.float64)
            got = p.grad.cpu().numpy().astype(np.float64)
            assert got.shape == ref.shape, k
            _close(got, ref, 2e-6, 2e-4, f"grad {pre}.{k}")
            nrm = np.linalg.norm(ref)
            if nrm > 0:
                assert np.linalg.norm(got - ref) <= 2e-5 * nrm + 1e-9, f"grad {pre}.{k}: rel err {np.linalg.norm(got - ref) / nrm:.2e}"
            checked += 1
    assert checked == sum(1 for _ in model.graph_encoder.parameters())


@pytest.mark.parametrize("name", ["cora_sync", "cora_async"])
def test_cora_size_training_step_matches_reference(dev, name):
    """BASELINE.json configs[0] / [1] shape on the training path: the reference's own log_p_t at every
    t, autograd backward at the last interior t and val_step at N=2708, F=1433 with the hidden sizes and batch size of
    configs/cora/train_{Sync,Async}.yaml (fixtures train_cora_sync / train_cora_async, tools/gen_golden_train.py; the
    reference's own run-to-run spread on these gradients is 4e-7).  Losses and validation terms rtol 1e-5; of every parameter
    gradient the L2 norm (rel 2e-5) and 2048 fixed entries (2e-6 + 2e-4 |ref|)."""
    cfg = gu.LARGE_CASES[name]
    gold = _load_train(name)
    model, d = _model(cfg, dev), _data(cfg, dev, B=gu.LARGE_TRAIN_BATCH)
    for t, (lX_ref, lE_ref) in enumerate(gold["losses"]):
        lX, lE = _log_p_t(model, cfg, d, t, dev)
        assert abs(lX.item() - lX_ref) <= 1e-5 * abs(lX_ref), (t, lX.item(), lX_ref)
        assert abs(lE.item() - lE_ref) <= 1e-5 * abs(lE_ref), (t, lE.item(), lE_ref)
    t = int(gold["grad_t"][0])
    model.zero_grad()
    lX, lE = _log_p_t(model, cfg, d, t, dev)
    (lX + lE).backward()
    checked = 0
    for pre, mod in (("X", model.graph_encoder.pred_X), ("E", model.graph_encoder.pred_E)):
        for k, p in mod.named_parameters():
            got = p.grad.cpu().numpy().astype(np.float64).reshape(-1)
            idx = gu.grad_sample_index(cfg, f"{pre}.{k}", got.size)
            _close(got[idx], gold[f"gsamp_{pre}.{k}"], 2e-6, 2e-4, f"grad {pre}.{k} (sampled entries)")
            nrm = float(gold[f"gnorm_{pre}.{k}"][0])
            # (absolute floor 1e-8: the time-embedding gradients are sums over all nodes that cancel to ~4e-5, and their last digits
            # follow the summation order)
            assert abs(np.linalg.norm(got) - nrm) <= 2e-5 * nrm + 1e-8, f"grad {pre}.{k}: norm {np.linalg.norm(got):.6e} vs {nrm:.6e}"
            checked += 1
    assert checked == sum(1 for _ in model.graph_encoder.parameters())
    model = model.eval()
    nE, nX = model._noise_E, model._noise_X
    model._noise_E = lambda bits0, ab, seed, step, u=None: nE(bits0, ab, seed, step,
                                                              u=torch.from_numpy(gu.train_uniforms(cfg, "E", step)).to(dev))
    model._noise_X = lambda x0, ab, seed, step, u=None: nX(x0, ab, seed, step,
                                                           u=torch.from_numpy(gu.train_uniforms(cfg, "X", step)).to(dev))
    if cfg["variant"] == "Sync":
        v = model.val_step(d["X3"], d["E1h"], d["Y"], d["bsrc"], d["bdst"], d["bE"])
    else:
        v = model.val_step(d["X3"], d["E1h"], d["Y"], d["X2"], d["bsrc"], d["bdst"], d["bE"])
    for got, ref, what in zip(v, gold["val_step"], ("denoise_match_E", "denoise_match_X", "log_p_0_E", "log_p_0_X")):
        assert abs(got - ref) <= 1e-5 * abs(ref) + 1e-9, (what, got, ref)


@pytest.mark.parametrize("name", gu.TRAIN_CASES)
def test_val_step_matches_reference(dev, name):
    cfg = gu.CASES[name]
    gold = _load_train(name)
    model, d = _model(cfg, dev, train=False), _data(cfg, dev)
    sync = cfg["variant"] == "Sync"

    # feed the fixture's uniforms: val_step draws through the same internal noising calls, one per t
    nE, nX = model._noise_E, model._noise_X
    model._noise_E = lambda bits0, ab, seed, step, u=None: nE(bits0, ab, seed, step,
                                                              u=torch.from_numpy(gu.train_uniforms(cfg, "E", step)).to(dev))
    model._noise_X = lambda x0, ab, seed, step, u=None: nX(x0, ab, seed, step,
                                                           u=torch.from_numpy(gu.train_uniforms(cfg, "X", step)).to(dev))
    if sync:
        v = model.val_step(d["X3"], d["E1h"], d["Y"], d["bsrc"], d["bdst"], d["bE"])
    else:
        v = model.val_step(d["X3"], d["E1h"], d["Y"], d["X2"], d["bsrc"], d["bdst"], d["bE"])
    for got, ref, what in zip(v, gold["val_step"], ("denoise_match_E", "denoise_match_X", "log_p_0_E", "log_p_0_X")):
        assert abs(got - ref) <= 1e-5 * abs(ref) + 1e-9, (what, got, ref)


# ------------------------------------------------------------------ a short training run
def test_training_steps_are_reproducible_and_reduce_the_loss(dev):
    cfg = gu.CASES["sync_small"]
    d = _data(cfg, dev)

    def run():
        model = _model(cfg, dev)
        opt_X = torch.optim.AdamW(model.graph_encoder.pred_X.parameters(), lr=1e-3, weight_decay=0, amsgrad=True)
        opt_E = torch.optim.AdamW(model.graph_encoder.pred_E.parameters(), lr=3e-4, weight_decay=0, amsgrad=True)
        hist = []
        for it in range(12):                                                 # train_sync.py:94-117
            lX, lE = model.log_p_t(d["X3"], d["E1h"], d["Y"], d["bsrc"], d["bdst"], d["bE"], torch.tensor([it % 4], device=dev),
                                   seed=100 + it)
            opt_X.zero_grad()
            opt_E.zero_grad()
            (lX + lE).backward()
            torch.nn.utils.clip_grad_norm_(model.graph_encoder.pred_X.parameters(), 10)
            torch.nn.utils.clip_grad_norm_(model.graph_encoder.pred_E.parameters(), 10)
            opt_X.step()
            opt_E.step()
            hist.append((lX.item(), lE.item()))
        return hist, [p.detach().clone() for p in model.graph_encoder.parameters()]

    h1, p1 = run()
    h2, p2 = run()
    assert h1 == h2, "two identical runs must give identical losses"
    assert all(torch.equal(a, b) for a, b in zip(p1, p2)), "two identical runs must give identical parameters"
    assert np.mean([a + b for a, b in h1[-4:]]) < np.mean([a + b for a, b in h1[:4]]), h1


def test_tape_forward_equals_fused_forward(dev):
    """The operator-by-operator forward used under autograd and the fused gm_tower_forward give the same bits."""
    cfg = gu.CASES["sync_small"]
    model, d = _model(cfg, dev, train=False), _data(cfg, dev)
    ops = _ops()
    n = cfg["N"]
    x0 = torch.from_numpy(d["x0"]).to(dev)
    oh = ops.onehot_from_x(x0)
    A = ops.adjacency_from_bitmap(ops.bitmap_from_dense(torch.from_numpy(d["E0"]).to(dev)), n)
    t = torch.tensor([2 / 3], device=dev)
    for tower in (model.graph_encoder.pred_X, model.graph_encoder.pred_E.gnn_encoder):
        with torch.no_grad():
            fused = tower(t, oh, d["Y"], A)
        taped = tower(t, oh, d["Y"], A)
        assert taped.requires_grad and not fused.requires_grad
        assert torch.equal(taped.detach(), fused)


# ------------------------------------------------------------------ the two callers end to end (local files)
def test_cli_train_then_sample(dev, tmp_path):
    """python -m graphmaker_amd train ... -> reference-format .pth -> python -m graphmaker_amd sample ..."""
    import os
    from graphmaker_amd import __main__ as cli
    from graphmaker_amd import data as gdata
    from test_host_io_cpu import SYNC_YAML
    cfg = gu.CASES["sync_small"]
    x0, Y, E0 = gu.real_graph(cfg)
    src, dst = np.nonzero(E0)
    gpath, ypath = os.path.join(tmp_path, "g.npz"), os.path.join(tmp_path, "train_Sync.yaml")
    gdata.save_graph_npz(gpath, x0, Y, src, dst)
    with open(ypath, "w") as fh:
        fh.write(SYNC_YAML)
    out = os.path.join(tmp_path, "cpts")
    path = cli.main(["train", "--config", ypath, "--data", gpath, "--out", out, "--dataset", "toy"])
    state = torch.load(path, weights_only=True)
    assert state["dataset"] == "toy" and np.isfinite(state["best_val_nll"])
    sdir = os.path.join(tmp_path, "samples")
    cli.main(["sample", "--model_path", path, "--data", gpath, "--num_samples", "2", "--out", sdir])
    for k in range(2):
        g = gdata.load_graph_npz(os.path.join(sdir, f"sample_{k}.npz"))
        assert g["num_nodes"] == cfg["N"] and g["src"].numel() % 2 == 0
