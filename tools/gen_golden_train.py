#!/usr/bin/env python3
"""Generate tests/golden/train_*.npz by RUNNING THE REFERENCE's training-side functions (this container only):
apply_noise / log_p_t (+ autograd backward) / val_step of model/diffusion.py on a synthetic real graph
(tests/golden_util.py real_graph) with the synthetic checkpoints of the sampling fixtures.

Seams as in tools/gen_golden.py: the torch-only dgl.sparse stand-in, and Tensor.multinomial swapped for the
threshold rule on host-supplied uniforms (golden_util.train_uniforms) so both sides consume the same draws.
Only reference OUTPUTS are stored: noised states, losses, gradients, validation terms.

Usage:  python tools/gen_golden_train.py [case ...]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F_

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import golden_util as gu  # noqa: E402
import gen_golden as gg  # noqa: E402


class Ctx:
    cfg = None
    t_E = 0
    t_X = 0


def _shim_multinomial(self, num_samples, replacement=False, *, generator=None):
    assert num_samples == 1
    p = self.detach().float()
    R, Cc = p.shape
    cfg = Ctx.cfg
    assert Cc == 2
    if R == cfg["N"] * cfg["N"]:
        u = gu.train_uniforms(cfg, "E", Ctx.t_E)
    elif R == cfg["N"] * cfg["F"]:
        u = gu.train_uniforms(cfg, "X", Ctx.t_X)
    else:
        raise RuntimeError(f"unexpected multinomial of {R} rows")
    u = torch.from_numpy(u)
    s = p[:, 0] + p[:, 1]
    return (u * s < p[:, 1]).long().unsqueeze(1)


def build(cfg, diffusion):
    Xm, Ym, Em = gu.marginals(cfg)
    sdX, sdE = gu.make_checkpoint(cfg)
    if cfg["variant"] == "Sync":
        model = diffusion.ModelSync(T=cfg["T"], X_marginal=torch.from_numpy(Xm), Y_marginal=torch.from_numpy(Ym),
                                    E_marginal=torch.from_numpy(Em), gnn_X_config=cfg["gnn_X"],
                                    gnn_E_config=cfg["gnn_E"], num_nodes=cfg["N"])
    else:
        model = diffusion.ModelAsync(T_X=cfg["T_X"], T_E=cfg["T_E"], X_marginal=torch.from_numpy(Xm),
                                     Y_marginal=torch.from_numpy(Ym), E_marginal=torch.from_numpy(Em),
                                     mlp_X_config=cfg["mlp_X"], gnn_E_config=cfg["gnn_E"], num_nodes=cfg["N"])
    model.graph_encoder.pred_X.load_state_dict(gg.to_t(sdX))
    model.graph_encoder.pred_E.load_state_dict(gg.to_t(sdE))
    return model


def run_case(name, diffusion):
    cfg = gu.CASES[name]
    Ctx.cfg = cfg
    N, F = cfg["N"], cfg["F"]
    sync = cfg["variant"] == "Sync"
    model = build(cfg, diffusion)
    model.train()                                                      # dropout is 0 in these cases
    x0, Y, E0 = gu.real_graph(cfg)
    X3 = F_.one_hot(torch.from_numpy(x0.T.astype(np.int64)), 2).float()          # (F,N,2)  data.py:65-71
    X2 = torch.transpose(X3, 0, 1).reshape(N, -1)                                 # (N,2F)   train_async.py
    E1h = F_.one_hot(torch.from_numpy(E0), 2).float()                             # (N,N,2)  data.py:73-76
    Yt = torch.from_numpy(Y)
    bdst, bsrc = (torch.from_numpy(a) for a in gu.train_batch(cfg))
    bE = E1h[bdst, bsrc]
    out = {}

    # wrap the noising entry points so the shim knows which step's uniforms to hand out
    if sync:
        orig = model.apply_noise

        def apply_noise(X, E, t=None):
            Ctx.t_E = Ctx.t_X = int(t.item())
            r = orig(X, E, t)
            out[f"t{Ctx.t_E}_E_t_nz"] = r[2].nonzero().numpy().astype(np.int32)
            out[f"t{Ctx.t_E}_x_t"] = r[1].reshape(N, F, 2)[:, :, 1].numpy().astype(np.uint8)
            return r
        model.apply_noise = apply_noise
        T_list = list(range(cfg["T"] + 1))
    else:
        oX, oE = model.apply_noise_X, model.apply_noise_E

        def apply_noise_X(X, t=None):
            Ctx.t_X = int(t.item())
            r = oX(X, t)
            out[f"tX{Ctx.t_X}_x_t"] = r[1].reshape(N, F, 2)[:, :, 1].numpy().astype(np.uint8)
            return r

        def apply_noise_E(E, t=None):
            Ctx.t_E = int(t.item())
            r = oE(E, t)
            out[f"tE{Ctx.t_E}_E_t_nz"] = r[1].nonzero().numpy().astype(np.int32)
            return r
        model.apply_noise_X, model.apply_noise_E = apply_noise_X, apply_noise_E
        T_list = list(range(min(cfg["T_X"], cfg["T_E"]) + 1))

    saved = torch.Tensor.multinomial
    torch.Tensor.multinomial = _shim_multinomial
    try:
        losses = []
        for t in T_list:
            model.zero_grad()
            tt = torch.LongTensor([t])
            if sync:
                lX, lE = model.log_p_t(X3, E1h, Yt, bsrc, bdst, bE, tt)
            else:
                lX, lE = model.log_p_t(X3, E1h, Yt, X2, bsrc, bdst, bE, tt, tt)
            losses.append([lX.item(), lE.item()])
            if t == T_list[-2]:                                        # gradients at one interior step
                (lX + lE).backward()
                out["grad_t"] = np.array([t], np.int32)
                for pre, mod in (("X", model.graph_encoder.pred_X), ("E", model.graph_encoder.pred_E)):
                    for k, p in mod.named_parameters():
                        out[f"grad_{pre}.{k}"] = p.grad.detach().numpy().astype(np.float32).copy()
        out["losses"] = np.array(losses, np.float64)
        model.eval()
        if sync:
            v = model.val_step(X3, E1h, Yt, bsrc, bdst, bE)
        else:
            v = model.val_step(X3, E1h, Yt, X2, bsrc, bdst, bE)
        out["val_step"] = np.array(v, np.float64)          # denoise_match_E, denoise_match_X, log_p_0_E, log_p_0_X
    finally:
        torch.Tensor.multinomial = saved
    path = os.path.join(gu.GOLDEN_DIR, f"train_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: wrote {path} ({os.path.getsize(path) / 1e3:.0f} kB) losses={losses} val={v}")


def run_large_case(name, diffusion):
    """A BASELINE.json-sized training step (gu.LARGE_CASES): the reference's log_p_t at every t, autograd backward at the last
    interior t and val_step at N=2708, F=1433 with the yaml's hidden sizes and batch size (configs/cora/train_{Sync,Async}.yaml).
    Stored: the losses, the validation terms, and of every parameter gradient its L2 norm and the values at 2048 fixed
    positions (golden_util.grad_sample_index) -- the full gradients are 80 MB."""
    cfg = dict(gu.LARGE_CASES[name])
    Ctx.cfg = cfg
    N, F = cfg["N"], cfg["F"]
    sync = cfg["variant"] == "Sync"
    model = build(cfg, diffusion)
    model.train()
    x0, Y, E0 = gu.real_graph(cfg)
    X3 = F_.one_hot(torch.from_numpy(x0.T.astype(np.int64)), 2).float()
    X2 = torch.transpose(X3, 0, 1).reshape(N, -1)
    E1h = F_.one_hot(torch.from_numpy(E0), 2).float()
    Yt = torch.from_numpy(Y)
    bdst, bsrc = (torch.from_numpy(a) for a in gu.train_batch(cfg, B=gu.LARGE_TRAIN_BATCH))
    bE = E1h[bdst, bsrc]
    out = {}
    if sync:
        orig = model.apply_noise

        def apply_noise(X, E, t=None):
            Ctx.t_E = Ctx.t_X = int(t.item())
            return orig(X, E, t)
        model.apply_noise = apply_noise
        T_list = list(range(cfg["T"] + 1))
    else:
        oX, oE = model.apply_noise_X, model.apply_noise_E

        def apply_noise_X(X, t=None):
            Ctx.t_X = int(t.item())
            return oX(X, t)

        def apply_noise_E(E, t=None):
            Ctx.t_E = int(t.item())
            return oE(E, t)
        model.apply_noise_X, model.apply_noise_E = apply_noise_X, apply_noise_E
        T_list = list(range(min(cfg["T_X"], cfg["T_E"]) + 1))
    saved = torch.Tensor.multinomial
    torch.Tensor.multinomial = _shim_multinomial
    try:
        losses = []
        for t in T_list:
            model.zero_grad()
            tt = torch.LongTensor([t])
            if sync:
                lX, lE = model.log_p_t(X3, E1h, Yt, bsrc, bdst, bE, tt)
            else:
                lX, lE = model.log_p_t(X3, E1h, Yt, X2, bsrc, bdst, bE, tt, tt)
            losses.append([lX.item(), lE.item()])
            if t == T_list[-2]:
                (lX + lE).backward()
                out["grad_t"] = np.array([t], np.int32)
                for pre, mod in (("X", model.graph_encoder.pred_X), ("E", model.graph_encoder.pred_E)):
                    for k, p in mod.named_parameters():
                        g = p.grad.detach().numpy().astype(np.float64).reshape(-1)
                        out[f"gnorm_{pre}.{k}"] = np.array([np.linalg.norm(g)])
                        out[f"gsamp_{pre}.{k}"] = g[gu.grad_sample_index(cfg, f"{pre}.{k}", g.size)].astype(np.float32)
        out["losses"] = np.array(losses, np.float64)
        model.eval()
        v = model.val_step(X3, E1h, Yt, bsrc, bdst, bE) if sync else model.val_step(X3, E1h, Yt, X2, bsrc, bdst, bE)
        out["val_step"] = np.array(v, np.float64)
    finally:
        torch.Tensor.multinomial = saved
    path = os.path.join(gu.GOLDEN_DIR, f"train_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: wrote {path} ({os.path.getsize(path) / 1e3:.0f} kB) losses={losses} val={v}")


def run_preprocess(name):
    """The reference's data.py preprocess() on the synthetic real graph (a stand-in object with the three DGLGraph
    members it touches: ndata, num_nodes(), edges()); `dgl` / `dgl.data` are empty stand-in modules for the imports."""
    import importlib.util
    import types
    cfg = gu.CASES[name]
    dgl = sys.modules.get("dgl") or types.ModuleType("dgl")
    dd = types.ModuleType("dgl.data")
    for cls in ("AmazonCoBuyPhotoDataset", "AmazonCoBuyComputerDataset", "CoraGraphDataset"):
        setattr(dd, cls, object)
    dgl.data = dd
    sys.modules["dgl"], sys.modules["dgl.data"] = dgl, dd
    spec = importlib.util.spec_from_file_location("gmref_data", os.path.join(gg.REF, "data.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    x0, Y, E0 = gu.real_graph(cfg)
    src, dst = np.nonzero(E0)

    class G:
        ndata = {"feat": torch.from_numpy(x0).float(), "label": torch.from_numpy(Y)}

        def num_nodes(self):
            return cfg["N"]

        def edges(self):
            return torch.from_numpy(src), torch.from_numpy(dst)

    X3, Yo, E1h, Xm, Ym, Em, XcY = mod.preprocess(G())
    path = os.path.join(gu.GOLDEN_DIR, f"preprocess_{name}.npz")
    np.savez_compressed(path, X_marginal=Xm.numpy(), Y_marginal=Ym.numpy(), E_marginal=Em.numpy(),
                        X_cond_Y_marginals=XcY.numpy(), X_one_hot_sum=X3.sum((1,)).numpy(), E_ones=E1h[:, :, 1].sum().numpy())
    print(f"{name}: wrote {path}")


def main():
    gnn, diffusion = gg.load_reference()
    names = sys.argv[1:] or gu.TRAIN_CASES
    if not sys.argv[1:]:
        run_preprocess("sync_small")
    for name in names:
        if name in gu.LARGE_CASES:
            run_large_case(name, diffusion)
        else:
            run_case(name, diffusion)


if __name__ == "__main__":
    main()
