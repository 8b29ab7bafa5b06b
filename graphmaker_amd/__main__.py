"""Command line of the two callers either side of the denoising step (reference: sample.py, train_sync.py,
train_async.py), for LOCAL files only (no hub / dataset downloads):

    python -m graphmaker_amd sample --model_path M.pth --data G.npz [--num_samples 10] [--out DIR] [--seed 0]
    python -m graphmaker_amd train  --config train_Sync.yaml --data G.npz [--dataset NAME] [--out DIR] [--epochs N]

`sample` mirrors sample.py:26-93 up to the evaluator (out of scope): it writes each generated graph as
DIR/sample_<k>.npz (feat, label, src, dst).  `train` mirrors train_sync.py / train_async.py (AdamW per tower, gradient
clipping, ReduceLROnPlateau, validation every val_every_epochs, early stopping, best checkpoint in the reference's
.pth format); wandb logging becomes JSON lines on stdout.
"""
import argparse
import json
import os
import sys
import time
from copy import deepcopy

import numpy as np
import torch

from . import checkpoint as ckpt
from . import data as gdata


def _device():
    if not torch.cuda.is_available():
        raise SystemExit("graphmaker_amd needs a ROCm device (there is no CPU path)")
    return torch.device("cuda:0")


def cmd_sample(args):
    dev = _device()
    g = gdata.load_graph_npz(args.data)
    _, _, _, Xm, Ym, Em, _ = gdata.preprocess(g, with_E_one_hot=False)
    model, state = ckpt.load_checkpoint(args.model_path, Xm.to(dev), Ym.to(dev), Em.to(dev), g["num_nodes"])
    variant = state["train_yaml_data"]["meta_data"]["variant"]
    print(f"Loaded GraphMaker-{variant} model trained on {state.get('dataset')}; val nll {state.get('best_val_nll')}")
    torch.manual_seed(args.seed)                                                   # setup_utils.set_seed, sample.py:84
    os.makedirs(args.out, exist_ok=True)
    for k in range(args.num_samples):
        t0 = time.perf_counter()
        X_0_one_hot, Y_0_one_hot, edges = model.sample(dense=False)                # sample.py:87
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        i, j = edges[0].cpu().numpy(), edges[1].cpu().numpy()
        gdata.save_graph_npz(os.path.join(args.out, f"sample_{k}.npz"), X_0_one_hot[:, :, 1].T.cpu().numpy(),
                             Y_0_one_hot.argmax(1).cpu().numpy(), np.concatenate([i, j]), np.concatenate([j, i]))
        print(json.dumps({"sample": k, "seconds": round(dt, 4), "undirected_edges": int(i.size),
                          "attributes_on": int(X_0_one_hot[:, :, 1].sum().item())}))


def cmd_train(args):
    dev = _device()
    y = ckpt.load_train_yaml(args.config)
    variant = y["meta_data"]["variant"]
    g = gdata.load_graph_npz(args.data)
    X3, Y, E1h, Xm, Ym, Em, _ = gdata.preprocess(g)
    X3, Y, E1h = X3.to(dev), Y.to(dev), E1h.to(dev)
    N = g["num_nodes"]
    X2 = torch.transpose(X3, 0, 1).reshape(N, -1).contiguous()                     # train_async.py:39-41
    model = ckpt.build_model(y, Xm.to(dev), Ym.to(dev), Em.to(dev), N)
    dst, src = torch.triu_indices(N, N, offset=1, device=dev)                      # train_sync.py:44-46
    tc = y["train"]
    torch.manual_seed(args.seed)
    opt = {k: torch.optim.AdamW(getattr(model.graph_encoder, f"pred_{k}").parameters(), **y[f"optimizer_{k}"])
           for k in ("X", "E")}
    sch_kw = {k: v for k, v in y["lr_scheduler"].items() if k != "verbose"}
    sch = {k: torch.optim.lr_scheduler.ReduceLROnPlateau(opt[k], mode="min", **sch_kw) for k in opt}
    best = {k: dict(epoch=0, sd=deepcopy(getattr(model.graph_encoder, f"pred_{k}").state_dict()), nll=float("inf"),
                    log_p_0=float("inf"), dm=float("inf")) for k in ("X", "E")}
    os.makedirs(args.out, exist_ok=True)
    name = f"{variant}_T{y['diffusion']['T']}" if variant == "Sync" else \
        f"{variant}_TX{y['diffusion']['T_X']}_TE{y['diffusion']['T_E']}"
    path = os.path.join(args.out, name + ".pth")
    P = dst.numel()

    def step_args(bd, bs):
        bE = E1h[bd, bs]
        return (X3, E1h, Y, bs, bd, bE) if variant == "Sync" else (X3, E1h, Y, X2, bs, bd, bE)

    patient = 0
    epochs = args.epochs if args.epochs is not None else tc["num_epochs"]
    for epoch in range(epochs):
        model.train()
        perm = torch.randperm(P, device=dev)                                       # DataLoader(shuffle=True)
        t0 = time.perf_counter()
        for b0 in range(0, P, tc["batch_size"]):
            idx = perm[b0:b0 + tc["batch_size"]]
            loss_X, loss_E = model.log_p_t(*step_args(dst[idx], src[idx]))         # train_sync.py:98-103
            for o in opt.values():
                o.zero_grad()
            (loss_X + loss_E).backward()
            if tc.get("max_grad_norm") is not None:
                for k in opt:
                    torch.nn.utils.clip_grad_norm_(getattr(model.graph_encoder, f"pred_{k}").parameters(), tc["max_grad_norm"])
            for o in opt.values():
                o.step()
            print(json.dumps({"epoch": epoch, "train/loss_X": loss_X.item(), "train/loss_E": loss_E.item()}))
        torch.cuda.synchronize()
        print(json.dumps({"epoch": epoch, "epoch_seconds": round(time.perf_counter() - t0, 3)}))
        if (epoch + 1) % tc["val_every_epochs"] != 0:
            continue
        model.eval()
        patient += 1
        acc = []
        for b0 in range(0, P, tc["val_batch_size"]):                               # train_sync.py:133-148
            acc.append(model.val_step(*step_args(dst[b0:b0 + tc["val_batch_size"]], src[b0:b0 + tc["val_batch_size"]])))
        dm_E, dm_X, l0_E, l0_X = (float(np.mean(c)) for c in zip(*acc))
        val = {"X": dm_X + l0_X, "E": dm_E + l0_E}
        save = False
        for k in ("X", "E"):
            if val[k] < best[k]["nll"]:
                best[k].update(nll=val[k], epoch=epoch, sd=deepcopy(getattr(model.graph_encoder, f"pred_{k}").state_dict()))
                save = True
        if save:
            ckpt.save_checkpoint(path, args.dataset, y, best["X"]["nll"] + best["E"]["nll"], best["X"]["epoch"],
                                 best["E"]["epoch"], best["X"]["sd"], best["E"]["sd"])
            print("model saved")
        for k, l0, dm in (("X", l0_X, dm_X), ("E", l0_E, dm_E)):
            if l0 < best[k]["log_p_0"]:
                best[k]["log_p_0"], patient = l0, 0
            if dm < best[k]["dm"]:
                best[k]["dm"], patient = dm, 0
        print(json.dumps({"epoch": epoch, "val/denoise_match_X": dm_X, "val/denoise_match_E": dm_E, "val/log_p_0_X": l0_X,
                          "val/log_p_0_E": l0_E, "val/best_val_X": best["X"]["nll"], "val/best_val_E": best["E"]["nll"],
                          "patience": patient}))
        if patient == tc["patient_epochs"]:
            break
        sch["X"].step(l0_X)
        sch["E"].step(l0_E)
    return path


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m graphmaker_amd")
    sub = ap.add_subparsers(dest="cmd", required=True)
    s = sub.add_parser("sample", help="sample.py for a local checkpoint")
    s.add_argument("--model_path", required=True)
    s.add_argument("--data", required=True, help="local .npz of the training graph (marginals and N come from it)")
    s.add_argument("--num_samples", type=int, default=10)
    s.add_argument("--out", default="samples")
    s.add_argument("--seed", type=int, default=0)
    t = sub.add_parser("train", help="train_sync.py / train_async.py for a local graph and yaml")
    t.add_argument("--config", required=True)
    t.add_argument("--data", required=True)
    t.add_argument("--dataset", default="local")
    t.add_argument("--out", default="cpts")
    t.add_argument("--epochs", type=int, default=None)
    t.add_argument("--seed", type=int, default=0)
    args = ap.parse_args(argv)
    return {"sample": cmd_sample, "train": cmd_train}[args.cmd](args)


if __name__ == "__main__":
    main(sys.argv[1:])
