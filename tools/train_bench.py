#!/usr/bin/env python3
"""tools/train_bench.py -- time the training step (log_p_t + backward + AdamW, train_sync.py:94-117) at amazon_computer
shapes on one GPU.  Not the headline bench (bench.py is); used to find the slow kernels of the training path.
    python tools/train_bench.py [--steps 5] [--batch 2097152] [--workload amazon_computer|cora_shape]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=2097152)           # configs/amazon_computer/train_Sync.yaml
    ap.add_argument("--workload", default="amazon_computer")
    args = ap.parse_args()
    wl = bench.WORKLOAD if args.workload == "amazon_computer" else bench.SMALL
    dev = torch.device("cuda:0")
    from graphmaker_amd import ops
    model = bench.build_model(wl, dev).train()
    n, Fa = wl["N"], wl["F"]
    # a synthetic "real" graph drawn from the priors
    bits0 = ops.prior_E(n, 1.0 - wl["density"], wl["density"], dev, seed=1)
    E0 = ops.dense_from_bitmap(bits0, n)
    E0.fill_diagonal_(0)
    E1h = torch.nn.functional.one_hot(E0, 2).float()
    del E0
    x0, _ = ops.prior_X(n, model.X_marginal, seed=1)
    X3 = torch.nn.functional.one_hot(x0.T.long().contiguous(), 2).float()
    Y = ops.prior_Y(n, model.Y_marginal, seed=1).long()
    dst, src = torch.triu_indices(n, n, offset=1, device=dev)
    optX = torch.optim.AdamW(model.graph_encoder.pred_X.parameters(), lr=1e-3, weight_decay=0, amsgrad=True)
    optE = torch.optim.AdamW(model.graph_encoder.pred_E.parameters(), lr=3e-4, weight_decay=0, amsgrad=True)
    P = dst.numel()
    B = min(args.batch, P)
    ev = {}

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev.setdefault(name, []).append(e)

    def step(k):
        idx = torch.randint(0, P, (B,), device=dev)
        bd, bs = dst[idx], src[idx]
        mark("start")
        lX, lE = model.log_p_t(X3, E1h, Y, bs, bd, E1h[bd, bs], seed=k)
        mark("forward")
        optX.zero_grad()
        optE.zero_grad()
        (lX + lE).backward()
        mark("backward")
        torch.nn.utils.clip_grad_norm_(model.graph_encoder.pred_X.parameters(), 10)
        torch.nn.utils.clip_grad_norm_(model.graph_encoder.pred_E.parameters(), 10)
        optX.step()
        optE.step()
        mark("optim")
        return lX, lE

    for k in range(args.warmup):
        step(k)
    torch.cuda.synchronize()
    ev.clear()
    t0 = time.perf_counter()
    for k in range(args.steps):
        lX, lE = step(args.warmup + k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    names = ["start", "forward", "backward", "optim"]
    phases = {b: float(np.mean([x.elapsed_time(y) for x, y in zip(ev[a], ev[b])])) for a, b in zip(names, names[1:])}
    print(json.dumps({"workload": wl["name"], "batch_pairs": B, "ms_per_train_step": dt * 1e3, "phases_ms": phases,
                      "loss_X": lX.item(), "loss_E": lE.item(),
                      "max_mem_GB": torch.cuda.max_memory_allocated() / 1e9}))


if __name__ == "__main__":
    main()
