#!/usr/bin/env python3
"""tools/async_bench.py -- wall time of a whole GraphMaker-Async sample (attributes first: T_X MLP-tower steps, then
structure: T_E steps) at Cora shapes on one GPU: BASELINE.json configs[1].  Hyper-parameters of
configs/cora/train_Async.yaml (T_X=6, T_E=9, hidden 512/64/32/128, 2 layers); random-init weights with the output
biases of bench.build_model; synthetic marginals.  Not the headline bench (bench.py is).
    python tools/async_bench.py [--samples 5]"""
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
    ap.add_argument("--samples", type=int, default=5)
    args = ap.parse_args()
    import graphmaker_amd as gm
    wl = bench.SMALL
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(0)
    p1 = rs.uniform(0.002, 0.02, size=wl["F"]).astype(np.float32)
    Xm = torch.from_numpy(np.stack([1 - p1, p1], 1).astype(np.float32)).to(dev)
    Ym = torch.full((wl["C"],), 1.0 / wl["C"], dtype=torch.float32, device=dev)
    d = np.float32(wl["density"])
    Em = torch.tensor([1.0 - d, d], dtype=torch.float32, device=dev)
    torch.manual_seed(0)
    mlp_X = dict(hidden_t=32, hidden_X=512, hidden_Y=64, num_mlp_layers=2, dropout=0.0)
    model = gm.ModelAsync(T_X=6, T_E=9, X_marginal=Xm, Y_marginal=Ym, E_marginal=Em, mlp_X_config=mlp_X,
                          gnn_E_config=wl["gnn_E"], num_nodes=wl["N"])
    with torch.no_grad():
        b = model.graph_encoder.pred_E.mlp_out[2].bias
        b[0] += 3.0
        b[1] -= 3.0
        bx = model.graph_encoder.pred_X.mlp_out[2].bias
        bx[0::2] += 2.0
        bx[1::2] -= 2.0
    model = model.to(dev).eval()
    model.sample(seed=1, dense=False)                      # warm-up: first-use allocations
    torch.cuda.synchronize()
    ts = []
    for k in range(args.samples):
        t0 = time.perf_counter()
        X, Y, E = model.sample(seed=2 + k, dense=False)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    n = wl["N"]
    P = n * (n - 1) // 2
    ms = float(np.median(ts))
    print(json.dumps({"workload": f"cora_shape GraphMaker-Async sample: N={n} F={wl['F']} T_X=6 T_E=9", "ms_per_sample": ms,
                      "samples_per_sec": 1e3 / ms, "pair_scores_per_sec": 9 * P / (ms * 1e-3), "edges_last": int(E.shape[1]),
                      "all_ms": [round(t, 3) for t in ts]}))


if __name__ == "__main__":
    main()
