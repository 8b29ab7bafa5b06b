#!/usr/bin/env python3
"""Time the pair kernel at amazon_computer size in its three precisions (GPU box): f32 (exact, default), bf16x3, bf16."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphmaker_amd import ops
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 13752
g = torch.Generator(device="cpu").manual_seed(2)
h = (torch.randn(n, 128, generator=g) * 0.4).to(dev)
W1 = (torch.randn(128, 128, generator=g) / 11).to(dev)
b1 = (torch.randn(128, generator=g) * 0.1).to(dev)
W2 = (torch.randn(2, 128, generator=g) / 11).to(dev)
b2 = torch.tensor([3.0, -3.0], device=dev)
bits = ops.prior_E(n, 1 - 2.6e-3, 2.6e-3, dev, seed=3)
tab = [0.99, 0.3, 0.01, 0.7, 0.8, 0.01, 0.2, 0.99]
ws = ops.pair_workspace(dev)
out = torch.zeros_like(bits)
P = n * (n - 1) // 2
for prec in ("f32", "bf16x3", "bf16"):
    f = lambda: ops.pair_score_sample(h, W1, b1, W2, b2, bits, tab, seed=1, step=1, out_upper=out, workspace=ws, precision=prec)
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    ops.check_pair_workspace(ws)
    print(f"{prec:7s} {ms:8.3f} ms  {P / ms / 1e6:8.2f} G pairs/s  {33408 * P / ms / 1e9:7.1f} TFLOP/s (algorithmic)")
