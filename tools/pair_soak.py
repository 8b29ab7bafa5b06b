#!/usr/bin/env python3
"""tools/pair_soak.py -- repeat the pair kernel many times per size and check that every launch returns the same bitmap
and never raises the workspace's trouble flag (the tile hand-over inside the kernel is ordered by LDS flags, not barriers;
this is the run-to-run determinism check for it).  GPU box:  python tools/pair_soak.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphmaker_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
W1 = (torch.randn(128, 128, generator=g) / 11).to(dev); b1 = (torch.randn(128, generator=g) * 0.1).to(dev)
W2 = (torch.randn(2, 128, generator=g) / 11).to(dev); b2 = torch.tensor([1.0, -1.0], device=dev)
tab = [0.998, 0.998, 0.002, 0.002, 0.85, 0.85, 0.15, 0.15]
bad = 0
for n, reps in ((65, 200), (777, 200), (1000, 200), (2708, 300), (4097, 100), (7650, 40), (13752, 25)):
    h = (torch.randn(n, 128, generator=g) * 0.5).to(dev)
    bits = ops.prior_E(n, 0.99, 0.01, dev, seed=n)
    ws = ops.pair_workspace(dev)
    ranges = [[(0, n)], [(0, n // 3), (n // 3, n // 2), (n // 2 + 1, n)]]
    for rr in ranges:
        ref = None
        for k in range(reps):
            up = ops.pair_score_sample(h, W1, b1, W2, b2, bits, tab, seed=5, step=1, row_ranges=rr, workspace=ws)
            if ref is None:
                ref = up.clone()
            elif not torch.equal(up, ref):
                bad += 1
                print(f"n={n} ranges={rr} launch {k}: output differs from launch 0 in {(up != ref).sum().item()} words")
        ops.check_pair_workspace(ws)
        print(f"n={n:6d} ranges={len(rr)} launches={reps}: identical, trouble flag clear, edges {int(ops.csr_from_bitmap(ops.symmetrize_upper(ref[:n], n), n)[0][n])}", flush=True)
print("FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
