#!/usr/bin/env python3
"""A/B timing of pair-kernel build variants (exploration tool; run on the GPU box).
usage: python tools/pair_ab.py libA.so libB.so ...   (paths relative to repo root)"""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(os.environ.get("PAIR_N", "13752"))
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
h = (torch.randn(n, 128, generator=g) * 0.5).to(dev)
W1 = (torch.randn(128, 128, generator=g) / 11).to(dev); b1 = (torch.randn(128, generator=g) * 0.1).to(dev)
W2 = (torch.randn(2, 128, generator=g) / 11).to(dev); b2 = torch.tensor([3.0, -3.0], device=dev)
wpr = ((n + 31) // 32 + 3) // 4 * 4
bits = torch.zeros((n, wpr), dtype=torch.int32, device=dev)
out = torch.zeros((n, wpr), dtype=torch.int32, device=dev)
ws = torch.zeros(256, dtype=torch.uint8, device=dev)
tab = (C.c_float * 8)(0.998, 0.998, 0.002, 0.002, 0.85, 0.85, 0.15, 0.15)
libs = [(p, C.CDLL(os.path.join(ROOT, p))) for p in sys.argv[1:]]
P = n * (n - 1) // 2
def run(lib):
    rc = lib.gm_pair_score_sample(C.c_int64(n), C.c_int64(128), C.c_void_p(h.data_ptr()), C.c_void_p(W1.data_ptr()),
        C.c_void_p(b1.data_ptr()), C.c_void_p(W2.data_ptr()), C.c_void_p(b2.data_ptr()), C.c_void_p(bits.data_ptr()),
        C.c_int64(wpr), tab, C.c_int64(0), C.c_int64(n), None, C.c_uint64(1), C.c_uint32(3), C.c_void_p(out.data_ptr()),
        None, None, C.c_void_p(ws.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
res = {p: [] for p, _ in libs}
sums = {}
for rnd in range(4):
    for p, lib in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(lib); e1.record(); torch.cuda.synchronize()
        if rnd: res[p].append(e0.elapsed_time(e1))
        sums[p] = int(out.to(torch.int64).abs().sum().item())
for p, _ in libs:
    ms = np.median(res[p]); print(f"{p:60s} {ms:8.3f} ms  {33408*P/ms/1e9:7.1f} TF  min {min(res[p]):.3f}  checksum {sums[p]}  stall {int(ws[64:68].view(torch.int32).item())}")
