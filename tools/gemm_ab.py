#!/usr/bin/env python3
"""A/B timing of gm_linear across library builds for the tower GEMM shapes (exploration tool; GPU box).
usage: python tools/gemm_ab.py [--m M1,M2,...] [--nk "N,K;N,K"] libA.so libB.so ...   (outputs of every library are compared bit for bit with the first one's)"""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
argv = sys.argv[1:]
MS = [13752]
if argv and argv[0] == "--m":
    MS = [int(x) for x in argv[1].split(",")]
    argv = argv[2:]
NK = None
if argv and argv[0] == "--nk":          # e.g. --nk 128,128;2,128
    NK = [tuple(int(v) for v in x.split(",")) for x in argv[1].split(";")]
    argv = argv[2:]
SHAPES = [(M, n, k) for M in MS for (n, k) in NK] if NK else [(M, n, k) for M in MS for (n, k) in [(512, 1534), (512, 512), (512, 592), (1744, 1744), (128, 1744), (1534, 1744), (767, 1744), (64, 64)]]
libs = [(p, C.CDLL(os.path.join(ROOT, p))) for p in argv]
tot = {p: 0.0 for p, _ in libs}
for (M, N, K) in SHAPES:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    line = f"{M}x{N}x{K:5d}: "
    ref = None
    for p, lib in libs:
        ts = []
        for rnd in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.gm_linear(C.c_int64(M), C.c_int64(N), C.c_int64(K), C.c_void_p(A.data_ptr()), C.c_int64(K), C.c_void_p(W.data_ptr()),
                               C.c_int64(K), C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr()), C.c_int64(N), C.c_int(1),
                               C.c_void_p(torch.cuda.current_stream().cuda_stream))
            e1.record(); torch.cuda.synchronize(); assert rc == 0
            if rnd: ts.append(e0.elapsed_time(e1))
        ms = float(np.median(ts)); tot[p] += ms
        if ref is None: ref = out.clone()
        elif not torch.equal(ref, out): line += " [DIFFERS] "
        line += f"  {os.path.basename(p):>14s} {ms:7.3f} ms {2*M*N*K/ms/1e9:6.1f} TF |"
    print(line)
print("total ms:", {os.path.basename(p): round(v, 3) for p, v in tot.items()})
