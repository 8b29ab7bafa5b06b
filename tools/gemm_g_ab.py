#!/usr/bin/env python3
"""A/B timing of gm_gemm (the general GEMM of the training path) across library builds on the backward shapes of an
amazon_computer training step (exploration tool; GPU box).   usage: python tools/gemm_g_ab.py libA.so libB.so ..."""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
n, B = 13752, 2097152
SHAPES = [  # (a_red, b_red, M, N, K)
    (0, 1, n, 1744, 1744), (0, 1, n, 1744, 128), (0, 1, n, 1744, 1534), (0, 1, n, 592, 512), (0, 1, n, 512, 512),
    (0, 1, B, 128, 128), (0, 1, B, 128, 2),
    (1, 1, 1744, 1744, n), (1, 1, 128, 1744, n), (1, 1, 1534, 1744, n), (1, 1, 512, 592, n), (1, 1, 512, 1534, n),
    (1, 1, 512, 512, n), (1, 1, 128, 128, B), (1, 1, 2, 128, B)]
libs = []
for p in sys.argv[1:]:
    lib = C.CDLL(os.path.join(ROOT, p)); lib.gm_gemm_workspace_bytes.restype = C.c_int64; libs.append((p, lib))
tot = {p: 0.0 for p, _ in libs}
I64 = C.c_int64
for (ar, br, M, N, K) in SHAPES:
    A = torch.randn((K, M) if ar else (M, K), device=dev); Bm = torch.randn((K, N) if br else (N, K), device=dev) / K ** 0.5
    out = torch.empty(M, N, device=dev)
    line = f"{'T' if ar else 'N'}{'T' if br else 'N'} {M:8d}x{N:5d}x{K:8d}: "
    ref = None
    for p, lib in libs:
        wsb = lib.gm_gemm_workspace_bytes(I64(M), I64(N), I64(K))
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        ts = []
        for rnd in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.gm_gemm(C.c_int(ar), C.c_int(br), I64(M), I64(N), I64(K), C.c_void_p(A.data_ptr()), I64(A.shape[1]),
                             C.c_void_p(Bm.data_ptr()), I64(Bm.shape[1]), None, None, I64(0), C.c_int(0), C.c_void_p(out.data_ptr()),
                             I64(N), C.c_void_p(ws.data_ptr()), I64(ws.numel()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
            e1.record(); torch.cuda.synchronize(); assert rc == 0
            if rnd: ts.append(e0.elapsed_time(e1))
        ms = float(np.median(ts)); tot[p] += ms
        line += f"  {os.path.basename(p)[3:-3]:>10s} {ms:7.3f} ms {2*M*N*K/ms/1e9:6.1f} TF |"
        if ref is None: ref = out.clone()
        else: line += f" maxrel {float(((out-ref).abs().max()/ref.abs().max()).item()):.1e} "
    print(line)
print("total ms:", {os.path.basename(p): round(v, 3) for p, v in tot.items()})
