#!/usr/bin/env python3
"""Run gm_linear a few times on one shape (for rocprofv3 --pmc passes).  usage: python tools/gemm_one.py M N K [reps]"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M, N, K = (int(x) for x in sys.argv[1:4]); reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
lib = C.CDLL(os.path.join(ROOT, "graphmaker_amd/lib/libgraphmaker_hip.so"))
dev = torch.device("cuda:0")
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev)
for _ in range(reps):
    rc = lib.gm_linear(C.c_int64(M), C.c_int64(N), C.c_int64(K), C.c_void_p(A.data_ptr()), C.c_int64(K), C.c_void_p(W.data_ptr()), C.c_int64(K),
                       C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr()), C.c_int64(N), C.c_int(1), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
torch.cuda.synchronize()
