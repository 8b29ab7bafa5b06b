import ctypes as C, os, sys, numpy as np, torch
ROOT="/root/repo"
dev=torch.device("cuda:0")
M,N,K=13752,1744,1744
A=torch.randn(M,K,device=dev); W=torch.randn(N,K,device=dev)/K**0.5; b=torch.randn(N,device=dev); out=torch.empty(M,N,device=dev)
for p in sys.argv[1:]:
    lib=C.CDLL(os.path.join(ROOT,p)); ts=[]
    for rnd in range(6):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        rc=lib.gm_linear(C.c_int64(M),C.c_int64(N),C.c_int64(K),C.c_void_p(A.data_ptr()),C.c_int64(K),C.c_void_p(W.data_ptr()),C.c_int64(K),C.c_void_p(b.data_ptr()),C.c_void_p(out.data_ptr()),C.c_int64(N),C.c_int(1),C.c_void_p(torch.cuda.current_stream().cuda_stream))
        e1.record(); torch.cuda.synchronize(); assert rc==0
        if rnd: ts.append(e0.elapsed_time(e1))
    ms=float(np.median(ts)); print(f"{os.path.basename(p):24s} {ms:7.3f} ms {2*M*N*K/ms/1e9:6.1f} TF")
