import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
lib = C.CDLL(os.path.join(ROOT, "graphmaker_amd/lib/libgraphmaker_hip.so")); lib.gm_gemm_workspace_bytes.restype = C.c_int64
dev = torch.device("cuda:0"); I64 = C.c_int64
n = 13752
def timeit(fn):
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if r: ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K) in [(n, 1744, 1744), (n, 1744, 1534), (n, 1744, 128), (n, 592, 512), (n, 512, 512), (2097152, 128, 128)]:
    # dX[M][N] = sum_k dY[M][k] W[k][N]   (W is (K_out=K, N_in=N) row-major: b_red layout)
    dY = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev) / K ** 0.5
    out = torch.empty(M, N, device=dev); out2 = torch.empty(M, N, device=dev)
    wsb = lib.gm_gemm_workspace_bytes(I64(M), I64(N), I64(K)); ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    def g():
        rc = lib.gm_gemm(C.c_int(0), C.c_int(1), I64(M), I64(N), I64(K), C.c_void_p(dY.data_ptr()), I64(K), C.c_void_p(W.data_ptr()), I64(N),
                         None, None, I64(0), C.c_int(0), C.c_void_p(out.data_ptr()), I64(N), C.c_void_p(ws.data_ptr()), I64(ws.numel()), st()); assert rc == 0
    Wt = W.t().contiguous()
    def l():
        rc = lib.gm_linear(I64(M), I64(N), I64(K), C.c_void_p(dY.data_ptr()), I64(K), C.c_void_p(Wt.data_ptr()), I64(K), None,
                           C.c_void_p(out2.data_ptr()), I64(N), C.c_int(0), st()); assert rc == 0
    tg, tl, tt = timeit(g), timeit(l), timeit(lambda: W.t().contiguous())
    print(f"{M}x{N}x{K}: gemm_g {tg:.3f} ms {2*M*N*K/tg/1e9:.0f} TF | k_linear {tl:.3f} ms {2*M*N*K/tl/1e9:.0f} TF + transpose {tt:.3f} ms | equal bits {torch.equal(out, out2)}")
