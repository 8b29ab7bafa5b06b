#!/usr/bin/env python3
"""tools/spmm_traffic.py lib.so [lib2.so ...] -- a few isolated gm_spmm_mean launches per library at amazon_computer shape, to be
run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (tools/run_spmm_traffic.sh): HBM-side bytes per launch of the
mean-aggregation kernel without the rest of the step around it.  Launch order: for each library, for (d, ld) in ((512,512), (576,576), (64,64), (576,1744), (576,1760)): 3
launches on the Erdos-Renyi prior graph."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphmaker_amd import ops
dev = torch.device("cuda:0")
n, dens = 13752, 2.6e-3
bits = ops.prior_E(n, 1.0 - dens, dens, dev, seed=0)
rowptr, colidx = ops.csr_from_bitmap(bits, n)
print("nnz", int(rowptr[n]))
for p in sys.argv[1:]:
    lib = C.CDLL(os.path.join(ROOT, p))
    for d, ld in ((512, 512), (576, 576), (64, 64), (576, 1744), (576, 1760)):
        H = torch.randn(n, ld, device=dev)
        out = torch.empty((n, 592), device=dev)
        for _ in range(3):
            rc = lib.gm_spmm_mean(C.c_int64(n), C.c_void_p(rowptr.data_ptr()), C.c_void_p(colidx.data_ptr()), C.c_void_p(H.data_ptr()),
                                  C.c_int64(ld), C.c_int64(d), C.c_void_p(out.data_ptr()), C.c_int64(592),
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0
        torch.cuda.synchronize()
