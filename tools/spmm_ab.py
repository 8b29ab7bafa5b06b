#!/usr/bin/env python3
"""A/B timing of gm_spmm_mean across library builds at amazon_computer / Cora shapes (exploration tool; GPU box).
usage: python tools/spmm_ab.py libA.so libB.so ..."""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphmaker_amd import ops
dev = torch.device("cuda:0")
libs = [(p, C.CDLL(os.path.join(ROOT, p))) for p in sys.argv[1:]]
def skewed_bits(n, mean_deg, max_deg, seed=0):
    """Chung-Lu style power-law graph (tests/test_gpu_parity.py skewed_sym_dense) as a device bitmap."""
    rs = np.random.RandomState(seed)
    w = (np.arange(n) + 1.0) ** -0.75
    w = np.minimum(w / w.sum() * mean_deg * n, max_deg)
    rs.shuffle(w)
    p = w / w.sum()
    m = int(mean_deg * n / 2)
    i, j = rs.choice(n, size=m, p=p), rs.choice(n, size=m, p=p)
    keep = i != j
    ei = torch.from_numpy(np.minimum(i, j)[keep].astype(np.int32)).to(dev)
    ej = torch.from_numpy(np.maximum(i, j)[keep].astype(np.int32)).to(dev)
    return ops.bitmap_from_upper_edges(n, ei, ej)


for (n, dens, d, ld) in [(13752, 2.6e-3, 512, 1744), (13752, 2.6e-3, 576, 576), (13752, 2.6e-3, 64, 1744), (13752, -1, 512, 1744), (13752, -1, 576, 576), (13752, -1, 64, 1744),
                         (1719, 2.6e-3, 576, 576), (2708, 1.44e-3, 576, 576), (2708, 1.44e-3, 512, 1760), (100000, 4e-5, 512, 1760)]:
    if dens < 0:      # power-law degrees at amazon_computer's N / nnz, largest degree ~3000
        bits = skewed_bits(n, 36.0, 3000.0)
    else:
        bits = ops.prior_E(n, 1.0 - dens, dens, dev, seed=0)
    adj = ops.adjacency_from_bitmap(bits, n)
    if dens < 0:
        deg = (adj.rowptr[1:] - adj.rowptr[:-1])
        print(f"  skewed graph: max degree {int(deg.max())}, rows > 64 neighbours: {int((deg > 64).sum())}")
    if n == 1719:     # a rank's node block: 1719 output rows out of a 13752-row operand
        big = ops.adjacency_from_bitmap(ops.prior_E(13752, 1.0 - dens, dens, dev, seed=0), 13752)
        adj = ops.Adjacency(n, None, big.rowptr[:n + 1].contiguous(), big.colidx)
        adj._nnz = int(adj.rowptr[n])
        H = torch.randn(13752, ld, device=dev)
    else:
        H = None
    H = torch.randn(n, ld, device=dev) if H is None else H
    line = f"n={n} d={d} ld={ld} nnz={adj.nnz}: "
    ref = None
    for p, lib in libs:
        out = torch.zeros(n, d, device=dev)
        ts = []
        for rnd in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.gm_spmm_mean(C.c_int64(n), C.c_void_p(adj.rowptr.data_ptr()), C.c_void_p(adj.colidx.data_ptr()), C.c_void_p(H.data_ptr()),
                                  C.c_int64(ld), C.c_int64(d), C.c_void_p(out.data_ptr()), C.c_int64(d), C.c_void_p(torch.cuda.current_stream().cuda_stream))
            e1.record(); torch.cuda.synchronize(); assert rc == 0
            if rnd: ts.append(e0.elapsed_time(e1))
        ms = float(np.median(ts))
        line += f" {os.path.basename(p)[3:-3]:>14s} {ms*1e3:8.1f} us |"
        if ref is None: ref = out.clone()
        elif not torch.equal(ref, out): line += " [DIFFERS] "
    print(line)
