#!/usr/bin/env python3
"""Reads the cycle stamps of a -DGM_PAIR_STAMP diagnostic build (tools/build_pair_variant.sh stamp -DGM_PAIR_STAMP)."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = 13752; dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
h = (torch.randn(n, 128, generator=g) * 0.5).to(dev)
W1 = (torch.randn(128, 128, generator=g) / 11).to(dev); b1 = (torch.randn(128, generator=g) * 0.1).to(dev)
W2 = (torch.randn(2, 128, generator=g) / 11).to(dev); b2 = torch.tensor([3.0, -3.0], device=dev)
wpr = ((n + 31) // 32 + 3) // 4 * 4
bits = torch.zeros((n, wpr), dtype=torch.int32, device=dev); out = torch.zeros_like(bits)
ws = torch.zeros(256, dtype=torch.uint8, device=dev)
dbg = torch.zeros(64, dtype=torch.float32, device=dev)
tab = (C.c_float * 8)(0.998, 0.998, 0.002, 0.002, 0.85, 0.85, 0.15, 0.15)
lib = C.CDLL(os.path.join(ROOT, sys.argv[1]))
for _ in range(2):
    rc = lib.gm_pair_score_sample(C.c_int64(n), C.c_int64(128), C.c_void_p(h.data_ptr()), C.c_void_p(W1.data_ptr()),
        C.c_void_p(b1.data_ptr()), C.c_void_p(W2.data_ptr()), C.c_void_p(b2.data_ptr()), C.c_void_p(bits.data_ptr()),
        C.c_int64(wpr), tab, C.c_int64(0), C.c_int64(n), None, C.c_uint64(1), C.c_uint32(3), C.c_void_p(out.data_ptr()),
        None, C.c_void_p(dbg.data_ptr()), C.c_void_p(ws.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
torch.cuda.synchronize()
d = dbg.cpu().numpy()
for g_ in (0, 1):
    mma, epi, phase, cnt, init, cyc, rt, wait = d[8 * g_: 8 * g_ + 8]
    print(f"group {g_}: workgroup 0 lived {cyc:.0f} shader cycles = {rt / 100:.1f} us of the 100 MHz clock -> {cyc / rt * 100:.0f} MHz average")
    print(f"group {g_}: units {cnt:.0f}  per unit: wait for the turn {wait / cnt:.0f}  acc init {init / cnt:.0f}  mma loop {mma / cnt:.0f}  "
          f"epilogue {epi / cnt:.0f}  whole {phase / cnt:.0f} cycles (two waves share a SIMD: {phase / (2 * cnt):.0f} per unit of SIMD time)")
