"""gm_prior_E_static -- the one entry point of the C ABI that allocates nothing (include/graphmaker_hip.h): the prior
draw of E_T (model/diffusion.py:836-839, sample_E :188-212) with the scratch bitmap inside the library instead of an
argument.  Checked through the C ABI against gm_prior_E, bit for bit, on two calls in a row, the second on a smaller
graph (so the static scratch still holds the larger call's words), plus the size limit.  Collected last on purpose."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle_lib as orc

pytestmark = pytest.mark.gpu


def _prior_both(lib, n, p1, seed, dev, u=None):
    from graphmaker_amd._lib import words_per_row
    wpr = words_per_row(n)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    up = C.c_void_p(u.data_ptr()) if u is not None else None
    upper = torch.full((n, wpr), -1, dtype=torch.int32, device=dev)
    want = torch.full((n, wpr), -1, dtype=torch.int32, device=dev)
    got = torch.full((n, wpr), -1, dtype=torch.int32, device=dev)
    args = (C.c_int64(n), C.c_int64(wpr), C.c_float(1.0 - p1), C.c_float(p1), up, C.c_uint64(seed))
    rc = lib.gm_prior_E(*args, C.c_void_p(upper.data_ptr()), C.c_void_p(want.data_ptr()), stream)
    assert rc == 0, lib.gm_last_error()
    torch.cuda.synchronize()
    # "allocates nothing" at two levels: torch's caching allocator (which a ctypes call cannot touch anyway) and the
    # driver's free-byte count, which a hipMalloc / hipMallocAsync inside the library WOULD move.  The gm_prior_E call above
    # has already made the code object and its static scratch array resident.
    before = torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev)
    free_before = torch.cuda.mem_get_info(dev)[0]
    rc = lib.gm_prior_E_static(*args, C.c_void_p(got.data_ptr()), stream)
    assert rc == 0, lib.gm_last_error()
    torch.cuda.synchronize()
    assert (torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev)) == before
    assert torch.cuda.mem_get_info(dev)[0] >= free_before, "device memory was allocated inside gm_prior_E_static"
    return got, want


def test_prior_E_static_equals_prior_E_twice_second_smaller():
    from graphmaker_amd import _lib
    assert torch.cuda.is_available(), "these tests need the MI355X"
    dev = torch.device("cuda:0")
    lib = _lib.load()
    # first call: a graph large enough for the LDS-tiled symmetrise (n > 7680), dense enough to leave many bits behind
    got, want = _prior_both(lib, 9000, 0.3, 11, dev)
    assert torch.equal(got, want)
    assert int(got.ne(0).sum()) > 0
    # second call, smaller graph, different row stride: reads nothing of the first call's words
    got2, want2 = _prior_both(lib, 777, 0.01, 12, dev)
    assert torch.equal(got2, want2)
    np.testing.assert_array_equal(got2.cpu().numpy().view(np.uint32), orc.prior_E(777, [0.99, 0.01], seed=12))
    # host-supplied uniforms take the same path
    u = torch.from_numpy(orc.uniform_E_prior(12, 777)).to(dev)
    got3, want3 = _prior_both(lib, 777, 0.01, 0, dev, u=u)
    assert torch.equal(got3, want3) and torch.equal(got3, got2)
    # and the first size again after the small one
    got4, _ = _prior_both(lib, 9000, 0.3, 11, dev)
    assert torch.equal(got4, got)


def test_prior_E_static_rejects_graphs_beyond_its_scratch():
    from graphmaker_amd import _lib
    from graphmaker_amd import ops
    dev = torch.device("cuda:0")
    n = 16385
    bits = torch.zeros((n, _lib.words_per_row(n)), dtype=torch.int32, device=dev)
    with pytest.raises(_lib.GraphMakerHipError, match="at most 16384"):
        ops.prior_E_static(n, 0.99, 0.01, bits)
    assert int(bits.ne(0).sum()) == 0
    n = 16384                                                   # the largest graph it serves
    bits = torch.zeros((n, _lib.words_per_row(n)), dtype=torch.int32, device=dev)
    ops.prior_E_static(n, 0.999, 0.001, bits, seed=5)
    assert torch.equal(bits, ops.prior_E(n, 0.999, 0.001, dev, seed=5))
