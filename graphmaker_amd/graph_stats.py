"""Statistics of a generated graph on the GPU (reference: Evaluator.preprocess_g, eval_utils.py:539-596, minus the ORCA
orbit counts): triangle count, clustering coefficients, degrees, LINKX homophily of A and of A @ A.  All counting is
integer work on the bitmap + CSR the sampler already holds (no A.to_dense(), no networkx), so it runs at 100k nodes."""
import torch

from . import _lib as L
from . import ops
from ._lib import I32, I64, ptr, require, stream_ptr

__all__ = ["graph_statistics", "linkx_from_counts"]


def linkx_from_counts(same_deg, deg, y, num_classes):
    """linkx_homophily (eval_utils.py:23-74) from per-node same-class degree and degree (float32 ratios like the
    reference; the sums are exact integers here)."""
    n = y.numel()
    value = 0.0
    for k in range(num_classes):
        m = y == k
        same_k = same_deg[m].sum().float()
        deg_k = deg[m].sum().float()
        n_k = m.sum()
        value += max(0.0, float(same_k / deg_k - n_k / n))
    return value / (num_classes - 1)


def graph_statistics(bits, Y, num_classes=None):
    """bits: symmetric bitmap (N, wpr) int32 without self loops (what sample() ends with); Y: (N,) labels.
    Returns dict(triangle_count, linkx_A, linkx_A_pow_2, degs (N,) int64, cluster_coefs (N,) float64)."""
    require(bits, torch.int32, "bits")
    n, wpr = Y.numel(), bits.shape[1]
    dev = bits.device
    y32 = Y.to(torch.int32).contiguous()
    C = int(num_classes if num_classes is not None else int(y32.max().item()) + 1)      # eval_utils.py:61
    adj = ops.adjacency_from_bitmap(bits, n)
    s = stream_ptr()
    tri2 = torch.empty(n, dtype=torch.int64, device=dev)
    L.call("gm_triangles_per_node", I64(n), I64(wpr), ptr(bits), ptr(adj.rowptr), ptr(adj.colidx), ptr(tri2), s)
    degs = (adj.rowptr[1:] - adj.rowptr[:-1]).long()
    same = torch.empty(n, dtype=torch.int32, device=dev)
    L.call("gm_same_class_degree", I64(n), ptr(adj.rowptr), ptr(adj.colidx), ptr(y32), ptr(same), s)
    bits2 = torch.empty_like(bits)
    L.call("gm_bitmap_square", I64(n), I64(wpr), ptr(bits), ptr(adj.rowptr), ptr(adj.colidx), ptr(bits2), s)
    rows = torch.empty((C, wpr), dtype=torch.int32, device=dev)
    L.call("gm_class_bitmaps", I64(n), I64(wpr), I32(C), ptr(y32), ptr(rows), s)
    deg2 = torch.empty(n, dtype=torch.int32, device=dev)
    same2 = torch.empty(n, dtype=torch.int32, device=dev)
    L.call("gm_masked_degree", I64(n), I64(wpr), ptr(bits2), ptr(rows), ptr(y32), ptr(deg2), ptr(same2), s)
    denom = (degs * (degs - 1)).double()
    cc = torch.where(denom > 0, tri2.double() / torch.where(denom > 0, denom, torch.ones_like(denom)), torch.zeros_like(denom))
    yl = y32.long()
    return {"triangle_count": int(tri2.sum().item()) // 6,
            "linkx_A": linkx_from_counts(same.long(), degs, yl, C),
            "linkx_A_pow_2": linkx_from_counts(same2.long(), deg2.long(), yl, C),
            "degs": degs, "cluster_coefs": cc}
