"""GPU graph statistics against networkx (the reference's own tool for triangles / clustering,
eval_utils.py:19-21, 585) and a dense numpy restatement of linkx_homophily (eval_utils.py:23-74) and k_order_g
(:521-529).  The reference's versions need dgl; "parity unpinned" at that boundary, the arithmetic is restated here."""
import networkx as nx
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _linkx_dense(A, y):
    """eval_utils.py:52-74 on a dense 0/1 matrix (edges u->v wherever A[u,v] != 0)."""
    A = torch.from_numpy(A).float()
    y = torch.from_numpy(y)
    same = (A * (y[:, None] == y[None, :]).float()).sum(0)          # in-edges with the same class
    deg = A.sum(0)
    n, C = y.numel(), int(y.max()) + 1
    value = torch.tensor(0.0)
    for k in range(C):
        m = y == k
        value += max(0, same[m].sum() / deg[m].sum() - m.sum() / n)
    return value.item() / (C - 1)


@pytest.mark.parametrize("n,density,C", [(150, 0.05, 4), (333, 0.02, 7), (64, 0.3, 2)])
def test_graph_statistics(n, density, C):
    from graphmaker_amd import graph_stats, ops
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(n)
    up = np.triu(rs.uniform(size=(n, n)) < density, k=1)
    A = (up | up.T).astype(np.int64)
    A[5, :] = 0
    A[:, 5] = 0                                                      # an isolated node
    y = rs.randint(0, C, size=n).astype(np.int64)
    y[:C] = np.arange(C)
    bits = ops.bitmap_from_dense(torch.from_numpy(A).to(dev))
    st = graph_stats.graph_statistics(bits, torch.from_numpy(y).to(dev))
    g = nx.from_numpy_array(A)
    assert st["triangle_count"] == sum(nx.triangles(g).values()) // 3
    assert np.array_equal(st["degs"].cpu().numpy(), A.sum(1))
    cc = np.array([nx.clustering(g)[v] for v in range(n)])
    assert np.allclose(st["cluster_coefs"].cpu().numpy(), cc, rtol=1e-12, atol=0)
    assert abs(st["linkx_A"] - _linkx_dense(A, y)) < 1e-6
    A2 = ((A @ A) != 0).astype(np.int64)
    assert abs(st["linkx_A_pow_2"] - _linkx_dense(A2, y)) < 1e-6
