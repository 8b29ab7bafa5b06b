"""Local datasets and the reference's preprocess() on plain arrays (reference: data.py:8-108).

The reference pulls Cora / Amazon graphs through dgl.data (a download); there is no network and no dgl here, so a
dataset is a local .npz with
    feat  (N, F_raw) any dtype, non-zero = attribute present      (g.ndata['feat'], data.py:19-20)
    label (N,) integer class ids                                   (g.ndata['label'])
    src, dst (M,) integer endpoints, either direction(s)           (g.edges())
`load_graph_npz` applies data.py:17-29 (drop self loops, binarise, drop constant columns); `preprocess` returns the
seven tensors of data.py:32-108 in the same order, shapes and dtypes.
"""
import numpy as np
import torch
import torch.nn.functional as F

__all__ = ["load_graph_npz", "preprocess", "save_graph_npz"]


def load_graph_npz(path):
    """-> dict(X (N,F) float 0/1, Y (N,) int64, src, dst int64 without self loops, num_nodes)."""
    z = np.load(path, allow_pickle=False)
    X = torch.from_numpy(np.asarray(z["feat"])).float()
    X[X != 0] = 1.0                                                        # data.py:19-20
    X = X[:, X.sum(dim=0) != 0]                                            # data.py:23-24
    X = X[:, X.sum(dim=0) != X.size(0)]                                    # data.py:26-27
    Y = torch.from_numpy(np.asarray(z["label"]).astype(np.int64))
    src = torch.from_numpy(np.asarray(z["src"]).astype(np.int64))
    dst = torch.from_numpy(np.asarray(z["dst"]).astype(np.int64))
    keep = src != dst                                                      # dgl.remove_self_loop, data.py:17
    return dict(X=X, Y=Y, src=src[keep], dst=dst[keep], num_nodes=int(Y.numel()))


def save_graph_npz(path, X, Y, src, dst):
    np.savez_compressed(path, feat=np.asarray(X, dtype=np.uint8), label=np.asarray(Y, dtype=np.int64),
                        src=np.asarray(src, dtype=np.int64), dst=np.asarray(dst, dtype=np.int64))


def preprocess(g, with_E_one_hot=True):
    """data.py:32-108 -> (X_one_hot (F,N,2), Y (N,), E_one_hot (N,N,2), X_marginal (F,2), Y_marginal (C,),
    E_marginal (2,), X_cond_Y_marginals (F,C,2)).  with_E_one_hot=False returns None for the N x N x 2 tensor (the
    sampler does not need it) while E_marginal is still the reference's count ratio."""
    X, Y, N = g["X"], g["Y"], g["num_nodes"]
    src, dst = g["src"], g["dst"]
    X_one_hot = F.one_hot(X.long().T.contiguous(), num_classes=2).float()            # (F,N,2)  :65-71
    E = torch.zeros(N, N)
    E[dst, src] = 1.0                                                                # :73-74
    E_one_hot = F.one_hot(E.long(), num_classes=2).float() if with_E_one_hot else None
    X_count = X_one_hot.sum(dim=1)                                                   # :79-81
    X_marginal = X_count / X_count.sum(dim=1, keepdim=True)
    Y_count = F.one_hot(Y).float().sum(dim=0)                                        # :84-88
    Y_marginal = Y_count / Y_count.sum()
    ones = float(E.sum().item())
    E_count = torch.tensor([float(N * N) - ones, ones])                              # :91-92
    E_marginal = E_count / E_count.sum()
    cond = []
    for k in range(Y_marginal.size(-1)):                                             # :95-106
        cnt = X_one_hot[:, Y == k].sum(dim=1)
        cond.append(cnt / cnt.sum(dim=1, keepdim=True))
    return X_one_hot, Y, E_one_hot, X_marginal, Y_marginal, E_marginal, torch.stack(cond, dim=1)
