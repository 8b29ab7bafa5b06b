"""Deterministic synthetic checkpoints + case definitions shared by tools/gen_golden.py and the tests.

Weights come from numpy's frozen legacy generator (RandomState / MT19937), so the fixtures under
tests/golden/ need to hold only the reference's OUTPUTS, not the weights.  Key names and shapes follow the
reference's state_dicts (train_sync.py:173-181).
"""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name -> config.  hidden_E stays 128 (every reference yaml uses 128; other widths: tests/test_gpu_parity.py, against the oracle).
CASES = {
    "sync_small": dict(variant="Sync", N=67, F=11, C=3, T=3, density=0.06, seed=11,
                       gnn_X=dict(hidden_t=8, hidden_X=32, hidden_Y=16, num_gnn_layers=2, dropout=0.0),
                       gnn_E=dict(hidden_t=8, hidden_X=32, hidden_Y=16, hidden_E=128, num_gnn_layers=2,
                                  dropout=0.0)),
    "sync_mid": dict(variant="Sync", N=200, F=33, C=7, T=3, density=0.03, seed=12,
                     gnn_X=dict(hidden_t=16, hidden_X=64, hidden_Y=32, num_gnn_layers=2, dropout=0.0),
                     gnn_E=dict(hidden_t=16, hidden_X=64, hidden_Y=32, hidden_E=128, num_gnn_layers=2,
                                dropout=0.0)),
    "async_mid": dict(variant="Async", N=130, F=20, C=4, T_X=2, T_E=3, density=0.04, seed=13,
                      mlp_X=dict(hidden_t=8, hidden_X=48, hidden_Y=16, num_mlp_layers=2, dropout=0.0),
                      gnn_E=dict(hidden_t=8, hidden_X=32, hidden_Y=16, hidden_E=128, num_gnn_layers=2,
                                 dropout=0.0)),
    # full hidden sizes of configs/*/train_Sync.yaml at a small node count: towers + pair stage only
    "tower_full": dict(variant="Sync", N=96, F=40, C=10, T=3, density=0.05, seed=14,
                       gnn_X=dict(hidden_t=16, hidden_X=512, hidden_Y=64, num_gnn_layers=2, dropout=0.0),
                       gnn_E=dict(hidden_t=16, hidden_X=512, hidden_Y=64, hidden_E=128, num_gnn_layers=2,
                                  dropout=0.0)),
    # the step counts of the shipped Async configs (configs/cora|amazon_photo/train_Async.yaml: T_X=6, T_E=9).  T_X=6 is the
    # one schedule length on which the reference's denoise_match_X recovers a truncated time index (t=5 -> 4,
    # model/diffusion.py:681), so its val_step fixture pins that behaviour
    "async_T6": dict(variant="Async", N=60, F=9, C=3, T_X=6, T_E=9, density=0.08, seed=15,
                     mlp_X=dict(hidden_t=8, hidden_X=32, hidden_Y=16, num_mlp_layers=2, dropout=0.0),
                     gnn_E=dict(hidden_t=8, hidden_X=32, hidden_Y=16, hidden_E=128, num_gnn_layers=2,
                                dropout=0.0)),
    # the amazon_computer Async denoiser (configs/amazon_computer/train_Async.yaml:4-21): MLPTower with hidden_X=1024
    # (hidden_cat 3280), T_X=7, T_E=9, at a small node count
    "async_wide": dict(variant="Async", N=96, F=40, C=10, T_X=7, T_E=9, density=0.05, seed=16,
                       mlp_X=dict(hidden_t=16, hidden_X=1024, hidden_Y=64, num_mlp_layers=2, dropout=0.0),
                       gnn_E=dict(hidden_t=16, hidden_X=512, hidden_Y=64, hidden_E=128, num_gnn_layers=2,
                                  dropout=0.0)),
}


# BASELINE.json-sized cases: the reference runs them once in the build container (tools/gen_golden.py), only the sampled
# states of every step are stored (edge lists, bit-packed attributes); not part of CASES (whose fixtures hold every
# intermediate tensor and every pair's logits)
LARGE_CASES = {
    # BASELINE.json configs[0]: Cora shape with configs/cora/train_Sync.yaml hyper-parameters
    "cora_sync": dict(variant="Sync", N=2708, F=1433, C=7, T=3, density=1.44e-3, seed=21, x_p1=(0.002, 0.02),
                      edge_bias=3.0, attr_bias=2.0,
                      gnn_X=dict(hidden_t=32, hidden_X=512, hidden_Y=64, num_gnn_layers=2, dropout=0.0),
                      gnn_E=dict(hidden_t=32, hidden_X=512, hidden_Y=64, hidden_E=128, num_gnn_layers=2,
                                 dropout=0.0)),
    # BASELINE.json configs[1]: Cora shape with configs/cora/train_Async.yaml hyper-parameters (T_X=6, T_E=9)
    "cora_async": dict(variant="Async", N=2708, F=1433, C=7, T_X=6, T_E=9, density=1.44e-3, seed=22, x_p1=(0.002, 0.02),
                       edge_bias=3.0, attr_bias=2.0,
                       mlp_X=dict(hidden_t=32, hidden_X=512, hidden_Y=64, num_mlp_layers=2, dropout=0.0),
                       gnn_E=dict(hidden_t=32, hidden_X=512, hidden_Y=64, hidden_E=128, num_gnn_layers=2,
                                  dropout=0.0)),
    # BASELINE.json configs[2]: amazon_photo shape with configs/amazon_photo/train_Sync.yaml hyper-parameters
    "photo_sync": dict(variant="Sync", N=7650, F=745, C=8, T=3, density=4.07e-3, seed=23, x_p1=(0.002, 0.02),
                       edge_bias=3.0, attr_bias=2.0,
                       gnn_X=dict(hidden_t=16, hidden_X=512, hidden_Y=64, num_gnn_layers=2, dropout=0.0),
                       gnn_E=dict(hidden_t=16, hidden_X=512, hidden_Y=64, hidden_E=128, num_gnn_layers=2,
                                  dropout=0.0)),
}


def marginals(cfg):
    rs = np.random.RandomState(cfg["seed"] + 1000)
    F, C = cfg["F"], cfg["C"]
    lo, hi = cfg.get("x_p1", (0.05, 0.3))
    p1 = rs.uniform(lo, hi, size=F).astype(np.float32)
    X_marginal = np.stack([1.0 - p1, p1], axis=1).astype(np.float32)
    y = rs.uniform(0.5, 1.5, size=C)
    Y_marginal = (y / y.sum()).astype(np.float32)
    d = np.float32(cfg["density"])
    E_marginal = np.array([np.float32(1.0) - d, d], np.float32)
    return X_marginal, Y_marginal, E_marginal


def _linear(rs, out_f, in_f):
    bound = 1.0 / np.sqrt(in_f)
    return (rs.uniform(-bound, bound, size=(out_f, in_f)).astype(np.float32),
            rs.uniform(-bound, bound, size=(out_f,)).astype(np.float32))


def tower_state_dict(rs, in_X, C, hidden_t, hidden_X, hidden_Y, num_layers, out_size, is_mlp, prefix=""):
    sd = {}

    def put(name, w, b):
        sd[prefix + name + ".weight"] = w
        sd[prefix + name + ".bias"] = b

    put("mlp_in_t.0", *_linear(rs, hidden_t, 1))
    put("mlp_in_t.2", *_linear(rs, hidden_t, hidden_t))
    put("mlp_in_X.0", *_linear(rs, hidden_X, in_X))
    put("mlp_in_X.2", *_linear(rs, hidden_X, hidden_X))
    sd[prefix + "emb_Y.weight"] = rs.normal(0, 1, size=(C, hidden_Y)).astype(np.float32)
    lay = "mlp_layers" if is_mlp else "gnn_layers"
    for l in range(num_layers):
        put(f"{lay}.{l}.update_X.0", *_linear(rs, hidden_X, hidden_X + hidden_Y + hidden_t))
        sd[prefix + f"{lay}.{l}.update_X.2.weight"] = (1 + 0.1 * rs.normal(size=hidden_X)).astype(np.float32)
        sd[prefix + f"{lay}.{l}.update_X.2.bias"] = (0.1 * rs.normal(size=hidden_X)).astype(np.float32)
        put(f"{lay}.{l}.update_Y.0", *_linear(rs, hidden_Y, hidden_Y))
        sd[prefix + f"{lay}.{l}.update_Y.2.weight"] = (1 + 0.1 * rs.normal(size=hidden_Y)).astype(np.float32)
        sd[prefix + f"{lay}.{l}.update_Y.2.bias"] = (0.1 * rs.normal(size=hidden_Y)).astype(np.float32)
    Hc = (num_layers + 1) * (hidden_X + hidden_Y) + hidden_t
    put("mlp_out.0", *_linear(rs, Hc, Hc))
    put("mlp_out.2", *_linear(rs, out_size, Hc))
    return sd


def make_checkpoint(cfg, edge_bias=None, attr_bias=None):
    """Returns (pred_X_state_dict, pred_E_state_dict) as name -> float32 ndarray, reference key names.

    Output biases are shifted so the generated graphs stay sparse-ish along the trajectory (a trained
    model's regime) instead of the p=0.5 coin flips an untrained head gives."""
    rs = np.random.RandomState(cfg["seed"])
    F, C = cfg["F"], cfg["C"]
    edge_bias = cfg.get("edge_bias", 2.0) if edge_bias is None else edge_bias
    attr_bias = cfg.get("attr_bias", 1.0) if attr_bias is None else attr_bias
    if cfg["variant"] == "Sync":
        g = cfg["gnn_X"]
        sdX = tower_state_dict(rs, 2 * F, C, g["hidden_t"], g["hidden_X"], g["hidden_Y"],
                               g["num_gnn_layers"], 2 * F, is_mlp=False)
    else:
        g = cfg["mlp_X"]
        sdX = tower_state_dict(rs, 2 * F, C, g["hidden_t"], g["hidden_X"], g["hidden_Y"],
                               g["num_mlp_layers"], 2 * F, is_mlp=True)
    b = sdX["mlp_out.2.bias"]
    b[0::2] += attr_bias
    b[1::2] -= attr_bias
    e = cfg["gnn_E"]
    sdE = tower_state_dict(rs, 2 * F, C, e["hidden_t"], e["hidden_X"], e["hidden_Y"], e["num_gnn_layers"],
                           e["hidden_E"], is_mlp=False, prefix="gnn_encoder.")
    hE = e["hidden_E"]
    w, bb = _linear(rs, hE, hE)
    sdE["mlp_out.0.weight"], sdE["mlp_out.0.bias"] = w, bb
    w, bb = _linear(rs, 2, hE)
    bb[0] += edge_bias
    bb[1] -= edge_bias
    sdE["mlp_out.2.weight"], sdE["mlp_out.2.bias"] = w, bb
    return sdX, sdE


def golden_path(name):
    return os.path.join(GOLDEN_DIR, name + ".npz")


def load_golden(name):
    z = np.load(golden_path(name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def case_json(name):
    return json.dumps(CASES[name], sort_keys=True)


# ---------------------------------------------------------------- training fixtures (tools/gen_golden_train.py)
TRAIN_CASES = ("sync_small", "async_mid", "async_T6")


def real_graph(cfg):
    """A synthetic "real" graph for the training path: x0 (N,F) uint8 drawn from the marginals, labels Y (N,) int64,
    symmetric adjacency E0 (N,N) int64 with the case's density and an empty diagonal (data.py:17 removes self loops)."""
    rs = np.random.RandomState(cfg["seed"] + 2000)
    N, F, C = cfg["N"], cfg["F"], cfg["C"]
    Xm, _, _ = marginals(cfg)
    x0 = (rs.uniform(size=(N, F)) < Xm[:, 1][None, :]).astype(np.uint8)
    Y = rs.randint(0, C, size=N).astype(np.int64)
    up = np.triu(rs.uniform(size=(N, N)) < cfg["density"], k=1)
    E0 = (up | up.T).astype(np.int64)
    return x0, Y, E0


def train_batch(cfg, B=1500):
    """(batch_dst, batch_src) as train_sync.py:44-46,97 builds them: rows of triu_indices(N, N, offset=1), a fixed
    random subset in random order."""
    rs = np.random.RandomState(cfg["seed"] + 3000)
    N = cfg["N"]
    iu, ju = np.triu_indices(N, k=1)
    pick = rs.permutation(len(iu))[:min(B, len(iu))]
    return iu[pick].astype(np.int64), ju[pick].astype(np.int64)       # dst = row index, src = column index


LARGE_TRAIN_BATCH = 16384      # configs/cora/train_Sync.yaml:42


def grad_sample_index(cfg, key, numel, k=2048):
    """The fixed positions of a parameter gradient a large training fixture keeps (all of a tensor of at most k entries)."""
    if numel <= k:
        return np.arange(numel)
    rs = np.random.RandomState((cfg["seed"] + 5000 + sum(map(ord, key))) % (2 ** 31))
    return np.sort(rs.choice(numel, size=k, replace=False))


def train_uniforms(cfg, kind, t):
    """Host-supplied uniforms (24-bit grid) of the forward-noising draws: kind "E" -> (N*N,), "X" -> (F*N,)."""
    rs = np.random.RandomState(cfg["seed"] + 4000 + 17 * t + (0 if kind == "E" else 7))
    n = cfg["N"] * cfg["N"] if kind == "E" else cfg["N"] * cfg["F"]
    return (rs.randint(0, 1 << 24, size=n).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


# ---------------------------------------------------------------- BASELINE.json-sized fixtures (LARGE_CASES)
def large_bits(n, upper_nz, diag=None):
    """Symmetric uint32 bitmap (n, wpr) of an upper edge list (+ optional diagonal entries: the prior keeps them)."""
    wpr = ((n + 31) // 32 + 3) // 4 * 4
    bits = np.zeros((n, wpr), np.uint32)
    i, j = upper_nz[:, 0].astype(np.int64), upper_nz[:, 1].astype(np.int64)
    np.bitwise_or.at(bits, (i, j >> 5), np.uint32(1) << (j & 31).astype(np.uint32))
    np.bitwise_or.at(bits, (j, i >> 5), np.uint32(1) << (i & 31).astype(np.uint32))
    if diag is not None and len(diag):
        d = diag.astype(np.int64)
        np.bitwise_or.at(bits, (d, d >> 5), np.uint32(1) << (d & 31).astype(np.uint32))
    return bits


def large_x(packed, F):
    """(N,F) uint8 attributes from the bit-packed rows of a large fixture."""
    return np.ascontiguousarray(np.unpackbits(packed, axis=1)[:, :F])


def upper_bit(bits, i, j):
    return (bits[i, j >> 5] >> (j & 31).astype(np.uint32)) & 1
