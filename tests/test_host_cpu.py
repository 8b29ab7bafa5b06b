"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol include/graphmaker_hip.h
declares (no compute calls), the Python mirror keeps the reference's constructor / state_dict layout, the host-side
schedule and posterior tables equal the reference's and the oracle's bit for bit, and the product refuses to run
without its HIP path (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import golden_util as gu
import oracle_lib as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from graphmaker_amd import _lib
    header = open(os.path.join(ROOT, "include", "graphmaker_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(gm_[A-Za-z0-9_]+)\s*\(", header)))
    assert declared, "no declarations found"
    assert sorted(_lib.EXPORTS) == declared, "graphmaker_amd/_lib.py EXPORTS out of sync with the header"
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported by {_lib.LIB_PATH}"
    assert lib.gm_version() >= 0x10000
    # the struct crossing the ABI has the C layout (8 ints)
    assert ctypes.sizeof(_lib.TowerDims) == 32


def test_product_does_not_link_or_import_the_oracle():
    pkg = os.path.join(ROOT, "graphmaker_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_lib" not in text and "gm_oracle" not in text.replace("oracle/gm_oracle.c", "").replace(
                    "oracle gmo_", ""), f"{f} references the oracle"


def _model(name, device="cpu"):
    import graphmaker_amd as gm
    cfg = gu.CASES[name]
    Xm, Ym, Em = (torch.from_numpy(a).to(device) for a in gu.marginals(cfg))
    if cfg["variant"] == "Sync":
        return cfg, gm.ModelSync(T=cfg["T"], X_marginal=Xm, Y_marginal=Ym, E_marginal=Em, gnn_X_config=cfg["gnn_X"],
                                 gnn_E_config=cfg["gnn_E"], num_nodes=cfg["N"])
    return cfg, gm.ModelAsync(T_X=cfg["T_X"], T_E=cfg["T_E"], X_marginal=Xm, Y_marginal=Ym, E_marginal=Em,
                              mlp_X_config=cfg["mlp_X"], gnn_E_config=cfg["gnn_E"], num_nodes=cfg["N"])


@pytest.mark.parametrize("name", ["sync_small", "async_mid", "tower_full"])
def test_state_dict_layout_matches_reference_checkpoints(name):
    cfg, model = _model(name)
    sdX, sdE = gu.make_checkpoint(cfg)          # key names / shapes the reference's load_state_dict accepted (gen_golden)
    mine_X = model.graph_encoder.pred_X.state_dict()
    mine_E = model.graph_encoder.pred_E.state_dict()
    assert list(mine_X) == list(sdX) and list(mine_E) == list(sdE)          # same keys, same order
    for k, v in sdX.items():
        assert tuple(mine_X[k].shape) == v.shape, k
    for k, v in sdE.items():
        assert tuple(mine_E[k].shape) == v.shape, k
    # strict load, as sample.py:64-65 does
    model.graph_encoder.pred_X.load_state_dict({k: torch.from_numpy(v) for k, v in sdX.items()})
    model.graph_encoder.pred_E.load_state_dict({k: torch.from_numpy(v) for k, v in sdE.items()})
    # a train_sync.py:173-181 style checkpoint dict round-trips through torch.save / torch.load
    import io
    buf = io.BytesIO()
    torch.save({"dataset": "synthetic", "train_yaml_data": {"meta_data": {"variant": cfg["variant"]}},
                "pred_X_state_dict": mine_X, "pred_E_state_dict": mine_E}, buf)
    buf.seek(0)
    ck = torch.load(buf, weights_only=True)
    model.graph_encoder.pred_E.load_state_dict(ck["pred_E_state_dict"])


@pytest.mark.parametrize("name", ["sync_small", "sync_mid", "async_mid"])
def test_schedule_and_tables_bit_exact(name):
    cfg, model = _model(name)
    g = gu.load_golden(name)
    Xm, _, Em = gu.marginals(cfg)
    if cfg["variant"] == "Sync":
        scheds = [(model.noise_schedule, "alphas", "alpha_bars", cfg["T"])]
    else:
        scheds = [(model.noise_schedule_X, "alphas_X", "alpha_bars_X", cfg["T_X"]),
                  (model.noise_schedule_E, "alphas_E", "alpha_bars_E", cfg["T_E"])]
    for ns, ka, kb, T in scheds:
        np.testing.assert_array_equal(ns.alphas_host, g[ka])            # the reference's own tables, bit for bit
        np.testing.assert_array_equal(ns.alpha_bars_host, g[kb])
        for t in range(1, T + 1):
            a, ab_s, ab_t = ns.alphas_host[t], ns.alpha_bars_host[t - 1], ns.alpha_bars_host[t]
            np.testing.assert_array_equal(model.transition.table_E(a, ab_s, ab_t), orc.posterior_table_E(Em, a, ab_s, ab_t))
            np.testing.assert_array_equal(model.transition.table_X(a, ab_s, ab_t), orc.posterior_tables_X(Xm, a, ab_s, ab_t))
    # get_Q_bar_E keeps the reference's formula (model/diffusion.py:65)
    T_E = cfg["T"] if cfg["variant"] == "Sync" else cfg["T_E"]
    nsE = model.noise_schedule if cfg["variant"] == "Sync" else model.noise_schedule_E
    np.testing.assert_array_equal(model.transition.get_Q_bar_E(nsE.alphas[T_E]).numpy(), g[f"e{T_E}_Q"][0])


def test_posterior_method_matches_reference_outputs():
    cfg, model = _model("sync_small")
    g = gu.load_golden("sync_small")
    Q = torch.from_numpy(g[f"e{cfg['T']}_Q"])
    z = torch.nn.functional.one_hot(torch.from_numpy(g["post_z"]).long(), 2).float()
    out = model.posterior(z, Q[0], Q[1], Q[2], torch.from_numpy(g["post_prior"]))
    np.testing.assert_array_equal(out.numpy(), g["post_out"])


def test_sampling_without_a_gpu_fails_loudly():
    from graphmaker_amd import ops
    from graphmaker_amd._lib import GraphMakerHipError
    cfg, model = _model("sync_small")
    with pytest.raises(RuntimeError, match="no CPU path"):
        model.sample()
    with pytest.raises(GraphMakerHipError):
        ops.bitmap_from_dense(torch.zeros((4, 4), dtype=torch.int64))
    with pytest.raises(GraphMakerHipError):
        ops.linear(torch.zeros(2, 2), torch.zeros(2, 2))


def test_words_per_row_convention():
    from graphmaker_amd._lib import words_per_row
    for n in (1, 31, 32, 33, 127, 128, 129, 13752, 100000):
        w = words_per_row(n)
        assert w % 4 == 0 and w * 32 >= n and w == orc.words_per_row(n)


@pytest.mark.parametrize("n,world", [(67, 2), (2708, 2), (7650, 4), (13752, 8), (100000, 8), (130, 3), (50, 8)])
def test_shard_plan_covers_every_pair_once_and_balances(n, world):
    from graphmaker_amd.parallel import ShardPlan
    plans = [ShardPlan.make(n, world, r) for r in range(world)]
    owner = np.full(n, -1)
    for r, p in enumerate(plans):
        for b, e in p.pair_chunks():
            assert 0 <= b <= e <= n and (b % 64 == 0 or b == n)
            assert (owner[b:e] == -1).all()
            owner[b:e] = r
    assert (owner >= 0).all()                                    # every row of the pair matrix has exactly one owner
    pairs = [p.pairs_of(r) for r, p in enumerate(plans)]
    assert sum(pairs) == n * (n - 1) // 2
    if n >= 64 * 2 * world * 4:
        assert max(pairs) <= 1.05 * (sum(pairs) / world)         # folded chunks: within a few % of perfect balance
    rows = np.zeros(n, int)
    for r, p in enumerate(plans):
        b, e = p.node_range()
        rows[b:e] += 1
    assert (rows == 1).all()


def test_pair_workspace_trouble_flag_is_reported():
    """include/graphmaker_hip.h, gm_pair_score_sample: word 16 of the workspace is the sticky trouble flag the host tests."""
    from graphmaker_amd import ops
    ws = ops.pair_workspace(torch.device("cpu"))
    assert ws.numel() >= 256 and int(ws.sum()) == 0
    ops.check_pair_workspace(ws)                       # clean workspace: nothing to report
    ws[0:4].view(torch.int32)[0] = 12345               # the tile counter is not a trouble sign
    ops.check_pair_workspace(ws)
    ws[64:68].view(torch.int32)[0] = 1
    with pytest.raises(RuntimeError, match="hand-shake"):
        ops.check_pair_workspace(ws)


def test_committed_bench_line_follows_the_contract():
    """profiles/r1_bench_line.json is bench.py's own output on an MI355X: the keys the driver and the judge read."""
    import json
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r1_bench_line.json")) as fh:
        d = json.loads(fh.read())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1 and r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


def test_pair_kernel_handshake_protocol_is_bounded(tmp_path):
    """The pair kernel's LDS hand-shake (graphmaker_amd/csrc/gm_pair_protocol.h), run on the host with threads for waves:
    a clean run never gives up; ONE lost publication makes every wave leave the tile loop within
    worst_case_polls_per_wave() unsatisfied polls and sets the sticky stall word; a launch that starts with the word set
    does nothing (tests/host/pair_protocol_test.cpp)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "pair_protocol_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(root, "graphmaker_amd", "csrc"),
                    os.path.join(root, "tests", "host", "pair_protocol_test.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr


def test_relu63_guard_warns_only_for_weights_that_allow_out_of_range_activations():
    """The all-pairs kernel's hidden ReLU is exact for activations <= 0 or within [2^-63, 2^63] (the reference's
    nn.ReLU, model/gnn.py:229, is exact everywhere).  The weights-only bound of LinkPredictor.relu63_bounds() must stay
    silent for a reference-initialised denoiser and warn when a weight makes an activation of 2^64 possible."""
    import warnings
    import graphmaker_amd as gm
    cfg = gu.CASES["sync_small"]
    Xm, Ym, Em = gu.marginals(cfg)
    model = gm.ModelSync(T=3, X_marginal=torch.from_numpy(Xm), Y_marginal=torch.from_numpy(Ym), E_marginal=torch.from_numpy(Em),
                         gnn_X_config=cfg["gnn_X"], gnn_E_config=cfg["gnn_E"], num_nodes=cfg["N"])
    pred_E = model.graph_encoder.pred_E
    hid, w2 = pred_E.relu63_bounds()
    assert 0 < hid < 2.0 ** 40 and 0 < w2 < 10
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        pred_E.check_relu63()
    with torch.no_grad():
        pred_E.mlp_out[0].bias[0] = 2.0 ** 64
    with pytest.warns(RuntimeWarning, match="2\\^63"):
        pred_E.check_relu63()
    with warnings.catch_warnings():                    # once per weight version
        warnings.simplefilter("error")
        pred_E.check_relu63()
