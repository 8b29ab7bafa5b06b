"""CPU tests of the callers' file formats: preprocess() against the reference's own output (fixture from
tools/gen_golden_train.py), the .pth checkpoint layout (train_sync.py:173-181) and the yaml key check."""
import os

import numpy as np
import pytest
import torch

import golden_util as gu
from graphmaker_amd import checkpoint as ckpt
from graphmaker_amd import data as gdata

SYNC_YAML = """
meta_data :
  variant: "Sync"
gnn_X : {hidden_t: 8, hidden_X: 32, hidden_Y: 16, num_gnn_layers: 2, dropout: 0.}
gnn_E : {hidden_t: 8, hidden_X: 32, hidden_Y: 16, hidden_E: 128, num_gnn_layers: 2, dropout: 0.}
diffusion : {T: 3}
optimizer_X : {lr: 0.001, weight_decay: 0, amsgrad: true}
optimizer_E : {lr: 0.0003, weight_decay: 0, amsgrad: true}
lr_scheduler : {factor: 0.9, patience: 0, verbose: true}
train : {num_epochs: 2, val_every_epochs: 1, patient_epochs: 15, max_grad_norm: 10, batch_size: 1024, val_batch_size: 2048}
"""


def _graph_npz(tmp_path, cfg):
    x0, Y, E0 = gu.real_graph(cfg)
    src, dst = np.nonzero(E0)
    path = os.path.join(tmp_path, "g.npz")
    # add a self loop and a constant column: load_graph_npz must drop both (data.py:17-27)
    feat = np.concatenate([x0, np.zeros((cfg["N"], 1), np.uint8), np.ones((cfg["N"], 1), np.uint8)], axis=1)
    gdata.save_graph_npz(path, feat, Y, np.concatenate([src, [3]]), np.concatenate([dst, [3]]))
    return path


def test_preprocess_matches_reference(tmp_path):
    cfg = gu.CASES["sync_small"]
    gold = np.load(gu.golden_path("preprocess_sync_small"), allow_pickle=False)
    g = gdata.load_graph_npz(_graph_npz(str(tmp_path), cfg))
    assert g["X"].shape == (cfg["N"], cfg["F"]) and not bool((g["src"] == g["dst"]).any())
    X3, Y, E1h, Xm, Ym, Em, XcY = gdata.preprocess(g)
    assert X3.shape == (cfg["F"], cfg["N"], 2) and E1h.shape == (cfg["N"], cfg["N"], 2)
    assert np.array_equal(Xm.numpy(), gold["X_marginal"])
    assert np.array_equal(Ym.numpy(), gold["Y_marginal"])
    assert np.array_equal(Em.numpy(), gold["E_marginal"])
    assert np.array_equal(XcY.numpy(), gold["X_cond_Y_marginals"], equal_nan=True)
    assert float(E1h[:, :, 1].sum()) == float(gold["E_ones"])
    assert gdata.preprocess(g, with_E_one_hot=False)[2] is None


def test_checkpoint_roundtrip_and_yaml(tmp_path):
    cfg = gu.CASES["sync_small"]
    ypath = os.path.join(str(tmp_path), "train_Sync.yaml")
    with open(ypath, "w") as fh:
        fh.write(SYNC_YAML)
    y = ckpt.load_train_yaml(ypath)
    assert y["meta_data"]["variant"] == "Sync" and y["diffusion"]["T"] == 3
    Xm, Ym, Em = (torch.from_numpy(a) for a in gu.marginals(cfg))
    sdX, sdE = gu.make_checkpoint(cfg)
    path = os.path.join(str(tmp_path), "Sync_T3.pth")
    ckpt.save_checkpoint(path, "toy", y, 1.25, 3, 4, {k: torch.from_numpy(v) for k, v in sdX.items()},
                         {k: torch.from_numpy(v) for k, v in sdE.items()})
    raw = torch.load(path, weights_only=True)
    assert sorted(raw) == sorted(["dataset", "train_yaml_data", "best_val_nll", "best_epoch_X", "best_epoch_E",
                                  "pred_X_state_dict", "pred_E_state_dict"])                    # train_sync.py:173-181
    model, state = ckpt.load_checkpoint(path, Xm, Ym, Em, cfg["N"])
    assert not model.training and state["dataset"] == "toy"
    for k, v in model.graph_encoder.pred_E.state_dict().items():
        assert np.array_equal(v.numpy(), sdE[k]), k
    bad = dict(y)
    bad["gnn_E"] = {k: v for k, v in y["gnn_E"].items() if k != "hidden_E"}
    with pytest.raises(ValueError):
        ckpt.build_model(bad, Xm, Ym, Em, cfg["N"])


@pytest.mark.parametrize("numpy1_path", [False, True])
def test_checkpoint_written_the_way_the_reference_writes_it_loads(tmp_path, numpy1_path):
    """train_sync.py:150-181 stores best_val_nll = np.mean(...) -- a numpy.float64 -- straight into the dict it torch.saves.
    The weights-only loader must take it (numpy-scalar reconstruction is data only), under numpy 2's module path and under
    the numpy.core path that the reference's pinned numpy 1.x wrote."""
    import zipfile
    cfg = gu.CASES["sync_small"]
    ypath = os.path.join(str(tmp_path), "train_Sync.yaml")
    with open(ypath, "w") as fh:
        fh.write(SYNC_YAML)
    y = ckpt.load_train_yaml(ypath)
    Xm, Ym, Em = (torch.from_numpy(a) for a in gu.marginals(cfg))
    sdX, sdE = gu.make_checkpoint(cfg)
    path = os.path.join(str(tmp_path), "Sync_T3.pth")
    torch.save({"dataset": "toy", "train_yaml_data": y, "best_val_nll": np.mean([1.0, 1.5]), "best_epoch_X": 3,
                "best_epoch_E": 4, "pred_X_state_dict": {k: torch.from_numpy(v) for k, v in sdX.items()},
                "pred_E_state_dict": {k: torch.from_numpy(v) for k, v in sdE.items()}}, path)
    if numpy1_path:
        old = os.path.join(str(tmp_path), "numpy1.pth")
        with zipfile.ZipFile(path) as zin, zipfile.ZipFile(old, "w") as zout:
            for it in zin.infolist():
                data = zin.read(it.filename)
                if it.filename.endswith("data.pkl"):
                    assert b"numpy._core.multiarray" in data
                    data = data.replace(b"numpy._core.multiarray", b"numpy.core.multiarray")
                zout.writestr(it, data)
        path = old
    with pytest.raises(Exception):
        torch.load(path, weights_only=True)                # the bare loader refuses the numpy scalar ...
    model, state = ckpt.load_checkpoint(path, Xm, Ym, Em, cfg["N"])      # ... ours takes it, still weights-only
    assert float(state["best_val_nll"]) == 1.25 and state["best_epoch_E"] == 4
    for k, v in model.graph_encoder.pred_X.state_dict().items():
        assert np.array_equal(v.numpy(), sdX[k]), k


def test_sampling_without_a_gpu_fails_loudly():
    cfg = gu.CASES["sync_small"]
    Xm, Ym, Em = (torch.from_numpy(a) for a in gu.marginals(cfg))
    model = ckpt.build_model({"meta_data": {"variant": "Sync"}, "gnn_X": cfg["gnn_X"], "gnn_E": cfg["gnn_E"],
                              "diffusion": {"T": 3}}, Xm, Ym, Em, cfg["N"])
    with pytest.raises(RuntimeError):
        model.sample()
