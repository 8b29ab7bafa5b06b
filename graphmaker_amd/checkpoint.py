"""The reference's .pth checkpoint (train_sync.py:173-181, train_async.py) and yaml configuration
(setup_utils.py:11-88), read and written without executing anything from the file.

    {"dataset", "train_yaml_data", "best_val_nll", "best_epoch_X", "best_epoch_E",
     "pred_X_state_dict", "pred_E_state_dict"}
"""
import numpy
import torch
import yaml

from .diffusion import ModelAsync, ModelSync

__all__ = ["load_checkpoint", "build_model", "save_checkpoint", "load_train_yaml"]

_SECTIONS = {
    "Sync": {"meta_data": ["variant"], "gnn_X": ["hidden_t", "hidden_X", "hidden_Y", "num_gnn_layers", "dropout"],
             "gnn_E": ["hidden_t", "hidden_X", "hidden_Y", "hidden_E", "num_gnn_layers", "dropout"], "diffusion": ["T"]},
    "Async": {"meta_data": ["variant"], "mlp_X": ["hidden_t", "hidden_X", "hidden_Y", "num_mlp_layers", "dropout"],
              "gnn_E": ["hidden_t", "hidden_X", "hidden_Y", "hidden_E", "num_gnn_layers", "dropout"],
              "diffusion": ["T_X", "T_E"]},
}


def _validate(yaml_data):
    variant = yaml_data.get("meta_data", {}).get("variant")
    if variant not in _SECTIONS:
        raise ValueError(f"meta_data.variant must be Sync or Async, got {variant!r}")
    for sec, keys in _SECTIONS[variant].items():
        missing = [k for k in keys if k not in yaml_data.get(sec, {})]
        if missing:
            raise ValueError(f"configuration section {sec!r} lacks {missing}")
    return variant


def load_train_yaml(path):
    """setup_utils.py:81-88 for an explicit path (yaml.SafeLoader; the pydantic models become a key check)."""
    with open(path) as fh:
        data = yaml.load(fh, Loader=yaml.loader.SafeLoader)
    _validate(data)
    data.setdefault("optimizer_X", {}).setdefault("weight_decay", 0.0)
    data.setdefault("optimizer_E", {}).setdefault("weight_decay", 0.0)
    return data


def build_model(yaml_data, X_marginal, Y_marginal, E_marginal, num_nodes):
    """sample.py:53-76 / train_sync.py:58-64: the model on the marginals' device."""
    variant = _validate(yaml_data)
    if variant == "Sync":
        return ModelSync(X_marginal=X_marginal, Y_marginal=Y_marginal, E_marginal=E_marginal,
                         gnn_X_config=yaml_data["gnn_X"], gnn_E_config=yaml_data["gnn_E"], num_nodes=num_nodes,
                         **yaml_data["diffusion"]).to(X_marginal.device)
    return ModelAsync(X_marginal=X_marginal, Y_marginal=Y_marginal, E_marginal=E_marginal,
                      mlp_X_config=yaml_data["mlp_X"], gnn_E_config=yaml_data["gnn_E"], num_nodes=num_nodes,
                      **yaml_data["diffusion"]).to(X_marginal.device)


def _numpy_scalar_globals():
    """The data-only numpy globals a reference checkpoint needs: train_sync.py:150-181 stores best_val_nll as the
    numpy.float64 that np.mean returns, pickled as numpy.core.multiarray.scalar(dtype, bytes) (numpy 1.x module path; numpy 2
    writes numpy._core).  Reconstructing a scalar from a dtype and raw bytes executes nothing from the file."""
    scalar = numpy._core.multiarray.scalar if hasattr(numpy, "_core") else numpy.core.multiarray.scalar
    dtypes = [getattr(numpy.dtypes, n) for n in ("Float64DType", "Float32DType", "Int64DType", "Int32DType")
              if hasattr(getattr(numpy, "dtypes", None), n)]
    return [scalar, (scalar, "numpy.core.multiarray.scalar"), (scalar, "numpy._core.multiarray.scalar"), numpy.dtype] + dtypes


def load_checkpoint(path, X_marginal, Y_marginal, E_marginal, num_nodes):
    """sample.py:26-81.  torch.load(weights_only=True): tensors, dicts and scalars only (numpy scalars included), nothing
    is unpickled into code.  Returns (model in eval mode, checkpoint dict)."""
    with torch.serialization.safe_globals(_numpy_scalar_globals()):
        state = torch.load(path, map_location="cpu", weights_only=True)
    for key in ("train_yaml_data", "pred_X_state_dict", "pred_E_state_dict"):
        if key not in state:
            raise KeyError(f"{path}: not a GraphMaker checkpoint (no {key!r})")
    model = build_model(state["train_yaml_data"], X_marginal, Y_marginal, E_marginal, num_nodes)
    model.graph_encoder.pred_X.load_state_dict(state["pred_X_state_dict"])          # strict, sample.py:64-65
    model.graph_encoder.pred_E.load_state_dict(state["pred_E_state_dict"])
    return model.eval(), state


def save_checkpoint(path, dataset, yaml_data, best_val_nll, best_epoch_X, best_epoch_E, state_dict_X, state_dict_E):
    """train_sync.py:173-181 (same keys; tensors moved to the CPU)."""
    def cpu(sd):
        return {k: v.detach().cpu() for k, v in sd.items()}

    torch.save({"dataset": dataset, "train_yaml_data": yaml_data, "best_val_nll": float(best_val_nll),
                "best_epoch_X": int(best_epoch_X), "best_epoch_E": int(best_epoch_E),
                "pred_X_state_dict": cpu(state_dict_X), "pred_E_state_dict": cpu(state_dict_E)}, path)
