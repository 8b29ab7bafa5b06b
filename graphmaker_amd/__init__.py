"""graphmaker_amd -- MI355X-native reverse-diffusion denoising step of GraphMaker.

Drop-in for the reference's `model` package on the sampling path:
    from graphmaker_amd import ModelSync, ModelAsync      # reference: model/__init__.py, model/diffusion.py
"""
from .diffusion import ModelAsync, ModelSync  # noqa: F401
from .gnn import GNN, GNNAsymm, GNNTower, LinkPredictor, MLPTower  # noqa: F401

__all__ = ["ModelSync", "ModelAsync", "GNN", "GNNAsymm", "GNNTower", "LinkPredictor", "MLPTower"]
__version__ = "0.1.0"
