"""ctypes binding of graphmaker_amd/lib/libgraphmaker_hip.so (C ABI: include/graphmaker_hip.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  If the shared object is missing or a call
fails, this module raises -- it never routes around the HIP path.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgraphmaker_hip.so")

# every symbol include/graphmaker_hip.h declares (tests check the .so exports all of them)
EXPORTS = [
    "gm_version", "gm_last_error", "gm_device_cus",
    "gm_bitmap_from_dense_i64", "gm_dense_i64_from_bitmap", "gm_symmetrize_upper", "gm_csr_rowptr", "gm_csr_rowptr_capped", "gm_csr_fill",
    "gm_bitmap_from_upper_edges", "gm_upper_edges_ranges", "gm_bitmap_from_edge_segments", "gm_spmm_mean", "gm_linear", "gm_layernorm", "gm_tower_workspace_bytes",
    "gm_tower_forward", "gm_tower_stage", "gm_tower_prep_floats", "gm_tower_prepare", "gm_onehot_prep_floats", "gm_onehot_prepare", "gm_onehot_linear", "gm_hadamard_gather", "gm_pair_score_sample", "gm_pair_score_sample_ranges", "gm_pair_score_sample_bf16", "gm_attr_posterior_sample", "gm_attr_posterior_sample_diff", "gm_head_diff_prepare", "gm_onehot_from_x",
    "gm_x_from_onehot", "gm_prior_Y", "gm_prior_X", "gm_prior_E", "gm_prior_E_static",
    "gm_noise_E", "gm_noise_X", "gm_bitmap_from_onehot_f32", "gm_x_from_onehot3d",
    "gm_gemm_workspace_bytes", "gm_gemm", "gm_mlp2_hidden_grad_workspace_bytes", "gm_mlp2_hidden_grad", "gm_colsum_workspace_bytes", "gm_colsum", "gm_layernorm_fwd",
    "gm_layernorm_bwd_workspace_bytes", "gm_layernorm_bwd", "gm_spmm_mean_bwd", "gm_embedding_bwd_workspace_bytes", "gm_embedding_bwd",
    "gm_hadamard_gather_bwd", "gm_softmax_ce2_workspace_bytes", "gm_softmax_ce2", "gm_ce2_diff", "gm_head_diff_bwd",
    "gm_relu_bwd", "gm_dropout", "gm_embedding_fwd", "gm_denoise_match2", "gm_bitmap_lookup",
    "gm_triangles_per_node", "gm_same_class_degree", "gm_bitmap_square", "gm_class_bitmaps", "gm_masked_degree",
]


class GraphMakerHipError(RuntimeError):
    pass


class TowerDims(C.Structure):
    _fields_ = [("in_X", C.c_int), ("num_classes_Y", C.c_int), ("hidden_t", C.c_int), ("hidden_X", C.c_int),
                ("hidden_Y", C.c_int), ("num_layers", C.c_int), ("out_size", C.c_int), ("is_mlp", C.c_int)]

    @property
    def hidden_cat(self):
        return (self.num_layers + 1) * (self.hidden_X + self.hidden_Y) + self.hidden_t


_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GraphMakerHipError(
                f"{LIB_PATH} is missing: build it with `make -C graphmaker_amd/csrc` (hipcc, gfx950). "
                "graphmaker_amd has no fallback path.")
        lib = C.CDLL(LIB_PATH)
        lib.gm_last_error.restype = C.c_char_p
        lib.gm_version.restype = C.c_int
        for name in ("gm_tower_prep_floats", "gm_onehot_prep_floats", "gm_gemm_workspace_bytes", "gm_mlp2_hidden_grad_workspace_bytes", "gm_colsum_workspace_bytes", "gm_layernorm_bwd_workspace_bytes",
                     "gm_softmax_ce2_workspace_bytes", "gm_embedding_bwd_workspace_bytes"):
            getattr(lib, name).restype = C.c_int64
        _lib = lib
    return _lib


def _check(rc, what):
    if rc != 0:
        msg = load().gm_last_error().decode(errors="replace")
        raise GraphMakerHipError(f"{what} failed (rc={rc}): {msg}")


def words_per_row(n):
    return ((n + 31) // 32 + 3) // 4 * 4


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def require(t, dtype, what, device=None):
    if t.dtype != dtype:
        raise TypeError(f"{what}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{what}: must be contiguous")
    if not t.is_cuda:
        raise GraphMakerHipError(f"{what}: graphmaker_amd runs on a ROCm device only (tensor is on {t.device})")
    if device is not None and t.device != device:
        raise ValueError(f"{what}: on {t.device}, expected {device}")
    return t


def call(name, *args):
    fn = getattr(load(), name)
    _check(fn(*args), name)


I64 = C.c_int64
I32 = C.c_int
U64 = C.c_uint64
U32 = C.c_uint32
F32 = C.c_float
