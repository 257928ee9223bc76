"""Loader of the native operator library `libmmf_amd_ops.so` (mmf_amd/csrc/torch_ops.cpp: TORCH_LIBRARY(mmf_amd), C++ autograd nodes
over the C ABI) and the switchboard between it and the Python package.

On a GPU box the native library IS `torch.ops.mmf_amd.*` — a scripted / saved model runs after `torch.ops.load_library` alone, and the
eager step costs ~60 host calls instead of ~450 ctypes launches — and it owns the state the operators need (bf16 weight shadows, dropout
keys, deferred LayerNorm reductions); the Python package reaches that state through the `_`-prefixed service operators, so both sides
see one cache.  A missing library on a GPU box is an error (no fallback).  Without a GPU (the build container) nothing native can run:
the operators are then declared from Python (mmf_amd/ops.py) so that the host logic can be dry-run against kernel stubs
(tests/native_stub.py); `MMF_AMD_PY_OPS=1` forces that form on a GPU box too (A/B measurements of the host path).
"""
import os

import torch

OPS_LIB_PATH = os.environ.get("MMF_AMD_OPS_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmmf_amd_ops.so")


def _decide():
    if os.environ.get("MMF_AMD_PY_OPS") == "1" or not torch.cuda.is_available():
        return False
    from mmf_amd._native import NativeLibraryError, lib
    lib()           # the kernels' C ABI first (raises when it is missing)
    if not os.path.exists(OPS_LIB_PATH):
        raise NativeLibraryError("mmf_amd operator library not found at %s. Build it with `python -m mmf_amd.csrc.build` "
                                 "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback path." % OPS_LIB_PATH)
    torch.ops.load_library(OPS_LIB_PATH)
    return True


NATIVE = _decide()

# operators whose kernels the native library provides; with NATIVE the Python implementation of each is registered as `_py_<name>`
NATIVE_OPS = ("additive_mask", "visual_masks", "visio_linguistic_embeddings", "transformer_layer", "linear", "layer_norm", "dense_gelu", "linear_tanh",
              "gather_rows", "dropout", "pair_halves", "logit_bce", "masked_lm_head", "masked_region_head")
PY_TWINS = ("visio_linguistic_embeddings", "transformer_layer", "linear", "layer_norm", "dense_gelu", "linear_tanh", "gather_rows",
            "dropout", "pair_halves", "masked_lm_head", "masked_region_head")

# Python-only modes: bit 0 = the fp32-accurate forward path is on, bit 1 = an opt-in experiment hook of mmf_amd/utils/graph.py is active.
# While any is set the native operators forward to their `_py_` twins.
_counts = [0, 0]


def push_mode(bit):
    _counts[bit] += 1
    _sync()


def pop_mode(bit):
    _counts[bit] -= 1
    _sync()


def _sync():
    if NATIVE:
        torch.ops.mmf_amd._set_py_mode((1 if _counts[0] > 0 else 0) | (2 if _counts[1] > 0 else 0))
