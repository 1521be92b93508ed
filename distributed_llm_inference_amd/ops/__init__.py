"""HIP kernel library (`torch.ops.dlia.*`) for gfx950.

`load()` loads the in-tree `_C.so`. On a machine with a GPU a missing or unloadable extension
is a hard error — the engine never silently falls back to eager PyTorch for its hot path.
"""
from __future__ import annotations

import os
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DLIA_OPS_LIB: another build of the same library (same-box A/B runs of a kernel variant, bench/ab_build.py)
_LIB = os.environ.get("DLIA_OPS_LIB") or os.path.join(_HERE, "_C.so")
_loaded = False


def lib_path() -> str:
    return _LIB


def is_loaded() -> bool:
    return _loaded


def load(build_if_missing: bool = False):
    """Load the kernel library and return `torch.ops.dlia`."""
    global _loaded
    if not _loaded:
        if not os.path.exists(_LIB):
            if build_if_missing:
                from . import build as _b
                _b.build()
            else:
                raise RuntimeError(
                    f"HIP kernel library {_LIB} is missing. Build it with "
                    f"`python -m distributed_llm_inference_amd.ops.build` (hipcc, --offload-arch=gfx950).")
        torch.ops.load_library(_LIB)
        _loaded = True
    return torch.ops.dlia


def available() -> bool:
    """True when the kernels can run here: library present AND a GPU visible."""
    return os.path.exists(_LIB) and torch.cuda.is_available()
