"""Loader for the in-tree native libraries.

``core`` is the pybind11 module with the protocol state machine, parsers, reducers,
KV manager and scheduler. ``kernels()`` returns the ctypes handle of the gfx950 HIP
kernel library. Both are built in place by ``build_native.py`` (or
``__graft_entry__.build()``); nothing is installed into site-packages.
"""
from __future__ import annotations

import ctypes
import importlib
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
_ROOT = _HERE.parent.parent


def _import_core():
    try:
        return importlib.import_module("llm_consensus_amd._native._core")
    except ImportError:
        # Build on first use (g++ only, a few seconds); keeps a fresh checkout usable.
        import sys

        sys.path.insert(0, str(_ROOT))
        try:
            import build_native

            build_native.build_core()
        finally:
            sys.path.pop(0)
        return importlib.import_module("llm_consensus_amd._native._core")


core = _import_core()

_KERNELS = None


class KernelLibraryMissing(RuntimeError):
    pass


def kernels_path() -> Path:
    return _HERE / "liblc_kernels.so"


def kernels() -> ctypes.CDLL:
    """ctypes handle of liblc_kernels.so. Raises loudly when the library is absent:
    GPU code paths never fall back to PyTorch ops."""
    global _KERNELS
    if _KERNELS is None:
        p = kernels_path()
        if not p.exists():
            raise KernelLibraryMissing(
                f"{p} not found: run `python build_native.py kernels` (hipcc --offload-arch=gfx950) first"
            )
        _KERNELS = ctypes.CDLL(os.fspath(p), mode=ctypes.RTLD_GLOBAL)
    return _KERNELS
