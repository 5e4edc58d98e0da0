"""ctypes bindings of the gfx950 kernel library (csrc/kernels/*.hip).

Every launcher takes raw device pointers plus the current HIP stream, so launches are
captured by ``torch.cuda.graph`` (hipGraph) like any other work on the stream. Nothing
here falls back to PyTorch: a missing library or a non-zero launcher status raises.
"""
from __future__ import annotations

import ctypes
import os
import sys
from ctypes import c_float, c_int, c_longlong, c_void_p
from typing import Optional

import torch

from .._native import kernels

_P, _I, _F, _L = c_void_p, c_int, c_float, c_longlong

_SIGNATURES = {
    "lc_embed_gather": [_P, _P, _P, _I, _I, _I, _P, _P, _P],
    "lc_rmsnorm": [_P, _P, _P, _P, _I, _I, _F, _P, _I, _P],
    "lc_copy_blocks": [_P, _P, _P, _I, _L, _L, _I, _P],
    "lc_embed_pool": [_P, _P, _P, _P, _I, _I, _I, _P],
    "lc_l2_normalize": [_P, _P, _I, _I, _P],
    "lc_gemm_tile": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "lc_gemm256": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "lc_gemm256w": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "lc_gemm_panel": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "lc_kgemm": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "lc_splitk_reduce": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "lc_gemm_grouped": [_P, _L, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "lc_gemm256_splitk": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "lc_gemm256_grouped": [_P, _L, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "lc_gemm_panel_grouped": [_P, _L, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "lc_rope_kv_write": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P],
    "lc_attn_prefill": [_P, _I, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _I, _P],
    "lc_attn_decode": [_P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _I, _P],
    "lc_attn_prefix": [_P, _I, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _F, _I, _P],
    "lc_attn_merge": [_P, _P, _P, _I, _I, _I, _P],
    "lc_fill_f32": [_P, _F, _L, _P],
    "lc_sample": [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "lc_logprob_gather": [_P, _I, _I, _P, _P, _I, _I, _P, _I, _P],
    "lc_logit_targets_shard": [_P, _I, _I, _P, _P, _I, _I, _I, _I, _P, _I, _P],
    "lc_logprob_merge": [_P, _I, _I, _I, _I, _P, _P],
    "lc_accumulate_f32": [_P, _P, _L, _P],
    "lc_concat_cols": [_P, _I, _I, _I, _I, _P, _P],
    "lc_pad_cols_f32": [_P, _P, _I, _I, _I, _F, _P],
    "lc_majority_vote": [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "lc_cosine_medoid": [_P, _I, _I, _I, _L, _P, _P, _P, _P],
    "lc_judge_reduce": [_P, _I, _I, _I, _P, _P, _P, _P],
    "lc_answer_extract": [_P, _P, _I, _I, _P, _I, _P, _I, _I, _P, _P],
    "lc_moe_router": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P],
    "lc_moe_permute": [_P, _I, _I, _I, _P, _P, _P, _P],
    "lc_moe_combine": [_P, _P, _P, _P, _I, _I, _I, _P],
    "lc_ep_pack": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "lc_ep_recv_index": [_P, _I, _I, _I, _P, _P, _P],
    "lc_scatter_rows": [_P, _P, _P, _I, _I, _P, _P],
}

_lib = None


class KernelLaunchError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        l = kernels()
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = argtypes
            fn.restype = c_int
        _lib = l
    return _lib


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# While a list is installed here, every launch is also appended to it as (function, args with
# the stream): the engine records the launches of one decode step whose buffers are all static
# and replays them later without going through the ops layer again (see replay()).
_recording: Optional[list] = None


def start_recording() -> list:
    global _recording
    _recording = []
    return _recording


def stop_recording() -> None:
    global _recording
    _recording = None


# LC_TRACE_LAUNCHES=1: every launch is announced on stderr before it is issued and the device is
# synchronised after it, so that when a kernel faults the last line names it (a fault ends the
# process before anything else can be reported). Debugging aid only: it serialises everything.
_TRACE = os.environ.get("LC_TRACE_LAUNCHES", "0") == "1"


def call(name: str, *args) -> None:
    fn = getattr(lib(), name)
    st = stream()
    if _TRACE:
        sys.stderr.write(f"[launch] {name} {[a for a in args if isinstance(a, (int, float))][-10:]}\n")
        sys.stderr.flush()
    rc = fn(*args, st)
    if _TRACE and not torch.cuda.is_current_stream_capturing():
        torch.cuda.synchronize()
    if rc != 0:
        what = "invalid arguments for this kernel" if rc < 0 else f"HIP error {rc}"
        raise KernelLaunchError(f"{name} failed: {what}")
    if _recording is not None:
        _recording.append((fn, args + (st,), name))


def replay(cmds: list) -> None:
    """Re-issues recorded launches: one ctypes call each, no allocation, no shape logic."""
    for fn, args, name in cmds:
        rc = fn(*args)
        if rc != 0:
            raise KernelLaunchError(f"{name} failed on replay: {rc}")
