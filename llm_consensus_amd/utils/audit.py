"""Audit of what a stretch of code launches on the GPU besides the kernel library.

Every kernel of ``csrc/kernels`` is launched through ctypes and is invisible to PyTorch's
dispatcher; every kernel PyTorch itself launches (``at::native::*``) comes from an aten
operator call on a CUDA tensor. ``AtenAudit`` records those calls, leaving out the operators
that only allocate memory or make views, and plain same-dtype copies (which run as
``hipMemcpyAsync``). A serving step that records nothing consists of the kernel library's
launches and buffer copies only. The reference's members do no local compute at all
(/root/reference/src/main.rs:82-86); this is the check that ours do theirs in our own kernels.
"""
from __future__ import annotations

from typing import Dict, List

import torch
from torch.utils._python_dispatch import TorchDispatchMode

# allocation and metadata-only operators: no device work
_NO_KERNEL = {
    "aten.empty.memory_format", "aten.empty_strided.default", "aten.empty_like.default", "aten.new_empty.default",
    "aten.new_empty_strided.default", "aten.view.default", "aten.view.dtype", "aten._unsafe_view.default",
    "aten.slice.Tensor", "aten.select.int", "aten.as_strided.default", "aten.detach.default", "aten.alias.default",
    "aten.unsqueeze.default", "aten.squeeze.dim", "aten.squeeze.default", "aten.t.default", "aten.transpose.int",
    "aten.permute.default", "aten.expand.default", "aten._reshape_alias.default", "aten.split.Tensor",
    "aten.unbind.int", "aten.lift_fresh.default", "aten.unfold.default", "aten.narrow.default",
    "aten.set_.source_Storage_storage_offset", "aten.set_.source_Storage", "aten.is_pinned.default",
    "aten.record_stream.default", "aten.resize_.default", "aten._local_scalar_dense.default",
}


def _tensors(x, out: list) -> None:
    if isinstance(x, torch.Tensor):
        out.append(x)
    elif isinstance(x, (list, tuple)):
        for y in x:
            _tensors(y, out)
    elif isinstance(x, dict):
        for y in x.values():
            _tensors(y, out)


class AtenAudit(TorchDispatchMode):
    """``with AtenAudit() as a: ...`` then ``a.kernels`` maps operator name -> call count for
    every aten call that made PyTorch launch device work of its own."""

    def __init__(self) -> None:
        super().__init__()
        self.kernels: Dict[str, int] = {}
        self.copies = 0

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        out = func(*args, **kwargs)
        name = str(func)
        if name in _NO_KERNEL:
            return out
        ts: List[torch.Tensor] = []
        _tensors(args, ts)
        _tensors(kwargs, ts)
        _tensors(out, ts)
        if not any(t.is_cuda for t in ts):
            return out
        if name in ("aten.copy_.default", "aten._to_copy.default", "aten.clone.default"):
            src = args[1] if name == "aten.copy_.default" else args[0]
            dst = args[0] if name == "aten.copy_.default" else out
            if (isinstance(src, torch.Tensor) and isinstance(dst, torch.Tensor) and src.dtype == dst.dtype
                    and src.is_contiguous() and dst.is_contiguous()):
                self.copies += 1          # a buffer copy (hipMemcpyAsync), not a kernel of PyTorch's
                return out
        self.kernels[name] = self.kernels.get(name, 0) + 1
        return out
