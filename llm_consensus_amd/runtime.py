"""Python face of the native device runtime (csrc/runtime/device_runtime.hip): HBM and pinned
arenas, streams and events, asynchronous copies and the hipGraph cache that a member engine
owns on its GPU. The classes here only hold handles and turn arena memory into tensors / numpy
arrays without copying; every operation is one call into the C++ classes.

An engine on a GPU allocates through this module only: its KV pool is one arena sized at
start-up, the argument blocks of its decode graphs and its staging buffers come from chunked
arenas, its step stream is a runtime stream (handed to torch as an external stream so the op
layer launches on it), and its decode graphs are recorded and replayed by ``GraphCache``.
"""
from __future__ import annotations

import ctypes
from ctypes import c_float, c_int, c_longlong, c_void_p
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from ._native import kernels

_lib = None


class DeviceRuntimeError(RuntimeError):
    """A device-runtime call failed (HIP error or exhausted arena)."""


def _L():
    global _lib
    if _lib is None:
        l = kernels()
        sig = {
            "lc_rt_arena_create": ([c_longlong, c_int], c_void_p),
            "lc_rt_arena_destroy": ([c_void_p], None),
            "lc_rt_arena_alloc": ([c_void_p, c_longlong, c_longlong], c_void_p),
            "lc_rt_arena_mark": ([c_void_p], c_longlong),
            "lc_rt_arena_release": ([c_void_p, c_longlong], c_int),
            "lc_rt_arena_stats": ([c_void_p, ctypes.POINTER(c_longlong)], None),
            "lc_rt_mem_info": ([ctypes.POINTER(c_longlong), ctypes.POINTER(c_longlong)], c_int),
            "lc_rt_stream_create": ([c_int], c_void_p),
            "lc_rt_stream_destroy": ([c_void_p], None),
            "lc_rt_stream_handle": ([c_void_p], c_void_p),
            "lc_rt_stream_sync": ([c_void_p], c_int),
            "lc_rt_event_create": ([c_int], c_void_p),
            "lc_rt_event_destroy": ([c_void_p], None),
            "lc_rt_event_record": ([c_void_p, c_void_p], c_int),
            "lc_rt_event_sync": ([c_void_p], c_int),
            "lc_rt_event_query": ([c_void_p], c_int),
            "lc_rt_stream_wait_event": ([c_void_p, c_void_p], c_int),
            "lc_rt_event_elapsed_ms": ([c_void_p, c_void_p], c_float),
            "lc_rt_copy_async": ([c_void_p, c_void_p, c_longlong, c_int, c_void_p], c_int),
            "lc_rt_graphs_create": ([], c_void_p),
            "lc_rt_graphs_destroy": ([c_void_p], None),
            "lc_rt_graphs_begin": ([c_void_p, c_void_p], c_int),
            "lc_rt_graphs_end": ([c_void_p, c_void_p, c_longlong], c_longlong),
            "lc_rt_graphs_abort": ([c_void_p, c_void_p], None),
            "lc_rt_graphs_launch": ([c_void_p, c_longlong, c_void_p], c_int),
            "lc_rt_graphs_erase": ([c_void_p, c_longlong], c_int),
            "lc_rt_graphs_size": ([c_void_p], c_longlong),
            "lc_rt_graphs_launches": ([c_void_p], c_longlong),
        }
        for name, (args, res) in sig.items():
            fn = getattr(l, name)
            fn.argtypes, fn.restype = args, res
        _lib = l
    return _lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise DeviceRuntimeError(f"{what} failed: HIP error {rc}")


def current_stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


HBM_HEADROOM_BYTES = 1 << 30      # what an arena reservation leaves free for everybody's small allocations


def require_hbm(nbytes: int, what: str) -> None:
    """Raises unless ``nbytes`` (plus headroom) are free on the current device, all processes
    counted: the check in front of every large allocation that is not an Arena (model weights)."""
    free, total = mem_info()
    if nbytes > free - HBM_HEADROOM_BYTES:
        raise DeviceRuntimeError(f"{what} needs {nbytes} bytes of HBM; {free} of {total} are free on this device "
                                 f"(all processes counted)")


def mem_info() -> Tuple[int, int]:
    """(free, total) bytes of the current device."""
    f, t = c_longlong(0), c_longlong(0)
    _check(_L().lc_rt_mem_info(ctypes.byref(f), ctypes.byref(t)), "hipMemGetInfo")
    return int(f.value), int(t.value)


class _DeviceSpan:
    """A range of arena memory seen through ``__cuda_array_interface__``; torch builds a tensor
    on it without copying and keeps this object (and through it the arena) alive."""

    def __init__(self, owner, ptr: int, nbytes: int) -> None:
        self.owner = owner
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class Arena:
    """One region of HBM (``pinned=False``) or of pinned host memory, bump-allocated."""

    def __init__(self, nbytes: int, pinned: bool = False, device: Optional[torch.device] = None) -> None:
        self.pinned, self.device = pinned, device
        self.nbytes = int(nbytes)
        if not pinned:
            # Never ask for more than the card has free RIGHT NOW, other processes included. The
            # driver may grant such a request by evicting another process's memory, which stalls
            # that process's queues (two ranks rehearsing on one GPU did that to each other); a
            # refusal here is an ordinary error the caller can report.
            free, total = mem_info()
            if self.nbytes > free - HBM_HEADROOM_BYTES:
                raise DeviceRuntimeError(f"cannot reserve a HBM arena of {self.nbytes} bytes: {free} of {total} bytes "
                                         f"are free on this device (all processes counted)")
        self._h = _L().lc_rt_arena_create(self.nbytes, 1 if pinned else 0)
        if not self._h:
            kind = "pinned host" if pinned else "HBM"
            raise DeviceRuntimeError(f"cannot reserve a {kind} arena of {self.nbytes} bytes")

    def alloc_ptr(self, nbytes: int, align: int = 256) -> int:
        p = _L().lc_rt_arena_alloc(self._h, int(nbytes), int(align))
        if not p:
            raise DeviceRuntimeError(f"arena exhausted: {nbytes} bytes requested, {self.stats()}")
        return int(p)

    def try_alloc_ptr(self, nbytes: int, align: int = 256) -> int:
        return int(_L().lc_rt_arena_alloc(self._h, int(nbytes), int(align)) or 0)

    def tensor(self, shape: Sequence[int], dtype: torch.dtype, ptr: Optional[int] = None) -> torch.Tensor:
        """Device tensor of ``shape`` on arena memory (contents undefined)."""
        n = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
        p = self.alloc_ptr(max(n, 1)) if ptr is None else ptr
        flat = torch.as_tensor(_DeviceSpan(self, p, max(n, 1)), device=self.device)
        return flat[:n].view(dtype).view(*shape)

    def array(self, shape: Sequence[int], dtype, ptr: Optional[int] = None) -> np.ndarray:
        """numpy view of pinned arena memory."""
        if not self.pinned:
            raise DeviceRuntimeError("host arrays come from pinned arenas")
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) * dt.itemsize
        p = self.alloc_ptr(max(n, 1)) if ptr is None else ptr
        buf = (ctypes.c_uint8 * max(n, 1)).from_address(p)
        buf._arena = self               # the array keeps its buffer object alive, and that the arena
        return np.frombuffer(buf, dtype=np.uint8, count=n).view(dt).reshape(shape)

    def mark(self) -> int:
        return int(_L().lc_rt_arena_mark(self._h))

    def release(self, mark: int) -> None:
        if _L().lc_rt_arena_release(self._h, int(mark)) != 0:
            raise DeviceRuntimeError("arena release past the current top")

    def stats(self) -> dict:
        out = (c_longlong * 4)()
        _L().lc_rt_arena_stats(self._h, out)
        return {"capacity": int(out[0]), "used": int(out[1]), "peak": int(out[2]), "allocations": int(out[3])}

    def __del__(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.lc_rt_arena_destroy(h)


class ChunkedArena:
    """Arenas added on demand, for the many small long-lived blocks of an engine (argument blocks
    of decode graphs, staging buffers): a request that does not fit the newest chunk opens a new
    one of ``chunk_bytes`` (or of the request's size if that is larger)."""

    def __init__(self, chunk_bytes: int, pinned: bool = False, device: Optional[torch.device] = None) -> None:
        self.chunk_bytes, self.pinned, self.device = int(chunk_bytes), pinned, device
        self.chunks: List[Arena] = []

    def _place(self, nbytes: int, align: int = 256) -> Tuple[Arena, int]:
        if self.chunks:
            p = self.chunks[-1].try_alloc_ptr(nbytes, align)
            if p:
                return self.chunks[-1], p
        self.chunks.append(Arena(max(self.chunk_bytes, nbytes + align), self.pinned, self.device))
        return self.chunks[-1], self.chunks[-1].alloc_ptr(nbytes, align)

    def tensor(self, shape: Sequence[int], dtype: torch.dtype) -> torch.Tensor:
        n = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
        a, p = self._place(max(n, 1))
        return a.tensor(shape, dtype, ptr=p)

    def array(self, shape: Sequence[int], dtype) -> np.ndarray:
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        a, p = self._place(max(n, 1))
        return a.array(shape, dtype, ptr=p)

    def stats(self) -> dict:
        s = [c.stats() for c in self.chunks]
        return {"chunks": len(s), "capacity": sum(x["capacity"] for x in s), "used": sum(x["used"] for x in s)}


class Event:
    def __init__(self, timing: bool = False) -> None:
        self._h = _L().lc_rt_event_create(1 if timing else 0)
        if not self._h:
            raise DeviceRuntimeError("hipEventCreate failed")

    def record(self, stream_ptr: Optional[int] = None) -> "Event":
        _check(_L().lc_rt_event_record(self._h, stream_ptr if stream_ptr is not None else current_stream_ptr()),
               "hipEventRecord")
        return self

    def synchronize(self) -> None:
        _check(_L().lc_rt_event_sync(self._h), "hipEventSynchronize")

    def query(self) -> bool:
        r = _L().lc_rt_event_query(self._h)
        if r < 0:
            raise DeviceRuntimeError(f"hipEventQuery failed: HIP error {-r}")
        return r == 1

    def elapsed_ms(self, later: "Event") -> float:
        return float(_L().lc_rt_event_elapsed_ms(self._h, later._h))

    def __del__(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.lc_rt_event_destroy(h)


class Stream:
    """A runtime-owned HIP stream. ``torch`` is the same stream as a ``torch.cuda.ExternalStream``:
    ``with torch.cuda.stream(s.torch)`` makes the op layer launch on it."""

    def __init__(self, device: torch.device, priority: int = 0) -> None:
        device = indexed(device)
        with torch.cuda.device(device):
            self._h = _L().lc_rt_stream_create(int(priority))
        if not self._h:
            raise DeviceRuntimeError("hipStreamCreate failed")
        self.ptr = int(_L().lc_rt_stream_handle(self._h))
        self.torch = torch.cuda.ExternalStream(self.ptr, device=device)

    def synchronize(self) -> None:
        _check(_L().lc_rt_stream_sync(self.ptr), "hipStreamSynchronize")

    def wait_event(self, e: Event) -> None:
        _check(_L().lc_rt_stream_wait_event(self.ptr, e._h), "hipStreamWaitEvent")

    def __del__(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.lc_rt_stream_destroy(h)


H2D, D2H, D2D = 0, 1, 2


def copy_async(dst_ptr: int, src_ptr: int, nbytes: int, kind: int, stream_ptr: Optional[int] = None) -> None:
    """hipMemcpyAsync on the given (default: current) stream. Host sides must be pinned arena memory
    for the copy to be asynchronous."""
    _check(_L().lc_rt_copy_async(dst_ptr, src_ptr, int(nbytes), kind,
                                 stream_ptr if stream_ptr is not None else current_stream_ptr()), "hipMemcpyAsync")


class GraphCache:
    """hipGraph executables by integer key. ``capture(key)`` is a context manager that records
    everything launched on the current stream; tensors the recorded code allocates come from the
    cache's private torch memory pool, so their addresses stay valid for every replay and are
    never handed to other allocations. Captures are thread-local (see device_runtime.hip)."""

    def __init__(self, device: torch.device) -> None:
        self.device = device
        self._h = _L().lc_rt_graphs_create()
        self.pool = torch.cuda.MemPool()
        self.nodes: dict = {}

    class _Capture:
        def __init__(self, cache: "GraphCache", key: int) -> None:
            self.cache, self.key = cache, key

        def __enter__(self):
            c = self.cache
            torch.cuda.synchronize(c.device)
            self.stream = current_stream_ptr()
            if self.stream == 0:
                raise DeviceRuntimeError("the legacy default stream cannot be captured: run under an own stream")
            self.mem = torch.cuda.use_mem_pool(c.pool, device=c.device)
            self.mem.__enter__()
            rc = _L().lc_rt_graphs_begin(c._h, self.stream)
            if rc != 0:
                self.mem.__exit__(None, None, None)
                raise DeviceRuntimeError(f"hipStreamBeginCapture failed: HIP error {rc}")
            return self

        def __exit__(self, et, ev, tb):
            c = self.cache
            try:
                if et is not None:
                    _L().lc_rt_graphs_abort(c._h, self.stream)     # never leave a capture open behind an error
                    return False
                n = int(_L().lc_rt_graphs_end(c._h, self.stream, int(self.key)))
                if n <= 0:
                    raise DeviceRuntimeError(f"hipStreamEndCapture / hipGraphInstantiate failed: HIP error {-n}")
                c.nodes[self.key] = n
            finally:
                self.mem.__exit__(None, None, None)
            return False

    def capture(self, key: int) -> "GraphCache._Capture":
        return GraphCache._Capture(self, key)

    def launch(self, key: int, stream_ptr: Optional[int] = None) -> None:
        rc = _L().lc_rt_graphs_launch(self._h, int(key), stream_ptr if stream_ptr is not None else current_stream_ptr())
        if rc != 0:
            raise DeviceRuntimeError(f"hipGraphLaunch of graph {key} failed: " + ("unknown key" if rc < 0 else f"HIP error {rc}"))

    def erase(self, key: int) -> bool:
        self.nodes.pop(key, None)
        return _L().lc_rt_graphs_erase(self._h, int(key)) == 0

    def __len__(self) -> int:
        return int(_L().lc_rt_graphs_size(self._h))

    @property
    def launches(self) -> int:
        return int(_L().lc_rt_graphs_launches(self._h))

    def __del__(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.lc_rt_graphs_destroy(h)


def indexed(device) -> torch.device:
    """``device`` with an explicit index ("cuda" means the current device)."""
    d = torch.device(device)
    return d if d.index is not None else torch.device(d.type, torch.cuda.current_device())


class DeviceRuntime:
    """Everything one engine owns on its GPU: the step stream, the KV arena, chunked arenas for
    small device and pinned blocks, and the graph cache."""

    def __init__(self, device: torch.device, own_stream: bool = True) -> None:
        self.device = indexed(device)
        with torch.cuda.device(self.device):
            self.stream = Stream(self.device) if own_stream else None
            self.small = ChunkedArena(32 << 20, pinned=False, device=self.device)
            self.pinned = ChunkedArena(8 << 20, pinned=True)
            self.graphs = GraphCache(self.device)
        self.kv: Optional[Arena] = None
        self._next_key = 0

    def reserve_kv(self, nbytes: int) -> Arena:
        with torch.cuda.device(self.device):
            self.kv = Arena(nbytes, pinned=False, device=self.device)
        return self.kv

    def new_key(self) -> int:
        self._next_key += 1
        return self._next_key

    def stats(self) -> dict:
        s = {"rt.graphs": len(self.graphs), "rt.graph_launches": self.graphs.launches,
             "rt.small_arena_bytes": self.small.stats()["used"], "rt.pinned_arena_bytes": self.pinned.stats()["used"]}
        if self.kv is not None:
            s["rt.kv_arena_bytes"] = self.kv.stats()["capacity"]
        return s
