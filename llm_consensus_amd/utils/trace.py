"""Span tracing: a per-process timeline in Chrome trace-event format plus roctx ranges.

The reference's only observability is ``log``/``env_logger`` lines
(/root/reference/src/main.rs:2, :352; per-event messages at :93, :111-114, :263, :464).
A GPU-resident ensemble needs to answer different questions: which phase of a consensus
batch the time went to (member generation vs. decision vs. vote), how full the engine
batches were, and how those phases line up with kernels in a profiler capture. So:

* ``span(name, **args)`` is a context manager that records begin/duration/thread/args
  into an in-memory buffer when tracing is on (``CONSENSUS_TRACE=/path/trace.json`` or
  ``enable()``); ``save()`` writes the ``{"traceEvents": [...]}`` file that Perfetto and
  chrome://tracing load.
* Every span is also a roctx range (``roctxRangePushA`` / ``roctxRangePop`` from
  libroctx64, loaded lazily with ctypes), so ``rocprofv3 --marker-trace`` shows the same
  names on the GPU timeline. roctx calls cost tens of nanoseconds when no profiler is
  attached; if the library is absent they are skipped.

Disabled tracing costs one attribute check per span.
"""
from __future__ import annotations

import atexit
import ctypes
import json
import os
import threading
import time
from contextlib import contextmanager
from typing import Any, Dict, List, Optional


class _Roctx:
    def __init__(self) -> None:
        self.lib = None
        self.tried = False

    def load(self):
        if not self.tried:
            self.tried = True
            for name in ("libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so"):
                try:
                    lib = ctypes.CDLL(name)
                    lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                    lib.roctxRangePushA.restype = ctypes.c_int
                    lib.roctxRangePop.restype = ctypes.c_int
                    self.lib = lib
                    break
                except (OSError, AttributeError):
                    continue
        return self.lib


class Tracer:
    def __init__(self) -> None:
        self.enabled = False
        self.use_roctx = False
        self.path: Optional[str] = None
        self.events: List[Dict[str, Any]] = []
        self._lock = threading.Lock()
        self._roctx = _Roctx()
        self._t0 = time.perf_counter()
        self.max_events = 1_000_000

    def enable(self, path: Optional[str] = None, roctx: bool = True) -> None:
        self.enabled, self.path = True, path
        self.use_roctx = roctx and self._roctx.load() is not None

    def disable(self) -> None:
        self.enabled = self.use_roctx = False

    def clear(self) -> None:
        with self._lock:
            self.events = []

    @contextmanager
    def span(self, name: str, cat: str = "consensus", **args):
        if not self.enabled:
            yield
            return
        if self.use_roctx:
            self._roctx.lib.roctxRangePushA(name.encode())
        t0 = time.perf_counter()
        try:
            yield
        finally:
            t1 = time.perf_counter()
            if self.use_roctx:
                self._roctx.lib.roctxRangePop()
            ev = {"name": name, "cat": cat, "ph": "X", "ts": (t0 - self._t0) * 1e6, "dur": (t1 - t0) * 1e6,
                  "pid": os.getpid(), "tid": threading.get_ident() & 0xFFFF}
            if args:
                ev["args"] = args
            with self._lock:
                if len(self.events) < self.max_events:
                    self.events.append(ev)

    def instant(self, name: str, cat: str = "consensus", **args) -> None:
        if not self.enabled:
            return
        ev = {"name": name, "cat": cat, "ph": "i", "s": "p", "ts": (time.perf_counter() - self._t0) * 1e6,
              "pid": os.getpid(), "tid": threading.get_ident() & 0xFFFF, "args": args}
        with self._lock:
            if len(self.events) < self.max_events:
                self.events.append(ev)

    def summary(self) -> Dict[str, Dict[str, float]]:
        """Total / count / mean duration (ms) per span name."""
        out: Dict[str, Dict[str, float]] = {}
        with self._lock:
            for ev in self.events:
                if ev.get("ph") != "X":
                    continue
                s = out.setdefault(ev["name"], {"count": 0, "total_ms": 0.0})
                s["count"] += 1
                s["total_ms"] += ev["dur"] / 1e3
        for s in out.values():
            s["mean_ms"] = s["total_ms"] / max(1, s["count"])
        return out

    def save(self, path: Optional[str] = None) -> Optional[str]:
        path = path or self.path
        if not path:
            return None
        rank = os.environ.get("RANK")
        if rank is not None and "{rank}" not in path and int(os.environ.get("WORLD_SIZE", "1")) > 1:
            root, ext = os.path.splitext(path)
            path = f"{root}.rank{rank}{ext}"
        path = path.replace("{rank}", rank or "0")
        with self._lock:
            data = {"traceEvents": list(self.events), "displayTimeUnit": "ms"}
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(data, f)
        return path


TRACER = Tracer()
span = TRACER.span
instant = TRACER.instant

if os.environ.get("CONSENSUS_TRACE"):
    TRACER.enable(os.environ["CONSENSUS_TRACE"])
    atexit.register(TRACER.save)
