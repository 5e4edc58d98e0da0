"""Member engine: one model on one device (or TP group), served with paged KV,
continuous batching, forked sequence groups and hipGraph-captured decode steps.

This is what replaces the reference's ``call_gemini`` (/root/reference/src/main.rs:82-86):
instead of one HTTPS round trip per message, requests from every consensus session
and persona that shares this model are queued (``submit``), batched by the native
scheduler (csrc/core/scheduler.cpp) and advanced together (``step``). The host only
moves small int32 arrays per step; tokens are sampled on device.
"""
from __future__ import annotations

import contextlib
import os
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .. import ops
from .. import runtime as rt
from .._native import core
from ..models.transformer import AttnMeta, KVCache, TransformerModel
from ..utils.log import get_logger
from ..utils.trace import span

log = get_logger("consensus.engine")
PK = core.StepPlan.Kind
# One member engine prefills at a time while the others decode (see run_engines).
STAGGER_PREFILL = os.environ.get("CONSENSUS_STAGGER_PREFILL", "1") == "1"
PREFIX_MIN_GROUP_ROWS = int(os.environ.get("CONSENSUS_PREFIX_MIN_GROUP_ROWS", "6"))


def serialize_streams() -> bool:
    return os.environ.get("CONSENSUS_SERIALIZE_STREAMS", "0") == "1"


@dataclass
class RequestOutput:
    request_id: int
    kind: "core.RequestKind"
    token_ids: List[List[int]] = field(default_factory=list)        # per chain
    cum_logprobs: List[float] = field(default_factory=list)
    finish_reasons: List[str] = field(default_factory=list)
    decision_logprobs: Optional[List[List[float]]] = None            # [chain][target]
    scores: Optional[List[float]] = None                             # Score requests
    embedding: Optional[np.ndarray] = None                           # Embed requests
    prompt_len: int = 0
    cached_tokens: int = 0
    submitted_at: float = 0.0
    finished_at: float = 0.0


@dataclass
class _ReqState:
    kind: "core.RequestKind"
    n: int
    decision_ids: Optional[List[int]] = None
    score_ids: Optional[List[int]] = None
    decisions: Dict[int, List[float]] = field(default_factory=dict)  # chain seed -> latest decision log-probs
    scores: Optional[List[float]] = None
    embed_sum: Optional[torch.Tensor] = None
    embed_count: int = 0
    seed: int = 0
    submitted_at: float = 0.0


class _KeepAlive(torch.utils._python_dispatch.TorchDispatchMode):
    """Holds a reference to every tensor produced while active, so that device buffers whose
    addresses were recorded in a launch list are never returned to the allocator."""

    def __init__(self, keep: list) -> None:
        super().__init__()
        self.keep = keep

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        if isinstance(out, torch.Tensor):
            self.keep.append(out)
        elif isinstance(out, (tuple, list)):
            self.keep.extend(o for o in out if isinstance(o, torch.Tensor))
        return out


class _DecodeGraph:
    """Static buffers + captured graph of one decode step for a batch bucket."""

    def __init__(self, eng: "Engine", bucket: int, with_prefix: bool, ctx_class: int) -> None:
        dev = eng.device
        self.bucket, self.with_prefix, self.ctx_class = bucket, with_prefix, ctx_class
        self.gmax = max(1, bucket // 2)
        B, MB, C, G = bucket, eng.max_blocks_cap, eng.max_decision, self.gmax
        # All per-step inputs live in ONE int32 arena: a pinned host mirror is filled with
        # numpy and shipped with a single H2D copy per step. Field order keeps the int64
        # seeds 8-byte aligned.
        fields = [("seeds", 2 * B), ("token_ids", B), ("positions", B), ("slot_mapping", B), ("ctx_lens", B),
                  ("top_k", B), ("steps", B), ("kv_begin", B), ("src_rows", B), ("temperature", B), ("top_p", B),
                  ("decision_ids", B * C), ("group_first", G), ("group_count", G), ("group_shared", G),
                  ("block_tables", B * MB)]
        total = sum(n for _, n in fields)
        if dev.type == "cuda":
            # device block and pinned host mirror from the engine's runtime arenas
            self.arena = ops.zero_(eng.rt.small.tensor((total,), torch.int32))
            hnp = eng.rt.pinned.array((total,), np.int32)
            hnp[:] = 0
            self.host = None
        else:
            self.arena = torch.zeros(total, dtype=torch.int32, device=dev)
            self.host = torch.zeros(total, dtype=torch.int32)
            hnp = self.host.numpy()
        self.host_np = hnp
        self.h: Dict[str, np.ndarray] = {}
        off = 0
        for name, n in fields:
            d, hv = self.arena[off:off + n], hnp[off:off + n]
            if name == "seeds":
                d, hv = d.view(torch.int64), hv.view(np.int64)
            elif name in ("temperature", "top_p"):
                d, hv = d.view(torch.float32), hv.view(np.float32)
            elif name == "decision_ids":
                d, hv = d.view(B, C), hv.reshape(B, C)
            elif name == "block_tables":
                d, hv = d.view(B, MB), hv.reshape(B, MB)
            setattr(self, name, d)
            self.h[name] = hv
            off += n
        self.h["slot_mapping"][:] = -1
        self.h["src_rows"][:] = -1
        self.h["top_p"][:] = 1.0
        self.feed_set = False                 # the mirror's src_rows hold something other than -1
        self.nsplit = ops.choose_kv_splits(bucket, eng.model.hkv, ctx_class)
        # The shared-prefix pass scans each group's prompt once per kv head; a wave that walks
        # the whole prompt alone is a chain of dependent loads, so the prompt is cut into pieces of
        # at least 128 keys: four per workgroup (merged on chip), and as many workgroups per group
        # as it takes to have ~1024 waves (a step of this bucket has about bucket * Hq / 32 row
        # tiles x kv-head pairs). Every further piece costs a partial slot per row in HBM: with 256
        # rows (16 prompts x 16 chains) one workgroup per (group, kv head, row tile) measured
        # 7.74 ms per step against 7.97 with two. Blocks of the (statically sized) group slots
        # that are empty at run time exit at once.
        prefix_waves = max(1, bucket * eng.model.hq // 8)
        hbm_pieces = max(1, min(-(-1024 // prefix_waves), ctx_class // 512))
        self.prefix_splits = max(1, min(4 * hbm_pieces, ctx_class // 128))
        # Two recorded forms of the step: without and with the gather of decision log-probs (a
        # step needs it only when a row with decision targets can end in it). graph[v] is the key
        # of the hipGraph in the engine's graph cache, outs[v] its static result tensors.
        self.graph: Dict[bool, int] = {}
        self.pieces: Dict[bool, List[tuple]] = {}
        self.lists = False
        self.outs: Dict[bool, tuple] = {}
        # Results land in pinned buffers (numpy views of pinned arena memory). Two steps of an
        # engine can be in flight (the second launched ahead of the first one's results), so
        # there are two landing sets, used alternately.
        self.land = [_Landing(eng, B, C), _Landing(eng, B, C)]
        self.copied = rt.Event() if dev.type == "cuda" else None   # the mirror's last upload
        self.copy_pending = False
        self.rows_key: Optional[np.ndarray] = None
        self._graphs = eng.rt.graphs if dev.type == "cuda" else None

    def begin_fill(self) -> None:
        """The host mirror may be rewritten once the copy that last shipped it has read it."""
        if self.copy_pending:
            self.copied.synchronize()
            self.copy_pending = False

    def upload(self) -> None:
        """The host mirror goes to the device block with one asynchronous copy on the step's stream."""
        if self.host is None:
            rt.copy_async(self.arena.data_ptr(), self.host_np.ctypes.data, self.host_np.nbytes, rt.H2D)
            self.copied.record()
            self.copy_pending = True
        else:
            self.arena.copy_(self.host)

    def meta(self, eng: "Engine") -> AttnMeta:
        return AttnMeta(
            is_prefill=False, block_tables=self.block_tables, ctx_lens=self.ctx_lens, nsplit=self.nsplit,
            groups=(self.group_first, self.group_count, self.group_shared) if self.with_prefix else None,
            kv_begin=self.kv_begin if self.with_prefix else None,
            max_group_rows=eng.max_group_rows if self.with_prefix else 0, prefix_splits=self.prefix_splits,
            token_feed=(eng.last_tokens, self.src_rows))

    def run_body(self, eng: "Engine", with_dec: bool = True):
        """One decode step over the static buffers. The sampled tokens go to the engine's
        ``last_tokens`` (row b of the step at index b), where the next step can pick them up on
        the device (rows of the mirror's ``src_rows`` that are >= 0)."""
        hidden = eng.model.forward(self.token_ids, self.positions, self.slot_mapping, self.meta(eng), eng.kv)
        logits = eng.model.logits(hidden)
        tok, lp = ops.sample(logits, self.temperature, self.top_p, self.top_k, self.seeds, self.steps,
                             out_tok=eng.last_tokens)
        dec = ops.logprob_gather(logits, self.decision_ids) if with_dec else None
        return tok, lp, dec

    def ready(self, eng: "Engine", with_dec: bool) -> bool:
        """Whether this step form can be launched without recording anything first."""
        if not (eng.use_graphs or eng.piecewise) or not eng.model.decode_capturable(self.bucket):
            return True
        return with_dec in self.graph or with_dec in self.pieces

    def capture(self, eng: "Engine", with_dec: bool = True) -> None:
        """Records the step into the engine's graph cache (runtime.GraphCache: thread-local stream
        capture, so HIP calls made meanwhile by other threads -- RCCL's watchdog, the HTTP front
        end, another engine's loopback rank -- do not invalidate it; tensors allocated while
        recording come from the cache's private pool and keep their addresses)."""
        self.run_body(eng, with_dec)                # warm-up: allocator, lazy init, peer channels
        torch.cuda.synchronize()
        key = eng.rt.new_key()
        with eng._capture_stream():
            with self._graphs.capture(key):
                self.outs[with_dec] = self.run_body(eng, with_dec)
        self.graph[with_dec] = key

    def capture_piecewise(self, eng: "Engine", with_dec: bool = True) -> None:
        """Tensor-parallel decode step as a chain of hipGraphs cut at the collectives: every
        stretch of kernels between two all-reduces (and up to the logits all-gather) is one
        graph, the collectives themselves are issued eagerly between the replays. A step of an
        L-layer model then costs 2L+2 graph launches and 2L+1 RCCL calls on the host instead
        of ~10 kernel launches per layer, and no RCCL call is ever inside a stream capture."""
        model = eng.model
        inner = model.comm
        s = torch.cuda.Stream(device=eng.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.run_body(eng, with_dec)            # warm-up: allocator, lazy init, RCCL channels
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        pieces: List[tuple] = []
        cur: List[Optional[object]] = [None]

        def begin() -> None:
            cap = self._graphs.capture(eng.rt.new_key())
            cap.__enter__()
            cur[0] = cap

        def end() -> None:
            cap, cur[0] = cur[0], None
            cap.__exit__(None, None, None)
            pieces.append((cap.key, None))

        class _Cut:
            """Stands in for the model's communicator during the capture: ends the running
            graph at each collective, performs it (keeps the ranks in step and sizes the
            static result buffers), records how to repeat it, and starts the next graph."""
            rank, world = inner.rank, inner.world

            @staticmethod
            def all_reduce(t: torch.Tensor) -> torch.Tensor:
                end()

                def again(t=t) -> None:
                    r = inner.all_reduce(t)
                    if r is not t:
                        t.copy_(r)

                again()
                pieces.append((None, again))
                begin()
                return t

            @staticmethod
            def all_gather_cols(t: torch.Tensor, total_cols: Optional[int] = None) -> torch.Tensor:
                end()
                out = inner.all_gather_cols(t, total_cols).contiguous()

                def again(t=t, out=out) -> None:
                    out.copy_(inner.all_gather_cols(t, total_cols))

                pieces.append((None, again))
                begin()
                return out

        s.wait_stream(torch.cuda.current_stream())
        model.comm = _Cut
        try:
            with torch.cuda.stream(s):
                begin()
                self.outs[with_dec] = self.run_body(eng, with_dec)
                end()
        except BaseException as e:
            if cur[0] is not None:          # never leave a stream capture open behind an error
                try:
                    with torch.cuda.stream(s):
                        cur[0].__exit__(type(e), e, None)
                except Exception:  # noqa: BLE001
                    pass
            raise
        finally:
            model.comm = inner
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.pieces[with_dec] = pieces

    def capture_launch_lists(self, eng: "Engine", with_dec: bool = True) -> None:
        """Same cut as capture_piecewise, but the stretches between collectives are kept as
        recorded launch lists (kernel entry point + argument tuple) instead of hipGraphs: a
        small hipGraph costs tens of microseconds to launch, more than the handful of kernels
        it holds costs to launch directly. What the replay saves is everything the ops layer
        does around a launch: output allocation, shape checks, policy decisions. The buffers
        the recorded launches point at are kept alive here. After recording, the step is
        replayed once and compared with the recorded run; a step that contains work the
        recorder cannot see (a kernel not launched through ops.hip) fails that check and the
        engine stays eager."""
        from ..ops import hip as _hip

        model = eng.model
        inner = model.comm
        self.run_body(eng, with_dec)                # warm-up
        pieces: List[tuple] = []
        keep: List[torch.Tensor] = []
        cur = [_hip.start_recording()]

        def cut(coll) -> None:
            pieces.append((cur[0], None))
            pieces.append((None, coll))
            cur[0] = _hip.start_recording()

        class _Cut:
            rank, world = inner.rank, inner.world

            @staticmethod
            def all_reduce(t: torch.Tensor) -> torch.Tensor:
                def again(t=t) -> None:
                    r = inner.all_reduce(t)
                    if r is not t:
                        t.copy_(r)

                again()
                cut(again)
                return t

            @staticmethod
            def all_gather_cols(t: torch.Tensor, total_cols: Optional[int] = None) -> torch.Tensor:
                out = inner.all_gather_cols(t, total_cols).contiguous()

                def again(t=t, out=out) -> None:
                    out.copy_(inner.all_gather_cols(t, total_cols))

                cut(again)
                return out

        # Tensors created while recording must outlive it: the recorded pointers refer to them.
        model.comm = _Cut
        alloc_hook = _KeepAlive(keep)
        try:
            with alloc_hook:
                self.outs[with_dec] = self.run_body(eng, with_dec)
            pieces.append((cur[0], None))
        finally:
            _hip.stop_recording()
            model.comm = inner
        self._keep = getattr(self, "_keep", []) + keep
        self.pieces[with_dec] = pieces
        self.lists = True
        out_tokens, out_lps, _ = self.outs[with_dec]
        want = (out_tokens.clone(), out_lps.clone())
        self.replay(with_dec)
        if not (torch.equal(want[0], out_tokens) and torch.allclose(want[1], out_lps, atol=1e-4)):
            del self.pieces[with_dec]
            self.lists = bool(self.pieces)
            raise RuntimeError("recorded decode step does not reproduce the eager step")

    def replay(self, with_dec: bool = True) -> None:
        if with_dec in self.graph:
            self._graphs.launch(self.graph[with_dec])
            return
        if self.lists:
            from ..ops.hip import replay as replay_list

            for cmds, coll in self.pieces[with_dec]:
                if cmds is not None:
                    replay_list(cmds)
                else:
                    coll()
            return
        for key, coll in self.pieces[with_dec]:
            if key is not None:
                self._graphs.launch(key)
            else:
                coll()


class _Landing:
    """Where the results of one decode step arrive on the host, and the event that says they have."""

    def __init__(self, eng: "Engine", B: int, C: int) -> None:
        if eng.device.type == "cuda":
            self.tok = eng.rt.pinned.array((B,), np.int32)
            self.lp = eng.rt.pinned.array((B,), np.float32)
            self.dec = eng.rt.pinned.array((B, C), np.float32)
            self.done = rt.Event()
        else:
            self.tok = self.lp = self.dec = None
            self.done = None


class _StepStaging:
    """Host arrays of one prefill step, packed into ONE pinned buffer and shipped with ONE
    asynchronous copy (the decode path has its own fixed-layout arena, _DecodeGraph). ``put``
    returns a ticket; after ``ship`` the ticket's device view is ``view(ticket)``. The device
    buffer is rewritten by the next step's copy on the same stream, i.e. in order. The PINNED
    HOST buffer is rewritten by the host: ``begin`` first waits for the previous copy to have
    read it. A step that samples nothing (prompt chunks that end no prompt, embedding requests)
    is never waited for by its engine, so without this the next step's packing raced with the
    copy still in flight and the kernels read block tables and slot maps of two steps mixed."""

    _TORCH = {np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64, np.dtype(np.float32): torch.float32}

    def __init__(self, device: torch.device) -> None:
        self.device = device
        self.cap = 0
        self.host = self.dev = None          # numpy view of pinned memory / uint8 device tensor
        self.items: List[tuple] = []
        self.used = 0
        self.copied: Optional[rt.Event] = None
        self.in_flight = False

    def begin(self) -> None:
        if self.in_flight:
            self.copied.synchronize()
            self.in_flight = False
        self.items, self.used = [], 0

    def put(self, a: np.ndarray) -> int:
        a = np.ascontiguousarray(a)
        if a.dtype not in self._TORCH:
            raise TypeError(f"unsupported staging dtype {a.dtype}")
        off = (self.used + 15) & ~15
        self.items.append((off, a))
        self.used = off + a.nbytes
        return len(self.items) - 1

    def ship(self) -> None:
        need = max(self.used, 16)
        cuda = self.device.type == "cuda"
        if need > self.cap:
            # grown by replacing both buffers (arenas of their own: the old ones go back whole once
            # the tensors that view them are gone)
            self.cap = max(need * 2, 1 << 16)
            if cuda:
                with torch.cuda.device(rt.indexed(self.device)):
                    self.host = rt.Arena(self.cap + 256, pinned=True).array((self.cap,), np.uint8)
                    self.dev = rt.Arena(self.cap + 256, device=self.device).tensor((self.cap,), torch.uint8)
                if self.copied is None:
                    self.copied = rt.Event()
            else:
                self.host = np.empty(self.cap, dtype=np.uint8)
                self.dev = torch.empty(self.cap, dtype=torch.uint8)
        hv = self.host
        for off, a in self.items:
            hv[off:off + a.nbytes] = a.reshape(-1).view(np.uint8)
        if cuda:
            rt.copy_async(self.dev.data_ptr(), hv.ctypes.data, need, rt.H2D)
            self.copied.record()
            self.in_flight = True
        else:
            self.dev[:need].copy_(torch.from_numpy(hv[:need]))

    def view(self, ticket: int) -> torch.Tensor:
        off, a = self.items[ticket]
        return self.dev[off:off + a.nbytes].view(self._TORCH[a.dtype]).view(a.shape)


class _StepLanding:
    """Results of one prefill step on their way to the host: every tensor is copied into ONE
    pinned buffer with an asynchronous copy queued right behind the kernel that produced it, and
    the host waits once (``wait``) before it reads any of them. Two of these alternate per
    engine, so the buffer a finished step is still being read from is never the one the next
    step's copies land in."""

    _NP = {torch.int32: np.int32, torch.float32: np.float32}

    def __init__(self, device: torch.device) -> None:
        self.device = device
        self.cap = 0
        self.host: Optional[np.ndarray] = None
        self.used = 0
        self.items: List[tuple] = []
        self.done: Optional[rt.Event] = None

    def begin(self) -> None:
        self.used, self.items = 0, []

    def take(self, t: torch.Tensor) -> int:
        """Queues the copy of ``t`` (int32 / float32, contiguous) to the host; returns a ticket."""
        t = t.contiguous()
        nbytes = t.numel() * t.element_size()
        if self.device.type != "cuda":
            self.items.append((t.numpy().copy(), None, None))
            return len(self.items) - 1
        off = (self.used + 15) & ~15
        if off + nbytes > self.cap:
            if self.items:
                raise RuntimeError("landing buffer too small (reserve() it before the first take)")
            self.reserve(off + nbytes)
        rt.copy_async(self.host.ctypes.data + off, t.data_ptr(), nbytes, rt.D2H)
        self.used = off + nbytes
        self.items.append((off, tuple(t.shape), self._NP[t.dtype]))
        return len(self.items) - 1

    def reserve(self, nbytes: int) -> None:
        if nbytes > self.cap and self.device.type == "cuda":
            self.cap = max(2 * nbytes, 1 << 16)
            with torch.cuda.device(rt.indexed(self.device)):
                self.host = rt.Arena(self.cap + 256, pinned=True).array((self.cap,), np.uint8)
            if self.done is None:
                self.done = rt.Event()

    def mark(self) -> None:
        if self.done is not None and self.items:
            self.done.record()

    def wait(self) -> None:
        if self.done is not None and self.items:
            self.done.synchronize()

    def get(self, ticket: int) -> np.ndarray:
        a, shape, dt = self.items[ticket]
        if shape is None:
            return a
        n = int(np.prod(shape)) * np.dtype(dt).itemsize
        return self.host[a:a + n].view(dt).reshape(shape)


class Engine:
    def __init__(self, model: TransformerModel, num_kv_blocks: Optional[int] = None, block_size: int = 64,
                 max_batch_seqs: int = 256, max_prefill_tokens: int = 8192, max_model_len: int = 4096,
                 kv_fraction: float = 0.6, enable_prefix_cache: bool = True, use_graphs: Optional[bool] = None,
                 graph_buckets: Sequence[int] = (1, 2, 4, 8, 16, 32, 64, 128, 256, 512), max_decision: int = 8,
                 max_group_rows: int = 16, name: str = "engine", mix_decode: bool = True,
                 optimistic_admission: bool = False, pipeline_decode: Optional[bool] = None) -> None:
        self.model, self.device, self.name = model, model.device, name
        cfg = model.cfg
        self.block_size = block_size
        self.max_model_len = min(max_model_len, cfg.max_position)
        # On a GPU the engine owns a device runtime (runtime.DeviceRuntime): its step stream, the
        # KV arena, the arenas of small device / pinned blocks and the cache of decode graphs.
        own_stream = self.device.type == "cuda" and not serialize_streams()
        self.rt = rt.DeviceRuntime(self.device, own_stream) if self.device.type == "cuda" else None
        if num_kv_blocks is None:
            per_block = KVCache.bytes_per_block(cfg, block_size, model.hkv)
            if self.rt is not None:
                with torch.cuda.device(self.rt.device):
                    free, _ = rt.mem_info()
                num_kv_blocks = max(16, int(free * kv_fraction) // per_block)
            else:
                num_kv_blocks = 256
        kv_alloc = None
        if self.rt is not None:
            def kv_alloc(numel: int) -> torch.Tensor:
                return self.rt.reserve_kv(2 * numel + 256).tensor((numel,), torch.bfloat16)
        self.kv = KVCache(cfg, num_kv_blocks, block_size, self.device, model.hkv, alloc=kv_alloc)
        self.kvm = core.KVManager(num_kv_blocks, block_size, enable_prefix_cache)
        sc = core.SchedulerConfig()
        sc.max_batch_seqs, sc.max_prefill_tokens = max_batch_seqs, max_prefill_tokens
        # Mixed steps: chains that are already decoding ride along in the steps that prefill newly
        # arrived prompts (one pass over the weights serves both), so a steady flow of arrivals
        # never starves the chains that are already generating.
        sc.mix_decode = mix_decode
        sc.optimistic_admission = optimistic_admission
        self.max_blocks_cap = (self.max_model_len + block_size - 1) // block_size + 1
        sc.max_blocks_per_seq = self.max_blocks_cap        # the width of a block-table row in the step buffers
        self.sched = core.Scheduler(self.kvm, sc)
        self.max_batch_seqs = max_batch_seqs
        # Decode graphs are captured per (batch bucket, context class): the number of KV splits
        # and prefix pieces is a launch-time constant, and the right one depends on how long the
        # contexts of the step are. Classes are powers of two from 256 keys up.
        self.ctx_classes = [c for c in (256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072)
                            if c < 2 * self.max_model_len] or [256]
        self.max_decision = max_decision
        self.max_group_rows = max_group_rows
        # Decode steps are captured into hipGraphs by default. A step that contains collectives
        # (tensor / expert parallel members) is captured like any other when the group's
        # communicator is made of plain kernel launches (``capturable``: parallel/p2p.py).
        # torch.distributed collectives are captured only on request (use_graphs=True): they
        # record and replay correctly under the thread-local capture mode used here, and end the
        # process under the global mode (the process group's watchdog thread makes a HIP call
        # while the capture is open; see TorchDistComm in parallel/comm.py). Any other
        # communicator has no capture story and asking for graphs with it is an error.
        sharded = model.tp_size > 1 or model.ep_size > 1
        capturable = not sharded or bool(getattr(model.comm, "capturable", False))
        if os.environ.get("CONSENSUS_NO_GRAPHS", "0") == "1":
            use_graphs = False          # debugging: every launch goes through the ops layer, eagerly
        use_graphs_arg = use_graphs
        if use_graphs is None:
            # RCCL collectives are recorded too (thread-local capture mode), unless
            # LC_RCCL_CAPTURE=0 asks for eager steps: this is where a group lands whose
            # peer-to-peer collectives failed their first-contact check (parallel/p2p.py).
            use_graphs = capturable or (bool(getattr(model.comm, "capturable_thread_local", False))
                                        and os.environ.get("LC_RCCL_CAPTURE", "1") == "1"
                                        and os.environ.get("CONSENSUS_PIECEWISE", "0") not in ("lists", "graphs"))
        elif use_graphs and not capturable and self.device.type == "cuda" and \
                not getattr(model.comm, "capturable_thread_local", False):
            raise ValueError(f"{name}: use_graphs=True needs a group communicator whose collectives can be recorded "
                             f"into a hipGraph; {type(model.comm).__name__} declares none")
        self.use_graphs = bool(use_graphs) and self.device.type == "cuda"
        # Tensor-parallel members: graphs cut at the collectives (see capture_piecewise). Expert
        # parallel steps have data-dependent exchange sizes and stay eager.
        # CONSENSUS_PIECEWISE = 0 (default: eager) | lists (recorded launch lists) | graphs (hipGraph
        # pieces). One rank of Llama-3-70B at TP=4 (batch 64, 512-token context) takes 12.9 ms per
        # decode step eagerly, 13.0 ms from launch lists and 13.9 ms from graph pieces on one
        # MI355X: the step is bound by its ~1000 kernels on the GPU, not by the host issuing them,
        # so the cut modes only pay off where the host is the slower side.
        mode = os.environ.get("CONSENSUS_PIECEWISE", "0")
        ok = (not self.use_graphs and self.device.type == "cuda" and model.tp_size > 1 and model.ep_size == 1
              and use_graphs_arg is not False and mode in ("lists", "graphs"))
        self.piecewise = mode if ok else ""
        self.graph_buckets = sorted(b for b in graph_buckets if b <= max_batch_seqs) or [max_batch_seqs]
        # Every engine issues its GPU work on a stream of its own, so the engines of one
        # ensemble can have steps in flight at the same time (see run_engines).
        # CONSENSUS_SERIALIZE_STREAMS=1 is the debug mode for stream-order bugs: everything runs
        # on the default stream, one engine at a time, and results must not change.
        self.stream = self.rt.stream.torch if (self.rt is not None and self.rt.stream is not None) else None
        self._side: Optional[rt.Stream] = None          # capture stream of the serialised debug mode
        # Two steps can be in flight (see ``pipeline_decode``): their events alternate.
        self._step_events = [rt.Event(), rt.Event()] if self.rt is not None else None
        self._step_no = 0
        # Decode steps run ahead of their predecessor's results: while step N is on the GPU the
        # host plans step N + 1 (Scheduler.schedule_ahead), ships its inputs and launches it; the
        # tokens step N samples reach step N + 1 on the device (``last_tokens``), and the host
        # reads them one step late. What the GPU would otherwise wait for between two steps
        # (event wake-up, bookkeeping, scheduling, packing, the graph launch: ~0.3 ms against a
        # 5-8 ms step) is hidden. A chain that a stop token ends in step N still has a row in
        # step N + 1; that row's result is dropped.
        if pipeline_decode is None:
            pipeline_decode = os.environ.get("CONSENSUS_PIPELINE_DECODE", "1") == "1"
        self.pipeline_decode = bool(pipeline_decode)
        self._ahead = None          # (plan, token) of a step launched by end_step for the next begin_step
        rows_cap = max_batch_seqs              # no decode step (bucket padding included) has more rows
        if self.rt is not None:
            self.last_tokens = ops.zero_(self.rt.small.tensor((rows_cap,), torch.int32))
        else:
            self.last_tokens = torch.zeros(rows_cap, dtype=torch.int32, device=self.device)
        self._dec_cache = None      # (request ids, rows with decision targets) of the newest decode batch
        self._staging = _StepStaging(self.device)
        self._landing = [_StepLanding(self.device), _StepLanding(self.device)]
        self._cow_staging = _StepStaging(self.device)       # (source, destination) block pairs of copy-on-write
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))   # weights were written there
        self._graphs: Dict[tuple, _DecodeGraph] = {}
        self._reqs: Dict[int, _ReqState] = {}
        self._cancel: set = set()       # requests to drop at the next step boundary (cancel())
        self._done: Dict[int, RequestOutput] = {}
        self._last_dec = None   # (request ids, chain seeds, decision log-probs) of the newest decode step
        self.metrics = core.Metrics()

    # ------------------------------------------------------------------ submission
    def submit(self, prompt_ids: Sequence[int], kind: "core.RequestKind" = core.RequestKind.Generate, n: int = 1,
               params: Optional["core.SamplingParams"] = None, decision_ids: Optional[Sequence[int]] = None,
               score_ids: Optional[Sequence[int]] = None) -> int:
        p = params or core.SamplingParams()
        ids = np.asarray(prompt_ids, dtype=np.int32)
        if ids.size == 0:
            raise ValueError("empty prompt")
        vocab = self.model.cfg.vocab_size
        for what, arr in (("prompt", ids), ("decision target", decision_ids), ("score target", score_ids)):
            arr = None if arr is None else np.asarray(arr)
            if arr is not None and arr.size and (int(arr.min()) < 0 or int(arr.max()) >= vocab):
                # (the kernels clamp such ids rather than read outside the table: the answer would be
                # wrong without a trace. A tokenizer that does not belong to the model is the usual cause.)
                raise ValueError(f"{what} token id outside the model's vocabulary of {vocab}: "
                                 f"{int(arr.min())} .. {int(arr.max())}")
        if n < 1:
            raise ValueError(f"n = {n}: a request has at least one chain")
        new = p.max_new_tokens if kind == core.RequestKind.Generate else 0
        if kind == core.RequestKind.Generate and not 1 <= new < self.max_model_len:
            raise ValueError(f"max_new_tokens = {new} must lie in [1, {self.max_model_len - 1}]: the longest sequence "
                             f"served is {self.max_model_len} tokens and a prompt has at least one")
        budget = self.max_model_len - new
        if ids.size > budget:
            ids = ids[-budget:]  # keep the tail (the instructions and the answer slot are at the end)
        if decision_ids is not None and len(decision_ids) > self.max_decision:
            raise ValueError(f"at most {self.max_decision} decision targets")
        rid = self.sched.add_request(kind, ids, n, p)
        self._reqs[rid] = _ReqState(kind, n if kind == core.RequestKind.Generate else 1,
                                    list(decision_ids) if decision_ids is not None else None,
                                    list(score_ids) if score_ids is not None else None, seed=p.seed,
                                    submitted_at=time.perf_counter())
        return rid

    def has_work(self) -> bool:
        return self.sched.has_work()

    def cancel(self, rid: int) -> bool:
        """Marks one request for removal: at the next step boundary (no step of this engine on
        the GPU) it leaves the scheduler whatever its state (waiting, prefilling, decoding), its
        KV blocks return to the pool and its result comes out with finish reason ``Aborted``.
        Returns False for a request the engine does not hold (any more). Call from the thread
        that steps the engine."""
        if rid not in self._reqs:
            return False
        self._cancel.add(rid)
        return True

    def _apply_cancels(self) -> None:
        rids, self._cancel = [r for r in self._cancel if r in self._reqs], set()
        if not rids:
            return
        for rid in rids:
            self.sched.abort_request(rid)
        self._collect()
        # cached views of the decode batch (row layout, decision rows) name rows that are gone
        self._dec_cache = None
        for g in self._graphs.values():
            g.rows_key = None
        self.metrics.inc("requests.cancelled", len(rids))

    def abort_all(self) -> int:
        """Drops every request the engine holds (waiting, prefilling or decoding) and returns
        their KV blocks: the recovery path after a failed step, whose requests cannot be
        resumed because the step's effects on the cache are unknown. Returns how many."""
        rids = list(self._reqs)
        for rid in rids:
            self.sched.abort_request(rid)
        self.sched.take_finished()
        # Prompt blocks enter the prefix cache when their step is scheduled, before the step has
        # written them; after a failed step such an entry may name KV that was never produced,
        # and the next prompt with the same prefix would attend over it. Forget them all.
        self.kvm.take_pending_copies()
        self.kvm.drop_cache()
        self._reqs.clear()
        self._cancel.clear()
        self._last_dec = None
        self._ahead = None
        self._dec_cache = None
        for g in self._graphs.values():
            g.rows_key = None
        self.metrics.inc("requests.aborted", len(rids))
        return len(rids)

    def warm_up(self, buckets: Optional[Sequence[int]] = None, ctx_classes: Optional[Sequence[int]] = None,
                with_prefix: Sequence[bool] = (False,), prompt_tokens: int = 320) -> Dict[str, float]:
        """Pays the first-use costs before the first real request does: one short request through
        the whole path (kernel code objects, launch attributes, allocator pools, the group's
        collectives), then every configured decode step recorded into its hipGraph, in both
        forms, for ``buckets`` x ``ctx_classes`` x ``with_prefix`` (defaults: every batch bucket,
        the context classes up to the engine's longest sequence, no shared-prefix pass). The
        recording runs over rows that are all padding (no KV slot, context length 0), so it
        touches no request state. A later request whose shape was warmed up finds its graph
        recorded (``decode.graph_captures`` does not move). In a tensor / expert parallel group
        every rank must call this with the same arguments. The reference answers its first
        question as fast as any other (/root/reference/src/main.rs:428-445)."""
        t0 = time.perf_counter()
        if self.rt is None:
            return {"graphs": 0, "seconds": 0.0}      # the CPU plumbing backend has no first-use costs worth paying early
        if self.has_work():
            raise RuntimeError("warm_up() runs on an idle engine")
        n_prompt = max(1, min(prompt_tokens, self.max_model_len - 4))
        p = core.SamplingParams()
        p.temperature, p.top_p, p.max_new_tokens, p.ignore_eos = 0.8, 0.95, 3, True
        vocab = self.model.cfg.vocab_size
        rid = self.submit([(7 + 13 * i) % vocab for i in range(n_prompt)], params=p, decision_ids=[0, 1])
        was, self.pipeline_decode = self.pipeline_decode, False
        try:
            self.run_until_done()
        finally:
            self.pipeline_decode = was
        self.pop_result(rid)
        self.kvm.drop_cache()            # the warm-up prompt must not answer prefix look-ups
        captured = 0
        if self.use_graphs or self.piecewise:
            with self._on_stream():
                for bucket in (buckets if buckets is not None else self.graph_buckets):
                    if bucket > self.max_batch_seqs or not self.model.decode_capturable(bucket):
                        continue
                    for ctx in (ctx_classes if ctx_classes is not None else self.ctx_classes):
                        if ctx not in self.ctx_classes:
                            continue
                        for wp in with_prefix:
                            key = (bucket, bool(wp), ctx)
                            g = self._graphs.get(key)
                            if g is None:
                                g = self._graphs[key] = _DecodeGraph(self, bucket, bool(wp), ctx)
                            for with_dec in (False, True):
                                if g.ready(self, with_dec):
                                    continue
                                g.begin_fill()
                                g.upload()          # the all-padding mirror of a fresh graph
                                if self.use_graphs:
                                    g.capture(self, with_dec)
                                elif self.piecewise == "graphs":
                                    g.capture_piecewise(self, with_dec)
                                else:
                                    g.capture_launch_lists(self, with_dec)
                                captured += 1
                                g.rows_key = None
            if self.rt is not None:
                torch.cuda.synchronize(self.device)
        took = time.perf_counter() - t0
        self.metrics.inc("warmup.graphs", captured)
        self.metrics.inc("warmup.seconds", took)
        return {"graphs": captured, "seconds": took}

    def pop_result(self, rid: int) -> Optional[RequestOutput]:
        return self._done.pop(rid, None)

    def run_until_done(self) -> None:
        while self.step():
            pass

    # ------------------------------------------------------------------ stepping
    def _copy_blocks(self, src: np.ndarray, dst: np.ndarray) -> None:
        """Copy-on-write block copies, queued on the step's stream ahead of its kernels; the block
        numbers reach the device through a pinned staging buffer (no blocking copy)."""
        cs = self._cow_staging
        cs.begin()
        a, b = cs.put(np.asarray(src, dtype=np.int32)), cs.put(np.asarray(dst, dtype=np.int32))
        cs.ship()
        self.kv.copy_blocks(cs.view(a), cs.view(b))

    def _on_stream(self):
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def _capture_stream(self):
        """The stream a decode graph is recorded on: the engine's own, or (serialised debug mode,
        where steps run on the legacy default stream, which cannot be captured) a side stream."""
        if self.stream is not None or self.rt is None:
            return contextlib.nullcontext()
        if self._side is None:
            self._side = rt.Stream(self.rt.device)
        return torch.cuda.stream(self._side.torch)

    def begin_step(self):
        """Schedules one step and enqueues its GPU work on the engine's stream without waiting
        for it. Returns an opaque token for ``end_step`` (None when idle)."""
        plan = self.plan_step()
        return None if plan is None else self.launch_step(plan)

    def plan_step(self):
        """Asks the scheduler for the next step without launching it (None when idle). The plan
        may be held back by a driver and launched later with ``launch_step``. When the previous
        ``end_step`` already launched the next decode step, that step's plan is returned."""
        if self._ahead is not None:
            return self._ahead[0]
        if self._cancel:
            self._apply_cancels()
        t0 = time.perf_counter()
        plan = self.sched.schedule()
        self.metrics.inc("host.schedule_seconds", time.perf_counter() - t0)
        return None if plan.kind == PK.Idle else plan

    def launch_step(self, plan):
        if self._ahead is not None and plan is self._ahead[0]:
            token, self._ahead = self._ahead[1], None
            return token
        t0 = time.perf_counter()
        with self._on_stream():
            if len(plan.copy_src):
                self._copy_blocks(plan.copy_src, plan.copy_dst)
                self.metrics.inc("kv.cow_copies", len(plan.copy_src))
            if plan.kind == PK.Prefill:
                with span(f"{self.name}.prefill", "engine", tokens=len(plan.token_ids)):
                    state = self._prefill_launch(plan)
            else:
                with span(f"{self.name}.decode", "engine", seqs=len(plan.seq_ids)):
                    state = self._decode_launch(plan)
            done = None
            if self.stream is not None:
                done = self._step_events[self._step_no & 1].record()
            self._step_no += 1
        return plan, state, t0, done

    def _launch_ahead(self, plan) -> None:
        """While the decode step ``plan`` is on the GPU: plans and launches the step after it."""
        if self._cancel:
            return            # a cancelled request leaves at a step boundary: let this step be the last in flight
        t0 = time.perf_counter()
        nxt = self.sched.schedule_ahead(plan)
        self.metrics.inc("host.schedule_seconds", time.perf_counter() - t0)
        if nxt.kind != PK.Decode:
            return
        if not self._decode_ready(nxt):
            # The step's graph would have to be recorded first, and recording runs the step once
            # for real (its sampler would overwrite the tokens the recorded step is to read):
            # this step takes the ordinary path. Nothing is lost but the overlap: the blocks
            # reserved for it are the ones the ordinary plan reserves again.
            if len(nxt.copy_src):
                with self._on_stream():
                    self._copy_blocks(nxt.copy_src, nxt.copy_dst)
            return
        self._ahead = (nxt, self.launch_step(nxt))
        self.metrics.inc("decode.ahead_steps")

    @staticmethod
    def step_done(token) -> bool:
        """True once the GPU work of a launched step has finished (``end_step`` will not block)."""
        return token[3] is None or token[3].query()

    def end_step(self, token) -> None:
        """Waits for the step started by ``begin_step`` and does its bookkeeping."""
        plan, state, t0, _ = token
        if self.pipeline_decode and plan.kind == PK.Decode and self._ahead is None:
            self._launch_ahead(plan)
        with self._on_stream():
            if plan.kind == PK.Prefill:
                self._prefill_finish(plan, state)
                self.metrics.inc("prefill.steps")
                self.metrics.inc("prefill.tokens", len(plan.token_ids))
                self.metrics.inc("prefill.seconds", time.perf_counter() - t0)
            else:
                self._decode_finish(plan, state)
                self.metrics.inc("decode.steps")
                self.metrics.inc("decode.tokens", len(plan.seq_ids))
                self.metrics.inc("decode.seconds", time.perf_counter() - t0)
            # A group collective that gave up waiting for a peer leaves garbage behind; the step's
            # results must not be used (the error word is read once per step, after its sync).
            check = getattr(self.model.comm, "check", None)
            if check is not None:
                check()
        self._collect()

    def step(self) -> bool:
        token = self.begin_step()
        if token is None:
            return False
        self.end_step(token)
        return True

    def _prefill_launch(self, plan):
        m = self.model
        rf = plan.rider_first
        n_seq = len(plan.seq_ids)
        n_riders = n_seq - rf if rf >= 0 else 0
        cu = plan.cu_q
        req_ids = plan.request_ids
        n_last = len(plan.last_rows)
        # ---- everything the step reads from the host goes into one staging buffer ----
        sg = self._staging
        sg.begin()
        t_tok, t_pos, t_slot = sg.put(plan.token_ids), sg.put(plan.positions), sg.put(plan.slot_mapping)
        t_bt, t_ctx = sg.put(plan.block_tables), sg.put(plan.context_lens)
        t_cu = sg.put(cu[:rf + 1] if n_riders else cu)
        # Embed requests: one pooling launch over the chunks of all of them in this step (fp32
        # sums of the final hidden states per chunk); a prompt cut into several chunks adds its
        # chunk sums up, the mean and the L2 normalisation happen when the request finishes.
        emb = [i for i, rid in enumerate(req_ids[:n_seq - n_riders])
               if self._reqs[int(rid)].kind == core.RequestKind.Embed]
        if emb:
            t_eb = sg.put(np.array([cu[i] for i in emb], dtype=np.int32))
            t_ee = sg.put(np.array([cu[i + 1] for i in emb], dtype=np.int32))
        t_need = t_samp = None
        slots = []
        if n_last or n_riders:
            # logits rows: the prompts that end in this step, then the riders (the step's last rows)
            need = np.concatenate([plan.last_rows, cu[rf:n_seq]]) if n_riders else plan.last_rows
            t_need = sg.put(np.asarray(need).astype(np.int32))
            slots = self.sched.prefill_sample_layout(plan)
            if slots or n_riders:
                ri = [s.last_row_index for s in slots] + list(range(n_last, n_last + n_riders))
                temp, top_p, top_k = ([getattr(s, f) for s in slots] for f in ("temperature", "top_p", "top_k"))
                seeds = [s.seed for s in slots]
                steps = [0] * len(slots)
                if n_riders:
                    cache: Dict[int, tuple] = {}
                    for rid in req_ids[rf:]:
                        e = cache.get(int(rid))
                        if e is None:
                            p = self.sched.params_of(int(rid))
                            e = cache[int(rid)] = (p.temperature, p.top_p, p.top_k)
                        temp.append(e[0]); top_p.append(e[1]); top_k.append(e[2])
                    seeds += [int(x) for x in plan.seeds]
                    steps += [int(x) for x in plan.step_index]
                t_samp = (sg.put(np.array(temp, dtype=np.float32)), sg.put(np.array(top_p, dtype=np.float32)),
                          sg.put(np.array(top_k, dtype=np.int32)),
                          sg.put(np.array(seeds, dtype=np.uint64).astype(np.int64)),
                          sg.put(np.array(steps, dtype=np.int32)), sg.put(np.array(ri, dtype=np.int32)))
        # Log-probs wanted at the last position of prompts that end here (Score requests, decision
        # targets) and of riders with decision targets: the target ids travel with the step's
        # inputs, the gathers are queued behind the logits, nothing of it waits for the host.
        want = self._score_wants(plan) if t_need is not None else []
        t_want, want_cols = None, 0
        if want:
            ids_w = self._padded_targets(want)
            want_cols = ids_w.shape[1]
            t_want = (sg.put(np.array([j for j, _, _ in want], dtype=np.int32)), sg.put(ids_w))
        rider_dec = None
        if n_riders:
            r_req = np.array(req_ids[rf:], copy=True)
            dec = [self._reqs[int(r)].decision_ids for r in r_req]
            if any(dec):
                ids_h = np.zeros((len(dec), self.max_decision), dtype=np.int32)
                for i, d in enumerate(dec):
                    if d:
                        ids_h[i, : len(d)] = d
                rider_dec = (r_req, sg.put(np.arange(n_last, n_last + len(dec), dtype=np.int32)), sg.put(ids_h))
        # A sharded model whose step only scores (nothing samples, no riders) exchanges the target
        # log-probs instead of all-gathering the logits rows.
        score_only = bool(want) and t_samp is None and m.tp_size > 1
        sg.ship()
        bt, ctx = sg.view(t_bt), sg.view(t_ctx)
        if n_riders:
            # Mixed step: sequences [rf, n_seq) are decoding chains with one token each.
            riders = AttnMeta(is_prefill=False, block_tables=bt[rf:], ctx_lens=ctx[rf:],
                              nsplit=ops.choose_kv_splits(n_riders, m.hkv, int(plan.context_lens[rf:].max()))
                              if self.device.type == "cuda" else 1)
            meta = AttnMeta(is_prefill=True, block_tables=bt[:rf], ctx_lens=ctx[:rf], cu_q=sg.view(t_cu),
                            max_q_len=int(np.diff(cu[:rf + 1]).max()), riders=riders, prefill_tokens=int(cu[rf]))
            self.metrics.inc("prefill.rider_tokens", n_riders)
        else:
            meta = AttnMeta(is_prefill=True, block_tables=bt, ctx_lens=ctx, cu_q=sg.view(t_cu),
                            max_q_len=int(np.diff(cu).max()))
        hidden = m.forward(sg.view(t_tok), sg.view(t_pos), sg.view(t_slot), meta, self.kv)
        if emb:
            sums = ops.embed_pool_ranges(hidden, sg.view(t_eb), sg.view(t_ee), mode=2)
            for j, i in enumerate(emb):
                st = self._reqs[int(req_ids[i])]
                if st.embed_sum is None:
                    st.embed_sum = sums[j]
                else:
                    ops.accumulate_(st.embed_sum, sums[j])
                st.embed_count += int(cu[i + 1] - cu[i])
        land = self._landing[self._step_no & 1]
        land.begin()
        k_tok = k_lp = k_want = k_rdec = None
        if t_need is not None:
            rows = ops.gather_rows(hidden, sg.view(t_need))
            if score_only:
                self.metrics.inc("prefill.target_only_score_rows", len(want))
                vals = m.target_logprobs(ops.gather_rows(rows, sg.view(t_want[0])), sg.view(t_want[1]))
                land.reserve(vals.numel() * 4 + 64)
                k_want = land.take(vals)
            else:
                logits = m.logits(rows)
                n_out = (len(slots) + n_riders) if t_samp is not None else 0
                land.reserve(8 * n_out + 4 * (want_cols * len(want) + self.max_decision * n_riders) + 256)
                if t_samp is not None:
                    toks, lps = ops.sample(logits, *(sg.view(t) for t in t_samp))
                    k_tok, k_lp = land.take(toks), land.take(lps)
                if want:
                    k_want = land.take(ops.logprob_gather(logits, sg.view(t_want[1]), sg.view(t_want[0])))
                if rider_dec is not None:
                    k_rdec = land.take(ops.logprob_gather(logits, sg.view(rider_dec[2]), sg.view(rider_dec[1])))
        land.mark()
        return land, k_tok, k_lp, (want, k_want), (rider_dec[0] if rider_dec is not None else None, k_rdec)

    def _score_wants(self, plan) -> list:
        """(index into the step's last rows, request state, target ids) of every prompt that ends
        in this step and wants log-probs at its last position (Score requests, decision targets)."""
        want = []
        for j, seq_i in enumerate(plan.last_row_seq):
            st = self._reqs[int(plan.request_ids[int(seq_i)])]
            if st.kind == core.RequestKind.Score and st.score_ids:
                want.append((j, st, st.score_ids))
            elif st.kind == core.RequestKind.Generate and st.decision_ids:
                want.append((j, st, st.decision_ids))
        return want

    @staticmethod
    def _padded_targets(want) -> np.ndarray:
        ids_h = np.zeros((len(want), max(len(ids) for _, _, ids in want)), dtype=np.int32)
        for i, (_, _, ids) in enumerate(want):
            ids_h[i, : len(ids)] = ids
        return ids_h

    @staticmethod
    def _store_scores(want, vals: np.ndarray) -> None:
        for i, (_, st, ids) in enumerate(want):
            v = vals[i, : len(ids)].tolist()
            if st.kind == core.RequestKind.Score:
                st.scores = v
            else:
                for k in range(st.n):
                    st.decisions[st.seed + k] = v

    def _prefill_finish(self, plan, state) -> None:
        land, k_tok, k_lp, (want, k_want), (r_req, k_rdec) = state
        land.wait()                       # the one wait for everything this step sends back
        first_tokens = np.zeros(0, dtype=np.int32)
        first_lps = np.zeros(0, dtype=np.float32)
        if k_tok is not None:
            first_tokens, first_lps = np.array(land.get(k_tok), copy=True), np.array(land.get(k_lp), copy=True)
        if k_want is not None:
            self._store_scores(want, land.get(k_want))
        if k_rdec is not None:
            # Riders with decision targets: same bookkeeping as a decode step (the newest row
            # block is kept, rows are resolved when their chain leaves the batch).
            cur = (r_req, np.array(plan.seeds, copy=True), np.array(land.get(k_rdec), copy=True))
            self._flush_decisions(keep=cur)
            self._last_dec = cur
        self.sched.complete_prefill(plan, first_tokens.astype(np.int32), first_lps.astype(np.float32))

    def _bucket(self, B: int) -> int:
        for b in self.graph_buckets:
            if b >= B:
                return b
        return B

    def _decode_key(self, plan):
        """(bucket, shared-prefix pass, context class) of a decode plan: which recorded step runs it."""
        B = len(plan.seq_ids)
        gshared = plan.group_shared_blocks
        gcount = plan.group_count
        # The shared-prefix pass pays when a group is large: it reads the prompt's KV once per
        # group instead of once per chain, but adds a second attention launch and the merge.
        # With 4 chains per prompt (64 rows, 512-token prompts) the plain pass takes 32 us per
        # layer against 43 us for prefix + suffix + merge; with 16 chains it is 56 us against
        # more than 100. The switch is the size of the step's largest group.
        with_prefix = bool(len(gshared)) and bool(((gshared > 0) & (gcount > 1)).any()) and \
            PREFIX_MIN_GROUP_ROWS <= int(gcount.max()) <= self.max_group_rows
        bucket = self._bucket(B) if (self.use_graphs or self.piecewise) else B
        max_ctx = int(plan.context_lens.max())
        ctx_class = next((c for c in self.ctx_classes if c >= max_ctx), self.ctx_classes[-1])
        return (bucket, with_prefix, ctx_class), gshared, gcount

    def _decision_rows(self, req_ids: np.ndarray):
        """Rows of a decode batch whose request has decision targets, with what the batch's rows
        need to know about ending: (row indices, max_new_tokens per row, whether any such row can
        be ended by a stop token). Cached while the batch keeps its rows."""
        c = self._dec_cache
        if c is not None and c[0].shape == req_ids.shape and np.array_equal(c[0], req_ids):
            return c[1], c[2], c[3]
        rows, max_new, can_stop = [], np.zeros(len(req_ids), dtype=np.int32), False
        seen: Dict[int, tuple] = {}
        for b, rid in enumerate(req_ids.tolist()):
            e = seen.get(rid)
            if e is None:
                p = self.sched.params_of(rid)
                e = seen[rid] = (bool(self._reqs[rid].decision_ids), p.max_new_tokens,
                                 (not p.ignore_eos and p.eos_id >= 0) or bool(p.stop_ids))
            max_new[b] = e[1]
            if e[0]:
                rows.append(b)
                can_stop = can_stop or e[2]
        rows = np.asarray(rows, dtype=np.int64)
        self._dec_cache = (np.array(req_ids, copy=True), rows, max_new, can_stop)
        return rows, max_new, can_stop

    def _needs_decisions(self, plan) -> bool:
        """A chain's decision log-probs are those of its LAST step, so the gather runs only in a
        step that can be some such chain's last: one that a stop token may end, or the one in
        which it reaches max_new_tokens."""
        rows, max_new, can_stop = self._decision_rows(plan.request_ids)
        if not len(rows):
            return False
        if can_stop:
            return True
        return bool((plan.step_index[rows] + 1 >= max_new[rows]).any())

    def _decode_ready(self, plan) -> bool:
        """Whether a decode plan can be launched without recording a graph first."""
        if not (self.use_graphs or self.piecewise):
            return True
        key, _, _ = self._decode_key(plan)
        g = self._graphs.get(key)
        return g is not None and g.ready(self, self._needs_decisions(plan))

    def _decode_launch(self, plan):
        t_fill = time.perf_counter()
        B = len(plan.seq_ids)
        key, gshared, gcount = self._decode_key(plan)
        bucket, with_prefix, ctx_class = key
        g = self._graphs.get(key)
        if g is None:
            g = _DecodeGraph(self, bucket, with_prefix, ctx_class)
            if not self.use_graphs:
                g.nsplit = 1 if self.device.type != "cuda" else g.nsplit
            self._graphs[key] = g
        with_dec = self._needs_decisions(plan)
        # ---- fill the pinned host mirror, then one H2D copy ----
        g.begin_fill()
        h = g.h

        def put(name: str, a: np.ndarray, fill=0):
            dst = h[name]
            n = a.shape[0]
            dst[:n] = a
            if n < dst.shape[0]:
                dst[n:] = fill

        put("token_ids", plan.token_ids)
        put("positions", plan.positions)
        put("slot_mapping", plan.slot_mapping, -1)
        put("ctx_lens", plan.context_lens)
        if plan.ahead:
            # the input tokens are still being sampled: row b reads last_tokens[src_rows[b]]
            put("src_rows", plan.src_rows, -1)
            g.feed_set = True
        elif g.feed_set:
            h["src_rows"][:] = -1
            g.feed_set = False
        bt = plan.block_tables
        h["block_tables"][:B, : bt.shape[1]] = np.maximum(bt, 0)
        req_ids = plan.request_ids
        # Per-request sampling parameters only change when the batch composition does; a decode
        # batch usually keeps its rows for many steps, so the host mirror is refilled only then.
        if g.rows_key is None or g.rows_key.shape[0] != B or not np.array_equal(g.rows_key, req_ids):
            cache: Dict[int, tuple] = {}
            for b in range(B):
                rid = int(req_ids[b])
                e = cache.get(rid)
                if e is None:
                    p = self.sched.params_of(rid)
                    d = self._reqs[rid].decision_ids
                    e = (p.temperature, p.top_p, p.top_k, d)
                    cache[rid] = e
                h["temperature"][b], h["top_p"][b], h["top_k"][b] = e[0], e[1], e[2]
                if e[3]:
                    h["decision_ids"][b, : len(e[3])] = e[3]
            if B < bucket:
                h["temperature"][B:] = 0.0
                h["top_p"][B:] = 1.0
                h["top_k"][B:] = 0
            g.rows_key = np.array(req_ids, copy=True)
        put("seeds", plan.seeds.astype(np.int64))
        put("steps", plan.step_index)
        if with_prefix:
            # Only groups of several rows with common blocks run the prefix pass (the per-row pass
            # skips the prefix for exactly those rows: kv_begin > 0 below). There are at most B / 2
            # of them, which is what the graph's group slots are sized for; single-row groups are
            # left out.
            sel = np.nonzero((gshared > 0) & (gcount > 1))[0]
            if len(sel) > g.gmax:
                raise RuntimeError("more sequence groups than the decode graph was sized for")
            gfirst = np.asarray(plan.group_first)
            put("group_first", gfirst[sel])
            put("group_count", gcount[sel].astype(np.int32))
            put("group_shared", gshared[sel])
            kvb = np.zeros(B, np.int32)
            for gi in sel:
                f, c = int(gfirst[gi]), int(gcount[gi])
                kvb[f:f + c] = int(gshared[gi]) * self.block_size
            put("kv_begin", kvb)
        g.upload()
        t_launch = time.perf_counter()
        self.metrics.inc("host.decode_fill_seconds", t_launch - t_fill)
        # ---- run ----
        if (self.use_graphs or self.piecewise) and self.model.decode_capturable(bucket):
            if not g.ready(self, with_dec):
                if plan.ahead:
                    raise RuntimeError("a decode step launched ahead must not record its graph")
                t_cap = time.perf_counter()
                if self.use_graphs:
                    g.capture(self, with_dec)
                elif self.piecewise == "graphs":
                    g.capture_piecewise(self, with_dec)
                else:
                    g.capture_launch_lists(self, with_dec)
                self.metrics.inc("decode.graph_captures")
                self.metrics.inc("decode.graph_capture_seconds", time.perf_counter() - t_cap)
            g.replay(with_dec)
            tok, lp, dl = g.outs[with_dec]
        else:
            tok, lp, dl = g.run_body(self, with_dec)
        # Results come back through pinned buffers: the copies are queued right behind the step,
        # and the host later waits once instead of doing three blocking reads.
        land = g.land[self._step_no & 1]
        if self.device.type == "cuda":
            rt.copy_async(land.tok.ctypes.data, tok.data_ptr(), 4 * bucket, rt.D2H)
            rt.copy_async(land.lp.ctypes.data, lp.data_ptr(), 4 * lp.numel(), rt.D2H)
            if with_dec:
                rt.copy_async(land.dec.ctypes.data, dl.data_ptr(), 4 * dl.numel(), rt.D2H)
            land.done.record()
            self.metrics.inc("host.decode_launch_seconds", time.perf_counter() - t_launch)
            return g, land, with_dec
        # the host copies are taken now: the next step overwrites last_tokens
        land.tok, land.lp = tok[:bucket].numpy().copy(), lp.numpy().copy()
        land.dec = dl.numpy().copy() if with_dec else None
        return g, land, with_dec

    def _decode_finish(self, plan, state) -> None:
        g, land, with_dec = state
        if land.done is not None:
            t_wait = time.perf_counter()
            land.done.synchronize()
            # time the host spent blocked on the GPU; decode.seconds minus this is host work
            self.metrics.inc("decode.wait_seconds", time.perf_counter() - t_wait)
        t_fin = time.perf_counter()
        B = len(plan.seq_ids)
        tok_h = land.tok[:B].astype(np.int32)
        lp_h = land.lp[:B].astype(np.float32)
        taken = self.sched.complete_decode(plan, tok_h, lp_h)
        if with_dec:
            # A chain's decision log-probs are those of its LAST step. Keeping the newest row
            # block and resolving rows only when they leave the batch avoids a Python loop
            # over every row at every step. Rows of a step launched ahead whose chain had
            # already ended (``taken`` 0) carry nothing.
            rids, seeds, vals = plan.request_ids, plan.seeds, land.dec[:B]
            if plan.ahead and not taken.all():
                live = np.nonzero(taken)[0]
                rids, seeds, vals = rids[live], seeds[live], vals[live]
            cur = (np.array(rids, copy=True), np.array(seeds, copy=True), np.array(vals, copy=True))
            self._flush_decisions(keep=cur)
            self._last_dec = cur
        self.metrics.inc("host.decode_finish_seconds", time.perf_counter() - t_fin)

    def _flush_decisions(self, keep=None, only_request: Optional[int] = None) -> None:
        """Moves rows of the remembered decode step into their requests' ``decisions``: rows that
        are absent from ``keep`` (the chain finished), or all rows of ``only_request``."""
        prev = self._last_dec
        if prev is None:
            return
        rids, seeds, vals = prev
        if only_request is not None:
            rows = np.nonzero(rids == only_request)[0]
        else:
            key_prev = (rids.astype(np.int64) << 32) ^ (seeds.astype(np.int64) & 0xFFFFFFFF)
            key_cur = (keep[0].astype(np.int64) << 32) ^ (keep[1].astype(np.int64) & 0xFFFFFFFF)
            if key_prev.shape == key_cur.shape and np.array_equal(key_prev, key_cur):
                return
            rows = np.nonzero(~np.isin(key_prev, key_cur))[0]
        for b in rows:
            st = self._reqs.get(int(rids[b]))
            if st is not None and st.decision_ids:
                st.decisions[int(seeds[b])] = vals[b, : len(st.decision_ids)].tolist()

    def _collect(self) -> None:
        for f in self.sched.take_finished():
            self._flush_decisions(only_request=f.request_id)
            st = self._reqs.pop(f.request_id)
            out = RequestOutput(f.request_id, f.kind, prompt_len=f.prompt_len, cached_tokens=f.cached_tokens,
                                submitted_at=st.submitted_at, finished_at=time.perf_counter())
            for c in f.chains:
                out.token_ids.append(list(c.tokens))
                out.cum_logprobs.append(c.cum_logprob)
                out.finish_reasons.append(c.reason.name)
            if st.decision_ids:
                out.decision_logprobs = [st.decisions.get(st.seed + k, [0.0] * len(st.decision_ids))
                                         for k in range(len(f.chains))]
            if st.kind == core.RequestKind.Score:
                out.scores = st.scores
            if st.kind == core.RequestKind.Embed and st.embed_sum is not None:
                out.embedding = ops.l2_normalize(st.embed_sum.view(1, -1))[0].cpu().numpy()
            self._done[f.request_id] = out
            self.metrics.inc("requests.finished")
            self.metrics.observe("request.latency_ms", (out.finished_at - out.submitted_at) * 1e3)

    # ------------------------------------------------------------------ introspection
    def stats(self) -> Dict[str, float]:
        s = dict(self.metrics.snapshot())
        k = self.kvm.stats()
        s.update({
            "kv.blocks_total": k.blocks_total, "kv.blocks_free": k.blocks_free, "kv.blocks_cached": k.blocks_cached,
            "kv.utilisation": 1.0 - k.blocks_free / max(1, k.blocks_total),
            "kv.prompt_tokens": k.prompt_tokens, "kv.prefix_hit_tokens": k.prefix_hit_tokens,
            "kv.prefix_hit_rate": k.prefix_hit_tokens / max(1, k.prompt_tokens),
            "kv.cow_copies_total": k.cow_copies, "kv.evictions": k.evictions,
            "sched.preemptions": self.sched.num_preemptions, "sched.head_skips": self.sched.num_head_skips,
            "sched.waiting": self.sched.num_waiting, "sched.running": self.sched.num_running,
        })
        if self.rt is not None:
            s.update(self.rt.stats())
        return s


class EngineDriver:
    """Steps several engines with their GPU work overlapped, one pass at a time, so a caller
    can admit new requests between passes. Each engine's next step is enqueued on its own
    stream as soon as its previous one has been collected, so up to len(engines) steps are in
    flight. Decode steps of 7-8B models at small batch are dominated by short kernels whose
    launch ramps and end-of-kernel cache write-backs leave CUs idle; kernels of the other
    members fill those gaps.

    Prefill steps are MFMA-bound, decode steps HBM-bound. With STAGGER_PREFILL only one engine
    prefills at a time, so the others' decode steps run against a prefill instead of against
    each other's prefills: the chip then has compute-bound and memory-bound kernels in flight
    together. An engine whose next step is a prefill simply holds its plan until the slot frees."""

    def __init__(self) -> None:
        self.pending: dict = {}     # engine -> token of its step in flight
        self.held: dict = {}        # engine -> prefill plan waiting for the prefill slot
        self.prefilling = None      # the engine that owns the prefill slot

    @property
    def in_flight(self) -> int:
        return len(self.pending) + len(self.held)

    def advance(self, engines: Sequence["Engine"]) -> bool:
        """One pass: collects the steps that have finished and launches each idle engine's next
        one. Returns False when nothing could be collected or launched (every step in flight
        is still running, or there is no work)."""
        progressed = False
        for e in engines:
            tok = self.pending.get(e)
            if tok is not None:
                # Never block on one engine's long prefill while another's decode step is done.
                if STAGGER_PREFILL and len(self.pending) > 1 and not e.step_done(tok):
                    continue
                del self.pending[e]
                if self.prefilling is e:
                    self.prefilling = None
                e.end_step(tok)
                progressed = True
            plan = self.held.pop(e, None) or e.plan_step()
            if plan is None:
                continue
            if STAGGER_PREFILL and plan.kind == PK.Prefill:
                if self.prefilling is not None and self.prefilling is not e:
                    self.held[e] = plan
                    continue
                self.prefilling = e
            self.pending[e] = e.launch_step(plan)
            progressed = True
        if not progressed and not self.pending:
            self.prefilling = None   # nothing in flight: never leave every engine waiting on a stale slot
        return progressed

    def reset(self) -> None:
        """Forgets steps in flight (after a failure whose requests the caller has dropped)."""
        self.pending.clear()
        self.held.clear()
        self.prefilling = None


def run_engines(engines: Sequence["Engine"]) -> None:
    """Drives several engines to completion with their steps overlapped (EngineDriver)."""
    live = [e for e in engines if e.has_work()]
    if len(live) <= 1 or serialize_streams():
        for e in live:
            e.run_until_done()
        return
    driver = EngineDriver()
    while True:
        if not driver.advance(live):
            if not driver.in_flight:
                live = [e for e in live if e.has_work()]
                if not live:
                    return
            else:
                time.sleep(20e-6)
