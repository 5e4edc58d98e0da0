"""Communication layer: one interface, RCCL in production, in-process loopback in tests.

The reference's "communication" is actor messaging on one thread
(/root/reference/src/main.rs:98, :144, :179, :234, :250-253, :282): point-to-point sends,
a broadcast of (question, answer) to all members, a gather of N verdicts with a fan-in
barrier (main.rs:265) and a logical AND over votes (main.rs:266). Hosting members on
GPUs turns those into collectives, plus the tensor/expert-parallel collectives inside a
model:

    all_reduce        TP partial sums after attention-O and MLP-down (2 per layer)
    all_gather_cols   vocab-parallel logits -> full logits
    all_gather_rows   candidate answers / votes / log-probs / embeddings to every rank
    broadcast         winning answer / next-round prompt
    all_to_all_equal  expert-parallel token dispatch and combine (fixed-capacity blocks)
    all_to_all_rows   ragged all-to-all (host-known split sizes)

``TorchDistComm`` runs them through ``torch.distributed`` (backend "nccl" is RCCL over
xGMI on ROCm; "gloo" on CPU for tests). ``LoopbackComm`` implements the same calls over
in-process buffers so every rank's shard can be run sequentially on one device and
compared with the unsharded model.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch


class Comm:
    rank: int = 0
    world: int = 1

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        return t

    def all_gather_cols(self, t: torch.Tensor, total_cols: Optional[int] = None) -> torch.Tensor:
        return t

    def all_gather_rows(self, t: torch.Tensor) -> torch.Tensor:
        return t

    def broadcast(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        return t

    def all_to_all_rows(self, t: torch.Tensor, send_counts: Sequence[int], recv_counts: Sequence[int]) -> torch.Tensor:
        return t

    def all_to_all_equal(self, t: torch.Tensor) -> torch.Tensor:
        """All-to-all with equal splits: ``t`` is ``world`` blocks of equal length along dim 0,
        block r goes to rank r; the result's block s is what rank s sent here."""
        return t

    def all_to_all_counted(self, t: torch.Tensor, counts: torch.Tensor, per: int):
        """All-to-all of row blocks of which only a counted prefix exists: ``t`` is ``world``
        blocks of equal height, block r bound for rank r; ``counts`` (int32 [world * per], on
        ``t``'s device) says per destination, in ``per`` slots, how many of its rows are real.
        Returns (received blocks, received counts): entry s * per + e of the latter is what rank
        s reported for (this rank, slot e). This default moves the whole blocks with two
        equal-split exchanges; communicators that can read the counts on the device move only the
        rows that exist (parallel/p2p.py)."""
        return self.all_to_all_equal(t), self.all_to_all_equal(counts)

    def counted_rows_cap(self, row_bytes: int) -> int:
        """Most rows per (source, destination) pair one all_to_all_counted call can carry."""
        return 1 << 30

    def barrier(self) -> None:
        pass

    def max_float(self, x: float) -> float:
        return x

    def gather_floats(self, xs: Sequence[float]) -> List[float]:
        return list(xs)

    def all_true(self, flag: bool) -> bool:
        """Logical AND across ranks: the collective form of 'every member voted Good'."""
        return flag

    def broadcast_object(self, obj, src: int = 0):
        """Host-side broadcast of a small picklable object (texts, vote records)."""
        return obj

    def all_gather_object(self, obj) -> list:
        return [obj]

    def all_gather_bytes(self, data: bytes) -> List[bytes]:
        """Every rank's byte string, in rank order (a serialised record per rank; the caller
        chooses the encoding). The default carries them as host objects."""
        return [bytes(x) for x in self.all_gather_object(bytes(data))]

    def subgroup(self, ranks: Sequence[int]) -> "Comm":
        """Communicator over ``ranks`` (ranks of THIS communicator, ascending). Collective:
        every rank of this communicator must call it with the same arguments in the same
        order; ranks outside ``ranks`` get None."""
        return self if list(ranks) == [self.rank] or self.world == 1 else None  # type: ignore[return-value]


class SingleComm(Comm):
    pass


class TorchDistComm(Comm):
    """torch.distributed collectives (RCCL on GPUs, gloo on the host).

    Recording these into a hipGraph: the process group runs a watchdog thread that queries the
    events of outstanding work. While a stream capture in GLOBAL mode is open, a HIP call from
    any thread is an error; the watchdog's query then throws inside its own thread
    (ProcessGroupNCCL.cpp, the watchdog's run loop) and the process ends with SIGABRT. A capture
    that is open for a few milliseconds usually slips between two polls; a decode step of a few
    hundred launches does not. Captured in THREAD_LOCAL mode the same all-reduce records and
    replays correctly: cold or warmed-up communicator, default group or sub-group
    (benchmarks/micro/rccl_capture_probe.py runs every combination in a child process; only
    "global mode, capture open for 2.5 s" aborts). The engine captures in thread-local mode,
    so ``capturable_thread_local`` communicators may be captured when the caller asks for it;
    the default for sharded members remains the peer-to-peer communicator (parallel/p2p.py),
    whose collectives are single kernel launches."""

    def __init__(self, dist, device: torch.device, group=None) -> None:
        self.dist, self.device, self.group = dist, device, group
        # only RCCL enqueues its collectives on the stream; gloo stages through the host
        self.capturable_thread_local = str(dist.get_backend(group)).lower() == "nccl"
        # gloo group, tensors on a GPU (several ranks rehearsing on one device, where RCCL refuses to
        # start): every collective goes through host copies of its operands
        self.staged = not self.capturable_thread_local and torch.device(device).type == "cuda"
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def _in(self, t: torch.Tensor) -> torch.Tensor:
        return t.cpu() if (self.staged and t.is_cuda) else t

    def _out(self, h: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
        return h.to(like.device) if h.device != like.device else h

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        h = self._in(t)
        self.dist.all_reduce(h, group=self.group)
        if h is not t:
            t.copy_(h)
        return t

    def all_gather_cols(self, t: torch.Tensor, total_cols: Optional[int] = None) -> torch.Tensor:
        # one collective into one buffer [world, ..., C], then one strided copy to [..., world * C]
        t = t.contiguous()
        h = self._in(t)
        buf = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=h.device)
        self.dist.all_gather_into_tensor(buf, h, group=self.group)
        buf = self._out(buf, t)
        out = buf.view((self.world,) + tuple(t.shape)).movedim(0, -2).reshape(
            tuple(t.shape[:-1]) + (self.world * t.shape[-1],))
        return out[..., :total_cols] if total_cols is not None else out

    def all_gather_rows(self, t: torch.Tensor) -> torch.Tensor:
        t = t.contiguous()
        h = self._in(t)
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=h.device)
        self.dist.all_gather_into_tensor(out, h, group=self.group)
        return self._out(out, t)

    def broadcast(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        h = self._in(t)
        self.dist.broadcast(h, src=self._global(src), group=self.group)
        if h is not t:
            t.copy_(h)
        return t

    def all_to_all_rows(self, t: torch.Tensor, send_counts: Sequence[int], recv_counts: Sequence[int]) -> torch.Tensor:
        h = self._in(t.contiguous())
        out = torch.empty((int(sum(recv_counts)),) + tuple(t.shape[1:]), dtype=t.dtype, device=h.device)
        self.dist.all_to_all_single(out, h, list(map(int, recv_counts)), list(map(int, send_counts)),
                                    group=self.group)
        return self._out(out, t)

    def all_to_all_equal(self, t: torch.Tensor) -> torch.Tensor:
        t = t.contiguous()
        h = self._in(t)
        out = torch.empty_like(h)
        self.dist.all_to_all_single(out, h, group=self.group)
        return self._out(out, t)

    def barrier(self) -> None:
        self.dist.barrier(group=self.group)

    def max_float(self, x: float) -> float:
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if self.staged else self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def gather_floats(self, xs: Sequence[float]) -> List[float]:
        objs: List[Optional[list]] = [None] * self.world
        self.dist.all_gather_object(objs, list(map(float, xs)), group=self.group)
        return [x for part in objs for x in (part or [])]

    def all_true(self, flag: bool) -> bool:
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cpu" if self.staged else self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return bool(t.item())

    def _global(self, group_rank: int) -> int:
        return self.dist.get_global_rank(self.group, group_rank) if self.group is not None else group_rank

    def broadcast_object(self, obj, src: int = 0):
        box = [obj if self.rank == src else None]
        self.dist.broadcast_object_list(box, src=self._global(src), group=self.group,
                                        device=torch.device("cpu") if self.staged else self.device)
        return box[0]

    def all_gather_object(self, obj) -> list:
        out: List[object] = [None] * self.world
        self.dist.all_gather_object(out, obj, group=self.group)
        return out

    def all_gather_bytes(self, data: bytes) -> List[bytes]:
        """Two tensor collectives on the group's own transport (device memory under RCCL): the
        lengths, then the byte strings padded to the longest. Nothing is pickled."""
        dev = "cpu" if self.staged else self.device
        n = torch.tensor([len(data)], dtype=torch.int64, device=dev)
        sizes = torch.empty(self.world, dtype=torch.int64, device=dev)
        self.dist.all_gather_into_tensor(sizes, n, group=self.group)
        sizes = sizes.cpu().tolist()
        cap = max(1, max(sizes))
        buf = torch.zeros(cap, dtype=torch.uint8)
        if data:
            buf[: len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8)
        out = torch.empty(self.world * cap, dtype=torch.uint8, device=dev)
        self.dist.all_gather_into_tensor(out, buf.to(dev), group=self.group)
        out = out.cpu()
        return [bytes(out[r * cap: r * cap + sizes[r]].numpy().tobytes()) for r in range(self.world)]

    def subgroup(self, ranks: Sequence[int]) -> Optional["Comm"]:
        ranks = list(ranks)
        g = self.dist.new_group([self._global(r) for r in ranks])   # collective over the default group
        if self.rank not in ranks:
            return None
        return SingleComm() if len(ranks) == 1 else TorchDistComm(self.dist, self.device, g)


class LoopbackHub:
    """Shared state of a set of LoopbackComm ranks that run as threads of one process
    (all on one device): a barrier plus one slot per rank. Every collective is
    deposit -> barrier -> combine -> barrier, so it is deterministic and needs no
    second GPU. Used by the TP/EP tests to compare sharded and unsharded models."""

    def __init__(self, world: int) -> None:
        import threading

        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots: List[Optional[torch.Tensor]] = [None] * world
        self.objs: List[object] = [None] * world
        self.lock = threading.Lock()
        self.children: dict = {}


class LoopbackComm(Comm):
    def __init__(self, hub: LoopbackHub, rank: int) -> None:
        self.hub, self.rank, self.world = hub, rank, hub.world

    def _exchange(self, t: torch.Tensor) -> List[torch.Tensor]:
        if t.is_cuda:
            torch.cuda.synchronize()
        self.hub.slots[self.rank] = t
        self.hub.barrier.wait()
        parts = [x.clone() for x in self.hub.slots]  # type: ignore[union-attr]
        if t.is_cuda:
            torch.cuda.synchronize()
        self.hub.barrier.wait()
        return parts

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        t.copy_(reduce_loopback(self._exchange(t)))
        return t

    def all_gather_cols(self, t: torch.Tensor, total_cols: Optional[int] = None) -> torch.Tensor:
        out = torch.cat(self._exchange(t.contiguous()), dim=-1)
        return out[..., :total_cols] if total_cols is not None else out

    def all_gather_rows(self, t: torch.Tensor) -> torch.Tensor:
        return torch.cat(self._exchange(t.contiguous()), dim=0)

    def broadcast(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        t.copy_(self._exchange(t)[src])
        return t

    def all_to_all_rows(self, t: torch.Tensor, send_counts: Sequence[int], recv_counts: Sequence[int]) -> torch.Tensor:
        self.hub.objs[self.rank] = list(map(int, send_counts))
        parts = self._exchange(t.contiguous())
        counts = [list(c) for c in self.hub.objs]  # type: ignore[arg-type]
        self.hub.barrier.wait()
        out = []
        for src in range(self.world):
            off = sum(counts[src][: self.rank])
            out.append(parts[src][off: off + counts[src][self.rank]])
        return torch.cat(out, dim=0)

    def all_to_all_equal(self, t: torch.Tensor) -> torch.Tensor:
        parts = self._exchange(t.contiguous())
        n = t.shape[0] // self.world
        return torch.cat([p[self.rank * n:(self.rank + 1) * n] for p in parts], dim=0)

    def barrier(self) -> None:
        self.hub.barrier.wait()

    def max_float(self, x: float) -> float:
        return float(max(p.item() for p in self._exchange(torch.tensor([x], dtype=torch.float64))))

    def gather_floats(self, xs: Sequence[float]) -> List[float]:
        self.hub.objs[self.rank] = list(xs)
        self.hub.barrier.wait()
        out = [x for part in self.hub.objs for x in part]  # type: ignore[union-attr]
        self.hub.barrier.wait()
        return out

    def all_true(self, flag: bool) -> bool:
        return all(bool(p.item()) for p in self._exchange(torch.tensor([1 if flag else 0])))

    def broadcast_object(self, obj, src: int = 0):
        if self.rank == src:
            self.hub.objs[src] = obj
        self.hub.barrier.wait()
        out = self.hub.objs[src]
        self.hub.barrier.wait()
        return out

    def all_gather_object(self, obj) -> list:
        self.hub.objs[self.rank] = obj
        self.hub.barrier.wait()
        out = list(self.hub.objs)
        self.hub.barrier.wait()
        return out

    def subgroup(self, ranks: Sequence[int]) -> Optional["Comm"]:
        ranks = list(ranks)
        # Every rank counts every call (member of the new group or not), so the counter
        # names the same group on all ranks, as creation order does for process groups.
        seq = self._subgroup_calls = getattr(self, "_subgroup_calls", 0) + 1
        if self.rank not in ranks:
            return None
        if len(ranks) == 1:
            return SingleComm()
        with self.hub.lock:
            hub = self.hub.children.setdefault((tuple(ranks), seq), LoopbackHub(len(ranks)))
        return LoopbackComm(hub, ranks.index(self.rank))


def run_loopback(world: int, fn) -> list:
    """Runs ``fn(comm)`` on ``world`` threads sharing a LoopbackHub; returns per-rank results
    (re-raises the first exception)."""
    import threading

    hub = LoopbackHub(world)
    results: List[object] = [None] * world
    errors: List[Optional[BaseException]] = [None] * world

    def work(r: int) -> None:
        try:
            results[r] = fn(LoopbackComm(hub, r))
        except BaseException as e:  # noqa: BLE001
            errors[r] = e
            hub.barrier.abort()

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errors:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in errors:
        if e is not None:
            raise e
    return results


def make_comm(dist, device: torch.device, group=None) -> Comm:
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return SingleComm()
    return TorchDistComm(dist, device, group)


def reduce_loopback(parts: List[torch.Tensor]) -> torch.Tensor:
    """Sum in rank order with one rounding per step, as a ring/tree all-reduce in bf16 would
    (used by the loopback TP runner so sharded-vs-unsharded comparisons are fair)."""
    acc = parts[0].float()
    for p in parts[1:]:
        acc = acc + p.float()
    return acc.to(parts[0].dtype)
