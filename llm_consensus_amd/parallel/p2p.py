"""Peer-to-peer collectives of one model group (csrc/kernels/p2p.hip) behind the ``Comm`` interface.

Every rank of a tensor- or expert-parallel group allocates one exchange region, hands its
HIP IPC handle to the other ranks (one host-side all-gather at start-up, through whatever
communicator bootstraps the group: RCCL or gloo process groups, or the in-process loopback
hub), and maps theirs. From then on the group's hot collectives are single kernel launches
that read and write the peers' regions over xGMI:

    allreduce_rmsnorm  sum of the ranks' partial products + residual add + RMSNorm (the two
                       per-layer reductions of a tensor-parallel transformer block)
    all_reduce         plain sum of bf16 [T, d] tensors
    all_gather_cols / all_gather_rows / all_to_all_equal   byte exchanges

They contain no RCCL call and no host synchronisation, so a decode step that uses them is
captured into one hipGraph like an unsharded one (``capturable``). Everything else
(broadcasts of Python objects, barriers, ragged all-to-all) goes to the bootstrap
communicator.

The reference's members never talk to each other directly; all of its traffic is
coordinator <-> member messages on one thread (/root/reference/src/main.rs:234, 250-253).
This module exists because one *member* here can span several GPUs.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_double, c_float, c_int, c_longlong, c_void_p
from typing import List, Optional

import torch

from .comm import Comm

_lib = None


def _L():
    global _lib
    if _lib is None:
        from .._native import kernels

        l = kernels()
        l.lc_p2p_region_bytes.argtypes, l.lc_p2p_region_bytes.restype = [c_longlong], c_longlong
        l.lc_p2p_alloc.argtypes, l.lc_p2p_alloc.restype = [c_longlong, ctypes.POINTER(c_void_p), c_void_p], c_int
        l.lc_p2p_open.argtypes, l.lc_p2p_open.restype = [c_void_p, ctypes.POINTER(c_void_p)], c_int
        l.lc_p2p_close.argtypes, l.lc_p2p_close.restype = [c_void_p], c_int
        l.lc_p2p_free.argtypes, l.lc_p2p_free.restype = [c_void_p], c_int
        l.lc_p2p_handle_bytes.argtypes, l.lc_p2p_handle_bytes.restype = [], c_int
        l.lc_p2p_comm_create.argtypes = [c_int, c_int, ctypes.POINTER(c_void_p), c_longlong, c_double, c_int]
        l.lc_p2p_comm_create.restype = c_void_p
        l.lc_p2p_comm_destroy.argtypes, l.lc_p2p_comm_destroy.restype = [c_void_p], None
        l.lc_p2p_comm_error.argtypes, l.lc_p2p_comm_error.restype = [c_void_p], c_int
        l.lc_p2p_comm_capacity.argtypes, l.lc_p2p_comm_capacity.restype = [c_void_p], c_longlong
        l.lc_p2p_comm_set_timeout.argtypes, l.lc_p2p_comm_set_timeout.restype = [c_void_p, c_double], None
        l.lc_p2p_allreduce.argtypes = [c_void_p, c_void_p, c_int, c_longlong, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_int, c_int, c_float, c_int, c_int, c_void_p]
        l.lc_p2p_allreduce.restype = c_int
        l.lc_p2p_exchange.argtypes = [c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p]
        l.lc_p2p_exchange.restype = c_int
        l.lc_p2p_counted_cap.argtypes, l.lc_p2p_counted_cap.restype = [c_void_p, c_longlong], c_longlong
        l.lc_p2p_exchange_counted.argtypes = [c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_void_p, c_int, c_void_p,
                                              c_void_p]
        l.lc_p2p_exchange_counted.restype = c_int
        _lib = l
    return _lib


# Messages (bytes per rank) above which the all-reduce runs as reduce-scatter + all-gather.
TWO_SHOT_BYTES = int(os.environ.get("LC_P2P_TWO_SHOT_BYTES", str(1 << 20)))
DEFAULT_STAGE_BYTES = int(os.environ.get("LC_P2P_STAGE_BYTES", str(64 << 20)))
# How long a collective waits for a peer before it flags the group as failed. Ranks of a group
# reach the same step at different times (a rank that also hosts another member may be in that
# member's prefill, the first use of a batch shape records a graph): generous, because the only
# cost of a long limit is how late a dead peer is noticed.
TIMEOUT_S = float(os.environ.get("LC_P2P_TIMEOUT_S", "60"))
# Most blocks one collective launches (0 = the kernel library's default of 64). Lower it when
# several ranks share one GPU (tests): waiting blocks hold CU slots.
GRID_CAP = int(os.environ.get("LC_P2P_GRID_CAP", "0"))
# First contact: how long the self-check of a newly built group waits for a peer.
SELF_CHECK_TIMEOUT_S = float(os.environ.get("LC_P2P_SELF_CHECK_TIMEOUT_S", "5"))
# Fault injection for the first-contact path: "stall" (the chosen rank never launches its part of
# the self-check, its peers time out) or "corrupt" (the chosen rank contributes wrong values),
# optionally ":<rank>" (default: the group's last rank).
FAIL_INJECT = os.environ.get("LC_P2P_FAIL_INJECT", "")


class P2PError(RuntimeError):
    pass


def supported(world: int, device) -> bool:
    return world in (2, 4, 8) and torch.device(device).type == "cuda"


class P2PComm(Comm):
    """``base`` bootstraps the group (its ``all_gather_object`` carries the IPC handles) and
    serves the collectives this class does not implement. All ranks of the group must
    construct it at the same point (collective)."""

    capturable = True

    # Measurement aid: this rank only stages and flags its contribution to every all-reduce and
    # never waits (see lc_p2p_allreduce); its own results are meaningless. Forces the one-shot form
    # on THIS rank only, so set TWO_SHOT_BYTES high on the ranks under test too.
    publish_only = False

    def __init__(self, base: Comm, device, stage_bytes: int = DEFAULT_STAGE_BYTES, timeout_s: float = TIMEOUT_S,
                 grid_cap: int = GRID_CAP) -> None:
        if not supported(base.world, device):
            raise ValueError(f"P2PComm needs a CUDA device and 2, 4 or 8 ranks (got {base.world} on {device})")
        self.base, self.rank, self.world = base, base.rank, base.world
        self.device = torch.device(device)
        self.timeout_s = timeout_s
        self._h = None
        self._own = None
        self._opened: List[int] = []
        l = _L()
        # Every step that can fail on one rank only (allocation, mapping a peer's region) is
        # followed by a collective on which all ranks go on: a rank that raised on its own would
        # leave the others waiting in the next collective for ever. The outcome is agreed with
        # all_true and raised on EVERY rank.
        problem: Optional[str] = None
        with torch.cuda.device(self.device):
            nbytes = l.lc_p2p_region_bytes(stage_bytes)
            ptr = c_void_p()
            handle = ctypes.create_string_buffer(l.lc_p2p_handle_bytes())
            rc = l.lc_p2p_alloc(nbytes, ctypes.byref(ptr), handle)
            if rc != 0:
                problem = f"exchange region allocation of {nbytes} bytes failed: HIP error {rc}"
            else:
                self._own = ptr.value
            # (process id, device index) tells ranks that are threads of one process on one
            # device apart from real peers: they share an address space and use the pointer.
            my_dev = self.device.index if self.device.index is not None else torch.cuda.current_device()
            me = (os.getpid(), my_dev, self._own, bytes(handle.raw))
            peers = base.all_gather_object(me)
            bases = (c_void_p * self.world)()
            for r, (pid, dev, p, hbytes) in enumerate(peers):
                if problem is not None:
                    break
                if p is None:
                    problem = f"rank {r} could not allocate its exchange region"
                elif r == self.rank:
                    bases[r] = self._own
                elif pid == os.getpid() and dev == my_dev:
                    bases[r] = p
                elif pid == os.getpid():
                    # thread-ranks of one process on different devices: the raw pointer is only
                    # usable with peer access enabled and the uncached region mapped on the peer,
                    # neither of which this path sets up
                    problem = (f"ranks {self.rank} and {r} are threads of one process on different devices "
                               f"({my_dev}, {dev}); use one process per GPU")
                else:
                    q = c_void_p()
                    rc = l.lc_p2p_open(hbytes, ctypes.byref(q))
                    if rc != 0:
                        problem = (f"rank {self.rank}: cannot map the exchange region of rank {r}: HIP error {rc} "
                                   "(HSA_ENABLE_IPC_MODE_LEGACY=0 must be set in every rank's environment)")
                    else:
                        self._opened.append(q.value)
                        bases[r] = q.value
            if problem is None:
                self._h = l.lc_p2p_comm_create(self.world, self.rank, bases, stage_bytes, timeout_s, grid_cap)
                if not self._h:
                    problem = "lc_p2p_comm_create failed"
                else:
                    self.capacity = int(l.lc_p2p_comm_capacity(self._h))
        # the all_true doubles as the barrier: nobody launches before every rank has mapped every region
        if not base.all_true(problem is None):
            self._release()
            why = [w for w in base.all_gather_object(problem) if w]
            raise P2PError("; ".join(why) or "a peer failed to set up its exchange region")
        self.calls = 0

    # ------------------------------------------------------------------ first contact
    def self_check(self, timeout_s: float = SELF_CHECK_TIMEOUT_S) -> Optional[str]:
        """One tiny all-reduce (one-shot), one of the two-shot size, one fused all-reduce +
        residual + RMSNorm, one all-gather and one equal-split all-to-all, with a short timeout,
        compared with the values every rank can compute for itself (each rank's contribution is
        a function of its rank) and with the bootstrap communicator's all-reduce of the same
        tensor. Collective: returns None on every rank when all ranks saw what they should, else
        the same description of what went wrong. After a failure the communicator must be
        discarded (its error word is set and its call counters may disagree between ranks)."""
        l = _L()
        dev, W, r = self.device, self.world, self.rank
        inject, _, who = FAIL_INJECT.partition(":")
        victim = int(who) if who else W - 1
        stall = inject == "stall" and r == victim
        corrupt = inject == "corrupt" and r == victim
        problem: Optional[str] = None
        l.lc_p2p_comm_set_timeout(self._h, min(timeout_s, self.timeout_s))
        try:
            with torch.cuda.device(dev):
                if not stall:
                    def contrib(rank: int, shape) -> torch.Tensor:
                        n = shape[0] * shape[1]
                        base_vals = (torch.arange(n, dtype=torch.float32) % 7 - 3).view(shape)   # small integers:
                        return base_vals + float(rank + 1)                                     # exact in bf16 sums

                    for shape in ((4, 64), (max(1, TWO_SHOT_BYTES // (2 * 1024)) + 16, 1024)):
                        if shape[0] * shape[1] * 2 > self.capacity:
                            continue
                        mine = contrib(r, shape) + (1.0 if corrupt else 0.0)
                        want = sum(contrib(q, shape) for q in range(W))
                        x = mine.to(dev, torch.bfloat16)
                        ref_x = x.clone()
                        got = self.all_reduce(x).float().cpu()
                        if not torch.equal(got, want):
                            problem = problem or f"all-reduce of {shape[0]}x{shape[1]} gave wrong sums on rank {r}"
                        # the bootstrap communicator's answer to the same question (fp32, so both are exact)
                        ref_sum = self.base.all_reduce(ref_x.float()).cpu()
                        if not torch.equal(ref_sum, got):
                            problem = problem or f"all-reduce of {shape[0]}x{shape[1]} disagrees with the bootstrap communicator on rank {r}"
                    # fused form: h += sum(delta); y = RMSNorm(h) * w
                    shape = (3, 128)
                    h0 = torch.ones(shape, dtype=torch.bfloat16, device=dev)
                    delta = (contrib(r, shape) + (1.0 if corrupt else 0.0)).to(dev, torch.bfloat16)
                    wgt = torch.full((shape[1],), 2.0, dtype=torch.bfloat16, device=dev)
                    y = self.allreduce_rmsnorm(h0, delta, wgt, 1e-6).float().cpu()
                    hsum = 1.0 + sum(contrib(q, shape) for q in range(W))
                    want_y = hsum * torch.rsqrt((hsum * hsum).mean(-1, keepdim=True) + 1e-6) * 2.0
                    if not torch.equal(h0.float().cpu(), hsum) or not torch.allclose(y, want_y, atol=0.05, rtol=0.02):
                        problem = problem or f"fused all-reduce + RMSNorm gave wrong values on rank {r}"
                    rows = torch.arange(2 * 8, dtype=torch.int32).view(2, 8) + 1000 * (r + 1) + (1 if corrupt else 0)
                    got = self.all_gather_rows(rows.to(dev)).cpu()
                    want = torch.cat([torch.arange(2 * 8, dtype=torch.int32).view(2, 8) + 1000 * (q + 1) for q in range(W)])
                    if not torch.equal(got, want):
                        problem = problem or f"all-gather gave wrong rows on rank {r}"
                    send = (torch.arange(W * 2 * 8, dtype=torch.int32).view(W * 2, 8) + 100000 * (r + 1)
                            + (1 if corrupt else 0))
                    got = self.all_to_all_equal(send.to(dev)).cpu()
                    want = torch.cat([(torch.arange(W * 2 * 8, dtype=torch.int32).view(W * 2, 8) + 100000 * (q + 1))[2 * r:2 * r + 2]
                                      for q in range(W)])
                    if not torch.equal(got, want):
                        problem = problem or f"all-to-all gave wrong rows on rank {r}"
                else:
                    # keep the bootstrap communicator's call sequence in step with the peers
                    for shape in ((4, 64), (max(1, TWO_SHOT_BYTES // (2 * 1024)) + 16, 1024)):
                        if shape[0] * shape[1] * 2 <= self.capacity:
                            self.base.all_reduce(torch.zeros(shape, dtype=torch.float32, device=dev))
                    problem = "injected stall: this rank did not take part"
                torch.cuda.synchronize(dev)
                e = l.lc_p2p_comm_error(self._h)
                if e != 0:
                    problem = (f"rank {r}: timed out after {min(timeout_s, self.timeout_s):.0f} s waiting for rank {e - 1}"
                               if e > 0 else f"rank {r}: cannot read the communicator's error word")
        except (P2PError, RuntimeError) as exc:
            problem = problem or f"rank {r}: {type(exc).__name__}: {exc}"
        finally:
            if self._h:
                l.lc_p2p_comm_set_timeout(self._h, self.timeout_s)
        if self.base.all_true(problem is None):
            return None
        return "; ".join(w for w in self.base.all_gather_object(problem) if w)

    # ------------------------------------------------------------------ fused all-reduce
    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            raise P2PError(f"{what} failed: {'invalid arguments' if rc < 0 else f'HIP error {rc}'}")

    def _allreduce(self, part, xin, h, w, y, T: int, d: int, eps: float) -> None:
        l = _L()
        rows_cap = max(1, self.capacity // (d * 2))
        S = part.shape[0] if part is not None else 0
        for r0 in range(0, T, rows_cap):           # same split on every rank: capacity and shapes are equal
            n = min(rows_cap, T - r0)
            two = 1 if (n * d * 2 > TWO_SHOT_BYTES and not self.publish_only) else 0
            xr = xin[r0:r0 + n] if xin is not None else None
            # a row range of the slab stack: same slabs, first row r0, slabs still T * d apart
            pp = part.data_ptr() + 4 * r0 * d if part is not None else None
            rc = l.lc_p2p_allreduce(
                self._h, pp, S, T * d, xr.data_ptr() if xr is not None else None,
                h[r0:r0 + n].data_ptr() if h is not None else None,
                w.data_ptr() if w is not None else None, y[r0:r0 + n].data_ptr(), n, d, float(eps), two,
                1 if self.publish_only else 0, self._stream())
            self._check(rc, "lc_p2p_allreduce")
            self.calls += 1

    def allreduce_rmsnorm(self, h: torch.Tensor, delta, w: torch.Tensor, eps: float) -> torch.Tensor:
        """h += sum over ranks of ``delta`` (a bf16 [T, d] tensor or an ``ops.PartialSum`` of fp32
        split-K slabs); returns RMSNorm(h) * w. One launch."""
        from ..ops import PartialSum

        T, d = h.shape
        y = torch.empty_like(h)
        if isinstance(delta, PartialSum):
            if delta.bias is not None:
                delta = delta.reduce()
            else:
                self._allreduce(delta.part, None, h, w, y, T, d, eps)
                return y
        self._allreduce(None, delta.contiguous(), h, w, y, T, d, eps)
        return y

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        if not (t.is_cuda and t.dtype == torch.bfloat16 and t.dim() == 2 and t.is_contiguous() and t.shape[1] % 8 == 0):
            return self.base.all_reduce(t)
        y = torch.empty_like(t)
        self._allreduce(None, t, None, None, y, t.shape[0], t.shape[1], 0.0)
        t.copy_(y)
        return t

    # ------------------------------------------------------------------ byte exchanges
    def _exchange(self, src: torch.Tensor, all_to_all: bool) -> Optional[torch.Tensor]:
        """Returns [world, ...block shape], or None when the message is not a device tensor or
        does not fit the staging slot. Blocks of any byte length travel in one launch (the
        kernel has a narrow path for lengths that are not a multiple of 16)."""
        src = src.contiguous()
        nbytes = src.numel() * src.element_size()
        nblk = self.world if all_to_all else 1
        m = nbytes // nblk
        if not src.is_cuda or m == 0 or nbytes % nblk:
            return None
        if (m + 15) // 16 * 16 * nblk > self.capacity:
            return None
        block_shape = ((src.shape[0] // self.world,) + tuple(src.shape[1:])) if all_to_all else tuple(src.shape)
        out = torch.empty((self.world,) + block_shape, dtype=src.dtype, device=src.device)
        rc = _L().lc_p2p_exchange(self._h, src.data_ptr(), out.data_ptr(), m, 1 if all_to_all else 0,
                                  1 if self.publish_only else 0, self._stream())
        self._check(rc, "lc_p2p_exchange")
        self.calls += 1
        return out

    def all_gather_rows(self, t: torch.Tensor) -> torch.Tensor:
        out = self._exchange(t, False)
        if out is None:
            return self.base.all_gather_rows(t)
        return out.view((self.world * t.shape[0],) + tuple(t.shape[1:]))

    def all_gather_cols(self, t: torch.Tensor, total_cols: Optional[int] = None) -> torch.Tensor:
        out = self._exchange(t, False)
        if out is None:
            return self.base.all_gather_cols(t, total_cols)
        # [W, ..., c] -> [..., W*c], one launch
        from .. import ops

        lead = tuple(t.shape[:-1])
        res = ops.concat_cols(out.view(self.world, -1, t.shape[-1]), total_cols)
        return res.view(lead + (res.shape[-1],))

    def all_to_all_equal(self, t: torch.Tensor) -> torch.Tensor:
        out = self._exchange(t, True) if t.shape[0] % self.world == 0 else None
        if out is None:
            return self.base.all_to_all_equal(t)
        return out.view(t.shape)

    def counted_rows_cap(self, row_bytes: int) -> int:
        return int(_L().lc_p2p_counted_cap(self._h, row_bytes))

    def all_to_all_counted(self, t: torch.Tensor, counts: torch.Tensor, per: int):
        """``t`` [world * cap, d]: block r holds the rows bound for rank r, of which only the
        first sum(counts[r * per : (r + 1) * per]) exist (``counts`` int32 [world * per] on the
        device). Returns (received [world * cap, d], recv_counts int32 [world * per]): block s
        holds what rank s sent here, rows beyond its count are unspecified. Only rows that exist
        cross the fabric, and the host never learns the counts."""
        W = self.world
        cap = t.shape[0] // W
        row_bytes = t.shape[1] * t.element_size()
        if (not t.is_cuda or t.shape[0] % W or row_bytes % 16 or W * per * 4 > 4096
                or cap > self.counted_rows_cap(row_bytes)):
            return super().all_to_all_counted(t, counts, per)
        t, counts = t.contiguous(), counts.contiguous()
        out = torch.empty_like(t)
        recv_counts = torch.empty_like(counts)
        rc = _L().lc_p2p_exchange_counted(self._h, t.data_ptr(), out.data_ptr(), row_bytes, cap, counts.data_ptr(), per,
                                          recv_counts.data_ptr(), self._stream())
        self._check(rc, "lc_p2p_exchange_counted")
        self.calls += 1
        return out, recv_counts

    # ------------------------------------------------------------------ everything else
    def broadcast(self, t, src=0):
        return self.base.broadcast(t, src)

    def all_to_all_rows(self, t, send_counts, recv_counts):
        return self.base.all_to_all_rows(t, send_counts, recv_counts)

    def barrier(self) -> None:
        self.base.barrier()

    def max_float(self, x):
        return self.base.max_float(x)

    def gather_floats(self, xs):
        return self.base.gather_floats(xs)

    def all_true(self, flag):
        return self.base.all_true(flag)

    def broadcast_object(self, obj, src=0):
        return self.base.broadcast_object(obj, src)

    def all_gather_object(self, obj):
        return self.base.all_gather_object(obj)

    def subgroup(self, ranks):
        return self.base.subgroup(ranks)

    def check(self) -> None:
        """Raises if a kernel of this communicator gave up waiting for a peer (synchronises)."""
        e = _L().lc_p2p_comm_error(self._h)
        if e != 0:
            raise P2PError(f"rank {self.rank}: peer-to-peer collective timed out waiting for rank {e - 1}"
                           if e > 0 else "cannot read the communicator's error word")

    def _release(self) -> None:
        l = _L()
        if self._h:
            l.lc_p2p_comm_destroy(self._h)
            self._h = None
        for p in self._opened:
            l.lc_p2p_close(p)
        self._opened = []
        if self._own:
            l.lc_p2p_free(self._own)
            self._own = None

    def close(self) -> None:
        if getattr(self, "_h", None):
            torch.cuda.synchronize(self.device)
            self.base.barrier()          # no peer may still be reading this rank's region
            self._release()


def wrap_group_comm(comm: Optional[Comm], device, stage_bytes: int = DEFAULT_STAGE_BYTES) -> Optional[Comm]:
    """The model-group communicator to hand to a sharded model: peer-to-peer kernels when the
    group is 2, 4 or 8 ranks on GPUs (LC_P2P=0 keeps the bootstrap communicator).

    First contact: the new group runs ``P2PComm.self_check``. A group whose regions cannot be
    mapped, whose peers do not answer within a few seconds or whose sums come out wrong does not
    serve: all ranks agree on that, release the regions and go on IN THIS PROCESS with the
    bootstrap communicator (RCCL collectives, recorded into the decode graphs in thread-local
    capture mode, see TorchDistComm). ``comm.collectives`` / ``comm.fallback_reason`` say which
    path a group ended up on. The reference's only answer to a reply that never comes is to
    spin for ever (/root/reference/src/main.rs:450-459)."""
    if comm is None or comm.world == 1:
        return comm
    if os.environ.get("LC_P2P", "1") != "1" or not supported(comm.world, device):
        comm.collectives = _describe(comm)
        return comm
    why: Optional[str]
    p: Optional[P2PComm] = None
    try:
        p = P2PComm(comm, device, stage_bytes)
        why = p.self_check()
    except P2PError as e:          # raised on every rank (see P2PComm.__init__)
        why = str(e)
    if why is None:
        p.collectives = "p2p kernels over IPC-mapped memory"
        return p
    from ..utils.log import get_logger

    get_logger("consensus.parallel").warning(
        "rank %d: peer-to-peer collectives failed their first-contact check (%s); this group uses %s instead",
        comm.rank, why, _describe(comm))
    if p is not None and p._h:
        try:
            p.close()
        except Exception:  # noqa: BLE001  a device that faulted may refuse the synchronisation
            p._release()
    comm.collectives = _describe(comm)
    comm.fallback_reason = why
    return comm


def _describe(comm: Comm) -> str:
    if getattr(comm, "capturable_thread_local", False):
        return "rccl"
    if getattr(comm, "staged", False):
        return "gloo (host-staged)"
    dist = getattr(comm, "dist", None)
    if dist is not None:
        return str(dist.get_backend(getattr(comm, "group", None))).lower()
    return type(comm).__name__
