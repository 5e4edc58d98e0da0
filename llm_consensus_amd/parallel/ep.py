"""Expert parallelism for block-sparse MoE members (Mixtral).

The reference has no model code at all (/root/reference/src/main.rs:82-86 calls a remote
API), so there is nothing to compare line by line; the contract here is numerical: an
expert-parallel forward must equal the single-device forward of the same weights.

Experts are split evenly over the ranks of one group (rank r owns experts
[r*E/ep, (r+1)*E/ep)); attention, norms and the router are replicated, and every rank
sees the same token batch (the engines of a group run in lock step). Two exchange
schemes produce the replicated FFN output:

``allreduce``  every rank runs its resident experts on the whole batch (pairs routed
               elsewhere contribute zero) and the partial outputs are summed with one
               all-reduce of [T, d]. No host synchronisation, so the step stays
               graph-capturable; traffic 2*T*d per rank on a ring.
``a2a``        each rank takes a 1/ep slice of the tokens, routes it, sends every
               (token, k) row to the expert's owner with all-to-all, receives the MLP
               outputs back the same way, combines, and the slices are all-gathered.
               Every (source, destination) pair owns a fixed number of rows (``cap``, the
               worst case: all pairs of a chunk bound for one rank) of the exchange
               buffers, and the per-expert row counts travel as data next to the rows
               (``Comm.all_to_all_counted``): the peer-to-peer communicator reads them on
               the device and moves only the rows that exist, so the worst-case capacity
               costs address space, not traffic. A slice too long for one exchange is cut
               into chunks of EP_CHUNK_ROWS routed rows. Nothing in a layer waits for the
               host at any size: prefill chunks and decode steps alike are plain kernel
               launches. No rank multiplies rows for experts it does not own.

On xGMI all-to-all maps to direct peer writes over the point-to-point links (every pair of
GPUs in a node has its own link), which is the collective this fabric is best at.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch

from .. import ops
from .comm import Comm


def expert_range(n_experts: int, ep_rank: int, ep_size: int) -> Tuple[int, int]:
    if n_experts % ep_size:
        raise ValueError(f"{n_experts} experts do not divide over ep={ep_size}")
    per = n_experts // ep_size
    return ep_rank * per, (ep_rank + 1) * per


def moe_allreduce(x, Wr, w_gate_up, w_down, topk: int, comm: Comm, expert_lo: int) -> torch.Tensor:
    out = ops.moe_ffn(x, Wr, w_gate_up, w_down, topk, expert_lo)
    comm.all_reduce(out)
    return out


# Most routed rows (tokens * top-k) of a rank's slice that go into one exchange; the buffers of an
# exchange hold ep times as many rows (the worst case per destination). Also bounded by what the
# communicator's staging slot carries.
EP_CHUNK_ROWS = int(os.environ.get("LC_EP_CHUNK_ROWS", "1024"))

_zero_counts: dict = {}


def _zeros_i32(n: int, device) -> torch.Tensor:
    """A cached int32 zero vector (the count table of a rank that has no rows in a chunk)."""
    key = (str(device), n)
    z = _zero_counts.get(key)
    if z is None:
        z = torch.empty(n, dtype=torch.int32, device=device)
        if z.is_cuda:
            from ..ops import hip

            hip.call("lc_fill_f32", z.data_ptr(), 0.0, n)        # all-zero bits are the integer 0 too
        else:
            z.zero_()
        _zero_counts[key] = z
    return z


def moe_all_to_all(x, Wr, w_gate_up, w_down, topk: int, comm: Comm, expert_lo: int) -> torch.Tensor:
    T, d = x.shape
    ep, r = comm.world, comm.rank
    E = Wr.shape[0]
    per = E // ep
    tl = (T + ep - 1) // ep                      # slice length, the same on every rank
    lo = min(T, r * tl)
    n = min(T, (r + 1) * tl) - lo                # rows of the slice that exist on this rank
    # Rows [n, tl) of a short last slice are never written: after the gather they lie beyond T.
    out = torch.empty(tl, d, dtype=x.dtype, device=x.device)
    rows_cap = max(topk, min(EP_CHUNK_ROWS, comm.counted_rows_cap(d * x.element_size())))
    ct = max(1, rows_cap // topk)                # tokens per chunk
    for c0 in range(0, tl, ct):
        c1 = min(tl, c0 + ct)
        cap = (c1 - c0) * topk                   # every pair's share of the buffers (same on all ranks)
        m = max(0, min(n, c1) - c0)              # this rank's tokens in the chunk
        if m > 0:
            xs = x[lo + c0: lo + c0 + m]
            ids, wts = ops.moe_route(xs, Wr, topk)
            offsets, sorted_src, slot_of = ops.moe_permute(ids, E)   # expert order == destination order
            send, send_counts, slot_back = ops.ep_pack(xs, sorted_src, offsets, slot_of, per, ep, cap)
        else:
            send = torch.empty(ep * cap, d, dtype=x.dtype, device=x.device)
            send_counts = _zeros_i32(ep * per, x.device)
        recv, recv_counts = comm.all_to_all_counted(send, send_counts, per)     # block s came from rank s
        row_index, off_local = ops.ep_recv_index(recv_counts, ep, per, cap)
        y = ops.experts_on_sorted(recv, row_index, off_local, 0, w_gate_up, w_down, ep * cap)
        # way back: to every source exactly the rows it sent, in its order
        back, _ = comm.all_to_all_counted(ops.scatter_rows(y, row_index, off_local[per:]), recv_counts, per)
        if m > 0:
            ops.moe_combine(back, slot_back, wts, m, out=out[c0:c0 + m])
    full = comm.all_gather_rows(out)
    return full[:T] if ep * tl != T else full


def moe_ffn_ep(x, Wr, w_gate_up, w_down, topk: int, comm: Comm, expert_lo: int, mode: str = "allreduce"):
    if comm is None or comm.world == 1:
        return ops.moe_ffn(x, Wr, w_gate_up, w_down, topk)
    if mode == "a2a":
        return moe_all_to_all(x, Wr, w_gate_up, w_down, topk, comm, expert_lo)
    if mode == "allreduce":
        return moe_allreduce(x, Wr, w_gate_up, w_down, topk, comm, expert_lo)
    raise ValueError(f"unknown ep mode {mode!r}")
