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
               Every (source, destination) pair owns a fixed number of rows (``cap``) of
               the exchange buffers, so all three exchanges have equal splits and the
               per-expert row counts travel as data: up to EP_FIXED_MAX_ROWS routed rows
               per rank (decode steps) cap is the worst case and nothing waits for the
               host; above that (prefill chunks) the ranks agree on the largest actual
               count first, which costs one device->host read per layer and keeps the
               buffers close to the T/ep*k*d bytes a rank really has to send. No rank
               multiplies rows for experts it does not own.

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


# Routed rows per rank (tokens/ep * top-k) up to which the exchange buffers are sized for the
# worst case (every pair of a rank bound for one destination) instead of the measured maximum.
EP_FIXED_MAX_ROWS = int(os.environ.get("LC_EP_FIXED_ROWS", "256"))


def moe_all_to_all(x, Wr, w_gate_up, w_down, topk: int, comm: Comm, expert_lo: int) -> torch.Tensor:
    T, d = x.shape
    ep, r = comm.world, comm.rank
    E = Wr.shape[0]
    per = E // ep
    # ---- this rank's token slice, padded with zero rows to the common slice length
    tl = (T + ep - 1) // ep
    lo, hi = min(T, r * tl), min(T, (r + 1) * tl)
    xs = x[lo:hi]
    if hi - lo < tl:
        xs = torch.cat([xs, x.new_zeros(tl - (hi - lo), d)], 0)
    xs = xs.contiguous()
    ids, wts = ops.moe_route(xs, Wr, topk)
    offsets, sorted_src, slot_of = ops.moe_permute(ids, E)       # expert order == destination order
    A = tl * topk
    if A <= EP_FIXED_MAX_ROWS:
        cap = A
    else:
        off = offsets[::per].to(torch.int64)                      # [ep + 1] destination boundaries
        most = comm.all_gather_rows((off[1:] - off[:-1]).max().view(1, 1))
        cap = max(64, -(-int(most.max().item()) // 64) * 64)      # the one host read of this layer
    send, send_counts, slot_back = ops.ep_pack(xs, sorted_src, offsets, slot_of, per, ep, cap)
    recv = comm.all_to_all_equal(send)                            # [ep*cap, d]: block s came from rank s
    recv_counts = comm.all_to_all_equal(send_counts)              # [ep*per]
    row_index, off_local = ops.ep_recv_index(recv_counts, ep, per, cap)
    y = ops.experts_on_sorted(recv, row_index, off_local, 0, w_gate_up, w_down, ep * cap)
    back = comm.all_to_all_equal(ops.scatter_rows(y, row_index, off_local[per:]))
    out = ops.moe_combine(back, slot_back, wts, tl)
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
