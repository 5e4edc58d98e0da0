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
               Traffic 2*(T/ep)*k*d*(ep-1)/ep + T*d per rank: less than the all-reduce
               when k < ep, and no rank multiplies rows for experts it does not own.
               Split sizes are data dependent (one small D2H per layer).

On xGMI all-to-all maps to direct peer writes over the point-to-point links (every pair of
GPUs in a node has its own link), which is the collective this fabric is best at.
"""
from __future__ import annotations

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


def moe_all_to_all(x, Wr, w_gate_up, w_down, topk: int, comm: Comm, expert_lo: int) -> torch.Tensor:
    T, d = x.shape
    ep, r = comm.world, comm.rank
    E = Wr.shape[0]
    per = E // ep
    # ---- this rank's token slice (padded so all slices have equal length for the gather)
    tl = (T + ep - 1) // ep
    lo, hi = min(T, r * tl), min(T, (r + 1) * tl)
    xs = x[lo:hi]
    n = hi - lo
    ids, wts = ops.moe_route(xs, Wr, topk) if n else (
        torch.empty(0, topk, dtype=torch.int32, device=x.device), torch.empty(0, topk, device=x.device))
    # ---- sort (token, k) pairs by destination rank
    flat_e = ids.reshape(-1).long()
    dest = flat_e // per
    order = torch.argsort(dest, stable=True)
    send_counts = torch.bincount(dest, minlength=ep)
    counts = comm.all_gather_rows(send_counts.to(torch.int64).view(1, ep))      # [ep, ep]: row s = sends of rank s
    counts_host = counts.cpu()
    sc = counts_host[r].tolist()
    rc = counts_host[:, r].tolist()
    tok_of = torch.div(order, topk, rounding_mode="floor")
    send_rows = xs[tok_of]
    send_eid = (flat_e[order] - dest[order] * per).to(torch.int32).view(-1, 1)
    recv_rows = comm.all_to_all_rows(send_rows, sc, rc)
    recv_eid = comm.all_to_all_rows(send_eid, sc, rc)
    # ---- resident experts on the received rows (top-1 "routing" with unit weights)
    R = recv_rows.shape[0]
    if R:
        ones = torch.ones(R, 1, dtype=torch.float32, device=x.device)
        y = ops.moe_experts(recv_rows.contiguous(), recv_eid.contiguous(), ones, w_gate_up, w_down, 0, per)
    else:
        y = recv_rows
    # ---- return to the senders, combine per token
    back = comm.all_to_all_rows(y, rc, sc)
    out = torch.zeros(tl, d, dtype=torch.float32, device=x.device)
    if n:
        w_sorted = wts.reshape(-1)[order]
        out.index_add_(0, tok_of, back.float() * w_sorted[:, None])
    full = comm.all_gather_rows(out.to(x.dtype))
    return full[:T].contiguous()


def moe_ffn_ep(x, Wr, w_gate_up, w_down, topk: int, comm: Comm, expert_lo: int, mode: str = "allreduce"):
    if comm is None or comm.world == 1:
        return ops.moe_ffn(x, Wr, w_gate_up, w_down, topk)
    if mode == "a2a":
        return moe_all_to_all(x, Wr, w_gate_up, w_down, topk, comm, expert_lo)
    if mode == "allreduce":
        return moe_allreduce(x, Wr, w_gate_up, w_down, topk, comm, expert_lo)
    raise ValueError(f"unknown ep mode {mode!r}")
