"""Request-level data parallelism over ensemble replicas.

Every rank (one process per GPU) hosts a full ensemble replica -- three 7-8B models are
about 46 GB of the 288 GB of one MI355X -- so questions are simply sharded over ranks and
only the results cross the fabric: one small all-gather of class ids per batch. This is
the scaling axis the reference lacks entirely (its coordinator serves one question at a
time, /root/reference/src/main.rs:210-247 overwrites its single session state per
``AskQuestion``).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch

from .comm import Comm

PAD = -2  # filler for ragged shards; never a valid class id (-1 means "no valid vote")


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous near-equal split of ``n`` items: the first ``n % world`` ranks get one more."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def winners_tensor(results, choices: str, device=None) -> torch.Tensor:
    return torch.tensor([(choices.find(r.answer) if r.answer else -1) for r in results], dtype=torch.int32,
                        device=device)


def gather_winners(comm: Comm, results, choices: str, device=None) -> torch.Tensor:
    """All ranks' winning class ids, rank-major. Every rank must pass the same number of results."""
    return comm.all_gather_rows(winners_tensor(results, choices, device))


def ask_sharded(strategy, comm: Comm, questions: Sequence[str], choices: str, device=None) -> Tuple[List[str], list]:
    """Answers a global question list: each rank runs its shard through ``strategy`` and the
    winners are all-gathered. Returns (answers for ALL questions in order, this rank's
    ConsensusResults)."""
    n, world = len(questions), comm.world
    lo, hi = shard_bounds(n, comm.rank, world)
    local = strategy.ask_many(list(questions[lo:hi])) if hi > lo else []
    width = (n + world - 1) // world
    w = torch.full((width,), PAD, dtype=torch.int32, device=device)
    if local:
        w[: hi - lo] = winners_tensor(local, choices, device)
    allw = comm.all_gather_rows(w).view(world, width).cpu()
    answers: List[str] = []
    for r in range(world):
        rlo, rhi = shard_bounds(n, r, world)
        answers += [choices[c] if c >= 0 else "" for c in allw[r, : rhi - rlo].tolist()]
    return answers, local
