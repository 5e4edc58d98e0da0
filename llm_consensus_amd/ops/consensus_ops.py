"""Device execution of the consensus reducers (csrc/kernels/consensus.hip)."""
from __future__ import annotations

import numpy as np
import torch

from .._native import core
from . import hip as h


_TIE_CODE = {core.TieBreak.LowestClass: 0, core.TieBreak.FirstVoter: 1, core.TieBreak.HighestWeight: 2}


def majority_vote_device(votes: np.ndarray, num_classes: int, tie, device, weights=None):
    from ..consensus.reducers import VoteBatch, majority_vote_batch

    if num_classes > 64:
        return majority_vote_batch(votes, num_classes, tie, device=None, weights=weights)  # kernel: one lane per class
    R, N = votes.shape
    v = torch.from_numpy(votes).to(device)
    w = torch.from_numpy(np.ascontiguousarray(weights, dtype=np.float32)).to(device) if weights is not None else None
    winner = torch.empty(R, dtype=torch.int32, device=device)
    count = torch.empty_like(winner)
    valid = torch.empty_like(winner)
    unan = torch.empty_like(winner)
    agree = torch.empty(R, dtype=torch.float32, device=device)
    h.call("lc_majority_vote", v.data_ptr(), R, N, num_classes, _TIE_CODE[tie],
           winner.data_ptr(), count.data_ptr(), valid.data_ptr(), agree.data_ptr(), unan.data_ptr(), h.ptr(w))
    return VoteBatch(winner.cpu().numpy(), count.cpu().numpy(), valid.cpu().numpy(), agree.cpu().numpy(),
                     unan.cpu().numpy().astype(bool))


def majority_vote_tensors(votes: torch.Tensor, num_classes: int, tie_first_voter: bool = False,
                          weights: "torch.Tensor | None" = None):
    """Device-resident variant (votes int32 [R, N] already on the GPU, weights float32 [N] or
    None); returns device tensors (winner, count, valid, agreement, unanimous) without a host
    round trip."""
    R, N = votes.shape
    dev = votes.device
    winner = torch.empty(R, dtype=torch.int32, device=dev)
    count = torch.empty_like(winner)
    valid = torch.empty_like(winner)
    unan = torch.empty_like(winner)
    agree = torch.empty(R, dtype=torch.float32, device=dev)
    h.call("lc_majority_vote", votes.data_ptr(), R, N, num_classes, 1 if tie_first_voter else 0, winner.data_ptr(),
           count.data_ptr(), valid.data_ptr(), agree.data_ptr(), unan.data_ptr(), h.ptr(weights))
    return winner, count, valid, agree, unan


class _Medoid:
    __slots__ = ("index", "score", "mean_pairwise")

    def __init__(self, index, score, mean_pairwise):
        self.index, self.score, self.mean_pairwise = index, score, mean_pairwise


def cosine_medoid_device(E: np.ndarray, device):
    n, d = E.shape
    if n > 64:
        return core.cosine_medoid(E)
    e = torch.from_numpy(E).to(device)
    idx = torch.empty(1, dtype=torch.int32, device=device)
    score = torch.empty(1, dtype=torch.float32, device=device)
    mean = torch.empty(1, dtype=torch.float32, device=device)
    h.call("lc_cosine_medoid", e.data_ptr(), 1, n, d, n * d, idx.data_ptr(), score.data_ptr(), mean.data_ptr())
    return _Medoid(int(idx.item()), float(score.item()), float(mean.item()))


def judge_reduce_device(scores: np.ndarray, device):
    from ..consensus.reducers import JudgeBatch

    R, C, J = scores.shape
    s = torch.from_numpy(scores).to(device)
    winner = torch.empty(R, dtype=torch.int32, device=device)
    score = torch.empty(R, dtype=torch.float32, device=device)
    unan = torch.empty(R, dtype=torch.int32, device=device)
    h.call("lc_judge_reduce", s.data_ptr(), R, C, J, winner.data_ptr(), score.data_ptr(), unan.data_ptr())
    return JudgeBatch(winner.cpu().numpy(), score.cpu().numpy(), unan.cpu().numpy().astype(bool))


def answer_extract_device(ids: torch.Tensor, lens: torch.Tensor, marker, choice_ids, window: int = 4) -> torch.Tensor:
    """Class id per generated sequence from token ids already on the GPU (int32 [R, Tmax],
    lens int32 [R]); see answer_extract_kernel. Returns int32 [R] on the device."""
    R, ld = ids.shape
    dev = ids.device
    mk = torch.tensor(list(marker) or [0], dtype=torch.int32, device=dev)
    ch = torch.tensor(list(choice_ids), dtype=torch.int32, device=dev)
    out = torch.empty(R, dtype=torch.int32, device=dev)
    h.call("lc_answer_extract", ids.data_ptr(), lens.data_ptr(), R, ld, mk.data_ptr(), len(marker), ch.data_ptr(),
           len(choice_ids), window, out.data_ptr())
    return out
