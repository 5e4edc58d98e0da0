"""Tensor-parallel decode as a chain of hipGraphs cut at the collectives, against the same
engine run eagerly.

One rank of a two-way tensor-parallel model runs alone here, with a stand-in communicator
whose collectives are deterministic functions of the local tensor (the partner is assumed to
hold identical values). That does not reproduce the unsharded model -- the loopback and gloo
tests do that -- but it makes eager and piecewise execution directly comparable in one
thread: every collective has to be issued, in order, on the right static buffer, between the
right two graphs, for the tokens and log-probs to agree."""
import numpy as np
import pytest
import torch

from llm_consensus_amd._native import core
from llm_consensus_amd.engine.engine import Engine
from llm_consensus_amd.models.config import get_preset
from llm_consensus_amd.models.transformer import TransformerModel
from llm_consensus_amd.parallel.comm import Comm

pytestmark = pytest.mark.gpu

PROMPTS = [list(range(3, 40)), list(range(100, 117)), [7, 7, 7, 9, 11]]


class MirrorComm(Comm):
    """Rank 0 of a world of two whose partner always holds the same tensor."""
    rank, world = 0, 2

    def __init__(self):
        self.calls = 0

    def all_reduce(self, t):
        self.calls += 1
        t.mul_(2.0)
        return t

    def all_gather_cols(self, t, total_cols=None):
        self.calls += 1
        out = torch.cat([t, t * 0.5], dim=-1)
        return out[..., :total_cols] if total_cols is not None else out


def _run(piecewise: str, monkeypatch):
    monkeypatch.setenv("CONSENSUS_PIECEWISE", piecewise or "0")
    cfg = get_preset("tiny-llama")
    comm = MirrorComm()
    m = TransformerModel.random_init(cfg, "cuda", seed=21, std=0.05, tp_rank=0, tp_size=2, comm=comm)
    eng = Engine(m, num_kv_blocks=64, block_size=16)
    assert eng.piecewise == piecewise and not eng.use_graphs
    p = core.SamplingParams()
    p.temperature, p.max_new_tokens, p.ignore_eos, p.seed = 0.0, 8, True, 0
    rids = [eng.submit(pr, params=p, decision_ids=[3, 9]) for pr in PROMPTS]
    eng.run_until_done()
    outs = [eng.pop_result(r) for r in rids]
    # one chain of pieces per recorded form of the step (with / without the decision gather)
    pieces = [chain for g in eng._graphs.values() for chain in g.pieces.values()]
    return [o.token_ids[0] for o in outs], [o.decision_logprobs for o in outs], pieces, comm.calls


@pytest.mark.parametrize("mode", ["graphs", "lists"])
def test_piecewise_decode_matches_eager_tensor_parallel_decode(mode, monkeypatch):
    tok_e, dec_e, pieces_e, calls_e = _run("", monkeypatch)
    tok_p, dec_p, pieces_p, calls_p = _run(mode, monkeypatch)
    cfg = get_preset("tiny-llama")
    assert not pieces_e and len(pieces_p) >= 1
    chain = pieces_p[0]
    n_coll = sum(1 for g, c in chain if c is not None)
    assert n_coll == 2 * cfg.num_layers + 1                      # two all-reduces per layer + the logits gather
    assert sum(1 for g, c in chain if g is not None) == n_coll + 1   # compute pieces and collectives alternate
    assert all((chain[i][0] is None) == (i % 2 == 1) for i in range(len(chain)))
    assert tok_p == tok_e
    for a, b in zip(dec_p, dec_e):
        assert np.allclose(np.asarray(a), np.asarray(b), atol=2e-3)
    assert calls_p >= calls_e                                    # warm-up and capture add calls, none is skipped
