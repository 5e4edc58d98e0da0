"""Continuous protocol driver (SessionPool) against the wave driver, and late admission."""
import numpy as np
import pytest

from llm_consensus_amd._native import core
from llm_consensus_amd.consensus.backend import GenParams, ScriptedBackend
from llm_consensus_amd.consensus.members import Member
from llm_consensus_amd.consensus.protocol import Coordinator, SessionPool
from llm_consensus_amd.engine.backend import EngineBackend
from llm_consensus_amd.engine.engine import Engine
from llm_consensus_amd.models.config import get_preset
from llm_consensus_amd.models.transformer import TransformerModel
from llm_consensus_amd.utils.tokenizer import ByteLevelTokenizer


def _engine_members(dissent=0.0, two_models=False):
    cfg = get_preset("tiny-llama")
    bes = []
    for i in range(2 if two_models else 1):
        model = TransformerModel.random_init(cfg, "cpu", seed=3 + i, std=0.1)
        eng = Engine(model, num_kv_blocks=512, block_size=16, use_graphs=False, max_model_len=1024, name=f"m{i}")
        bes.append(EngineBackend(eng, ByteLevelTokenizer(cfg.vocab_size), stop_at_eos=False))
    gp = GenParams(temperature=0.0, max_new_tokens=5, seed=1)
    return [Member(p, bes[i % len(bes)], params=gp, decision_mode="logprob", dissent_rate=dissent, dissent_seed=i)
            for i, p in enumerate(core.default_personas())], bes


def _cfg(seed=7):
    sc = core.SessionConfig()
    sc.seed = seed
    return sc


def test_continuous_driver_gives_the_same_answers_as_waves_scripted():
    qs = [f"question {i}" for i in range(9)]

    def run(continuous):
        be = ScriptedBackend(script={"evaluate": ["NeedsRefinement\nmore", "Good\nok", "Good\nok", "Good\nok"] * 40})
        members = [Member(p, be) for p in core.default_personas()]
        return Coordinator(members, _cfg()).ask_many(qs, continuous=continuous)

    a, b = run(False), run(True)
    assert [r.question for r in b] == qs
    assert all(r.answer for r in b) and all(r.rounds <= 5 for r in b)
    assert len(a) == len(b)


def test_continuous_driver_matches_waves_on_an_engine_backend():
    qs = ["why is the sky blue?", "what is two plus two?", "name a prime number"]
    members, _ = _engine_members()
    want = Coordinator(members, _cfg()).ask_many(qs)
    members, _ = _engine_members()
    got = Coordinator(members, _cfg()).ask_many(qs, continuous=True)
    for w, g in zip(want, got):
        assert (w.answer, w.rounds, w.proposer, w.member_calls) == (g.answer, g.rounds, g.proposer, g.member_calls)


def test_sessions_join_while_others_are_in_flight_and_two_engines_overlap():
    members, bes = _engine_members(dissent=0.4, two_models=True)
    pool = SessionPool(Coordinator(members, _cfg()))
    t0 = pool.submit("first question")
    done = []
    for _ in range(3):
        done += pool.pump()
    assert pool.active == 1 and not done                 # still proposing (5 decode steps)
    t1 = pool.submit("second question, admitted late")
    t2 = pool.submit("third")
    assert pool.active == 3
    done += pool.drain()
    res = dict(done)
    assert set(res) == {t0, t1, t2} and pool.active == 0
    assert res[t1].question.startswith("second") and all(r.answer and 1 <= r.rounds <= 5 for r in res.values())
    assert all(not be._pending for be in bes)
    st = bes[0].engine.stats()
    assert st["requests.finished"] >= 3


@pytest.mark.parametrize("strategy", ["reference_protocol", "majority_vote", "self_consistency", "judge_logprob",
                                      "embedding_medoid"])
@pytest.mark.parametrize("seed", range(2))
def test_pools_with_random_cancellations_answer_the_rest_and_leave_nothing_behind(strategy, seed):
    """Every kind of pool over real (CPU) engines: questions join at random pumps, some are
    cancelled at random moments (first or second phase, waiting or decoding). A cancelled ticket
    produces no result, every other ticket exactly one, and at the end no backend holds a pending
    request, a leftover or a KV block."""
    import numpy as np
    import torch

    from llm_consensus_amd.app import build_ensemble
    from llm_consensus_amd.config import config_from_dict
    from llm_consensus_amd.consensus.protocol import SessionPool

    roles = {"judge_logprob": ["member", "member", "judge"], "embedding_medoid": ["member", "member", "embedder"]}
    members = [{"name": f"m{i}", "model": "a" if i % 2 == 0 else "b", "role": r}
               for i, r in enumerate(roles.get(strategy, ["member"] * 3))]
    cfg = config_from_dict({"strategy": strategy, "decision_mode": "logprob", "k_chains": 3, "seed": 5,
                            "sampling": {"temperature": 0.7, "max_new_tokens": 5},
                            "models": {"a": {"preset": "tiny-llama", "seed": 1, "kv_blocks": 256},
                                       "b": {"preset": "tiny-qwen", "seed": 2, "kv_blocks": 256}},
                            "members": members})
    ens = build_ensemble(cfg, torch.device("cpu"))
    pool = SessionPool(ens.strategy) if strategy == "reference_protocol" else ens.strategy.pool()
    rng = np.random.default_rng(11 + seed)
    n_q = 14
    arrive = sorted(int(x) for x in rng.integers(0, 40, size=n_q))
    tickets, cancelled, results = [], set(), {}
    for pump in range(4000):
        while arrive and arrive[0] <= pump:
            arrive.pop(0)
            tickets.append(pool.submit(f"question number {len(tickets)}: which option, A, B, C or D?"))
        live = [t for t in tickets if t not in results and t not in cancelled]
        if live and rng.integers(6) == 0:
            t = live[int(rng.integers(len(live)))]
            if pool.cancel(t):
                cancelled.add(t)
        for t, res in pool.pump():
            assert t not in results and t not in cancelled
            results[t] = res
        if not arrive and not pool.active:
            break
    assert len(tickets) == n_q and cancelled and set(results) | cancelled == set(tickets)
    assert not pool.cancel(tickets[0]) or tickets[0] not in results          # a finished ticket is not cancellable
    # cancelled member calls leave the engines at their next step
    for _ in range(200):
        if not any(b.has_leftovers() or b.engines() for b in ens.backends.values()):
            break
        pool.stepper.step()
    for name, b in ens.backends.items():
        st = b.engine.stats()
        assert not b._pending and not b.has_leftovers() and st["sched.running"] == 0, (name, st)
        assert st["kv.blocks_free"] + st["kv.blocks_cached"] == st["kv.blocks_total"], (name, st)
