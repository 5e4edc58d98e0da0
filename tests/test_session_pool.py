"""Continuous protocol driver (SessionPool) against the wave driver, and late admission."""
import numpy as np

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
