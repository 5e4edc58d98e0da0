"""Phase pipelining of the two-phase strategies: judges (or the embedder) work on one part of a
question batch while the proposers generate the next, and the answers are those of the serial
schedule."""
import dataclasses

import pytest
import torch

from llm_consensus_amd.app import build_ensemble
from llm_consensus_amd.config import ConsensusConfig, MemberConfig, ModelSpec, SamplingConfig
from llm_consensus_amd.consensus.strategies import _micro_batches
from llm_consensus_amd.utils.trace import TRACER

CPU = torch.device("cpu")
QUESTIONS = [f"Question {i}: pick one. A) a B) b C) c D) d" for i in range(6)]


def two_phase_config(strategy: str, micro_batches: int, warm_judges: bool = False) -> ConsensusConfig:
    models = {"p0": ModelSpec(name="p0", preset="tiny-llama", seed=1),
              "p1": ModelSpec(name="p1", preset="tiny-qwen", seed=2),
              "second": ModelSpec(name="second", preset="tiny-mistral", seed=3)}
    role = "judge" if strategy == "judge_logprob" else "embedder"
    members = [MemberConfig(name="p0", model="p0"), MemberConfig(name="p1", model="p1"),
               MemberConfig(name="second", model="second", role=role)]
    return ConsensusConfig(strategy=strategy, models=models, members=members, micro_batches=micro_batches,
                           min_micro_batch=1, warm_judges=warm_judges,
                           sampling=SamplingConfig(temperature=0.0, max_new_tokens=6))


def engine_spans():
    return sorted(((e["ts"], e["ts"] + e["dur"], e["name"].split(".")[0]) for e in TRACER.events
                   if e.get("cat") == "engine"), key=lambda s: s[0])


def run_traced(strategy: str, micro_batches: int, warm_judges: bool = False):
    ens = build_ensemble(two_phase_config(strategy, micro_batches, warm_judges), CPU)
    TRACER.clear()
    TRACER.enable(roctx=False)
    try:
        res = ens.ask_many(QUESTIONS)
        spans = engine_spans()
    finally:
        TRACER.disable()
        TRACER.clear()
    return res, spans


def test_micro_batches_cover_every_question_once():
    assert [list(r) for r in _micro_batches(5, 2)] == [[0, 1, 2], [3, 4]]
    assert [list(r) for r in _micro_batches(2, 4)] == [[0], [1]]
    assert _micro_batches(0, 2) == []


@pytest.mark.parametrize("strategy", ["judge_logprob", "embedding_medoid"])
def test_pipelined_schedule_overlaps_phases_and_keeps_the_answers(strategy):
    serial, s_spans = run_traced(strategy, 1)
    piped, p_spans = run_traced(strategy, 3)
    assert [r.answer for r in piped] == [r.answer for r in serial]
    assert [r.proposer for r in piped] == [r.proposer for r in serial]
    for a, b in zip(piped, serial):
        assert a.usage == b.usage

    def interleaved(spans):
        """The second-stage engine has work between two steps of a proposer engine."""
        first_second = min(t0 for t0, _, n in spans if n == "second")
        last_proposer = max(t1 for _, t1, n in spans if n in ("p0", "p1"))
        return first_second < last_proposer

    assert not interleaved(s_spans)            # serial: every proposal before the first score
    assert interleaved(p_spans)
    # One run_backends call drives both stages: within the window in which the second-stage
    # engine works on part k, the proposer engines step too.
    window = [(t0, t1) for t0, t1, n in p_spans if n == "second"]
    lo, hi = window[0][0], window[-1][1]
    assert sum(1 for t0, _, n in p_spans if n in ("p0", "p1") and lo < t0 < hi) >= 2


def test_judges_prefill_the_question_while_the_proposers_generate():
    """warm_judges: a scoring call on the leading part of the judge prompt (the question) is queued
    with the proposals, so the judge engine works during the proposer phase even in the serial
    schedule, the later scoring calls find the question's KV blocks in the judge's prefix cache,
    and the answers do not change."""
    long_q = [q + " " + "context " * 40 for q in QUESTIONS]           # several KV blocks per question

    def run(warm):
        ens = build_ensemble(two_phase_config("judge_logprob", 1, warm), CPU)
        TRACER.clear()
        TRACER.enable(roctx=False)
        try:
            res = ens.ask_many(long_q)
            spans = engine_spans()
        finally:
            TRACER.disable()
            TRACER.clear()
        return res, spans

    cold, c_spans = run(False)
    warm, w_spans = run(True)
    assert [r.answer for r in warm] == [r.answer for r in cold]
    assert [r.proposer for r in warm] == [r.proposer for r in cold]
    for a, b in zip(warm, cold):
        assert [x for v in a.votes for x in v["scores"]] == pytest.approx([x for v in b.votes for x in v["scores"]],
                                                                           abs=1e-3)

    def judge_before_last_proposal(spans):
        return min(t0 for t0, _, n in spans if n == "second") < max(t1 for _, t1, n in spans if n in ("p0", "p1"))

    assert judge_before_last_proposal(w_spans) and not judge_before_last_proposal(c_spans)
    # the candidates' scoring calls reuse the question's blocks: more cached prompt tokens than without
    assert sum(r.usage["cached_tokens"] for r in warm) > sum(r.usage["cached_tokens"] for r in cold)


def test_small_batches_are_not_cut():
    from llm_consensus_amd.consensus.strategies import MIN_MICRO_BATCH, _two_stage_pipeline

    seen = []
    _two_stage_pipeline(MIN_MICRO_BATCH + 5, 4, lambda rng: seen.append(("p", len(rng))), [],
                        lambda rng: seen.append(("j", len(rng))), [])
    assert seen == [("p", MIN_MICRO_BATCH + 5), ("j", MIN_MICRO_BATCH + 5)]
    seen.clear()
    _two_stage_pipeline(4 * MIN_MICRO_BATCH, 2, lambda rng: seen.append(("p", len(rng))), [],
                        lambda rng: seen.append(("j", len(rng))), [])
    assert [s for s in seen if s[0] == "p"] == [("p", 2 * MIN_MICRO_BATCH)] * 2


def vote_config() -> ConsensusConfig:
    models = {f"m{i}": ModelSpec(name=f"m{i}", preset=p, seed=i + 1)
              for i, p in enumerate(("tiny-llama", "tiny-qwen", "tiny-mistral"))}
    members = [MemberConfig(name=n, model=n) for n in models]
    return ConsensusConfig(strategy="majority_vote", decision_mode="logprob", models=models, members=members,
                           sampling=SamplingConfig(temperature=0.0, max_new_tokens=6))


def test_vote_pool_answers_match_batch_and_admits_while_running():
    """Questions that arrive while others are being answered join the running engines; every
    question gets the answer the batch form gives it."""
    ens = build_ensemble(vote_config(), CPU)
    batch = ens.ask_many(QUESTIONS)
    pool = ens.strategy.pool()
    tickets = {pool.submit(q): q for q in QUESTIONS[:2]}
    got, late, pumps_before_late = {}, list(QUESTIONS[2:]), 0
    while pool.active or late:
        for t, res in pool.pump():
            got[tickets[t]] = res
        pumps_before_late += 1
        if late and pumps_before_late >= 3:          # the first two are mid-generation by now
            assert pool.active
            q = late.pop(0)
            tickets[pool.submit(q)] = q
    assert [got[q].answer for q in QUESTIONS] == [r.answer for r in batch]
    assert [[v["class"] for v in got[q].votes] for q in QUESTIONS] == [[v["class"] for v in r.votes] for r in batch]
    assert all(got[q].latency_s > 0 for q in QUESTIONS)
    for k in ("prompt_tokens", "completion_tokens"):          # the second pass hits the prefix cache
        assert got[QUESTIONS[0]].usage[k] == batch[0].usage[k]
    assert got[QUESTIONS[0]].usage["cached_tokens"] > 0


def test_self_consistency_pool_matches_batch():
    models = {"m": ModelSpec(name="m", preset="tiny-llama", seed=4)}
    cfg = ConsensusConfig(strategy="self_consistency", decision_mode="logprob", k_chains=4, models=models,
                          members=[MemberConfig(name="m", model="m")],
                          sampling=SamplingConfig(temperature=0.9, top_p=0.95, max_new_tokens=5, seed=3))
    ens = build_ensemble(cfg, CPU)
    batch = ens.ask_many(QUESTIONS[:3])
    pool = ens.strategy.pool()
    tickets = {pool.submit(q): i for i, q in enumerate(QUESTIONS[:3])}
    got = {}
    while pool.active:
        for t, res in pool.pump():
            got[tickets[t]] = res
    for i in range(3):
        assert got[i].answer == batch[i].answer
        assert [v["class"] for v in got[i].votes] == [v["class"] for v in batch[i].votes]
        assert got[i].extra["chains"] == batch[i].extra["chains"]


@pytest.mark.parametrize("strategy", ["judge_logprob", "embedding_medoid"])
def test_two_phase_strategies_without_batch_boundaries(strategy):
    """PhasedPool: a question moves to its second phase the moment ITS proposals are in and is
    answered when its own second phase is done; questions admitted later join the engines'
    running batches. Answers equal those of the batch driver."""
    ens = build_ensemble(two_phase_config(strategy, 1), CPU)
    want = {r.question: (r.answer, r.proposer) for r in ens.ask_many(QUESTIONS)}
    pool = ens.strategy.pool()
    tickets = {pool.submit(q): q for q in QUESTIONS[:2]}
    got, pumps, late = {}, 0, list(QUESTIONS[2:])
    while pool.active or late:
        pumps += 1
        if late and pumps % 3 == 0:                       # arrivals while others are in flight
            q = late.pop(0)
            tickets[pool.submit(q)] = q
        for t, r in pool.pump():
            got[tickets[t]] = (r.answer, r.proposer)
    assert got == want
    assert pool.drain() == []


def test_embedding_medoid_drops_candidates_whose_embedding_failed():
    """A failed embedding call removes that candidate from the vote (batch driver and pool alike);
    only a question none of whose candidates could be embedded comes back empty, with the reason."""
    from llm_consensus_amd._native import core
    from llm_consensus_amd.consensus.backend import GenParams, ScriptedBackend
    from llm_consensus_amd.consensus.members import Member
    from llm_consensus_amd.consensus.strategies import EmbeddingMedoid

    P = core.default_personas()

    def strategy(embed_fail_every):
        gen = ScriptedBackend({"propose": lambda p: "candidate text"}, name="gen")
        emb = ScriptedBackend(fail_every=embed_fail_every, name="emb")
        props = [Member(P[i], gen, params=GenParams(max_new_tokens=4)) for i in range(3)]
        return EmbeddingMedoid(props, Member(P[3], emb, params=GenParams(max_new_tokens=4), role="embedder"))

    for run in (lambda s, qs: s.ask_many(qs), lambda s, qs: [r for _, r in sorted(_pool_drain(s, qs))]):
        res = run(strategy(2), ["q one", "q two"])                  # every second embedding fails
        assert [r.answer for r in res] == ["candidate text"] * 2
        assert all(1 <= len(r.votes) < 3 for r in res)
        res = run(strategy(1), ["q one"])                            # all of them fail
        assert res[0].answer == "" and "embedded" in res[0].extra["error"]


def _pool_drain(strategy, questions):
    pool = strategy.pool()
    for q in questions:
        pool.submit(q)
    return pool.drain()
