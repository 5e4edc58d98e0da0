"""A failing engine step (a HIP / RCCL error in production) fails the requests in flight, not
the process: handles carry the error, the protocol applies its failure policy, the engine
returns every KV block and serves the next batch. The reference dies on its first failed call
(`.expect()` at /root/reference/src/main.rs:85, 97, 138, 178)."""
import pytest

from llm_consensus_amd._native import core
from llm_consensus_amd.consensus.backend import BackendError, GenParams
from llm_consensus_amd.consensus.members import Member
from llm_consensus_amd.consensus.protocol import Coordinator
from llm_consensus_amd.engine.backend import EngineBackend
from llm_consensus_amd.engine.engine import Engine
from llm_consensus_amd.models.config import get_preset
from llm_consensus_amd.models.transformer import TransformerModel
from llm_consensus_amd.ops.hip import KernelLaunchError
from llm_consensus_amd.utils.tokenizer import ByteLevelTokenizer


def _backend():
    cfg = get_preset("tiny-llama")
    model = TransformerModel.random_init(cfg, "cpu", seed=4, std=0.1)
    eng = Engine(model, num_kv_blocks=256, block_size=16, use_graphs=False, max_model_len=1024)
    return EngineBackend(eng, ByteLevelTokenizer(cfg.vocab_size), stop_at_eos=False), eng, model


def _fail_on_call(model, n):
    real, calls = model.forward, [0]

    def forward(*a, **k):
        calls[0] += 1
        if calls[0] == n:
            raise KernelLaunchError("lc_gemm_panel failed: HIP error 719")
        return real(*a, **k)

    model.forward = forward
    return calls


def test_failed_step_fails_handles_frees_kv_and_engine_keeps_working():
    be, eng, model = _backend()
    total = eng.kvm.stats().blocks_total
    _fail_on_call(model, 3)                               # prefill ok, second decode step raises
    hs = [be.submit_generate(f"prompt number {i}", GenParams(temperature=0.0, max_new_tokens=6)) for i in range(3)]
    be.run()                                              # does not raise
    assert all(h.done and h.failed for h in hs)
    with pytest.raises(BackendError, match="HIP error 719"):
        hs[0].result()
    assert not eng.has_work() and eng.kvm.available_blocks() == total
    assert eng.stats()["requests.aborted"] == 3
    h = be.submit_generate("after the failure", GenParams(temperature=0.0, max_new_tokens=4))
    be.run()
    assert len(h.result().token_ids[0]) == 4


def test_protocol_finishes_sessions_when_a_member_call_dies():
    be, eng, model = _backend()
    gp = GenParams(temperature=0.0, max_new_tokens=4)
    members = [Member(p, be, params=gp, decision_mode="logprob") for p in core.default_personas()]
    _fail_on_call(model, 8)                               # somewhere inside the evaluate wave
    res = Coordinator(members).ask_many(["first question", "second question"])
    assert len(res) == 2 and all(r.answer for r in res)   # the proposals survive as the answer
    assert all(r.rounds <= 5 for r in res)
    ok = Coordinator(members).ask_many(["third question"])
    assert ok[0].answer and eng.kvm.available_blocks() == eng.kvm.stats().blocks_total


def test_continuous_driver_survives_a_failed_step():
    from llm_consensus_amd.consensus.protocol import SessionPool

    be, eng, model = _backend()
    gp = GenParams(temperature=0.0, max_new_tokens=4)
    members = [Member(p, be, params=gp, decision_mode="logprob") for p in core.default_personas()]
    _fail_on_call(model, 2)
    pool = SessionPool(Coordinator(members))
    t = pool.submit("a question")
    done = dict(pool.drain())
    assert t in done and not pool.active


def test_failure_with_several_engines_in_flight_fails_the_batch_not_the_process():
    from llm_consensus_amd.consensus.strategies import MajorityVote

    cfg = get_preset("tiny-llama")
    bes = []
    for i in range(2):
        model = TransformerModel.random_init(cfg, "cpu", seed=10 + i, std=0.1)
        eng = Engine(model, num_kv_blocks=128, block_size=16, use_graphs=False, name=f"m{i}")
        bes.append((EngineBackend(eng, ByteLevelTokenizer(cfg.vocab_size), stop_at_eos=False), eng, model))
    _fail_on_call(bes[1][2], 2)
    members = [Member(core.default_personas()[i], be, params=GenParams(temperature=0.0, max_new_tokens=3))
               for i, (be, _, _) in enumerate(bes)]
    vote = MajorityVote(members, decision_mode="logprob")
    res = vote.ask_many(["Q? A) a B) b C) c D) d", "R? A) a B) b C) c D) d"])
    assert len(res) == 2                                  # votes may be empty, the call returns
    for _, eng, _ in bes:
        assert not eng.has_work() and eng.kvm.available_blocks() == eng.kvm.stats().blocks_total
    again = vote.ask_many(["S? A) a B) b C) c D) d"])
    assert again[0].answer in list("ABCD")


def test_prompt_reused_after_a_failed_prefill_is_recomputed_not_served_from_unwritten_blocks():
    """Prompt blocks are registered in the prefix cache when their step is scheduled. If that
    step dies, the entries must not survive: the same prompt submitted again has to produce
    what a fresh engine produces."""
    prompt = "shared persona prefix " * 6 + "and the question"
    gp = GenParams(temperature=0.0, max_new_tokens=6)
    fresh, _, _ = _backend()
    want = fresh.submit_generate(prompt, gp)
    fresh.run()

    be, eng, model = _backend()
    _fail_on_call(model, 1)                               # the prefill step itself raises
    h = be.submit_generate(prompt, gp)
    be.run()
    assert h.failed and eng.kvm.stats().blocks_cached == 0
    again = be.submit_generate(prompt, gp)
    be.run()
    assert again.result().cached_tokens == 0
    assert again.result().token_ids == want.result().token_ids
