"""Prefix sharing between prompts that are admitted and prefilled TOGETHER.

The evaluate prompts of one protocol round put the question and the answer first and the
persona text after it (/root/reference/src/main.rs:111-136), so the N personas that share one
model submit N prompts with a common head in the same wave. The prefix cache must serve the
later prompts of a step from the blocks of the earlier ones, and results must not change."""
import numpy as np
import pytest

from llm_consensus_amd._native import core
from llm_consensus_amd.engine.engine import Engine
from llm_consensus_amd.models.config import get_preset
from llm_consensus_amd.models.transformer import TransformerModel


def toks(n, start=0):
    return np.arange(start, start + n, dtype=np.int32)


def make_sched(blocks=256, bs=4, max_batch=8, max_prefill=64, cache=True):
    kv = core.KVManager(blocks, bs, cache)
    cfg = core.SchedulerConfig()
    cfg.max_batch_seqs, cfg.max_prefill_tokens = max_batch, max_prefill
    return core.Scheduler(kv, cfg), kv


def test_extend_match_swaps_unwritten_blocks_for_cached_ones():
    kv = core.KVManager(32, 4, True)
    a, hit_a = kv.create_sequence(toks(10))
    b, hit_b = kv.create_sequence(np.concatenate([toks(8), toks(3, 50)]))
    assert hit_a == 0 and hit_b == 0
    free0 = kv.available_blocks()
    kv.commit(a, toks(10))                           # registers a's two full blocks
    assert kv.extend_match(b, np.concatenate([toks(8), toks(3, 50)])) == 8
    assert kv.block_table(b)[:2] == kv.block_table(a)[:2] and kv.block_table(b)[2] != kv.block_table(a)[2]
    assert kv.num_tokens(b) == 8
    assert kv.available_blocks() == free0 + 2        # b's two unwritten blocks went back
    assert kv.extend_match(b, np.concatenate([toks(8), toks(3, 50)])) == 0
    kv.commit(b, toks(3, 50))
    kv.free_sequence(a); kv.free_sequence(b)
    assert kv.available_blocks() == 32
    # A prompt equal to a cached one keeps its last token to run.
    c, hit_c = kv.create_sequence(toks(8))
    assert hit_c == 4


def test_prompts_admitted_together_share_their_common_head():
    sched, kv = make_sched()
    p = core.SamplingParams(); p.max_new_tokens = 2
    head = toks(16)
    prompts = [np.concatenate([head, toks(3 + i, 100 * (i + 1))]) for i in range(3)]
    ids = [sched.add_request(core.RequestKind.Generate, pr, 1, p) for pr in prompts]
    plan = sched.schedule()
    assert plan.kind == core.StepPlan.Kind.Prefill and len(plan.seq_ids) == 3
    lens = np.diff(plan.cu_q)
    assert lens.tolist() == [19, 4, 5]               # the 2nd and 3rd prompt only run their own tails
    assert plan.positions[plan.cu_q[1]] == 16 and plan.positions[plan.cu_q[2]] == 16
    bt = plan.block_tables
    assert (bt[1, :4] == bt[0, :4]).all() and (bt[2, :4] == bt[0, :4]).all() and bt[1, 4] != bt[0, 4]
    assert plan.context_lens.tolist() == [19, 20, 21]
    slots = sched.prefill_sample_layout(plan)
    sched.complete_prefill(plan, np.zeros(len(slots), np.int32), np.zeros(len(slots), np.float32))
    while sched.has_work():
        pl = sched.schedule()
        sched.complete_decode(pl, np.ones(len(pl.seq_ids), np.int32), np.zeros(len(pl.seq_ids), np.float32))
    fin = {f.request_id: f for f in sched.take_finished()}
    assert [fin[i].cached_tokens for i in ids] == [0, 16, 16]
    assert kv.stats().prefix_hit_tokens == 32 and kv.available_blocks() == 256


def test_sharing_stops_where_the_earlier_prompt_ran_out_of_step_budget():
    sched, kv = make_sched(max_prefill=10)
    head = toks(16)
    a = sched.add_request(core.RequestKind.Score, np.concatenate([head, toks(2, 100)]))
    b = sched.add_request(core.RequestKind.Score, np.concatenate([head, toks(2, 200)]))
    plan = sched.schedule()                          # only 10 tokens of the first prompt fit
    assert np.diff(plan.cu_q).tolist() == [10]
    sched.complete_prefill(plan, np.zeros(0, np.int32), np.zeros(0, np.float32))
    plan = sched.schedule()                          # rest of a (8), then b: its head is cached by now
    assert np.diff(plan.cu_q).tolist() == [8, 2] and plan.positions[plan.cu_q[1]] == 16
    sched.complete_prefill(plan, np.zeros(0, np.int32), np.zeros(0, np.float32))
    assert {f.request_id for f in sched.take_finished()} == {a, b} and not sched.has_work()


def test_engine_results_equal_with_and_without_in_step_sharing():
    cfg = get_preset("tiny-llama")
    model = TransformerModel.random_init(cfg, "cpu", seed=11, std=0.1)
    head = list(range(5, 5 + 40))
    prompts = [head + [300 + 7 * i + j for j in range(6 + i)] for i in range(4)]
    sp = core.SamplingParams(); sp.max_new_tokens = 6; sp.temperature = 0.0
    dec = [3, 9]

    def run(cache):
        eng = Engine(model, num_kv_blocks=128, block_size=16, use_graphs=False, enable_prefix_cache=cache)
        rids = [eng.submit(p, params=sp, decision_ids=dec) for p in prompts]
        sid = eng.submit(prompts[1][:-1] + [77], core.RequestKind.Score, score_ids=dec)
        eng.run_until_done()
        outs = [eng.pop_result(r) for r in rids]
        return outs, eng.pop_result(sid), eng.stats()

    shared, s_shared, st = run(True)
    cold, s_cold, st_cold = run(False)
    assert [o.cached_tokens for o in shared] == [0, 32, 32, 32] and st["kv.prefix_hit_rate"] > 0.4
    assert st_cold["kv.prefix_hit_tokens"] == 0
    for a, b in zip(shared, cold):
        assert a.token_ids == b.token_ids
        assert np.allclose(a.decision_logprobs, b.decision_logprobs, atol=2e-4)
    assert np.allclose(s_shared.scores, s_cold.scores, atol=2e-4)


@pytest.mark.gpu
def test_in_step_sharing_on_gpu_matches_cold_prefill():
    """On the GPU the shared blocks are written by the fused RoPE + KV-write kernel and read by
    the prefill attention kernel of the SAME step, for a different sequence."""
    cfg = get_preset("small-llama")
    model = TransformerModel.random_init(cfg, "cuda", seed=13, std=0.05)
    head = list(range(5, 5 + 200))
    prompts = [head + [300 + 7 * i + j for j in range(20 + 3 * i)] for i in range(6)]
    sp = core.SamplingParams(); sp.max_new_tokens = 6; sp.temperature = 0.0
    dec = [3, 9, 11]

    def run(cache):
        eng = Engine(model, num_kv_blocks=256, block_size=64, enable_prefix_cache=cache)
        rids = [eng.submit(p, params=sp, decision_ids=dec) for p in prompts]
        eng.run_until_done()
        return [eng.pop_result(r) for r in rids], eng.stats()

    shared, st = run(True)
    cold, st_cold = run(False)
    assert [o.cached_tokens for o in shared] == [0] + [192] * 5 and st_cold["kv.prefix_hit_tokens"] == 0
    for a, b in zip(shared, cold):
        assert a.token_ids == b.token_ids
        assert np.allclose(a.decision_logprobs, b.decision_logprobs, atol=5e-2)
