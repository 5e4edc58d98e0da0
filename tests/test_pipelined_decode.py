"""Decode steps launched ahead of their predecessor's results (Engine(pipeline_decode=True)):
the host plans and launches step N + 1 while step N runs, and step N + 1 reads the tokens step N
sampled on the device. Everything a request returns must be what the one-step-at-a-time engine
returns: tokens, log-probs, decision log-probs, finish reasons, with stop tokens cutting chains
short (their speculative row in the following step is dropped) and with chains of different
lengths leaving the batch at different steps."""
import numpy as np
import pytest
import torch

from llm_consensus_amd._native import core
from llm_consensus_amd.engine.engine import Engine, run_engines
from llm_consensus_amd.models.config import get_preset
from llm_consensus_amd.models.transformer import TransformerModel


def params(n_new, seed=0, temperature=0.9, eos=None):
    p = core.SamplingParams()
    p.temperature, p.top_p, p.max_new_tokens, p.seed = temperature, 0.95, n_new, seed
    if eos is None:
        p.ignore_eos = True
    else:
        p.eos_id = eos
    return p


@pytest.fixture(scope="module")
def model():
    return TransformerModel.random_init(get_preset("tiny-llama"), "cpu", seed=11, std=0.1)


def run(model, pipeline, eos=None, n_req=5, decisions=True):
    eng = Engine(model, num_kv_blocks=256, block_size=16, use_graphs=False, pipeline_decode=pipeline)
    rids = []
    for i in range(n_req):
        prompt = list(range(3 + i, 30 + 3 * i))
        dec = [5, 9, 77] if (decisions and i % 2 == 0) else None
        rids.append(eng.submit(prompt, n=1 + (i % 3), params=params(6 + 3 * i, seed=10 * i, eos=eos), decision_ids=dec))
    eng.run_until_done()
    outs = [eng.pop_result(r) for r in rids]
    return outs, eng.stats()


def same(a, b):
    assert a.token_ids == b.token_ids
    assert a.finish_reasons == b.finish_reasons
    assert np.allclose(a.cum_logprobs, b.cum_logprobs, atol=1e-5)
    if a.decision_logprobs is None:
        assert b.decision_logprobs is None
    else:
        assert np.allclose(np.array(a.decision_logprobs), np.array(b.decision_logprobs), atol=1e-5)


def test_pipelined_decode_equals_stepwise(model):
    base, st0 = run(model, False)
    piped, st1 = run(model, True)
    for a, b in zip(base, piped):
        same(a, b)
    assert st0.get("decode.ahead_steps", 0) == 0
    assert st1["decode.ahead_steps"] >= 10
    assert st1["decode.steps"] == st0["decode.steps"]          # no extra steps without stop tokens


def test_pipelined_decode_with_stop_tokens(model):
    # pick a stop token that some chain actually samples mid-way
    base, _ = run(model, False, decisions=False)
    counts = {}
    for o in base:
        for chain in o.token_ids:
            for t in chain[1:-2]:
                counts[t] = counts.get(t, 0) + 1
    eos = max(counts, key=counts.get)
    base, st0 = run(model, False, eos=eos)
    assert any(r == "Eos" for o in base for r in o.finish_reasons)
    piped, st1 = run(model, True, eos=eos)
    for a, b in zip(base, piped):
        same(a, b)
    assert st1["decode.ahead_steps"] >= 5


def test_pipelined_decode_with_late_submission_and_several_engines(model):
    """A request submitted while a step launched ahead is in flight waits for the ordinary
    path; several engines driven together keep their own look-ahead state."""
    def go(pipeline):
        engs = [Engine(model, num_kv_blocks=128, block_size=16, use_graphs=False, pipeline_decode=pipeline,
                       name=f"e{i}") for i in range(2)]
        rids = [e.submit(list(range(5, 40)), params=params(12, seed=3 + i)) for i, e in enumerate(engs)]
        for _ in range(4):
            for e in engs:
                e.step()
        late = engs[0].submit(list(range(50, 90)), params=params(5, seed=99))
        run_engines(engs)
        return [engs[i].pop_result(r).token_ids for i, r in enumerate(rids)] + [engs[0].pop_result(late).token_ids]

    assert go(False) == go(True)


@pytest.mark.gpu
@pytest.mark.parametrize("graphs", [True, False])
def test_pipelined_decode_equals_stepwise_on_gpu(graphs):
    """Same comparison on the real kernels: recorded decode graphs read the previous step's tokens
    from the engine's device buffer; chains of different lengths change the batch bucket on the
    way, forked chains go through the shared-prefix pass, some requests carry decision targets
    (their gather runs only in the steps that can end a chain)."""
    m = TransformerModel.random_init(get_preset("small-llama"), "cuda", seed=4, std=0.05)

    def go(pipeline):
        eng = Engine(m, num_kv_blocks=512, block_size=16, use_graphs=graphs, pipeline_decode=pipeline,
                     max_model_len=512)
        rids = []
        for i in range(7):
            prompt = list(range(3 + i, 60 + 5 * i))
            dec = [5, 9, 77] if i % 2 == 0 else None
            rids.append(eng.submit(prompt, n=(8 if i == 3 else 1), params=params(6 + 4 * i, seed=10 * i),
                                   decision_ids=dec))
        eng.run_until_done()
        torch.cuda.synchronize()
        return [eng.pop_result(r) for r in rids], eng.stats()

    base, st0 = go(False)
    piped, st1 = go(True)
    for a, b in zip(base, piped):
        same(a, b)
    assert st1["decode.ahead_steps"] >= 10 and st0.get("decode.ahead_steps", 0) == 0
    assert st1["decode.steps"] == st0["decode.steps"]
