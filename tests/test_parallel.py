"""Tensor / expert parallel models against the unsharded model, and the communicator
layer over real process groups (gloo, two processes). Ranks of the in-process cases are
threads sharing a loopback hub, so they run anywhere; the contract is numerical (the
reference has no model code: /root/reference/src/main.rs:82-86 calls a remote API)."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from llm_consensus_amd._native import core
from llm_consensus_amd.engine.engine import Engine
from llm_consensus_amd.models.config import get_preset
from llm_consensus_amd.models.transformer import TransformerModel
from llm_consensus_amd.parallel.comm import run_loopback

ROOT = Path(__file__).resolve().parent.parent


def _params(n_new):
    p = core.SamplingParams()
    p.temperature, p.max_new_tokens, p.ignore_eos, p.seed = 0.0, n_new, True, 0
    return p


def _probe(model, prompts, targets):
    """Teacher-forced log-probs of ``targets`` after each prompt + a short greedy decode."""
    eng = Engine(model, num_kv_blocks=64, block_size=16, use_graphs=False)
    score = [eng.submit(p, core.RequestKind.Score, score_ids=targets) for p in prompts]
    gen = [eng.submit(p, params=_params(4)) for p in prompts]
    eng.run_until_done()
    return (np.array([eng.pop_result(r).scores for r in score]),
            [eng.pop_result(r).token_ids[0] for r in gen])


PROMPTS = [list(range(3, 40)), list(range(100, 117)), [7, 7, 7, 9, 11]]
TARGETS = [1, 65, 66, 200, 411]


@pytest.mark.parametrize("preset", ["tiny-llama", "tiny-qwen"])
def test_tensor_parallel_matches_single(preset):
    cfg = get_preset(preset)
    want, want_tok = _probe(TransformerModel.random_init(cfg, "cpu", seed=11, std=0.08), PROMPTS, TARGETS)

    def rank(comm):
        m = TransformerModel.random_init(cfg, "cpu", seed=11, std=0.08, tp_rank=comm.rank, tp_size=2, comm=comm)
        assert m.hq == cfg.num_heads // 2 and m.hkv == cfg.num_kv_heads // 2
        return _probe(m, PROMPTS, TARGETS)

    res = run_loopback(2, rank)
    for got, tok in res:
        assert np.allclose(got, want, atol=0.05), np.abs(got - want).max()
    assert res[0][1] == res[1][1], "TP ranks must sample identical tokens"
    assert np.array_equal(res[0][0], res[1][0])


@pytest.mark.parametrize("mode", ["allreduce", "a2a"])
def test_expert_parallel_matches_single(mode):
    cfg = get_preset("tiny-mixtral")
    want, _ = _probe(TransformerModel.random_init(cfg, "cpu", seed=12, std=0.08), PROMPTS, TARGETS)

    def rank(comm):
        m = TransformerModel.random_init(cfg, "cpu", seed=12, std=0.08, comm=comm, ep_size=2, ep_mode=mode)
        assert (m.expert_lo, m.expert_hi) == (2 * comm.rank, 2 * comm.rank + 2)
        assert m.layers[0].experts_gate_up.data.shape[0] == 2
        return _probe(m, PROMPTS, TARGETS)

    res = run_loopback(2, rank)
    for got, _ in res:
        assert np.allclose(got, want, atol=0.05), np.abs(got - want).max()
    assert res[0][1] == res[1][1]


def test_expert_parallel_four_ranks_uneven_token_count():
    """a2a with more ranks than some steps have tokens (decode of 3 sequences on 4 ranks)."""
    cfg = get_preset("tiny-mixtral")
    want, _ = _probe(TransformerModel.random_init(cfg, "cpu", seed=13, std=0.08), PROMPTS, TARGETS)

    def rank(comm):
        m = TransformerModel.random_init(cfg, "cpu", seed=13, std=0.08, comm=comm, ep_size=4, ep_mode="a2a")
        return _probe(m, PROMPTS, TARGETS)

    for got, _ in run_loopback(4, rank):
        assert np.allclose(got, want, atol=0.05)


def test_invalid_parallel_layouts_are_rejected():
    with pytest.raises(ValueError):
        TransformerModel(get_preset("tiny-llama"), "cpu", tp_size=4)          # 2 KV heads
    with pytest.raises(ValueError):
        TransformerModel(get_preset("tiny-llama"), "cpu", ep_size=2)          # dense model
    with pytest.raises(ValueError):
        TransformerModel(get_preset("tiny-mixtral"), "cpu", ep_size=3)        # 4 experts


def test_gloo_two_process_collectives_and_replica_bench_path(tmp_path):
    """Real process groups: the communicator's collectives, a TP=2 model and the
    replica-sharded majority-vote flow of bench.py, on two gloo ranks."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", PYTHONPATH=str(ROOT))
    out = tmp_path / "ok"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29617", str(ROOT / "tests" / "dist_worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert sorted(p.name for p in tmp_path.glob("ok.*")) == ["ok.0", "ok.1"]
