"""RCCL path at world size 1 on the GPU box: the production communicator (torch.distributed
backend "nccl" = RCCL on ROCm) runs every collective the framework uses, on device tensors,
and the replica-sharded answer path of bench.py works through it. Multi-GPU worlds are the
driver's to launch; the sharding arithmetic itself is covered by the loopback and gloo tests."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def nccl_world():
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=dev)
    yield dist, dev
    dist.destroy_process_group()


def test_rccl_collectives_world1(nccl_world):
    from llm_consensus_amd.parallel.comm import SingleComm, TorchDistComm, make_comm

    dist, dev = nccl_world
    assert isinstance(make_comm(dist, dev), SingleComm)          # world 1 needs no collectives ...
    comm = TorchDistComm(dist, dev)                               # ... but the RCCL path must still work
    t = torch.arange(8, dtype=torch.bfloat16, device=dev)
    assert torch.equal(comm.all_reduce(t.clone()), t)
    assert torch.equal(comm.all_gather_rows(t.view(2, 4)), t.view(2, 4))
    assert torch.equal(comm.all_gather_cols(t.view(2, 4).float(), 3), t.view(2, 4).float()[:, :3])
    assert torch.equal(comm.all_to_all_rows(t.view(4, 2), [4], [4]), t.view(4, 2))
    assert torch.equal(comm.broadcast(t.clone(), 0), t)
    assert comm.all_true(True) and not comm.all_true(False)
    assert comm.max_float(3.5) == 3.5 and comm.gather_floats([1.0, 2.0]) == [1.0, 2.0]
    assert comm.broadcast_object({"a": 1}, 0) == {"a": 1} and comm.all_gather_object("x") == ["x"]
    comm.barrier()
    sub = comm.subgroup([0])
    assert isinstance(sub, SingleComm)


def test_replica_answer_gather_through_rccl(nccl_world):
    from llm_consensus_amd.app import build_ensemble
    from llm_consensus_amd.config import config_from_dict
    from llm_consensus_amd.parallel.comm import TorchDistComm
    from llm_consensus_amd.parallel.replica import ask_sharded, gather_winners

    dist, dev = nccl_world
    cfg = config_from_dict({"strategy": "majority_vote", "decision_mode": "logprob",
                            "sampling": {"temperature": 0.0, "max_new_tokens": 4},
                            "models": {"m0": {"preset": "tiny-llama", "seed": 1}, "m1": {"preset": "tiny-qwen", "seed": 2}},
                            "members": [{"name": "a", "model": "m0"}, {"name": "b", "model": "m1"}]})
    ens = build_ensemble(cfg, dev)
    comm = TorchDistComm(dist, dev)
    qs = [f"Q{i}? A) a B) b C) c D) d" for i in range(3)]
    answers, local = ask_sharded(ens.strategy, comm, qs, "ABCD", dev)
    assert answers == [r.answer for r in local] and all(a in "ABCD" for a in answers)
    w = gather_winners(comm, local, "ABCD", dev)
    assert w.is_cuda and w.tolist() == ["ABCD".index(a) for a in answers]



def test_piecewise_decode_graphs_with_rccl_between_them(nccl_world, monkeypatch):
    """One rank of a two-way tensor-parallel model whose collectives are real RCCL calls
    (world size 1, so they are identities on half-width tensors): the graphs cut at the
    collectives must replay correctly with RCCL work enqueued between them."""
    from llm_consensus_amd._native import core
    from llm_consensus_amd.engine.engine import Engine
    from llm_consensus_amd.models.config import get_preset
    from llm_consensus_amd.models.transformer import TransformerModel
    from llm_consensus_amd.parallel.comm import TorchDistComm

    dist, dev = nccl_world
    cfg = get_preset("tiny-llama")

    def run(piecewise):
        monkeypatch.setenv("CONSENSUS_PIECEWISE", piecewise or "0")
        m = TransformerModel.random_init(cfg, dev, seed=5, std=0.05, tp_rank=0, tp_size=2, comm=TorchDistComm(dist, dev))
        eng = Engine(m, num_kv_blocks=64, block_size=16)
        assert eng.piecewise == piecewise
        p = core.SamplingParams()
        p.temperature, p.max_new_tokens, p.ignore_eos = 0.0, 6, True
        rids = [eng.submit(list(range(2, 30 + 3 * i)), params=p) for i in range(3)]
        eng.run_until_done()
        return [eng.pop_result(r).token_ids[0] for r in rids]

    want = run("")
    assert run("graphs") == want and run("lists") == want
