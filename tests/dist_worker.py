"""Worker of tests/test_parallel.py::test_gloo_two_process_collectives_and_replica_bench_path.
Launched by torch.distributed.run with two ranks on the gloo backend (CPU); each rank
writes ``<argv[1]>.<rank>`` when every check passed."""
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from llm_consensus_amd._native import core  # noqa: E402
from llm_consensus_amd.app import build_ensemble  # noqa: E402
from llm_consensus_amd.config import load_config  # noqa: E402
from llm_consensus_amd.engine.engine import Engine  # noqa: E402
from llm_consensus_amd.models.config import get_preset  # noqa: E402
from llm_consensus_amd.models.transformer import TransformerModel  # noqa: E402
from llm_consensus_amd.parallel.comm import TorchDistComm, make_comm  # noqa: E402
from llm_consensus_amd.parallel.replica import ask_sharded, gather_winners, shard_bounds  # noqa: E402


def score(model, prompt, targets):
    eng = Engine(model, num_kv_blocks=32, block_size=16, use_graphs=False)
    rid = eng.submit(prompt, core.RequestKind.Score, score_ids=targets)
    eng.run_until_done()
    return np.array(eng.pop_result(rid).scores)


def main() -> None:
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    comm = make_comm(dist, torch.device("cpu"))
    assert isinstance(comm, TorchDistComm) and comm.world == world == 2

    # ---- collectives
    t = torch.full((4,), float(rank + 1))
    assert comm.all_reduce(t).tolist() == [3.0] * 4
    cols = comm.all_gather_cols(torch.full((2, 3), float(rank)), total_cols=5)
    assert cols.shape == (2, 5) and cols[0].tolist() == [0, 0, 0, 1, 1]
    rows = comm.all_gather_rows(torch.tensor([[rank, rank]], dtype=torch.int32))
    assert rows.tolist() == [[0, 0], [1, 1]]
    # rank 0 sends 1 row to itself and 2 to rank 1; rank 1 sends 3 to rank 0 and 1 to itself
    send = [[1, 2], [3, 1]][rank]
    recv = [[1, 3], [2, 1]][rank]
    payload = torch.arange(sum(send), dtype=torch.float32).view(-1, 1) + 10 * rank
    got = comm.all_to_all_rows(payload, send, recv)
    want = {0: [0, 10, 11, 12], 1: [1, 2, 13]}[rank]
    assert got.view(-1).tolist() == want, got
    assert comm.all_true(True) and not comm.all_true(rank == 0)
    assert comm.max_float(float(rank)) == 1.0
    assert comm.gather_floats([rank + 0.5]) == [0.5, 1.5]
    b = torch.tensor([float(rank)])
    assert comm.broadcast(b, src=1).item() == 1.0

    # ---- TP=2 over a real process group equals the unsharded model
    cfg = get_preset("tiny-llama")
    prompt, targets = list(range(5, 33)), [3, 65, 300]
    want_lp = score(TransformerModel.random_init(cfg, "cpu", seed=21, std=0.08), prompt, targets)
    tp = TransformerModel.random_init(cfg, "cpu", seed=21, std=0.08, tp_rank=rank, tp_size=2, comm=comm)
    got_lp = score(tp, prompt, targets)
    assert np.allclose(got_lp, want_lp, atol=0.05), (got_lp, want_lp)

    # ---- EP=2 (all-to-all exchange) equals the unsharded model
    mcfg = get_preset("tiny-mixtral")
    want_lp = score(TransformerModel.random_init(mcfg, "cpu", seed=22, std=0.08), prompt, targets)
    ep = TransformerModel.random_init(mcfg, "cpu", seed=22, std=0.08, comm=comm, ep_size=2, ep_mode="a2a")
    assert np.allclose(score(ep, prompt, targets), want_lp, atol=0.05)

    # ---- replica data parallelism: the flow bench.py times
    cfg_path = Path(__file__).resolve().parent.parent / "configs" / "plumbing_tiny.toml"
    ens = build_ensemble(load_config(str(cfg_path)), torch.device("cpu"))
    questions = [f"Question {i}? A) a B) b C) c D) d" for i in range(5)]     # ragged: 3 + 2
    answers, local = ask_sharded(ens.strategy, comm, questions, "ABCD")
    lo, hi = shard_bounds(5, rank, 2)
    assert len(local) == hi - lo and len(answers) == 5
    assert [r.answer for r in local] == answers[lo:hi]
    solo = [r.answer for r in ens.ask_many(questions)]          # every replica holds the same weights
    assert answers == solo, (answers, solo)
    allw = gather_winners(comm, local[:2], "ABCD")
    assert allw.numel() == 4

    dist.barrier()
    Path(f"{sys.argv[1]}.{rank}").write_text("ok")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
