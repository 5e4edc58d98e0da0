"""One rank of tests/test_p2p_gpu.py::test_first_contact_*: W processes on ONE GPU build a model
group through parallel.p2p.wrap_group_comm. Usage: p2p_firstcontact_worker.py <out_prefix> <mode>
with mode ok | stall | corrupt (the last two inject a fault into the first-contact check of the
peer-to-peer collectives; the group must then serve from the bootstrap communicator, in this
process, and produce the same tokens)."""
import json
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from llm_consensus_amd._native import core  # noqa: E402
from llm_consensus_amd.engine.engine import Engine  # noqa: E402
from llm_consensus_amd.models.config import get_preset  # noqa: E402
from llm_consensus_amd.models.transformer import TransformerModel  # noqa: E402
from llm_consensus_amd.parallel import p2p  # noqa: E402
from llm_consensus_amd.parallel.comm import TorchDistComm  # noqa: E402

DEV = torch.device("cuda", 0)


def main() -> None:
    out_prefix, mode = sys.argv[1], sys.argv[2]
    torch.cuda.set_device(DEV)
    dist.init_process_group("gloo")
    rank, W = dist.get_rank(), dist.get_world_size()
    base = TorchDistComm(dist, DEV)                 # gloo with device tensors: host-staged collectives
    p2p.FAIL_INJECT = "" if mode == "ok" else mode
    p2p.GRID_CAP = 32                                # the ranks share one GPU
    t0 = time.perf_counter()
    comm = p2p.wrap_group_comm(base, DEV, 16 << 20)
    took = time.perf_counter() - t0
    if mode == "ok":
        assert isinstance(comm, p2p.P2PComm) and comm.collectives.startswith("p2p"), comm
    else:
        assert comm is base and base.fallback_reason, "a failed self-check must leave the bootstrap communicator"
        assert took < 30, f"falling back took {took:.1f} s"
    cfg = get_preset("small-llama")
    sp = core.SamplingParams()
    sp.temperature, sp.max_new_tokens, sp.ignore_eos = 0.0, 6, True
    model = TransformerModel.random_init(cfg, DEV, seed=6, std=0.05, tp_rank=rank, tp_size=W, comm=comm)
    eng = Engine(model, num_kv_blocks=64, block_size=64)
    assert eng.use_graphs == (mode == "ok")          # the staged gloo fallback of this rehearsal runs eagerly
    rid = eng.submit(list(range(3, 43)), core.RequestKind.Generate, 1, sp)
    eng.run_until_done()
    toks = eng.pop_result(rid).token_ids[0]
    all_toks = base.all_gather_object(list(toks))
    assert all(t == all_toks[0] for t in all_toks), "ranks sampled different tokens"
    base.barrier()
    if isinstance(comm, p2p.P2PComm):
        comm.close()
    Path(f"{out_prefix}.{mode}.{rank}").write_text(json.dumps({
        "tokens": list(map(int, toks)), "seconds": round(took, 2), "collectives": getattr(comm, "collectives", ""),
        "reason": getattr(comm, "fallback_reason", None)}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
