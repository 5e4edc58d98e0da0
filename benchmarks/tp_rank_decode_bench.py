"""Decode step time of ONE rank of a tensor-parallel model (default: Llama-3-70B at TP=4, the
judge of the judge-consensus configuration), eager launches against graphs cut at the
collectives. The collectives are stand-ins that only touch the local tensor (there is one GPU
here), so this measures the host-launch and kernel side of the step, not the xGMI exchange.

  python benchmarks/tp_rank_decode_bench.py [preset] [tp] [batch] [ctx]
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_consensus_amd._native import core  # noqa: E402
from llm_consensus_amd.engine.engine import Engine  # noqa: E402
from llm_consensus_amd.models.config import get_preset  # noqa: E402
from llm_consensus_amd.models.transformer import TransformerModel  # noqa: E402
from llm_consensus_amd.parallel.comm import Comm  # noqa: E402


class LocalComm(Comm):
    """Rank 0 of `world` ranks; a collective costs one small kernel on the local tensor."""

    def __init__(self, world):
        self.rank, self.world = 0, world

    def all_reduce(self, t):
        t.mul_(1.0)
        return t

    def all_gather_cols(self, t, total_cols=None):
        out = torch.cat([t] * self.world, dim=-1)
        return out[..., :total_cols] if total_cols is not None else out


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "llama-3-70b"
    tp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    ctx = int(sys.argv[4]) if len(sys.argv) > 4 else 512
    new = 32
    cfg = get_preset(preset)
    dev = torch.device("cuda", 0)
    model = TransformerModel.random_init(cfg, dev, seed=1, tp_rank=0, tp_size=tp, comm=LocalComm(tp))
    out = {"model": preset, "tp": tp, "batch": B, "ctx": ctx, "weight_GB_per_rank": round(model.weight_bytes() / 1e9, 1)}
    for mode in (sys.argv[5].split(",") if len(sys.argv) > 5 else ("eager", "graphs", "lists")):
        os.environ["CONSENSUS_PIECEWISE"] = {"eager": "0", "graphs": "graphs", "lists": "lists"}[mode]
        eng = Engine(model, num_kv_blocks=B * ((ctx + new) // 64 + 2) + 16, block_size=64, max_batch_seqs=max(64, B),
                     max_model_len=ctx + new + 64, name=mode)
        p = core.SamplingParams()
        p.temperature, p.top_p, p.max_new_tokens, p.ignore_eos = 0.8, 0.95, new, True
        g = torch.Generator().manual_seed(0)
        for rep in range(2):                       # the first pass captures, the second is timed
            for b in range(B):
                p.seed = b
                eng.submit(torch.randint(5, 1000, (ctx,), generator=g).tolist(), params=p)
            t_dec = n_dec = 0
            while eng.has_work():
                plan_is_decode = not eng.sched.num_waiting and eng.metrics.counter("prefill.tokens") >= (rep + 1) * B * ctx
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                eng.step()
                torch.cuda.synchronize()
                if plan_is_decode:
                    t_dec += time.perf_counter() - t0
                    n_dec += 1
        out[mode + "_ms_per_step"] = round(t_dec / max(1, n_dec) * 1e3, 3)
        out[mode + "_steps"] = n_dec
        del eng
        torch.cuda.empty_cache()
    print(json.dumps(out))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/tp_rank_decode_bench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
