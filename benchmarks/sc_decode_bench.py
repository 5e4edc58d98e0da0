"""Decode step time of the self-consistency workload: Q questions, K chains each forked from one
prefill and sharing its KV blocks (default 16 x 16 on Llama-3-8B, prompt 512, 64 new tokens).

  python benchmarks/sc_decode_bench.py [preset] [questions] [chains] [prompt_len] [new_tokens]
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


def main():
    a = sys.argv[1:]
    preset = a[0] if len(a) > 0 else "llama-3-8b"
    Q = int(a[1]) if len(a) > 1 else 16
    K = int(a[2]) if len(a) > 2 else 16
    ctx = int(a[3]) if len(a) > 3 else 512
    new = int(a[4]) if len(a) > 4 else 64
    cfg = get_preset(preset)
    model = TransformerModel.random_init(cfg, torch.device("cuda", 0), seed=1)
    eng = Engine(model, num_kv_blocks=Q * (ctx // 64 + 1) + Q * K * (new // 64 + 2) + 64, block_size=64,
                 max_batch_seqs=max(64, Q * K), max_model_len=ctx + new + 64)
    p = core.SamplingParams()
    p.temperature, p.top_p, p.max_new_tokens, p.ignore_eos = 0.8, 0.95, new, True
    g = torch.Generator().manual_seed(0)
    out = {"model": preset, "questions": Q, "chains": K, "prompt_len": ctx, "new_tokens": new}
    for rep in range(2):                       # the first pass captures the graphs, the second is timed
        for q in range(Q):
            p.seed = q * 1000 + rep
            eng.submit(torch.randint(5, 1000, (ctx,), generator=g).tolist(), n=K, params=p)
        t_dec = n_dec = 0
        t_pre = 0.0
        while eng.has_work():
            is_decode = not eng.sched.num_waiting and eng.metrics.counter("prefill.tokens") >= (rep + 1) * Q * ctx
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.step()
            torch.cuda.synchronize()
            if is_decode:
                t_dec += time.perf_counter() - t0
                n_dec += 1
            else:
                t_pre += time.perf_counter() - t0
    out["decode_ms_per_step"] = round(t_dec / max(1, n_dec) * 1e3, 3)
    out["decode_steps"] = n_dec
    out["prefill_ms"] = round(t_pre * 1e3, 2)
    print(json.dumps(out))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/sc_decode_bench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
