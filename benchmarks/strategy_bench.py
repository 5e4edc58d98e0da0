"""The other two consensus strategies at 7-8B scale on one GPU: judge-logprob (three proposers,
one judge scoring every candidate with a Yes/No log-prob) and embedding-medoid (three
proposers, the pooled hidden state of an embedder picks the medoid answer).

  python benchmarks/strategy_bench.py [questions] [prompt_len] [new_tokens]
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_consensus_amd._native import core  # noqa: E402
from llm_consensus_amd.consensus.backend import GenParams  # noqa: E402
from llm_consensus_amd.consensus.members import Member  # noqa: E402
from llm_consensus_amd.consensus.strategies import EmbeddingMedoid, JudgeLogprob  # noqa: E402
from llm_consensus_amd.engine.backend import EngineBackend  # noqa: E402
from llm_consensus_amd.engine.engine import Engine  # noqa: E402
from llm_consensus_amd.models.config import get_preset  # noqa: E402
from llm_consensus_amd.models.transformer import TransformerModel  # noqa: E402
from llm_consensus_amd.utils.tokenizer import ByteLevelTokenizer  # noqa: E402


def main():
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    plen = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    new = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    dev = torch.device("cuda", 0)
    personas = core.default_personas()
    members = []
    for i, name in enumerate(["llama-3-8b", "qwen2.5-7b", "mistral-7b"]):
        cfg = get_preset(name)
        model = TransformerModel.random_init(cfg, dev, seed=10 + i)
        eng = Engine(model, num_kv_blocks=4096, block_size=64, max_model_len=2048, name=name)
        be = EngineBackend(eng, ByteLevelTokenizer(cfg.vocab_size), stop_at_eos=False)
        members.append(Member(personas[i], be, params=GenParams(temperature=0.8, top_p=0.95, max_new_tokens=new, seed=i)))
    rng = np.random.default_rng(0)

    def questions():
        return ["".join(chr(c) for c in rng.integers(97, 123, size=plen)) for _ in range(nq)]

    out = {"questions": nq, "prompt_len": plen, "new_tokens": new}
    for name, strat in (("judge_logprob", JudgeLogprob(members, members[:1], device=dev)),
                        ("embedding_medoid", EmbeddingMedoid(members, members[0], device=dev))):
        strat.ask_many(questions())                      # warm-up: graph capture, allocator
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = strat.ask_many(questions())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[name] = {"seconds": round(dt, 3), "answers_per_s": round(nq / dt, 2),
                     "member_calls": int(np.mean([r.member_calls for r in res]))}
        print(name, out[name], flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/strategy_bench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
