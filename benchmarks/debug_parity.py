import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from llm_consensus_amd._native import core
from llm_consensus_amd.engine.engine import Engine
from llm_consensus_amd.models.config import get_preset
from llm_consensus_amd.models.transformer import TransformerModel
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_engine_gpu import copy_weights, params

name = sys.argv[1] if len(sys.argv) > 1 else "tiny-llama"
cfg = get_preset(name)
cpu = TransformerModel.random_init(cfg, "cpu", seed=1, std=0.08)
gpu = TransformerModel.random_init(cfg, "cuda", seed=1, std=0.08)
copy_weights(cpu, gpu)
rng = np.random.default_rng(0)
prompts = [rng.integers(0, cfg.vocab_size, size=n).tolist() for n in (5, 70, 33, 64, 129)]
for which in ([3], [2, 3], [3, 4]):
    res = []
    for m in (cpu, gpu):
        eng = Engine(m, num_kv_blocks=128, block_size=16, max_prefill_tokens=96, use_graphs=False)
        rids = [eng.submit(prompts[i], params=params(6)) for i in which]
        sc = [eng.submit(prompts[i], core.RequestKind.Score, score_ids=[4, 5, 6, 7]) for i in which] if os.environ.get("SC") else []
        eng.run_until_done()
        res.append(([eng.pop_result(r).token_ids[0] for r in rids], [np.round(eng.pop_result(r).scores, 3).tolist() for r in sc]))
    print("subset", which)
    print(" cpu", res[0])
    print(" gpu", res[1])
