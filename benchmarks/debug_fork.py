import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from llm_consensus_amd._native import core
from llm_consensus_amd.engine.engine import Engine
from llm_consensus_amd.models.config import get_preset
from llm_consensus_amd.models.transformer import TransformerModel
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_engine_gpu import params

name, bs = sys.argv[1], int(sys.argv[2])
gpu = TransformerModel.random_init(get_preset(name), "cuda", seed=2, std=0.1)
prompt = list(range(1, 1 + 3 * bs + 7))
K, n_new = 16, 12
for temp in (0.0, 0.8):
    outs = []
    for mode in ("fork_graph", "fork_eager", "fork_noprefix", "indep"):
        eng = Engine(gpu, num_kv_blocks=512, block_size=bs, use_graphs=(mode == "fork_graph"),
                     enable_prefix_cache=(mode != "indep"), max_group_rows=(0 if mode == "fork_noprefix" else 16))
        if mode == "indep":
            rids = [eng.submit(prompt, params=params(n_new, seed=500 + k, temperature=temp, top_p=0.95)) for k in range(K)]
            eng.run_until_done()
            rs = [eng.pop_result(r) for r in rids]
            outs.append(([r.token_ids[0] for r in rs], [r.cum_logprobs[0] for r in rs]))
        else:
            rid = eng.submit(prompt, n=K, params=params(n_new, seed=500, temperature=temp, top_p=0.95))
            eng.run_until_done()
            r = eng.pop_result(rid)
            outs.append((r.token_ids, r.cum_logprobs))
        print(mode, temp, "graph keys", list(eng._graphs.keys()))
    base = outs[-1]
    for mi, mode in enumerate(("fork_graph", "fork_eager", "fork_noprefix")):
        div = [next((i for i, (a, b) in enumerate(zip(outs[mi][0][k], base[0][k])) if a != b), n_new) for k in range(K)]
        print(" ", mode, "first divergence per chain:", div)
