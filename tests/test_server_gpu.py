"""The HTTP front end on a real GPU: the server's driver thread owns the engines (runtime streams,
graph capture in that thread) while request threads admit questions at different times."""
import json
import threading
import time
import urllib.request

import pytest

pytestmark = pytest.mark.gpu


def _post(port, path, body, timeout=120):
    req = urllib.request.Request(f"http://127.0.0.1:{port}{path}", json.dumps(body).encode(),
                                 {"Content-Type": "application/json"})
    with urllib.request.urlopen(req, timeout=timeout) as r:
        return r.status, json.loads(r.read())


@pytest.mark.parametrize("strategy", ["reference_protocol", "majority_vote"])
def test_server_answers_staggered_questions_on_the_gpu(strategy):
    from llm_consensus_amd.app import build_ensemble
    from llm_consensus_amd.config import config_from_dict
    from llm_consensus_amd.serve.server import ConsensusServer

    models = {"a": {"preset": "tiny-llama", "seed": 1, "kv_blocks": 512},
              "b": {"preset": "tiny-qwen", "seed": 2, "kv_blocks": 512}}
    members = [{"name": f"m{i}", "model": "ab"[i % 2], "domain": "d", "tuning": "- t"} for i in range(3)]
    cfg = config_from_dict({"strategy": strategy, "decision_mode": "logprob", "seed": 3, "members": members,
                            "sampling": {"temperature": 0.7, "max_new_tokens": 6}, "models": models})
    ens = build_ensemble(cfg)
    assert ens.device.type == "cuda"
    srv = ConsensusServer(ens, port=0, max_batch=8).start()
    out = {}
    try:
        def ask(i, delay):
            time.sleep(delay)
            out[i] = _post(srv.port, "/v1/chat/completions",
                           {"messages": [{"role": "user", "content": f"question {i}: A. x B. y C. z D. w"}]})

        def complete():
            time.sleep(0.02)
            out["direct"] = _post(srv.port, "/v1/completions", {"model": "a", "prompt": "hello", "max_tokens": 40})

        ts = [threading.Thread(target=ask, args=(i, 0.01 * i)) for i in range(6)] + [threading.Thread(target=complete)]
        [t.start() for t in ts]
        [t.join() for t in ts]
    finally:
        srv.shutdown()
    for i in range(6):
        st, body = out[i]
        assert st == 200 and body["consensus"]["strategy"] == strategy
        assert body["usage"]["total_tokens"] > 0
    assert out["direct"][0] == 200
    stats = ens.stats()
    assert stats["consensus"]["answers"] == 6
    eng_stats = [v for k, v in stats.items() if isinstance(v, dict) and "rt.graphs" in v]
    assert eng_stats and all(s["rt.kv_arena_bytes"] > 0 for s in eng_stats)
    assert any(s["rt.graph_launches"] > 0 for s in eng_stats)
