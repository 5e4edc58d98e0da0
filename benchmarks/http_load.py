"""Closed-loop load on the HTTP front door: the majority-vote ensemble of bench.py's workload A behind
``ConsensusServer``, questions sent by a separate client process over real sockets.

``bench.py`` measures the closed loop in process (``VotePool``: a finished question is replaced at
the next engine step). This measures the same loop through everything a deployment adds: one
handler thread per connection, JSON both ways, the job queue between handler threads and the
driver thread, the watch for clients that hang up. The difference between the two numbers is the
cost of the front end.

    python benchmarks/http_load.py                       # 3 x 7-8B on cuda:0, 256 connections, 1024 questions
    python benchmarks/http_load.py --device cpu --models tiny-llama,tiny-llama --in-flight 8 --questions 16 \
        --prompt-len 96 --new-tokens 4                   # plumbing check without a GPU

The client half (``--client``) never touches a GPU: it is started as a child process, opens
``--in-flight`` connections' worth of threads, and each thread posts its next question as soon as
the previous answer is in. One JSON line comes out.
"""
from __future__ import annotations

import argparse
import json
import subprocess
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def client(a) -> int:
    import http.client

    import numpy as np

    from llm_consensus_amd.workloads import make_questions

    qs = make_questions(np.random.default_rng(a.seed), a.questions + a.warm, a.prompt_len)
    lock = threading.Lock()
    state = {"next": 0, "errors": 0}
    lat, done_at, ttft = [], [], []
    kinds = {}                                   # failed requests by exception type

    def worker():
        while True:
            with lock:
                i = state["next"]
                if i >= len(qs):
                    return
                state["next"] = i + 1
            if a.stream:      # one member model's tokens as server-sent events
                body = json.dumps({"model": a.stream, "prompt": qs[i], "max_tokens": a.new_tokens, "stream": True}).encode()
            else:
                body = json.dumps({"messages": [{"role": "user", "content": qs[i]}]}).encode()
            t0 = time.perf_counter()
            t_first = None
            try:
                conn = http.client.HTTPConnection("127.0.0.1", a.port, timeout=a.timeout)
                conn.request("POST", "/v1/completions" if a.stream else "/v1/chat/completions", body,
                             {"Content-Type": "application/json"})
                r = conn.getresponse()
                if a.stream and r.status == 200:
                    pieces, finished, done = 0, False, False
                    for raw in r:
                        line = raw.strip()
                        if line == b"data: [DONE]":
                            done = True
                            break
                        if line.startswith(b"data: {"):
                            c = json.loads(line[6:])["choices"][0]
                            if c.get("text"):
                                pieces += 1
                                t_first = t_first or time.perf_counter()
                            finished = finished or bool(c.get("finish_reason"))
                    conn.close()
                    ok = done and finished and pieces > 0
                else:
                    data = r.read()
                    conn.close()
                    ok = r.status == 200 and "consensus" in json.loads(data)
                if not ok:
                    with lock:
                        kinds[f"HTTP {r.status}"] = kinds.get(f"HTTP {r.status}", 0) + 1
            except (OSError, ValueError) as e:
                ok = False
                with lock:
                    kinds[type(e).__name__] = kinds.get(type(e).__name__, 0) + 1
            t1 = time.perf_counter()
            with lock:
                if not ok:
                    state["errors"] += 1
                elif i >= a.warm:                # the first questions fill the pipeline and are not counted
                    lat.append((t1 - t0) * 1e3)
                    done_at.append(t1)
                    if t_first is not None:
                        ttft.append((t_first - t0) * 1e3)

    ts = [threading.Thread(target=worker, daemon=True) for _ in range(a.in_flight)]
    t_start = time.perf_counter()
    [t.start() for t in ts]
    [t.join() for t in ts]
    n = len(lat)
    # rate over the span in which counted answers arrived (the ramp of the warm questions is outside it)
    span = (max(done_at) - min(done_at)) if n > 1 else float("nan")
    print(json.dumps({"answers": n, "errors": state["errors"], "error_kinds": kinds, "wall_s": round(time.perf_counter() - t_start, 2),
                      "answers_per_s": round((n - 1) / span, 3) if n > 1 and span > 0 else None,
                      "p50_latency_ms": round(float(np.percentile(lat, 50)), 1) if n else None,
                      "p95_latency_ms": round(float(np.percentile(lat, 95)), 1) if n else None,
                      "p99_latency_ms": round(float(np.percentile(lat, 99)), 1) if n else None,
                      **({"p50_first_token_ms": round(float(np.percentile(ttft, 50)), 1),
                          "p99_first_token_ms": round(float(np.percentile(ttft, 99)), 1)} if ttft else {})}), flush=True)
    return 0 if n and not state["errors"] else 1


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"])
    ap.add_argument("--models", default="llama-3-8b,qwen2.5-7b,mistral-7b")
    ap.add_argument("--in-flight", type=int, default=256, help="client connections, each a closed loop")
    ap.add_argument("--questions", type=int, default=1024, help="counted questions")
    ap.add_argument("--warm", type=int, default=None, help="uncounted questions that fill the pipeline first (default: --in-flight)")
    ap.add_argument("--prompt-len", type=int, default=512)
    ap.add_argument("--new-tokens", type=int, default=64)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--timeout", type=float, default=300.0)
    ap.add_argument("--stream", nargs="?", const="first", default=None, metavar="MODEL",
                    help="stream completions of ONE member model (default: the first) as server-sent events instead "
                         "of asking consensus questions; adds time to first token")
    ap.add_argument("--client", action="store_true", help="(internal) run the client half against --port")
    ap.add_argument("--port", type=int, default=0)
    a = ap.parse_args(argv)
    if a.warm is None:
        a.warm = a.in_flight
    if a.client:
        return client(a)

    import torch

    from llm_consensus_amd import workloads as W
    from llm_consensus_amd.app import build_ensemble
    from llm_consensus_amd.serve.server import ConsensusServer

    dev = torch.device("cuda", 0) if a.device == "cuda" else torch.device("cpu")
    if a.device == "cuda":
        torch.cuda.set_device(0)
    samp = W.sampling(0.8, 0.95, a.new_tokens)
    cfg = W.ensemble_vote_config(a.models.split(","), a.in_flight, a.prompt_len, a.new_tokens, samp,
                                 bs=64 if a.device == "cuda" else 16)
    t0 = time.perf_counter()
    ens = build_ensemble(cfg, dev)
    ens.warm_up()
    init_s = time.perf_counter() - t0
    srv = ConsensusServer(ens, port=0, max_batch=max(64, a.in_flight), max_pending=4 * a.in_flight,
                          request_timeout_s=a.timeout).start()
    extra = []
    if a.stream:
        extra = ["--stream", sorted(ens.backends)[0] if a.stream == "first" else a.stream, "--new-tokens", str(a.new_tokens)]
    try:
        p = subprocess.run([sys.executable, str(Path(__file__).resolve()), "--client", "--port", str(srv.port),
                            "--in-flight", str(a.in_flight), "--questions", str(a.questions), "--warm", str(a.warm),
                            "--prompt-len", str(a.prompt_len), "--seed", str(a.seed), "--timeout", str(a.timeout)] + extra,
                           capture_output=True, text=True, timeout=a.timeout * 4)
    finally:
        srv.shutdown()
    line = (p.stdout.strip().splitlines() or ["{}"])[-1]
    try:
        out = json.loads(line)
    except ValueError:
        out = {"error": (p.stderr or p.stdout)[-400:]}
    out.update({"mode": ("closed loop over HTTP: every connection streams its next completion when the last one ended"
                         if a.stream else
                         "closed loop over HTTP: every connection posts its next question when its answer is in"),
                "connections": a.in_flight, "models": a.models, "prompt_len": a.prompt_len, "new_tokens": a.new_tokens,
                "device": a.device, "init_s": round(init_s, 1),
                "server": {k: v for k, v in srv.counters.items()}})
    print(json.dumps(out), flush=True)
    return 0 if p.returncode == 0 else 1


if __name__ == "__main__":
    raise SystemExit(main())
