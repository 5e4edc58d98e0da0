"""The HTTP server's continuous driver: reference-protocol sessions and direct completions
admitted at different times share the engine, each answered as soon as it is done."""
import json
import threading
import time
import urllib.error
import urllib.request

import pytest

from llm_consensus_amd.app import build_ensemble
from llm_consensus_amd.config import config_from_dict
from llm_consensus_amd.serve.server import ConsensusServer


def _post(port, path, body, timeout=300):
    req = urllib.request.Request(f"http://127.0.0.1:{port}{path}", json.dumps(body).encode(),
                                 {"Content-Type": "application/json"})
    with urllib.request.urlopen(req, timeout=timeout) as r:
        return r.status, json.loads(r.read())


@pytest.fixture(scope="module")
def server():
    cfg = config_from_dict({"strategy": "reference_protocol", "decision_mode": "logprob", "seed": 3,
                            "sampling": {"temperature": 0.0, "max_new_tokens": 4},
                            "models": {"default": {"preset": "tiny-llama", "seed": 1}}})
    ens = build_ensemble(cfg)
    srv = ConsensusServer(ens, port=0, max_batch=8).start()
    assert srv._continuous_ok()
    yield srv, ens
    srv.shutdown()


def test_staggered_questions_and_a_direct_completion_are_all_answered(server):
    srv, ens = server
    out = {}

    def ask(i, delay):
        time.sleep(delay)
        out[i] = _post(srv.port, "/v1/chat/completions", {"messages": [{"role": "user", "content": f"question {i}"}]})

    def complete():
        time.sleep(0.05)
        out["direct"] = _post(srv.port, "/v1/completions", {"model": "default", "prompt": "hello", "max_tokens": 3})

    ts = [threading.Thread(target=ask, args=(i, 0.03 * i)) for i in range(5)] + [threading.Thread(target=complete)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for i in range(5):
        st, body = out[i]
        assert st == 200 and body["consensus"]["strategy"] == "reference_protocol"
        assert 1 <= body["consensus"]["rounds"] <= 5 and isinstance(body["choices"][0]["message"]["content"], str)
    assert out["direct"][0] == 200 and len(out["direct"][1]["choices"]) == 1
    st = ens.stats()["consensus"]
    assert st["answers"] == 5 and st["member_calls"] >= 25


def test_batched_mode_is_still_available(server):
    srv, ens = server
    batched = ConsensusServer(ens, port=0, continuous=False)
    assert not batched._continuous_ok()
    batched.httpd.server_close()


def test_direct_completion_outliving_the_vote_questions_is_stepped_by_one_driver():
    """A direct completion that is still generating when the last consensus question has been
    answered: the engine it shares with the vote members must go on being stepped by the SAME
    driver (a second driver would schedule a new step while the first one's step was still in
    flight, and the engine's bookkeeping would fall apart with a KeyError)."""
    cfg = config_from_dict({"strategy": "majority_vote", "decision_mode": "logprob", "seed": 3,
                            "sampling": {"temperature": 0.7, "max_new_tokens": 3},
                            "models": {"default": {"preset": "tiny-llama", "seed": 1}}})
    ens = build_ensemble(cfg)
    srv = ConsensusServer(ens, port=0, max_batch=8).start()
    out = {}
    try:
        def ask(i):
            out[i] = _post(srv.port, "/v1/chat/completions",
                           {"messages": [{"role": "user", "content": f"question {i}: A. x B. y C. z D. w"}]})

        def complete():
            out["direct"] = _post(srv.port, "/v1/completions", {"model": "default", "prompt": "hello", "max_tokens": 48})

        for rep in range(3):
            ts = [threading.Thread(target=complete)] + [threading.Thread(target=ask, args=(10 * rep + i,)) for i in range(3)]
            [t.start() for t in ts]
            [t.join() for t in ts]
            assert out["direct"][0] == 200 and out["direct"][1]["usage"]["completion_tokens"] == 48
            for i in range(3):
                st, body = out[10 * rep + i]
                assert st == 200 and body["usage"]["total_tokens"] > 0, body
    finally:
        srv.shutdown()


def test_streamed_text_that_diverges_after_a_preemption_is_restarted_not_silenced():
    """Token streaming decodes the whole chain every step and sends what is new. A preempted
    request is recomputed; should the regenerated chain differ from what was already sent, the
    client is told (restart marker + the text as it now stands) instead of getting nothing more."""
    from llm_consensus_amd.engine.backend import EngineBackend
    from llm_consensus_amd.utils.tokenizer import ByteLevelTokenizer

    be = EngineBackend.__new__(EngineBackend)
    be.tok = ByteLevelTokenizer()
    be._stops = {}
    got = []
    be._streams = {7: [lambda chain, text: got.append((chain, text)), [""]]}
    enc = lambda s: list(be.tok.encode(s, add_bos=False))  # noqa: E731
    be._emit(7, [enc("The answer")], final=False)
    be._emit(7, [enc("The answer is")], final=False)
    be._emit(7, [enc("The")], final=False)                    # preempted: the chain restarts, a prefix so far
    be._emit(7, [enc("The reply is B")], final=False)         # and then leaves what was sent
    be._emit(7, [enc("The reply is B.")], final=True)
    assert got[0] == (0, "The answer") and got[1] == (0, " is")
    assert got[2] == (0, EngineBackend.RESTART_MARKER + "The reply is B")
    assert got[3] == (0, ".") and len(got) == 4


def test_a_client_that_hangs_up_takes_its_work_out_of_the_engine():
    """A streaming completion whose client closes the socket after the first tokens, and a
    consensus question whose client gives up while it waits: the serving loop cancels both (the
    engine drops their requests at its next step and frees their blocks) instead of generating to
    the end for nobody, and a request that arrives afterwards is served as usual."""
    import socket

    cfg = config_from_dict({"strategy": "majority_vote", "seed": 3,
                            "sampling": {"temperature": 0.0, "max_new_tokens": 400},
                            "models": {"default": {"preset": "tiny-llama", "seed": 1, "kv_blocks": 512}}})
    ens = build_ensemble(cfg)
    srv = ConsensusServer(ens, port=0, max_batch=8).start()
    eng = next(iter(ens.backends.values())).engine
    try:
        def raw_post(path, body):
            s = socket.create_connection(("127.0.0.1", srv.port))
            data = json.dumps(body).encode()
            s.sendall(f"POST {path} HTTP/1.1\r\nHost: x\r\nContent-Type: application/json\r\n"
                      f"Content-Length: {len(data)}\r\n\r\n".encode() + data)
            return s

        # 1. token stream: read until the first data chunk, then hang up
        model = next(iter(ens.backends))
        s1 = raw_post("/v1/completions", {"model": model, "prompt": "tell me a long story", "stream": True,
                                          "max_tokens": 400})
        got = b""
        while b"data: " not in got:
            got += s1.recv(4096)
        s1.close()
        # 2. consensus question (not streamed): hang up while it is being answered
        s2 = raw_post("/v1/chat/completions", {"messages": [{"role": "user", "content": "A) x B) y C) z D) w ?"}]})
        time.sleep(0.3)
        s2.close()
        deadline = time.time() + 60
        while time.time() < deadline:
            st = eng.stats()
            if st.get("requests.cancelled", 0) >= 2 and st["sched.running"] == 0 and st["sched.waiting"] == 0:
                break
            time.sleep(0.05)
        st = eng.stats()
        assert st.get("requests.cancelled", 0) >= 2, st
        assert st["sched.running"] == 0 and st["sched.waiting"] == 0
        assert srv.counters["cancelled_total"] == 2
        # far fewer tokens were generated than the 400 per member call that were asked for (5 calls)
        assert st["decode.tokens"] < 1200, st["decode.tokens"]
        # 3. the server still serves
        code, out = _post(srv.port, "/v1/completions", {"model": model, "prompt": "hi", "max_tokens": 3})
        assert code == 200 and out["usage"]["completion_tokens"] == 3
    finally:
        srv.shutdown()


def test_requests_beyond_the_admission_limit_get_429_and_the_rest_are_answered():
    import urllib.error
    from concurrent.futures import ThreadPoolExecutor

    cfg = config_from_dict({"strategy": "majority_vote", "seed": 3,
                            "sampling": {"temperature": 0.0, "max_new_tokens": 60},
                            "models": {"default": {"preset": "tiny-llama", "seed": 1, "kv_blocks": 512}}})
    ens = build_ensemble(cfg)
    srv = ConsensusServer(ens, port=0, max_batch=8, max_pending=3).start()
    model = next(iter(ens.backends))
    try:
        def one(i):
            try:
                code, out = _post(srv.port, "/v1/completions", {"model": model, "prompt": f"question {i}", "max_tokens": 60})
                return code, out["usage"]["completion_tokens"]
            except urllib.error.HTTPError as e:
                return e.code, e.headers.get("Retry-After")

        with ThreadPoolExecutor(10) as ex:
            got = list(ex.map(one, range(10)))
        ok = [g for g in got if g[0] == 200]
        busy = [g for g in got if g[0] == 429]
        assert len(ok) + len(busy) == 10 and 3 <= len(ok) < 10 and busy, got
        assert all(n == 60 for _, n in ok) and all(ra == "1" for _, ra in busy)
        assert srv.counters["rejected_total"] == len(busy) and srv._outstanding == 0
        assert one(99)[0] == 200                     # and the server is free again
    finally:
        srv.shutdown()


def test_a_request_that_times_out_is_answered_504_and_cancelled():
    import urllib.error

    cfg = config_from_dict({"strategy": "majority_vote", "seed": 3,
                            "sampling": {"temperature": 0.0, "max_new_tokens": 400},
                            "models": {"default": {"preset": "tiny-llama", "seed": 1, "kv_blocks": 512}}})
    ens = build_ensemble(cfg)
    srv = ConsensusServer(ens, port=0, max_batch=8, request_timeout_s=0.2).start()       # 400 tokens take ~1 s here
    eng = next(iter(ens.backends.values())).engine
    model = next(iter(ens.backends))
    try:
        with pytest.raises(urllib.error.HTTPError) as ei:
            _post(srv.port, "/v1/completions", {"model": model, "prompt": "a very long answer please", "max_tokens": 400})
        assert ei.value.code == 504
        with pytest.raises(urllib.error.HTTPError) as bad:       # a budget the context cannot hold is the caller's error
            _post(srv.port, "/v1/completions", {"model": model, "prompt": "x", "max_tokens": 10 ** 6})
        assert bad.value.code == 400
        deadline = time.time() + 30
        while time.time() < deadline and eng.stats().get("requests.cancelled", 0) < 1:
            time.sleep(0.05)
        st = eng.stats()
        assert st["requests.cancelled"] == 1 and st["sched.running"] == 0 and st["decode.tokens"] < 400
    finally:
        srv.shutdown()


def test_api_key_guards_the_post_routes(server):
    import urllib.error

    srv, ens = server
    srv.api_key = "s3cret"
    try:
        body = {"model": next(iter(ens.backends)), "prompt": "hi", "max_tokens": 2}
        with pytest.raises(urllib.error.HTTPError) as e1:
            _post(srv.port, "/v1/completions", body)
        assert e1.value.code == 401
        req = urllib.request.Request(f"http://127.0.0.1:{srv.port}/v1/completions", json.dumps(body).encode(),
                                     {"Content-Type": "application/json", "Authorization": "Bearer s3cret"})
        with urllib.request.urlopen(req, timeout=120) as r:
            assert r.status == 200
        with urllib.request.urlopen(f"http://127.0.0.1:{srv.port}/health", timeout=5) as r:
            assert r.status == 200            # liveness stays open
    finally:
        srv.api_key = None


def test_soak_with_clients_that_hang_up_at_random_leaves_nothing_behind():
    """Eight clients for six seconds: consensus questions, direct completions and token streams,
    four in ten abandoned at a random moment. Afterwards nothing is in flight, nothing runs or
    waits in the engine, every KV block is free or kept by the prefix cache, and the server answers."""
    import random
    import socket

    cfg = config_from_dict({"strategy": "majority_vote", "seed": 3, "sampling": {"temperature": 0.7, "max_new_tokens": 30},
                            "models": {"default": {"preset": "tiny-llama", "seed": 1, "kv_blocks": 1024}}})
    ens = build_ensemble(cfg)
    srv = ConsensusServer(ens, port=0, max_batch=8, max_pending=64).start()
    eng = next(iter(ens.backends.values())).engine
    model = next(iter(ens.backends))
    stop = time.time() + 6.0
    counts = {"ok": 0, "hung_up": 0, "other": 0}
    lock = threading.Lock()

    def raw(path, body):
        s = socket.create_connection(("127.0.0.1", srv.port), timeout=60)
        data = json.dumps(body).encode()
        s.sendall(f"POST {path} HTTP/1.1\r\nHost: x\r\nContent-Type: application/json\r\n"
                  f"Content-Length: {len(data)}\r\n\r\n".encode() + data)
        return s

    def client(seed):
        rng = random.Random(seed)
        while time.time() < stop:
            kind = rng.choice(["consensus", "direct", "stream"])
            if kind == "consensus":
                path, body = "/v1/chat/completions", {"messages": [{"role": "user", "content": f"q{rng.random()} A) a B) b C) c D) d"}]}
            else:
                path, body = "/v1/completions", {"model": model, "prompt": f"p{rng.random()}",
                                                 "max_tokens": rng.choice([5, 20, 30]), "stream": kind == "stream"}
            try:
                s = raw(path, body)
                if rng.random() < 0.4:
                    time.sleep(rng.random() * 0.2)
                    s.close()
                    with lock:
                        counts["hung_up"] += 1
                    continue
                buf = b""
                while True:
                    chunk = s.recv(65536)
                    if not chunk:
                        break
                    buf += chunk
                    if kind == "stream":
                        if b"[DONE]" in buf:
                            break
                    elif b"\r\n\r\n" in buf:
                        head, rest = buf.split(b"\r\n\r\n", 1)
                        cl = [l for l in head.split(b"\r\n") if l.lower().startswith(b"content-length")]
                        if cl and len(rest) >= int(cl[0].split(b":")[1]):
                            break
                s.close()
                with lock:
                    counts["ok" if buf.split(b" ", 2)[1:2] == [b"200"] else "other"] += 1
            except OSError:
                with lock:
                    counts["other"] += 1

    try:
        ths = [threading.Thread(target=client, args=(i,)) for i in range(8)]
        [t.start() for t in ths]
        [t.join(120) for t in ths]
        deadline = time.time() + 20
        while time.time() < deadline and (srv._outstanding or eng.stats()["sched.running"] or eng.stats()["sched.waiting"]):
            time.sleep(0.05)
        st = eng.stats()
        assert counts["ok"] > 10 and counts["hung_up"] > 5 and counts["other"] == 0, counts
        assert srv._outstanding == 0 and st["sched.running"] == 0 and st["sched.waiting"] == 0
        assert st["kv.blocks_free"] + st["kv.blocks_cached"] == st["kv.blocks_total"]
        assert st.get("requests.cancelled", 0) > 0 and srv.counters["cancelled_total"] > 0
        code, out = _post(srv.port, "/v1/completions", {"model": model, "prompt": "still there?", "max_tokens": 3})
        assert code == 200 and out["usage"]["completion_tokens"] == 3
    finally:
        srv.shutdown()


def test_stop_sequences_end_a_completion_early_and_finish_reasons_are_reported(server):
    """``stop``: the text ends before the first stop sequence, the engine stops generating for the
    request (far fewer tokens than max_tokens), streamed chunks never contain the sequence or
    anything after it; without a stop sequence a completion that uses its whole budget says "length"."""
    srv, ens = server
    model = next(iter(ens.backends))
    code, full = _post(srv.port, "/v1/completions", {"model": model, "prompt": "count for me", "max_tokens": 40, "temperature": 0.0})
    text = full["choices"][0]["text"]
    assert code == 200 and full["choices"][0]["finish_reason"] == "length" and full["usage"]["completion_tokens"] == 40
    assert len(text) > 12
    stop = text[8:11]                                 # three characters the model will produce again (greedy)
    cut = text.find(stop)
    code, out = _post(srv.port, "/v1/completions", {"model": model, "prompt": "count for me", "max_tokens": 40,
                                                    "temperature": 0.0, "stop": stop})
    assert code == 200 and out["choices"][0]["text"] == text[:cut] and out["choices"][0]["finish_reason"] == "stop"
    assert out["usage"]["completion_tokens"] < 40 and out["usage"]["prompt_tokens"] == full["usage"]["prompt_tokens"]
    # streamed: the pieces add up to the same truncated text
    req = urllib.request.Request(f"http://127.0.0.1:{srv.port}/v1/completions",
                                 json.dumps({"model": model, "prompt": "count for me", "max_tokens": 40, "temperature": 0.0,
                                             "stop": [stop, "\\u0000never"], "stream": True}).encode(),
                                 {"Content-Type": "application/json"})
    pieces, reasons = [], []
    with urllib.request.urlopen(req, timeout=120) as r:
        for line in r.read().decode().splitlines():
            if line.startswith("data: {"):
                c = json.loads(line[6:])["choices"][0]
                pieces.append(c.get("text", ""))
                if c.get("finish_reason"):
                    reasons.append(c["finish_reason"])
    assert "".join(pieces) == text[:cut] and reasons == ["stop"]
    with pytest.raises(urllib.error.HTTPError) as bad:
        _post(srv.port, "/v1/completions", {"model": model, "prompt": "x", "stop": ["a", "b", "c", "d", "e"]})
    assert bad.value.code == 400
    eng = ens.backends[model].engine
    deadline = time.time() + 20
    while time.time() < deadline and (eng.stats()["sched.running"] or ens.backends[model]._discard):
        time.sleep(0.05)
    assert eng.stats()["sched.running"] == 0 and not ens.backends[model]._discard


def test_score_route_and_decision_targets_agree_with_the_backend(server):
    """``POST /v1/score`` returns what ``Backend.submit_score`` returns for the same prompt and
    targets; ``decision_targets`` on a completion attaches the log-probabilities of the targets at
    the position after each choice's reply; both reject malformed requests with 400 / 404."""
    srv, ens = server
    model = next(iter(ens.backends))
    targets = [" Good", " NeedsRefinement", " A"]
    code, out = _post(srv.port, "/v1/score", {"model": model, "prompt": "Is this fine?\n", "targets": targets})
    assert code == 200 and out["object"] == "score" and out["targets"] == targets and len(out["logprobs"]) == 3
    assert all(lp <= 0.0 for lp in out["logprobs"])
    # the same call once more gives the same numbers (greedy scoring has no randomness) ...
    code, again = _post(srv.port, "/v1/score", {"model": model, "prompt": "Is this fine?\n", "targets": targets})
    assert again["logprobs"] == pytest.approx(out["logprobs"], abs=1e-5)
    # ... and a different prompt does not
    code, other = _post(srv.port, "/v1/score", {"model": model, "prompt": "Something else entirely", "targets": targets})
    assert other["logprobs"] != pytest.approx(out["logprobs"], abs=1e-5)

    code, gen = _post(srv.port, "/v1/completions", {"model": model, "prompt": "pick one", "max_tokens": 5, "n": 2,
                                                    "temperature": 0.7, "seed": 5, "decision_targets": [" A", " B"]})
    assert code == 200 and len(gen["choices"]) == 2
    for c in gen["choices"]:
        assert len(c["decision_logprobs"]) == 2 and all(lp <= 0.0 for lp in c["decision_logprobs"])
    # (the decision is taken on the reply's TOKENS; re-scoring prompt + reply text would re-tokenise
    # a random-init model's bytes differently, so there is no exact cross-check through the text)
    code, plain = _post(srv.port, "/v1/completions", {"model": model, "prompt": "pick one", "max_tokens": 5})
    assert "decision_logprobs" not in plain["choices"][0]

    for path, body, want in (
        ("/v1/score", {"model": "nobody", "prompt": "x", "targets": ["a"]}, 404),
        ("/v1/score", {"model": model, "prompt": "", "targets": ["a"]}, 400),
        ("/v1/score", {"model": model, "prompt": "x", "targets": []}, 400),
        ("/v1/score", {"model": model, "prompt": "x", "targets": "a"}, 400),
        ("/v1/completions", {"model": model, "prompt": "x", "decision_targets": [""]}, 400),
        ("/v1/completions", {"model": model, "prompt": "x", "decision_targets": [" A"], "stream": True}, 400),
        ("/v1/chat/completions", {"messages": [{"role": "user", "content": "q"}], "decision_targets": [" A"]}, 400),
    ):
        with pytest.raises(urllib.error.HTTPError) as bad:
            _post(srv.port, path, body)
        assert bad.value.code == want, (path, body)


def test_malformed_and_stalling_requests_cost_the_server_nothing():
    """Bad Content-Length -> 400, an announced body over the limit -> 413 without reading it, a body
    that is not a JSON object -> 400, and a client that sends half a request and goes quiet gets its
    connection closed after the socket timeout; the server answers a normal request afterwards."""
    import socket

    cfg = config_from_dict({"strategy": "majority_vote", "sampling": {"temperature": 0.0, "max_new_tokens": 2},
                            "models": {"default": {"preset": "tiny-llama", "seed": 1}},
                            "members": [{"name": "a", "model": "default"}]})
    ens = build_ensemble(cfg)
    srv = ConsensusServer(ens, port=0, io_timeout_s=0.5, max_body_bytes=4096).start()
    try:
        def raw(request: bytes, wait=5.0) -> bytes:
            with socket.create_connection(("127.0.0.1", srv.port), timeout=wait) as c:
                c.sendall(request)
                out = b""
                try:
                    while True:
                        b = c.recv(65536)
                        if not b:
                            break
                        out += b
                except socket.timeout:
                    pass
                return out

        head = b"POST /v1/completions HTTP/1.1\r\nHost: x\r\nContent-Type: application/json\r\n"
        assert raw(head + b"Content-Length: banana\r\n\r\n").startswith(b"HTTP/1.1 400")
        assert raw(head + b"Content-Length: -5\r\n\r\n").startswith(b"HTTP/1.1 400")
        assert raw(head + b"Content-Length: 999999999\r\n\r\n").startswith(b"HTTP/1.1 413")      # nothing of the body sent
        assert raw(head + b"Content-Length: 7\r\n\r\n[1,2,3]").startswith(b"HTTP/1.1 400")
        assert raw(head + b"Content-Length: 4\r\n\r\n\xff\xfe\x00\x01").startswith(b"HTTP/1.1 400")
        # half a request, then silence: the server closes the connection by itself
        t0 = time.perf_counter()
        got = raw(head + b"Content-Length: 100\r\n\r\n{\"model\":", wait=10.0)
        assert time.perf_counter() - t0 < 5.0 and not got.startswith(b"HTTP/1.1 200")
        # a connection that never sends a byte
        t0 = time.perf_counter()
        assert raw(b"", wait=10.0) == b"" and time.perf_counter() - t0 < 5.0
        code, out = _post(srv.port, "/v1/completions", {"model": "default", "prompt": "still there?", "max_tokens": 2})
        assert code == 200 and len(out["choices"]) == 1
        assert threading.active_count() < 20          # no handler threads left behind
    finally:
        srv.shutdown()


def test_a_stream_reader_that_stalls_is_cut_off_and_its_work_cancelled(monkeypatch):
    """A client that asks for a token stream and then stops reading: once the socket buffers are
    full a write times out (io_timeout_s). The request is cancelled and the engine lets it go; the
    server keeps answering others. (Loopback buffers hold more than a tiny model's context can
    produce, so the timed-out write is injected at the first chunk.)"""
    cfg = config_from_dict({"strategy": "majority_vote", "sampling": {"temperature": 0.0, "max_new_tokens": 4},
                            "models": {"default": {"preset": "tiny-llama", "seed": 1, "kv_blocks": 512}},
                            "members": [{"name": "a", "model": "default"}]})
    ens = build_ensemble(cfg)
    srv = ConsensusServer(ens, port=0).start()
    eng = ens.backends["default"].engine
    real = ConsensusServer._stream_tokens_body

    def stalled(self, h, job, emit, started, deadline, chat):
        item = job.chunks.get(timeout=30)            # generation is under way
        assert item is not None
        raise TimeoutError("timed out")              # what socket.sendall raises when the reader has stalled

    monkeypatch.setattr(ConsensusServer, "_stream_tokens_body", stalled)
    try:
        req = urllib.request.Request(f"http://127.0.0.1:{srv.port}/v1/completions",
                                     json.dumps({"model": "default", "prompt": "tell me everything", "max_tokens": 400,
                                                 "stream": True}).encode(), {"Content-Type": "application/json"})
        try:
            urllib.request.urlopen(req, timeout=60).read()
        except Exception:  # noqa: BLE001  the connection is cut; how the client sees that does not matter here
            pass
        deadline = time.time() + 30
        while time.time() < deadline and eng.stats().get("requests.cancelled", 0) < 1:
            time.sleep(0.05)
        st = eng.stats()
        assert st.get("requests.cancelled", 0) == 1 and st["decode.tokens"] < 400, st
        monkeypatch.setattr(ConsensusServer, "_stream_tokens_body", real)
        code, out = _post(srv.port, "/v1/completions", {"model": "default", "prompt": "and you?", "max_tokens": 2})
        assert code == 200
        deadline = time.time() + 20
        while time.time() < deadline and eng.stats()["sched.running"]:
            time.sleep(0.05)
        assert eng.stats()["sched.running"] == 0
    finally:
        srv.shutdown()


def test_wrongly_typed_fields_get_400_never_a_dropped_connection():
    """Every field of every POST route set to values of the wrong type or range (null, numbers,
    strings, lists, objects, booleans, huge integers): the answer is 200, 400 or 404 with a JSON
    body; no request costs the client its connection or the server a handler thread."""
    cfg = config_from_dict({"strategy": "majority_vote", "sampling": {"temperature": 0.0, "max_new_tokens": 2},
                            "models": {"default": {"preset": "tiny-llama", "seed": 1}},
                            "members": [{"name": "a", "model": "default"}]})
    srv = ConsensusServer(build_ensemble(cfg), port=0, request_timeout_s=30).start()

    def post(path, body):
        req = urllib.request.Request(f"http://127.0.0.1:{srv.port}{path}", json.dumps(body).encode(),
                                     {"Content-Type": "application/json"})
        try:
            with urllib.request.urlopen(req, timeout=60) as r:
                r.read()                      # JSON, or server-sent events where "stream" happened to be true
                return r.status
        except urllib.error.HTTPError as e:
            assert "error" in json.loads(e.read())
            return e.code

    weird = [None, 5, -1, 1.5, "x", "", [], [1], {}, {"a": 1}, True, 10 ** 12]
    try:
        for path, base, keys in (
            ("/v1/chat/completions", {"messages": [{"role": "user", "content": "hi"}]},
             ["messages", "model", "n", "max_tokens", "temperature", "top_p", "top_k", "seed", "stop", "stream"]),
            ("/v1/completions", {"model": "default", "prompt": "hi", "max_tokens": 2},
             ["prompt", "model", "n", "max_tokens", "temperature", "top_p", "top_k", "seed", "stop", "stream", "decision_targets"]),
            ("/v1/embeddings", {"model": "default", "input": "hi"}, ["input", "model"]),
            ("/v1/score", {"model": "default", "prompt": "hi", "targets": ["a"]}, ["prompt", "targets", "model"]),
        ):
            for k in keys:
                for w in weird:
                    assert post(path, dict(base, **{k: w})) in (200, 400, 404), (path, k, w)
        for m in ([1], [{"role": 5, "content": 7}], [{"content": None}], [{"role": "user"}],
                  [{"role": "user", "content": [{"text": 5}]}], [[]], [{"role": "user", "content": [{"type": "text", "text": "ok"}]}]):
            assert post("/v1/chat/completions", {"messages": m}) in (200, 400), m
        assert srv.counters["errors_total"] > 50 and threading.active_count() < 20
    finally:
        srv.shutdown()


def test_connections_beyond_the_cap_are_told_503_without_a_thread_of_their_own():
    import socket

    cfg = config_from_dict({"strategy": "majority_vote", "sampling": {"temperature": 0.0, "max_new_tokens": 2},
                            "models": {"default": {"preset": "tiny-llama", "seed": 1}},
                            "members": [{"name": "a", "model": "default"}]})
    srv = ConsensusServer(build_ensemble(cfg), port=0, max_connections=4, io_timeout_s=5.0).start()
    idle = []
    try:
        for _ in range(4):                              # four connections that just sit there
            c = socket.create_connection(("127.0.0.1", srv.port), timeout=5)
            idle.append(c)
        deadline = time.time() + 5
        while time.time() < deadline and srv.httpd.active < 4:
            time.sleep(0.02)
        with pytest.raises(urllib.error.HTTPError) as e:
            _post(srv.port, "/v1/completions", {"model": "default", "prompt": "hi", "max_tokens": 2}, timeout=10)
        assert e.value.code == 503 and e.value.headers.get("Retry-After") == "1"
        assert "too many" in json.loads(e.value.read())["error"]["message"]
        for c in idle[:2]:
            c.close()
        deadline = time.time() + 10
        while time.time() < deadline and srv.httpd.active > 2:
            time.sleep(0.05)
        code, out = _post(srv.port, "/v1/completions", {"model": "default", "prompt": "hi", "max_tokens": 2})
        assert code == 200
    finally:
        for c in idle:
            c.close()
        srv.shutdown()
