"""Front ends on CPU: config files -> ensemble, the REPL, the OpenAI-compatible endpoint
and the placement planner. The REPL contract follows /root/reference/src/main.rs:428-471
(prompt text, ``exit``, reset between questions)."""
import io
import json
import threading
import urllib.error
import urllib.request
from pathlib import Path

import pytest
import torch

from llm_consensus_amd import cli
from llm_consensus_amd.app import build_ensemble
from llm_consensus_amd.config import ConsensusConfig, load_config
from llm_consensus_amd.consensus.backend import ScriptedBackend
from llm_consensus_amd.parallel.placement import plan_for_config
from llm_consensus_amd.serve.server import ConsensusServer

CONFIGS = Path(__file__).resolve().parent.parent / "configs"
CPU = torch.device("cpu")


@pytest.fixture(scope="module")
def tiny_ensemble():
    return build_ensemble(load_config(str(CONFIGS / "plumbing_tiny.toml")), CPU)


def test_every_example_config_loads():
    for f in sorted(CONFIGS.glob("*.toml")):
        cfg = load_config(str(f))
        assert cfg.members, f.name
        for m in cfg.members:
            assert m.model in cfg.models


def test_default_config_is_reference_topology():
    cfg = ConsensusConfig()
    assert [m.name for m in cfg.members] == ["High Society", "The Technician", "Art Boy", "Programming Nerd"]
    assert cfg.strategy == "reference_protocol" and cfg.max_rounds == 5
    assert len({m.model for m in cfg.members}) == 1


def test_plumbing_majority_vote_three_tiny_models(tiny_ensemble):
    res = tiny_ensemble.ask_many(["Which letter comes first? A) a B) b C) c D) d", "2+2? A) 3 B) 4 C) 5 D) 6"])
    assert len(res) == 2
    for r in res:
        assert r.strategy == "majority_vote" and r.answer in list("ABCD")
        assert len(r.votes) == 3 and r.member_calls == 3
        counts = [sum(v["class"] == c for v in r.votes) for c in range(4)]
        assert counts["ABCD".index(r.answer)] == max(counts)
    # deterministic: the same question gives the same votes again
    again = tiny_ensemble.ask(res[0].question)
    assert [v["class"] for v in again.votes] == [v["class"] for v in res[0].votes]


def scripted_reference_ensemble(**kw):
    be = ScriptedBackend({"propose": ["first draft"], "evaluate": ["NeedsRefinement\nmeh"] + ["Good\nfine"] * 16,
                          "refine": ["second draft"]}, **kw)
    return build_ensemble(ConsensusConfig(), CPU, backends={"default": be})


def test_repl_prompt_exit_and_answer():
    ens = scripted_reference_ensemble()
    out = io.StringIO()
    n = cli.repl(ens, io.StringIO("What is art?\n\nexit\nnever read\n"), out)
    text = out.getvalue()
    assert n == 1
    assert text.count(cli.PROMPT) == 3          # question, blank line, exit
    assert "second draft" in text or "first draft" in text
    assert "never read" not in text


def test_repl_eof_ends_loop_and_json_mode():
    ens = scripted_reference_ensemble()
    out = io.StringIO()
    assert cli.repl(ens, io.StringIO("q one\n"), out, as_json=True) == 1
    rec = json.loads(out.getvalue().split(cli.PROMPT)[1].strip().splitlines()[0])
    assert rec["question"] == "q one" and rec["strategy"] == "reference_protocol" and rec["rounds"] >= 1


def test_repl_survives_a_failing_session():
    class Boom:
        def ask(self, q):
            raise RuntimeError("engine fell over")

    out = io.StringIO()
    assert cli.repl(Boom(), io.StringIO("q\nexit\n"), out) == 0
    assert "System error: engine fell over" in out.getvalue()


def test_cli_questions_file(tmp_path, capsys):
    qf = tmp_path / "q.txt"
    qf.write_text("one? A) x B) y C) z D) w\n\nsecond? A) x B) y C) z D) w\n")
    rc = cli.main(["--config", str(CONFIGS / "plumbing_tiny.toml"), "--questions", str(qf), "--json", "--device", "cpu"])
    assert rc == 0
    lines = [json.loads(x) for x in capsys.readouterr().out.strip().splitlines()]
    assert len(lines) == 2 and all(x["answer"] in "ABCD" for x in lines)


def test_cli_warns_when_answers_come_from_random_weights(tmp_path, capsys):
    """The reference refuses to start without a usable backend (main.rs:354-357); a model
    with no checkpoint path still runs here, but not silently."""
    qf = tmp_path / "q.txt"
    qf.write_text("one? A) x B) y C) z D) w\n")
    cli.main(["--config", str(CONFIGS / "plumbing_tiny.toml"), "--questions", str(qf), "--device", "cpu"])
    err = capsys.readouterr().err
    assert "RANDOM-INIT" in err and "m0" in err


def test_removed_config_keys_are_rejected():
    from llm_consensus_amd.config import config_from_dict

    with pytest.raises(ValueError, match="proposer_policy"):
        config_from_dict({"proposer_policy": "all"})
    with pytest.raises(ValueError, match="unknown ModelSpec keys"):
        config_from_dict({"models": {"default": {"dtype": "fp8"}}})
    with pytest.raises(ValueError, match="admission"):
        config_from_dict({"models": {"default": {"admission": "greedy"}}})


# ---------------------------------------------------------------- HTTP endpoint
def _post(port, path, body, timeout=120):
    req = urllib.request.Request(f"http://127.0.0.1:{port}{path}", json.dumps(body).encode(),
                                 {"Content-Type": "application/json"})
    with urllib.request.urlopen(req, timeout=timeout) as r:
        return r.status, r.read()


def _get(port, path):
    with urllib.request.urlopen(f"http://127.0.0.1:{port}{path}", timeout=30) as r:
        return r.status, r.read()


@pytest.fixture(scope="module")
def server(tiny_ensemble):
    srv = ConsensusServer(tiny_ensemble, port=0, max_batch=16, batch_window_ms=30).start()
    yield srv
    srv.shutdown()


def test_http_health_models_metrics(server):
    st, body = _get(server.port, "/health")
    assert st == 200 and json.loads(body)["status"] == "ok"
    ids = [m["id"] for m in json.loads(_get(server.port, "/v1/models")[1])["data"]]
    assert ids == ["consensus", "m0", "m1", "m2"]
    assert b"consensus_http_requests_total" in _get(server.port, "/metrics")[1]


def test_http_chat_consensus_matches_direct_call(server, tiny_ensemble):
    q = "Pick one. A) red B) green C) blue D) grey"
    st, body = _post(server.port, "/v1/chat/completions",
                     {"model": "consensus", "messages": [{"role": "system", "content": "be brief"},
                                                         {"role": "user", "content": q}]})
    obj = json.loads(body)
    assert st == 200 and obj["object"] == "chat.completion"
    direct = tiny_ensemble.ask(q)
    assert obj["choices"][0]["message"]["content"] == direct.answer
    assert [v["class"] for v in obj["consensus"]["votes"]] == [v["class"] for v in direct.votes]


def test_http_concurrent_requests_are_batched(server):
    before = dict(server.counters)
    results = [None] * 8

    def call(i):
        results[i] = json.loads(_post(server.port, "/v1/chat/completions",
                                      {"messages": [{"role": "user", "content": f"Q{i}? A) a B) b C) c D) d"}]})[1])

    ts = [threading.Thread(target=call, args=(i,)) for i in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert all(r and r["choices"][0]["message"]["content"] in list("ABCD") for r in results)
    batches = server.counters["batches_total"] - before["batches_total"]
    assert batches < 8, "concurrent callers should share engine batches"


def test_http_member_model_completion_and_stream(server):
    st, body = _post(server.port, "/v1/completions", {"model": "m1", "prompt": "hello", "max_tokens": 5, "n": 2,
                                                      "temperature": 0.9, "seed": 7})
    obj = json.loads(body)
    assert st == 200 and len(obj["choices"]) == 2 and obj["usage"]["completion_tokens"] == 10
    st, raw = _post(server.port, "/v1/chat/completions",
                    {"model": "m0", "stream": True, "max_tokens": 4, "messages": [{"role": "user", "content": "hi"}]})
    events = [l[6:] for l in raw.decode().splitlines() if l.startswith("data: ")]
    assert events[-1] == "[DONE]"
    chunks = [json.loads(e) for e in events[:-1]]
    assert chunks[0]["choices"][0]["delta"]["role"] == "assistant"
    # the whole budget was used: "length" (a stop token or stop sequence would say "stop")
    assert chunks[-1]["choices"][0]["finish_reason"] == "length" and chunks[-1]["usage"]["completion_tokens"] == 4
    # tokens leave as they are sampled: several content chunks, and together they are the completion
    pieces = [c["choices"][0]["delta"].get("content", "") for c in chunks if c["choices"][0]["finish_reason"] is None]
    assert len(pieces) >= 2
    st, body = _post(server.port, "/v1/chat/completions",
                     {"model": "m0", "max_tokens": 4, "messages": [{"role": "user", "content": "hi"}]})
    assert "".join(pieces) == json.loads(body)["choices"][0]["message"]["content"]
    # text-completion flavour of the same stream
    st, raw = _post(server.port, "/v1/completions", {"model": "m1", "prompt": "hello", "max_tokens": 3, "stream": True})
    ev = [json.loads(l[6:]) for l in raw.decode().splitlines() if l.startswith("data: ") and l != "data: [DONE]"]
    assert ev[-1]["choices"][0]["finish_reason"] == "length" and len(ev) >= 2


def test_consensus_responses_report_token_usage(server, tiny_ensemble):
    """Usage of a consensus answer = the tokens of every member call behind it."""
    st, body = _post(server.port, "/v1/chat/completions",
                     {"messages": [{"role": "user", "content": "Pick. A) a B) b C) c D) d"}]})
    u = json.loads(body)["usage"]
    assert u["prompt_tokens"] > 3 * 10 and u["completion_tokens"] > 0
    assert u["total_tokens"] == u["prompt_tokens"] + u["completion_tokens"]
    direct = tiny_ensemble.ask("Pick. A) a B) b C) c D) d")
    assert direct.usage["prompt_tokens"] == u["prompt_tokens"]


def test_http_embeddings(server):
    st, body = _post(server.port, "/v1/embeddings", {"model": "m0", "input": ["alpha", "beta"]})
    data = json.loads(body)["data"]
    assert st == 200 and len(data) == 2
    v = torch.tensor(data[0]["embedding"])
    assert v.numel() == 128 and abs(float(v.norm()) - 1.0) < 1e-2


def test_http_errors(server):
    for path, body, code in [("/v1/chat/completions", {"model": "nope", "messages": [{"role": "user", "content": "x"}]}, 404),
                             ("/v1/chat/completions", {"messages": []}, 400),
                             ("/v1/completions", {"prompt": ""}, 400),
                             ("/v1/nothing", {}, 404)]:
        with pytest.raises(urllib.error.HTTPError) as ei:
            _post(server.port, path, body)
        assert ei.value.code == code
        assert "error" in json.loads(ei.value.read())


# ---------------------------------------------------------------- placement
def test_placement_three_small_models_pack_to_one_gpu_and_replicate():
    cfg = load_config(str(CONFIGS / "ensemble_3x8b.toml"))
    plan = plan_for_config(cfg, 8)
    assert plan.feasible and plan.gpus_per_replica == 1 and len(plan.replicas) == 8
    rep = plan.replicas[5]
    assert list(rep.devices) == [5] and sorted({cfg.models[m.model].name for m in cfg.members}) == ["llama", "mistral", "qwen"]
    assert all(list(m.devices) == [5] for m in rep.members)


def test_placement_70b_judge_spread():
    cfg = load_config(str(CONFIGS / "judge_70b_tp4.toml"))
    plan = plan_for_config(cfg, 8)
    assert plan.feasible
    rep = plan.replicas[0]
    judge = next(m for m in rep.members if m.name == "judge")
    assert len(judge.devices) == 4 and plan.gpus_per_replica == 8      # 4 proposers on a GPU each + TP=4 judge
    assert len({d for m in rep.members for d in m.devices}) == 8
    proposer_gpus = [d for m in rep.members if m.name != "judge" for d in m.devices]
    assert len(proposer_gpus) == 4 and not set(proposer_gpus) & set(judge.devices)


def test_placement_infeasible_reports_error():
    cfg = load_config(str(CONFIGS / "judge_70b_tp4.toml"))
    plan = plan_for_config(cfg, 2)
    assert not plan.feasible and plan.error


def test_ensemble_metrics_sections(tiny_ensemble):
    before = tiny_ensemble.stats()["consensus"].get("answers", 0)
    tiny_ensemble.ask_many(["One? A) a B) b C) c D) d", "Two? A) a B) b C) c D) d"])
    st = tiny_ensemble.stats()
    c = st["consensus"]
    assert c["answers"] == before + 2 and c["answers_per_s"] > 0
    assert c["latency_ms.count"] >= 2 and c["rounds.mean"] == 1.0 and 0 < c["agreement.mean"] <= 1
    assert any(k.startswith("decode_tokens_per_s.") for k in c)
    assert "kv.prefix_hit_rate" in st["m0"]


def test_packaging_manifest_names_real_entry_points():
    """pyproject.toml is this build's Cargo.toml (/root/reference/Cargo.toml:1-14): the console
    scripts it declares must resolve to callables, and the build hook must know both native targets."""
    import importlib

    import tomli

    root = CONFIGS.parent
    meta = tomli.loads((root / "pyproject.toml").read_text())
    assert meta["project"]["name"] == "llm-consensus-amd"
    scripts = meta["project"]["scripts"]
    assert set(scripts) == {"llm-consensus", "llm-consensus-serve", "llm-consensus-bench"}
    for target in scripts.values():
        mod, fn = target.split(":")
        assert callable(getattr(importlib.import_module(mod), fn))
    setup_src = (root / "setup.py").read_text()
    assert "build_core" in setup_src and "build_kernels" in setup_src
    # the package itself must be importable without the hyphenated compatibility link
    assert importlib.import_module("llm_consensus_amd.workloads").make_questions


def test_toml_configs_load_with_the_standard_library_parser(monkeypatch):
    """Python 3.11+ has no ``tomli`` unless it is installed separately; the loader must use
    ``tomllib`` there. Simulated by offering ``tomllib`` and hiding ``tomli``."""
    import builtins
    import sys
    import types

    import tomli

    fake = types.ModuleType("tomllib")
    fake.loads = tomli.loads
    monkeypatch.setitem(sys.modules, "tomllib", fake)
    real_import = builtins.__import__

    def no_tomli(name, *a, **k):
        if name == "tomli":
            raise ModuleNotFoundError("No module named 'tomli'")
        return real_import(name, *a, **k)

    monkeypatch.setattr(builtins, "__import__", no_tomli)
    for f in sorted(CONFIGS.glob("*.toml")):
        assert load_config(str(f)).members, f.name


def test_sigterm_drains_the_server_admitted_requests_are_answered_new_ones_refused(tmp_path):
    """`llm-consensus-serve` under SIGTERM: the request in flight gets its 200, a request that
    arrives during the drain gets 503 with Retry-After, and the process exits by itself."""
    import json as _json
    import signal
    import socket
    import subprocess
    import sys
    import threading
    import time
    import urllib.error
    import urllib.request
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    cfg = tmp_path / "slow.toml"
    cfg.write_text('strategy = "majority_vote"\n[sampling]\ntemperature = 0.0\nmax_new_tokens = 300\n'
                   '[models.default]\npreset = "tiny-llama"\nseed = 1\nkv_blocks = 512\n')
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import os
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(PYTHONPATH=str(root), OMP_NUM_THREADS="2")
    proc = subprocess.Popen([sys.executable, "-m", "llm_consensus_amd.serve.server", "--config", str(cfg), "--device", "cpu",
                             "--gpus", "1", "--port", str(port)], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                            text=True)

    def post(body, timeout=120):
        req = urllib.request.Request(f"http://127.0.0.1:{port}/v1/completions", _json.dumps(body).encode(),
                                     {"Content-Type": "application/json"})
        with urllib.request.urlopen(req, timeout=timeout) as r:
            return r.status, _json.loads(r.read())

    try:
        for _ in range(600):
            try:
                with urllib.request.urlopen(f"http://127.0.0.1:{port}/health", timeout=2) as r:
                    if r.status == 200:
                        break
            except OSError:
                assert proc.poll() is None, proc.stderr.read()[-2000:]
                time.sleep(0.1)
        got = {}
        t = threading.Thread(target=lambda: got.update(slow=post({"model": "default", "prompt": "long one", "max_tokens": 300})))
        t.start()
        time.sleep(0.5)                              # the slow request is being generated
        proc.send_signal(signal.SIGTERM)
        time.sleep(0.3)
        try:
            post({"model": "default", "prompt": "late", "max_tokens": 3}, timeout=10)
            late = 200
        except urllib.error.HTTPError as e:
            late = e.code
            assert e.headers.get("Retry-After") == "1"
        except OSError:
            late = "closed"
        t.join(60)
        assert got["slow"][0] == 200 and got["slow"][1]["usage"]["completion_tokens"] == 300
        assert late in (503, "closed")
        assert proc.wait(30) == 0
    finally:
        if proc.poll() is None:
            proc.kill()


def test_bad_config_values_are_refused_with_a_message_that_names_them():
    """A value that would fail deep inside an engine (or silently fall back to a default) is an
    error at load time; every shipped config passes the same checks."""
    import glob

    from llm_consensus_amd.config import config_from_dict, load_config

    for f in sorted(glob.glob(str(Path(__file__).resolve().parents[1] / "configs" / "*"))):
        load_config(f)
    for bad, word in (
        ({"sampling": {"temperature": "hot"}}, "temperature"), ({"sampling": {"max_new_tokens": -3}}, "max_new_tokens"),
        ({"sampling": {"top_p": 0}}, "top_p"), ({"models": {"default": {"preset": "tiny-llama", "tp": 0}}}, "tp"),
        ({"models": {"default": {"preset": "tiny-llama", "kv_block_size": 7}}}, "kv_block_size"),
        ({"members": [{"name": "a", "model": "default", "role": "wizard"}]}, "role"),
        ({"strategy": "self_consistency", "k_chains": 0}, "k_chains"), ({"tie_break": "coin"}, "tie_break"),
        ({"placement": "everywhere"}, "placement"), ({"members": [{"name": "a", "model": "default", "weight": -1}]}, "weight"),
        ({"choices": "A"}, "choices"), ({"deadline_s": -1}, "deadline_s"),
        ({"models": {"default": {"preset": "tiny-llama", "chat_format": "weird"}}}, "chat_format"),
        ({"models": {"default": {"preset": "tiny-llama", "ep_mode": "x"}}}, "ep_mode"),
        ({"models": {"default": {"preset": "tiny-llama", "kv_fraction": 1.5}}}, "kv_fraction"),
    ):
        with pytest.raises(ValueError, match=word):
            config_from_dict(bad)


def test_command_line_mistakes_end_with_one_line_and_status_2(tmp_path, capsys):
    """A missing or malformed config, a value the checks refuse, an unreadable --questions file: one
    line on stderr, exit status 2, and no model is loaded for it."""
    from llm_consensus_amd import cli
    from llm_consensus_amd.serve import server

    bad = tmp_path / "bad.toml"
    bad.write_text('strategy = "nonsense"\n')
    broken = tmp_path / "broken.toml"
    broken.write_text("strategy = \n")
    for argv, word in ((["--config", str(bad)], "unknown strategy"), (["--config", str(broken)], "configuration"),
                       (["--config", str(tmp_path / "missing.toml")], "No such file")):
        for main in (cli.main, server.main):
            with pytest.raises(SystemExit) as e:
                main(argv + ["--device", "cpu"])
            assert e.value.code == 2
            err = capsys.readouterr().err
            assert word in err and "Traceback" not in err and len(err.strip().splitlines()) == 1
    assert cli.main(["--device", "cpu", "--questions", str(tmp_path / "none.txt")]) == 2
    assert "cannot read --questions" in capsys.readouterr().err
    assert cli.main(["--device", "cpu", "--strategy", "nope"]) == 2


def test_server_finds_out_that_its_port_is_taken_before_loading_anything(capsys):
    import socket
    import time

    from llm_consensus_amd.serve import server

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        s.listen(1)
        t0 = time.perf_counter()
        assert server.main(["--device", "cpu", "--port", str(s.getsockname()[1])]) == 2
        assert "cannot listen" in capsys.readouterr().err and time.perf_counter() - t0 < 2.0


def test_prompt_template_files_must_keep_their_placeholders(tmp_path):
    from llm_consensus_amd.config import config_from_dict

    good = tmp_path / "propose.txt"
    good.write_text("Please answer briefly.\n{question}\n")
    bad = tmp_path / "evaluate.txt"
    bad.write_text("Judge this: {question}\n")                   # no {answer}
    cfg = config_from_dict({"prompt_propose": str(good)})
    assert "{question}" in cfg.templates().propose and "briefly" in cfg.templates().propose
    with pytest.raises(ValueError, match=r"prompt_evaluate.*\{answer\}"):
        config_from_dict({"prompt_evaluate": str(bad)})                  # at load time, before any model is built
    with pytest.raises(OSError):
        config_from_dict({"prompt_refine": str(tmp_path / "missing.txt")})
