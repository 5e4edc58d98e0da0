"""Front ends that start their own ranks: ``llm-consensus --gpus N`` and ``llm-consensus-serve
--gpus N`` spawn one process per rank (CPU processes on gloo here), rank 0 owns the terminal or
the socket, the others follow it in lock-step rounds. The reference's ``main`` brings up all of
its members itself (/root/reference/src/main.rs:350-426); a user of it should not have to learn
a launcher to use several GPUs."""
import json
import os
import signal
import socket
import subprocess
import sys
import time
import urllib.request
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CONFIGS = ROOT / "configs"
QUESTIONS = ["Which letter comes first? A) a B) b C) c D) d", "2+2? A) 3 B) 4 C) 5 D) 6", "Pick. A) w B) x C) y D) z"]

SPREAD = """
strategy = "majority_vote"
placement = "spread"
[sampling]
temperature = 0.0
max_new_tokens = 6
[models.m0]
preset = "tiny-llama"
seed = 1
[models.m1]
preset = "tiny-qwen"
seed = 2
[models.m2]
preset = "tiny-mistral"
seed = 3
[[members]]
name = "a"
model = "m0"
[[members]]
name = "b"
model = "m1"
[[members]]
name = "c"
model = "m2"
"""


def env():
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(PYTHONPATH=str(ROOT), OMP_NUM_THREADS="2")
    return e


def repl(cfg, gpus, extra=()):
    p = subprocess.run([sys.executable, "-m", "llm_consensus_amd.cli", "--config", str(cfg), "--device", "cpu",
                        "--gpus", str(gpus), "--json", *extra], input="\n".join(QUESTIONS) + "\nexit\n",
                       capture_output=True, text=True, env=env(), timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    recs = [json.loads(l[l.index("{"):]) for l in p.stdout.splitlines() if "{" in l]
    return [(r["question"], r["answer"], [v["class"] for v in r["votes"]]) for r in recs]


@pytest.fixture(scope="module")
def spread_cfg(tmp_path_factory):
    f = tmp_path_factory.mktemp("cfg") / "spread.toml"
    f.write_text(SPREAD)
    return f


def test_repl_starts_its_own_ranks_two_and_three(spread_cfg):
    one = repl(spread_cfg, 1)
    assert [q for q, _, _ in one] == QUESTIONS
    # three ranks: one model per rank, served in lock-step rounds; and as closed waves
    assert repl(spread_cfg, 3) == one
    assert repl(spread_cfg, 3, ("--batched",)) == one
    # two ranks with the packed example config: two replicas of the whole ensemble
    packed = CONFIGS / "plumbing_tiny.toml"
    assert repl(packed, 2) == repl(packed, 1)


def test_single_process_says_so_when_the_plan_wants_more_gpus(spread_cfg):
    p = subprocess.run([sys.executable, "-m", "llm_consensus_amd.cli", "--config", str(spread_cfg), "--gpus", "1",
                        "--no-warm-up"], input="exit\n", capture_output=True, text=True, env=env(), timeout=300)
    assert p.returncode == 0 and "planned for 3 GPUs" in p.stderr and "--gpus 3" in p.stderr


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _post(port, body, timeout=120):
    req = urllib.request.Request(f"http://127.0.0.1:{port}/v1/chat/completions", json.dumps(body).encode(),
                                 {"Content-Type": "application/json"})
    with urllib.request.urlopen(req, timeout=timeout) as r:
        return json.loads(r.read())


@pytest.mark.parametrize("gpus", [2, 3])
def test_http_server_starts_its_own_ranks(spread_cfg, gpus):
    cfg = spread_cfg if gpus == 3 else CONFIGS / "plumbing_tiny.toml"
    want = {q: a for q, a, _ in repl(cfg, 1)}
    port = _free_port()
    proc = subprocess.Popen([sys.executable, "-m", "llm_consensus_amd.serve.server", "--config", str(cfg), "--device", "cpu",
                             "--gpus", str(gpus), "--port", str(port)], env=env(), stdout=subprocess.DEVNULL,
                            stderr=subprocess.PIPE, text=True)
    try:
        for _ in range(600):
            try:
                with urllib.request.urlopen(f"http://127.0.0.1:{port}/health", timeout=2) as r:
                    if r.status == 200:
                        break
            except OSError:
                assert proc.poll() is None, proc.stderr.read()[-3000:]
                time.sleep(0.2)
        else:
            raise AssertionError("server did not come up")
        # concurrent callers: their questions meet in the engines' batches; a late one joins them
        with ThreadPoolExecutor(4) as ex:
            futs = [ex.submit(_post, port, {"messages": [{"role": "user", "content": q}]}) for q in QUESTIONS]
            time.sleep(0.05)
            futs.append(ex.submit(_post, port, {"messages": [{"role": "user", "content": QUESTIONS[0]}]}))
            outs = [f.result() for f in futs]
        for q, o in zip(QUESTIONS + [QUESTIONS[0]], outs):
            assert o["choices"][0]["message"]["content"] == want[q]
            assert o["consensus"]["strategy"] == "majority_vote" and len(o["consensus"]["votes"]) == 3
    finally:
        proc.send_signal(signal.SIGTERM)         # the supervisor takes its ranks with it
        try:
            proc.wait(30)
        except subprocess.TimeoutExpired:
            proc.kill()


JUDGE_TP2 = """
strategy = "judge_logprob"
[sampling]
temperature = 0.0
max_new_tokens = 6
[models.p]
preset = "tiny-llama"
seed = 1
kv_blocks = 512
[models.q]
preset = "tiny-qwen"
seed = 3
kv_blocks = 512
[models.judge]
preset = "small-llama"
seed = 2
kv_blocks = 512
tp = 2
[[members]]
name = "p0"
model = "p"
[[members]]
name = "p1"
model = "q"
[[members]]
name = "judge"
model = "judge"
role = "judge"
"""


@pytest.mark.gpu
def test_repl_with_two_ranks_and_a_tensor_parallel_judge_on_one_gpu(tmp_path):
    """The plain command on the real kernels: `llm-consensus --gpus 2` starts two ranks (rehearsal
    mode: both on GPU 0), the judge is a TP=2 group over the peer-to-peer kernels, the start-up
    warm-up records its decode graphs on both ranks, and the questions are served in lock-step
    rounds. The judge's scores agree with those of one process holding the unsharded judge up to
    the reduction order of bf16 partial sums, and a clear decision is the same decision."""
    cfg = tmp_path / "judge.toml"
    cfg.write_text(JUDGE_TP2)

    def run(gpus, extra_env):
        e = env()
        e.update(HSA_ENABLE_IPC_MODE_LEGACY="0", LC_P2P_GRID_CAP="32", **extra_env)
        p = subprocess.run([sys.executable, "-m", "llm_consensus_amd.cli", "--config", str(cfg), "--gpus", str(gpus),
                            "--json"], input="\n".join(QUESTIONS) + "\nexit\n", capture_output=True, text=True, env=e,
                           timeout=400)
        assert p.returncode == 0, p.stderr[-3000:]
        recs = [json.loads(l[l.index("{"):]) for l in p.stdout.splitlines() if "{" in l]
        return recs

    two = run(2, {"LC_SHARE_GPU": "1"})
    assert [r["question"] for r in two] == QUESTIONS and all(r["strategy"] == "judge_logprob" for r in two)
    one = run(1, {})
    for a, b in zip(two, one):
        assert [v["candidate"] for v in a["votes"]] == [v["candidate"] for v in b["votes"]] == ["p0", "p1"]
        sa = [sum(v["scores"]) for v in a["votes"]]
        sb = [sum(v["scores"]) for v in b["votes"]]
        assert all(abs(x - y) < 0.35 + 0.02 * abs(y) for x, y in zip(sa, sb)), (sa, sb)
        if abs(sb[0] - sb[1]) > 1.0:              # a clear decision survives the sharded reduction order
            assert a["answer"] == b["answer"]
