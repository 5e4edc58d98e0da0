"""bench.py's own plumbing on CPU: the launcher really starts one rank per requested device
(gloo, tiny presets), the JSON line reports what ran, and mismatches are errors. The
reference's defining behaviour is concurrent fan-out to all members
(/root/reference/src/main.rs:250-253); a benchmark that silently ran on one device would
measure none of it."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
TINY = ["--device", "cpu", "--models", "tiny-llama,tiny-qwen,tiny-mistral", "--questions", "2", "--prompt-len", "64",
        "--new-tokens", "4", "--steps", "1", "--warmup", "0", "--sc-questions", "1", "--k-chains", "4",
        "--proto-questions", "1", "--proto-prompt-len", "64", "--judge-questions", "2", "--judge-model", "tiny-qwen",
        "--proposer-model", "tiny-llama", "--moe-questions", "2", "--moe-model", "tiny-mixtral", "--moe-experts", "4",
        "--embed-model", "tiny-llama", "--moe-samples", "2", "--skip-stream"]


def run_bench(args, env=None, timeout=300):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, env=e,
                          timeout=timeout)


def json_line(p):
    """The result is the LAST line; every line before it is the same record at an earlier point of
    the run, marked provisional and growing section by section."""
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert lines, (p.stdout[-2000:], p.stderr[-2000:])
    assert "provisional" not in lines[-1]
    for earlier, later in zip(lines, lines[1:]):
        assert "provisional" in earlier and set(earlier) - {"provisional"} <= set(later)
    return lines[-1]


def coll(section) -> str:
    """A section's transport report: a label, or one description per sharded model."""
    c = section["collectives"]
    return c if isinstance(c, str) else "; ".join(c[k] for k in sorted(c))


def test_gpus_flag_launches_that_many_ranks_for_every_workload():
    p = run_bench(["--gpus", "2", "--workload", "A,D,E"] + TINY)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json_line(p)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["value"] > 0
    assert out["steps"] == 1 and out["warmup"] == 0 and out["higher_is_better"] is True
    d, e = out["judge_consensus"], out["moe_medoid"]
    assert d["judge_tp"] == 2 and d["placement"]["gpus_per_replica"] == 2 and d["value"] > 0
    assert e["expert_parallel"] == 2 and e["ep_mode"] == "a2a" and e["value"] > 0
    assert d["mean_member_calls"] == 4 + 4 * 1           # four proposals + four judge scores per question


def test_single_rank_line_has_every_section_and_the_contract_fields():
    p = run_bench(["--workload", "all"] + TINY)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json_line(p)
    assert out["n_gpus"] == 1 and out["config"]["global_batch"] == 2
    for key in ("metric", "value", "unit", "ms_per_step", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert out[key] is not None, key
    for sec in ("self_consistency", "reference_protocol", "reference_protocol_persona_last",
                "reference_protocol_persona_first", "judge_consensus",
                "moe_medoid"):
        assert out[sec]["value"] > 0 and out[sec]["p50_latency_ms"] > 0, sec
    assert out["judge_consensus"]["judge_tp"] == 1 and out["moe_medoid"]["expert_parallel"] == 1


def test_closed_loop_and_poisson_arrival_sections_of_workload_a():
    """Workload A as a service: the closed loop (a finished question is replaced at once) and the
    open loop with Poisson arrivals at 70 % of the closed-loop rate, whose latencies count from the
    arrival and whose achieved rate cannot exceed the offered one by more than the trace's noise."""
    args = [x for x in TINY if x != "--skip-stream"] + ["--workload", "A", "--in-flight", "4", "--poisson-seconds", "1.5"]
    p = run_bench(args)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json_line(p)
    sq, po = out["standing_queue"], out["poisson_arrivals"]
    assert sq["value"] > 0 and sq["questions_in_flight_per_gpu"] == 4
    assert po["questions_per_gpu"] >= 8 and po["value"] > 0
    assert po["p50_latency_ms"] > 0 and po["p99_latency_ms"] >= po["p95_latency_ms"] >= po["p50_latency_ms"]
    assert po["value"] < 2.0 * po["offered_rate_per_s"]


def test_spread_placement_puts_one_model_per_rank():
    p = run_bench(["--gpus", "3", "--workload", "A", "--placement", "spread"] + TINY)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json_line(p)
    assert out["n_gpus"] == 3 and out["config"]["placement"]["gpus_per_replica"] == 3
    assert out["config"]["global_batch"] == 2
    gpus = out["config"]["placement"]["replica0_member_gpus"]
    assert sorted(g[0] for g in gpus.values()) == [0, 1, 2]


def test_mismatches_are_errors_not_silent_single_device_runs():
    p = run_bench(["--gpus", "4", "--workload", "A"] + TINY, env={"WORLD_SIZE": "2", "RANK": "0"})
    assert p.returncode == 2 and "disagrees" in p.stderr and not p.stdout.strip()
    import torch

    if torch.cuda.device_count() < 3:
        q = run_bench(["--gpus", "3", "--workload", "A"])
        assert q.returncode == 2 and "visible" in q.stderr and not q.stdout.strip()
    r = run_bench(["--gpus", "2", "--workload", "A", "--placement", "spread"] + TINY)
    assert r.returncode != 0 and "spread" in r.stderr


def test_sharded_sections_run_isolated_and_a_failing_one_does_not_take_the_line_down():
    """At more than one rank D and E run in child processes with their own process group. The
    first test above passes through that path; here the children get no time at all, both
    attempts (peer-to-peer kernels, then RCCL) are abandoned, and the line still carries A."""
    p = run_bench(["--gpus", "2", "--workload", "A,D", "--section-timeout", "0.05"] + TINY)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json_line(p)
    assert out["value"] > 0 and out["n_gpus"] == 2
    err = out["judge_consensus"]["error"]
    assert len(err) == 2 and "timeout" in json.dumps(err)


def test_isolated_sections_share_one_time_budget():
    """All child-process sections together get --isolated-budget seconds: with less than a
    minute of it a section is not started at all, and the line still carries A."""
    p = run_bench(["--gpus", "2", "--workload", "A,D,E", "--isolated-budget", "30", "--skip-spread"] + TINY)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json_line(p)
    assert out["value"] > 0
    for key in ("judge_consensus", "moe_medoid"):
        assert "skipped" in json.dumps(out[key]["error"]), out[key]


def test_wall_budget_counts_from_process_start():
    """--wall-budget is measured from the start of the process (imports, model loading and the
    replica workloads included): with less than a minute of it left no child section is started."""
    p = run_bench(["--gpus", "2", "--workload", "A,D", "--wall-budget", "20", "--skip-spread"] + TINY)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json_line(p)
    assert out["value"] > 0 and "wall-budget" in json.dumps(out["judge_consensus"]["error"])


def test_isolated_section_reports_which_collectives_it_used():
    p = run_bench(["--gpus", "2", "--workload", "E"] + TINY)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json_line(p)
    # per sharded model: the transport its group ended up on (CPU ranks: the gloo process group, eager steps)
    assert coll(out["moe_medoid"]).startswith("gloo") and "eager" in coll(out["moe_medoid"])
    assert out["value"] == out["moe_medoid"]["value"]


def test_four_ranks_shard_the_judge_four_ways_next_to_the_proposers():
    """Workload D as the round-end driver starts it on a 4-GPU node (torch.distributed.run, one rank
    per device): the judge is a TP=4 group over all four ranks, each of which also hosts a proposer.
    `tiny-judge` is the smallest preset four ranks can shard (four kv heads)."""
    args = [a if a != "tiny-qwen" or TINY[i - 1] != "--judge-model" else "tiny-judge" for i, a in enumerate(TINY)]
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    import socket

    with socket.socket() as probe:                       # a port that is free now (a fixed one may be taken)
        probe.bind(("127.0.0.1", 0))
        port = probe.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "4",
                        "--workload", "D"] + args, capture_output=True, text=True, env=e, timeout=420)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json_line(p)
    d = out["judge_consensus"]
    assert out["n_gpus"] == 4 and "error" not in d and d["value"] > 0 and d["judge_tp"] == 4
    assert d["placement"]["replica0_member_gpus"]["judge"] == [0, 1, 2, 3]
    assert sorted(g for k, v in d["placement"]["replica0_member_gpus"].items() if k != "judge" for g in v) == [0, 1, 2, 3]


def test_three_ranks_also_report_one_model_per_gpu():
    p = run_bench(["--gpus", "3", "--workload", "A"] + TINY)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json_line(p)
    assert out["n_gpus"] == 3 and out["config"]["global_batch"] == 6          # the headline: a full ensemble per rank
    sp = out["ensemble_one_model_per_gpu"]
    assert sp["value"] > 0 and sp["global_batch"] == 2 and sp["placement"]["gpus_per_replica"] == 3


@pytest.mark.gpu
def test_two_ranks_rehearse_every_workload_on_one_gpu():
    """--share-gpu: the multi-GPU code paths of bench.py on a one-GPU machine. Two ranks (child
    processes) use GPU 0; the replica workloads exchange winners through the host-staged gloo
    group, D runs its judge tensor-parallel over both ranks and E its experts expert-parallel with
    all-to-all dispatch, both through the peer-to-peer kernels over IPC-mapped memory and inside
    isolated child processes, as on a real node."""
    args = [x for x in TINY if x not in ("--device", "cpu")] + ["--gpus", "2", "--share-gpu", "--workload", "A,B,D,E",
                                                                 "--skip-spread"]
    p = run_bench(args, env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"}, timeout=420)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json_line(p)
    assert out["n_gpus"] == 2 and "rehearsal" in out and out["value"] > 0
    d, e = out["judge_consensus"], out["moe_medoid"]
    assert d.get("value", 0) > 0 and d["judge_tp"] == 2, d
    assert e.get("value", 0) > 0 and e["expert_parallel"] == 2 and e["ep_mode"] == "a2a", e
    assert coll(d).startswith("p2p") and coll(e).startswith("p2p")
    assert "hipGraphs" in coll(d)


@pytest.mark.gpu
def test_stalled_peer_at_first_contact_falls_back_inside_the_section():
    """LC_P2P_FAIL_INJECT=stall: one rank of the judge's tensor-parallel group never takes part in
    the first-contact check of the peer-to-peer collectives. The group notices within seconds,
    serves from the bootstrap communicator in the same child processes, and the section still
    reports a number (no second attempt, no lost section)."""
    args = [x for x in TINY if x not in ("--device", "cpu")] + ["--gpus", "2", "--share-gpu", "--workload", "D"]
    p = run_bench(args, env={"HSA_ENABLE_IPC_MODE_LEGACY": "0", "LC_P2P_FAIL_INJECT": "stall",
                             "LC_P2P_SELF_CHECK_TIMEOUT_S": "3"}, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json_line(p)["judge_consensus"]
    assert d.get("value", 0) > 0 and d["judge_tp"] == 2, d
    assert "self-check failed" in coll(d) and "timed out" in coll(d) and "first_attempt" not in d, d

@pytest.mark.gpu
def test_four_ranks_rehearse_the_sharded_workloads_on_one_gpu():
    """The same with four ranks: the judge of D is a TP=4 group (`tiny-judge`: four kv heads) whose
    ranks also host the four proposers, E is expert parallel over four ranks, and workload A runs
    once more with one model per GPU; all through the peer-to-peer kernels."""
    args = [x for x in TINY if x not in ("--device", "cpu")]
    args[args.index("--judge-model") + 1] = "tiny-judge"
    args += ["--gpus", "4", "--share-gpu", "--workload", "A,D,E"]
    p = run_bench(args, env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"}, timeout=420)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json_line(p)
    assert out["n_gpus"] == 4 and out["value"] > 0
    sp, d, e = out["ensemble_one_model_per_gpu"], out["judge_consensus"], out["moe_medoid"]
    assert sp.get("value", 0) > 0 and sp["placement"]["gpus_per_replica"] == 3, sp
    assert d.get("value", 0) > 0 and d["judge_tp"] == 4, d
    assert e.get("value", 0) > 0 and e["expert_parallel"] == 4 and e["ep_mode"] == "a2a", e
    assert all(coll(x).startswith("p2p") for x in (sp, d, e))


def test_sharded_workloads_report_a_standing_queue_in_lock_step_rounds():
    """D and E as services under load: one process drives the strategy's pool; two ranks (TP=2
    judge, EP=2 experts) are driven in lock-step rounds with rank 0 feeding the questions."""
    args = [x for x in TINY if x != "--skip-stream"] + ["--workload", "D,E", "--poisson-load", "0"]
    for gpus in (1, 2):
        p = run_bench(args + ["--gpus", str(gpus)], timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        out = json_line(p)
        for key in ("judge_consensus", "moe_medoid"):
            sq = out[key]["standing_queue"]
            assert sq["value"] > 0 and sq["questions"] == 4 * 2 * gpus and sq["p50_latency_ms"] > 0, (gpus, key, sq)
            assert ("lock-step" in sq["mode"]) == (gpus == 2)


@pytest.mark.gpu
def test_two_ranks_serve_the_sharded_workloads_in_lock_step_rounds_on_one_gpu():
    """--share-gpu with the standing-queue sections on: the TP=2 judge (peer-to-peer kernels, decode
    steps recorded as hipGraphs and launched ahead of their results) and the EP=2 MoE are served
    continuously, both ranks advancing in lock-step rounds with rank 0 feeding the questions."""
    args = [x for x in TINY if x not in ("--device", "cpu", "--skip-stream")]
    args += ["--gpus", "2", "--share-gpu", "--workload", "D,E", "--poisson-load", "0"]
    p = run_bench(args, env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"}, timeout=420)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json_line(p)
    for key in ("judge_consensus", "moe_medoid"):
        sq = out[key]["standing_queue"]
        assert sq["value"] > 0 and "lock-step" in sq["mode"] and sq["questions"] == 16, (key, sq)
        assert coll(out[key]).startswith("p2p"), out[key]


def test_http_load_benchmark_runs_a_closed_loop_through_the_server():
    """benchmarks/http_load.py: the vote ensemble behind the HTTP server, a separate client process
    keeping 6 connections busy; every counted question is answered, none fails."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    p = subprocess.run([sys.executable, str(root / "benchmarks" / "http_load.py"), "--device", "cpu", "--models",
                        "tiny-llama,tiny-llama", "--in-flight", "6", "--questions", "12", "--prompt-len", "96",
                        "--new-tokens", "3"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["answers"] == 12 and out["errors"] == 0 and out["answers_per_s"] > 0
    assert out["p99_latency_ms"] >= out["p95_latency_ms"] >= out["p50_latency_ms"] > 0
    assert out["server"]["requests_total"] == 18 and out["server"]["errors_total"] == 0     # 6 warm + 12 counted
    # the same loop with every connection streaming one member model's tokens
    p = subprocess.run([sys.executable, str(root / "benchmarks" / "http_load.py"), "--device", "cpu", "--models",
                        "tiny-llama", "--in-flight", "4", "--questions", "8", "--prompt-len", "96", "--new-tokens", "6",
                        "--stream"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["answers"] == 8 and out["errors"] == 0 and 0 < out["p50_first_token_ms"] <= out["p50_latency_ms"]
