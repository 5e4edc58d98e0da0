#!/usr/bin/env python3
"""Headline benchmark: consensus answers/sec and latency percentiles on one MI355X node.

    python bench.py --gpus N --steps K --warmup W [--workload A,B,C,D,E] [--placement replica|spread]

With ``--gpus N`` > 1 and no torchrun environment this process only launches: it starts N
children (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on
127.0.0.1) before anything touches a GPU, relays rank 0's output and exits with the worst
child status. Under ``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N``
it is one of the ranks. Fewer visible GPUs than ``--gpus``, or a WORLD_SIZE that disagrees
with it, is an error (never a silent 1-GPU run). Rank 0 prints the result as ONE JSON line, the
LAST line of its output. As soon as the headline workload is measured, and again after every
further section, it also prints the line as it stands so far (marked ``"provisional"``): a run
that is cut short by a time limit has still reported everything it measured.

Time budget on a multi-GPU node (defaults): A, B and C take ~180 s (C with 32 questions in
flight and four timed steps); the sections that run in child processes (A with one model per
GPU, D, E) share ``--isolated-budget`` 360 s, each attempt at most ``--section-timeout`` 150 s,
and an attempt is only started with a minute left: the whole run stays under 180 + 360 = 540 s
whatever stalls; ``--wall-budget`` 570 s (counted from process start, so a slow start on a cold
machine is seen too) ends or skips child sections all the same. A model group's own collectives give
up waiting for a peer after LC_P2P_TIMEOUT_S (10 s here) and a group whose first exchange fails
its self-check falls back to RCCL inside the same process (parallel/p2p.py).

Workloads (llm_consensus_amd/workloads.py; random-init bf16 weights of the named shapes,
synthetic multiple-choice prompts, every new token really sampled):

  A  (the timed headline "step") Llama-3-8B + Qwen2.5-7B + Mistral-7B, majority vote over
     on-device log-prob decisions. One step = ``--questions`` questions per ensemble replica
     through tokenize -> 3 x (prefill + sampled decode) -> vote kernel -> cross-GPU gather of
     winners. ``--placement replica``: every GPU hosts a whole ensemble (47 GB of 288 GB) and
     its own share of the questions, ranks exchange only winners (weak scaling).
     ``--placement spread``: one model per GPU, three GPUs per replica.
  B  Llama-3-8B self-consistency, K=16 sampled chains forked from one prefill (shared-prefix
     paged KV), answer vote.
  C  the reference protocol itself: 4 personas on one Llama-3-8B, propose -> everyone
     evaluates -> a dissenter refines, <= 5 rounds (/root/reference/src/main.rs:220-314).
  D  4 x 8B proposers + a Llama-3-70B judge scoring candidates by log-prob; the judge is
     tensor parallel over min(4, N) GPUs (unsharded on one GPU).
  E  Mixtral-8x7B proposer, expert parallel over min(8, N) GPUs, sampling 4 candidates per
     question + an embedding model; the cosine medoid wins.

A and B are timed for exactly ``--steps`` steps; C, D and E for ``--aux-steps``.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

T_START = time.perf_counter()
ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# A peer that never shows up is noticed after this many seconds (the library's default of 60 s
# suits a server; a benchmark section has 150 s in all).
os.environ.setdefault("LC_P2P_TIMEOUT_S", "10")

REFERENCE_BOUND_ANSWERS_PER_S = 0.2  # derived upper bound for the reference (it publishes no numbers)
ALL_WORKLOADS = "ABCDE"


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="all", help="comma list out of A,B,C,D,E (default: all)")
    ap.add_argument("--placement", choices=("replica", "spread"), default="replica",
                    help="workload A: a full ensemble per GPU, or one model per GPU")
    ap.add_argument("--device", choices=("cuda", "cpu"), default="cuda",
                    help="cpu: plumbing run over gloo (use the tiny model names)")
    ap.add_argument("--models", default="llama-3-8b,qwen2.5-7b,mistral-7b")
    ap.add_argument("--questions", type=int, default=64, help="questions per replica per step (workload A)")
    ap.add_argument("--prompt-len", type=int, default=512)
    ap.add_argument("--new-tokens", type=int, default=64)
    ap.add_argument("--k-chains", type=int, default=16)
    ap.add_argument("--sc-questions", type=int, default=16, help="questions per GPU per step (workload B)")
    ap.add_argument("--sc-steps", type=int, default=None, help="timed steps of workload B (default: --steps)")
    ap.add_argument("--skip-stream", action="store_true", help="skip the standing-queue run of workload A")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="questions kept in flight per GPU in the standing-queue run (default 4 x --questions)")
    ap.add_argument("--poisson-load", type=float, default=0.7,
                    help="open-loop run of workload A: arrival rate as a fraction of the closed-loop rate (0 = skip)")
    ap.add_argument("--poisson-seconds", type=float, default=10.0, help="length of the arrival trace")
    ap.add_argument("--aux-steps", type=int, default=None, help="timed steps of C, D, E (default: min(steps, 6), min(steps, 4) with several GPUs)")
    ap.add_argument("--proto-questions", type=int, default=32,
                    help="questions per GPU per step (workload C): 128 member rows, where the protocol stops gaining from more "
                         "(8: 8.9 answers/s at the reference layout, 32: 11.1, 64: 11.4 with twice the latency)")
    ap.add_argument("--proto-prompt-len", type=int, default=128)
    ap.add_argument("--judge-questions", type=int, default=16, help="questions per GPU per step (workload D)")
    ap.add_argument("--judge-model", default="llama-3-70b")
    ap.add_argument("--proposer-model", default="llama-3-8b")
    ap.add_argument("--no-warm-judges", action="store_true",
                    help="workload D: judges do not prefill the question part of their prompt during the proposer phase")
    ap.add_argument("--moe-questions", type=int, default=16, help="questions per GPU per step (workload E)")
    ap.add_argument("--moe-model", default="mixtral-8x7b")
    ap.add_argument("--moe-experts", type=int, default=8, help="experts of --moe-model (bounds the EP degree)")
    ap.add_argument("--embed-model", default="qwen2.5-0.5b")
    ap.add_argument("--moe-samples", type=int, default=4)
    ap.add_argument("--temperature", type=float, default=0.8)
    ap.add_argument("--top-p", type=float, default=0.95)
    ap.add_argument("--prefill-tokens", type=int, default=8192, help="token budget of one prefill step")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-mix-decode", action="store_true",
                    help="A, B, C: prefill-only and decode-only steps (decoding chains do not ride along in prefill steps)")
    ap.add_argument("--out", default=None, help="also write the JSON line to this file")
    ap.add_argument("--skip-spread", action="store_true",
                    help="multi-GPU runs: skip the extra run of workload A with one model per GPU")
    ap.add_argument("--section-timeout", type=float, default=150.0,
                    help="multi-GPU runs: seconds a sharded workload (D, E) may take before it is abandoned")
    ap.add_argument("--isolated-budget", type=float, default=360.0,
                    help="multi-GPU runs: seconds all child-process sections together may take; what is left "
                         "bounds each attempt, and a section that finds less than a minute left is skipped")
    ap.add_argument("--wall-budget", type=float, default=570.0,
                    help="multi-GPU runs: seconds since this process started (imports and model loading included) "
                         "after which no child-process section is started or kept running; guards the run against "
                         "a slow start on a cold machine, which --isolated-budget alone does not see")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a one-GPU machine: all --gpus ranks use GPU 0, the process group is gloo with "
                         "host-staged tensors (RCCL refuses two ranks on one device), model groups still use the "
                         "peer-to-peer kernels over IPC-mapped memory. Not a measurement: the ranks share the GPU")
    ap.add_argument("--in-child", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args(argv)
    wl = ALL_WORKLOADS if a.workload.lower() == "all" else "".join(w.strip().upper() for w in a.workload.split(","))
    if not wl or any(w not in ALL_WORKLOADS for w in wl):
        ap.error(f"--workload takes letters out of {ALL_WORKLOADS}")
    a.workloads = wl
    if a.share_gpu and a.gpus > 1:
        # rehearsal: all ranks allocate on ONE card, so the per-rank pools must add up to what a
        # single GPU has (with 32 questions workload C alone reserves 43 GB of KV per rank)
        a.proto_questions = min(a.proto_questions, 8)
    if a.aux_steps is None:
        # multi-GPU runs time fewer steps of C, D and E: their sections share a fixed time budget
        a.aux_steps = min(a.steps, 6 if a.gpus == 1 else 4)
    return a


# ------------------------------------------------------------------------------------ launcher
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(a, argv) -> int:
    """Starts one child per GPU. Runs before this process has made any GPU call and never
    replaces a process: the children are fresh interpreters."""
    if a.device == "cuda" and not a.share_gpu:
        import torch

        seen = torch.cuda.device_count()          # counts devices without creating a context
        if seen < a.gpus:
            print(f"bench.py: --gpus {a.gpus} but only {seen} GPU(s) are visible", file=sys.stderr)
            return 2
    port = _free_port()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = set(range(a.gpus))
    while live:
        for r in list(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                rc = rc or code
                for o in live:                     # a rank died: the others would wait for it forever
                    procs[o].terminate()
        time.sleep(0.05)
    return rc


# ------------------------------------------------------------------------------------ helpers
class Ctx:
    def __init__(self, a):
        import torch

        self.a = a
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        self.cuda = a.device == "cuda"
        self.dist = None
        if self.cuda:
            if not torch.cuda.is_available():
                raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False); --device cpu is the "
                                 "plumbing mode")
            if a.share_gpu:
                local = 0
            if torch.cuda.device_count() <= local:
                raise SystemExit(f"bench.py: rank {self.rank} wants GPU {local}, {torch.cuda.device_count()} visible")
            torch.cuda.set_device(local)
            self.dev = torch.device("cuda", local)
        else:
            self.dev = torch.device("cpu")
            # ranks share the host's cores: without a cap every rank starts a full-size thread pool
            torch.set_num_threads(max(1, (os.cpu_count() or 1) // max(1, self.world)))
        if self.world > 1:
            import torch.distributed as dist

            # generous collective timeout: ranks wait for each other across whole isolated sections
            from datetime import timedelta

            if self.cuda and not a.share_gpu:
                dist.init_process_group("nccl", device_id=self.dev, timeout=timedelta(minutes=40))
            else:
                dist.init_process_group("gloo", timeout=timedelta(minutes=40))
            self.dist = dist
        from llm_consensus_amd.parallel.comm import make_comm

        self.comm = make_comm(self.dist, self.dev)
        self.bs = 64 if self.cuda else 16

    def sync(self):
        import torch

        if self.cuda:
            torch.cuda.synchronize()

    def free(self):
        import torch

        gc.collect()
        if self.cuda:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()


def pct(xs, q):
    import numpy as np

    return round(float(np.percentile(xs, q)), 1) if len(xs) else None


def timed(ctx: Ctx, ask, make_batch, steps: int, warmup: int, count):
    """``warmup`` untimed steps, then exactly ``steps`` timed ones between barrier + device
    synchronisation on both sides; elapsed time is the maximum over ranks. ``ask(batch)``
    returns this rank's results, ``count(results)`` the number of answers they stand for
    node-wide (it may run a collective)."""
    for _ in range(warmup):
        count(ask(make_batch()))
    ctx.comm.barrier()
    ctx.sync()
    t0 = time.perf_counter()
    n_ans, lat, kept = 0, [], []
    for _ in range(steps):
        res = ask(make_batch())
        n_ans += count(res)
        lat += [r.latency_s * 1e3 for r in res]
        kept += res
    ctx.sync()
    ctx.comm.barrier()
    el = ctx.comm.max_float(time.perf_counter() - t0)
    return el, n_ans, ctx.comm.gather_floats(lat), kept


def section(metric, el, n, lat, steps, **extra):
    d = {"metric": metric, "value": round(n / el, 3), "unit": "answers/s", "ms_per_step": round(el / steps * 1e3, 2),
         "steps": steps, "p50_latency_ms": pct(lat, 50), "p95_latency_ms": pct(lat, 95)}
    d.update(extra)
    return d


def section_done(ctx: Ctx, out: dict, label: str) -> None:
    """Marks a section as measured and reports the line so far when others are still to come."""
    todo = ctx.todo
    if label in todo:
        todo.remove(label)
    if todo:
        emit(ctx, out, ", ".join(todo))


def emit(ctx: Ctx, out: dict, pending: str = "") -> None:
    """Rank 0 prints the result line as it stands. ``pending`` names what is still to be
    measured (the line is then marked provisional); the final call passes nothing."""
    if ctx.rank != 0:
        return
    d = dict(out)
    if "A" not in ctx.a.workloads:
        # no headline run requested: promote the first measured section so the line stays well formed
        for key in ("self_consistency", "reference_protocol", "judge_consensus", "moe_medoid"):
            if key in d and "value" in d[key]:
                d.update({"metric": d[key]["metric"], "value": d[key]["value"],
                          "ms_per_step": d[key]["ms_per_step"], "steps": d[key]["steps"]})
                break
    if d.get("value") is not None:
        d["vs_baseline"] = round(d["value"] / REFERENCE_BOUND_ANSWERS_PER_S, 1)
    if pending:
        d["provisional"] = f"still to come: {pending}"
    d["elapsed_s"] = round(time.perf_counter() - T_START, 1)      # since this rank started (imports included)
    line = json.dumps(d)
    print(line, flush=True)
    if ctx.a.out:
        Path(ctx.a.out).parent.mkdir(parents=True, exist_ok=True)
        Path(ctx.a.out).write_text(line + "\n")


# ------------------------------------------------------------------------------------ A, B, C
def run_replica_workloads(ctx: Ctx, out: dict) -> None:
    """A (replica placement), B and C: every rank hosts its own engines; ranks exchange winners."""
    import numpy as np
    import torch

    from llm_consensus_amd._native import core
    from llm_consensus_amd.consensus.backend import GenParams
    from llm_consensus_amd.consensus.members import Member
    from llm_consensus_amd.consensus.protocol import Coordinator
    from llm_consensus_amd.consensus.strategies import MajorityVote, SelfConsistency
    from llm_consensus_amd.engine.backend import EngineBackend
    from llm_consensus_amd.engine.engine import Engine
    from llm_consensus_amd.models.config import get_preset
    from llm_consensus_amd.models.transformer import TransformerModel
    from llm_consensus_amd.parallel.replica import gather_winners
    from llm_consensus_amd.utils.tokenizer import ByteLevelTokenizer
    from llm_consensus_amd.workloads import make_questions

    a, dev, comm, bs = ctx.a, ctx.dev, ctx.comm, ctx.bs
    wl = a.workloads
    names = [n.strip() for n in a.models.split(",") if n.strip()]
    if "A" not in wl:
        names = names[:1]                                       # B and C run on the first model only
    seq_len = a.prompt_len + a.new_tokens
    blocks_per_seq = (seq_len + bs - 1) // bs + 1
    max_seqs = max(a.questions if "A" in wl else 1, a.sc_questions * a.k_chains if "B" in wl else 1,
                   (a.in_flight or 4 * a.questions) if ("A" in wl and not a.skip_stream) else 1)
    t_init = time.perf_counter()
    members = []
    personas = core.default_personas()
    for i, name in enumerate(names):
        cfg = get_preset(name)
        model = TransformerModel.random_init(cfg, dev, seed=1234 + i)
        eng = Engine(model, num_kv_blocks=(max_seqs + 8) * blocks_per_seq + 64, block_size=bs,
                     max_batch_seqs=max(256, max_seqs), max_prefill_tokens=a.prefill_tokens,
                     max_model_len=seq_len + 64, use_graphs=not a.no_graphs, name=name, mix_decode=not a.no_mix_decode)
        be = EngineBackend(eng, ByteLevelTokenizer(cfg.vocab_size), stop_at_eos=False)
        gp = GenParams(temperature=a.temperature, top_p=a.top_p, max_new_tokens=a.new_tokens, seed=17 * i)
        members.append(Member(personas[i % len(personas)], be, params=gp))
    ctx.sync()
    init_s = time.perf_counter() - t_init
    rng = np.random.default_rng(1000 + ctx.rank)

    def winners(results):
        return int(gather_winners(comm, results, "ABCD", dev).numel())

    if "A" in wl:
        vote = MajorityVote(members, choices="ABCD", decision_mode="logprob", device=dev if ctx.cuda else None)
        st0 = [m.backend.stats() for m in members]
        el, n, lat, _ = timed(ctx, vote.ask_many, lambda: make_questions(rng, a.questions, a.prompt_len), a.steps,
                              a.warmup, winners)
        st1 = [m.backend.stats() for m in members]
        out.update({
            "value": round(n / el, 3), "ms_per_step": round(el / a.steps * 1e3, 2),
            "p50_latency_ms": pct(lat, 50), "p95_latency_ms": pct(lat, 95), "init_s": round(init_s, 1),
        })
        out["config"].update({"global_batch": a.questions * ctx.world,
                              "parallelism": f"dp{ctx.world} (one full ensemble replica per GPU)"})
        # host-side view of each member engine over warmup + timed steps (the members' steps overlap
        # on the GPU, so these wall-clock spans add up to more than the step time)
        out["member_engines"] = {
            names[i]: {k: round(st1[i].get(k, 0.0) - st0[i].get(k, 0.0), 3)
                       for k in ("prefill.steps", "prefill.seconds", "decode.steps", "decode.seconds",
                                 "decode.wait_seconds", "prefill.rider_tokens", "decode.ahead_steps")}
            for i in range(len(names))}
        section_done(ctx, out, "A")
        if not a.skip_stream:
            # The same ensemble as a service under load: a closed loop keeps ``in_flight``
            # questions in the engines at all times; every finished question is replaced by a new
            # one at the next engine step (strategies.VotePool). Prompts are admitted as KV blocks
            # and batch slots free up, chains already decoding ride along in the steps that
            # prefill the newcomers, and each member's decode batch stays at ~in_flight rows
            # instead of draining at the end of every batch. The clock covers ramp-up and drain.
            def closed_loop(n_total, in_flight):
                qs = make_questions(rng, n_total, a.prompt_len)
                pool = vote.pool()
                comm.barrier()
                ctx.sync()
                t0 = time.perf_counter()
                res, nxt = [], 0
                while nxt < min(in_flight, n_total):
                    pool.submit(qs[nxt])
                    nxt += 1
                while pool.active:
                    for _, r in pool.pump():
                        res.append(r)
                        if nxt < n_total:
                            pool.submit(qs[nxt])
                            nxt += 1
                k = winners(res)
                ctx.sync()
                comm.barrier()
                return comm.max_float(time.perf_counter() - t0), k, comm.gather_floats([r.latency_s * 1e3 for r in res])

            inflight = a.in_flight or 4 * a.questions
            closed_loop(inflight, inflight)
            n_sq = inflight * max(2, min(a.steps, 4))
            el_s, n_s, lat_s = closed_loop(n_sq, inflight)
            out["standing_queue"] = {
                "mode": "closed loop: a finished question is replaced at the next engine step",
                "questions_in_flight_per_gpu": inflight, "questions_per_gpu": n_sq, "value": round(n_s / el_s, 3),
                "unit": "answers/s", "ms_per_batch_of_questions": round(el_s / n_sq * a.questions * 1e3, 2),
                "p50_latency_ms": pct(lat_s, 50), "p95_latency_ms": pct(lat_s, 95),
                "preemptions": int(sum(m.backend.stats().get("sched.preemptions", 0) for m in members)),
            }
            section_done(ctx, out, "A.standing_queue")
            if a.poisson_load > 0:
                # Open loop: questions arrive as a Poisson process at a fraction of the closed-loop
                # rate just measured (each rank draws its own arrival trace), are admitted the moment
                # they arrive, and their latency counts from the ARRIVAL, so time spent waiting for
                # the driver to come round is included. This is the latency a client sees when the
                # service is busy but not saturated.
                rate = a.poisson_load * (n_s / el_s) / ctx.world           # arrivals per second and rank
                n_open = max(8, int(min(a.poisson_seconds * rate, 4 * n_sq)))
                arrive = np.cumsum(rng.exponential(1.0 / rate, size=n_open))
                qs = make_questions(rng, n_open, a.prompt_len)
                pool = vote.pool()
                comm.barrier()
                ctx.sync()
                t0 = time.perf_counter()
                lat_o, late, res_o, nxt, peak = [], {}, [], 0, 0
                while nxt < n_open or pool.active:
                    now = time.perf_counter() - t0
                    while nxt < n_open and arrive[nxt] <= now:
                        late[pool.submit(qs[nxt])] = now - arrive[nxt]
                        nxt += 1
                    peak = max(peak, pool.active)
                    if pool.active:
                        for tk, r in pool.pump():
                            res_o.append(r)
                            lat_o.append((r.latency_s + late.pop(tk)) * 1e3)
                    else:
                        time.sleep(min(1e-3, max(0.0, arrive[nxt] - now)))
                k_o = winners(res_o)
                ctx.sync()
                comm.barrier()
                el_o = comm.max_float(time.perf_counter() - t0)
                lat_o = comm.gather_floats(lat_o)
                out["poisson_arrivals"] = {
                    "mode": "open loop: Poisson arrivals, latency from arrival to answer",
                    "load_fraction_of_closed_loop_rate": a.poisson_load,
                    "offered_rate_per_s": round(rate * ctx.world, 2), "questions_per_gpu": n_open,
                    "value": round(k_o / el_o, 3), "unit": "answers/s",
                    "p50_latency_ms": pct(lat_o, 50), "p95_latency_ms": pct(lat_o, 95),
                    "p99_latency_ms": pct(lat_o, 99), "peak_in_flight_per_gpu": int(peak),
                }
                section_done(ctx, out, "A.poisson_arrivals")
    if "B" in wl:
        sc = SelfConsistency(members[0], k_chains=a.k_chains, choices="ABCD", decision_mode="logprob",
                             device=dev if ctx.cuda else None)
        steps_b = a.sc_steps or a.steps
        s0 = members[0].backend.stats()
        el, n, lat, _ = timed(ctx, sc.ask_many, lambda: make_questions(rng, a.sc_questions, a.prompt_len), steps_b,
                              a.warmup, winners)
        s1 = members[0].backend.stats()

        def d(key):
            return s1.get(key, 0.0) - s0.get(key, 0.0)

        out["self_consistency"] = section(
            f"consensus answers/sec, {names[0]} K={a.k_chains} self-consistency", el, n, lat, steps_b,
            questions_per_gpu=a.sc_questions, chains_tokens_per_s=round(n * a.k_chains * a.new_tokens / el, 1),
            # the K chains of a question are forks of one prefilled prompt: they attend ONE copy of its
            # full KV blocks (shared-prefix decode kernel); the prefix cache is not involved
            chains_per_question=a.k_chains, prompt_kv_blocks_shared_by_the_chains=a.prompt_len // bs,
            prompt_kv_reads_saved=round(1.0 - 1.0 / a.k_chains, 3), cow_copies=int(d("kv.cow_copies_total")),
            decode_ms_per_step=round(1e3 * d("decode.seconds") / max(1.0, d("decode.steps")), 3),
            decode_host_us_per_step={k: round(1e6 * d(f"host.{k}_seconds") / max(1.0, d("decode.steps")), 1)
                                     for k in ("schedule", "decode_fill", "decode_launch", "decode_finish")},
            decode_steps_launched_ahead=round(d("decode.ahead_steps") / max(1.0, d("decode.steps")), 3))
        section_done(ctx, out, "B")
    if "C" in wl and a.proto_questions > 0:
        # The reference's own topology and loop: four personas share ONE model, a random member
        # proposes, everyone (proposer included) judges, a random dissenter refines, at most five
        # evaluation rounds (/root/reference/src/main.rs:220-311, 359-426). Verdicts are the
        # on-device two-way log-prob decision because random-init weights never spell a verdict.
        model0 = members[0].backend.engine.model
        plen = 2048 + 2 * a.proto_prompt_len + 2 * a.new_tokens        # evaluate prompt: template + question + answer
        n_seq = a.proto_questions * 4
        eng_p = Engine(model0, num_kv_blocks=(n_seq + 4) * ((plen + bs - 1) // bs + 1) + 64, block_size=bs,
                       max_batch_seqs=max(64, n_seq), max_prefill_tokens=a.prefill_tokens, max_model_len=plen,
                       use_graphs=not a.no_graphs, name=names[0] + ".panel")
        be_p = EngineBackend(eng_p, ByteLevelTokenizer(model0.cfg.vocab_size), stop_at_eos=False)
        gp_p = GenParams(temperature=a.temperature, top_p=a.top_p, max_new_tokens=a.new_tokens, seed=99)

        def finished(results):
            # Answers are free text here; ranks exchange one row per finished question.
            done = torch.tensor([[int(r.forced)] for r in results], dtype=torch.int32, device=dev)
            return int(comm.all_gather_rows(done).numel())

        # Three prompt layouts: the reference's (question and answer first, domain named a third of
        # the way into the evaluate prompt); the same instructions with the persona text moved to
        # the end, which lets the four evaluate prompts of a round share most of their prefill; and
        # with the question and the answer moved to the end, which makes a persona's whole head the
        # same for every question (served from the prefix cache after the first). Random-init
        # weights give all four personas the same verdict, so the loop ends after one round; the
        # last run injects seeded dissent (a Good verdict becomes NeedsRefinement with probability
        # 0.25) to load the refinement path as well.
        sess = core.SessionConfig()
        sess.max_rounds, sess.seed = 5, 4242 + ctx.rank
        for key, templates, dissent in (
                ("reference_protocol", core.PromptTemplates(), 0.0),
                ("reference_protocol_persona_last", core.PromptTemplates.persona_last(), 0.0),
                ("reference_protocol_persona_first", core.PromptTemplates.persona_first(), 0.0),
                ("reference_protocol_persona_last_dissent25", core.PromptTemplates.persona_last(), 0.25)):
            panel = [Member(p, be_p, templates, gp_p, decision_mode="logprob", dissent_rate=dissent,
                            dissent_seed=1000 * ctx.rank + i) for i, p in enumerate(personas[:4])]
            coord = Coordinator(panel, sess)
            s0 = be_p.stats()
            el, n, lat, kept = timed(ctx, coord.ask_many, lambda: make_questions(rng, a.proto_questions, a.proto_prompt_len),
                                     a.aux_steps, min(a.warmup, 2), finished)
            s1 = be_p.stats()
            hit = (s1["kv.prefix_hit_tokens"] - s0["kv.prefix_hit_tokens"]) / max(1, s1["kv.prompt_tokens"] - s0["kv.prompt_tokens"])
            out[key] = section(
                f"consensus answers/sec, reference protocol (4 personas on {names[0]}, unanimity loop, <= 5 rounds)",
                el, n, lat, a.aux_steps, questions_per_gpu=a.proto_questions, question_tokens=a.proto_prompt_len,
                mean_rounds=round(float(np.mean([r.rounds for r in kept])), 2),
                mean_member_calls=round(float(np.mean([r.member_calls for r in kept])), 2),
                forced_fraction=round(float(np.mean([r.forced for r in kept])), 3), injected_dissent_rate=dissent,
                prefix_cache_hit_rate=round(float(hit), 3))
        # Interactive use (the reference's REPL answers one question at a time): latency of a
        # single question through the whole protocol with nothing else in flight.
        for key, templates in (("reference_protocol", core.PromptTemplates()),
                               ("reference_protocol_persona_first", core.PromptTemplates.persona_first())):
            panel = [Member(p, be_p, templates, gp_p, decision_mode="logprob") for p in personas[:4]]
            solo = Coordinator(panel, sess)
            lat1 = []
            for i in range(4):
                r1 = solo.ask(make_questions(rng, 1, a.proto_prompt_len)[0])
                ctx.sync()
                if i:
                    lat1.append(r1.latency_s * 1e3)
            out[key]["single_question_latency_ms"] = round(float(np.median(lat1)), 1)
        del panel, solo, coord, be_p, eng_p          # the panel engine's KV pool goes back before D and E load their models
        ctx.free()
        section_done(ctx, out, "C")


# ------------------------------------------------------------------------------------ configured ensembles
def build_configured(ctx: Ctx, cfg):
    """One ensemble from a ConsensusConfig: in this process when there is one rank, placed
    over the ranks by the planner otherwise. Returns (ask_many over a GLOBAL question list,
    replicas, description of the placement)."""
    from llm_consensus_amd.app import build_ensemble
    from llm_consensus_amd.parallel.placement import plan_for_config

    if ctx.world == 1:
        ens = build_ensemble(cfg, ctx.dev)
        return ens, ens.strategy.ask_many, 1, {"gpus_per_replica": 1, "replicas": 1}
    from llm_consensus_amd.parallel.ensemble_parallel import DistributedEnsemble

    plan = plan_for_config(cfg, ctx.world)
    ens = DistributedEnsemble(cfg, ctx.comm, ctx.dev, plan)
    by_member = {m.name: list(m.devices) for m in plan.replicas[0].members}
    return ens, ens.ask_many, len(plan.replicas), {"gpus_per_replica": plan.gpus_per_replica,
                                                   "replicas": len(plan.replicas), "idle_gpus": len(plan.idle_devices),
                                                   "replica0_member_gpus": by_member}


def run_configured(ctx: Ctx, cfg, per_replica: int, prompt_len: int, steps: int, warmup: int, metric: str, **extra):
    import numpy as np

    from llm_consensus_amd.workloads import make_questions

    t0 = time.perf_counter()
    ens, ask, replicas, where = build_configured(ctx, cfg)
    ctx.sync()
    init_s = time.perf_counter() - t0
    rng = np.random.default_rng(77)                 # same questions on every rank: the ensemble is SPMD
    n_global = per_replica * replicas
    lead = ctx.rank == 0

    def ask_global(qs):
        res = ask(qs)
        return res if (ctx.world == 1 or lead) else []       # every rank holds all results; count them once

    el, n, lat, kept = timed(ctx, ask_global, lambda: make_questions(rng, n_global, prompt_len), steps, warmup,
                             lambda res: n_global)
    sec = section(metric, el, n, lat, steps, questions_per_step=n_global, placement=where, init_s=round(init_s, 1),
                  **extra)
    if kept:
        sec["mean_member_calls"] = round(float(np.mean([r.member_calls for r in kept])), 2)
    if hasattr(ens, "collectives"):
        used = ens.collectives()
        if used:
            sec["collectives"] = used
    if not ctx.a.skip_stream:
        sq = closed_loop_configured(ctx, ens, rng, 2 * n_global, 4 * n_global, prompt_len)
        if sq:
            sec["standing_queue"] = sq
    del ens, ask
    ctx.free()
    return sec


def closed_loop_configured(ctx: Ctx, ens, rng, in_flight: int, n_total: int, prompt_len: int):
    """The configured ensemble as a service under load: ``in_flight`` questions are kept in the
    engines at all times, every finished question is replaced by a new one. One process: the
    strategy's pool (SessionPool / VotePool / PhasedPool). Several ranks: lock-step rounds
    (parallel.ensemble_parallel.LockstepDriver), rank 0 feeding the questions; a question is
    sent to the other ranks in the round after it arrives and its answer returns in the round
    after its last member call, both inside the measured latency."""
    from llm_consensus_amd.workloads import make_questions

    strat = getattr(ens, "strategy", None)
    if ctx.world == 1 and not callable(getattr(strat, "pool", None)):
        return None
    qs = make_questions(rng, n_total, prompt_len)
    lat, n_done = [], 0
    ctx.comm.barrier()
    ctx.sync()
    t0 = time.perf_counter()
    if ctx.world == 1:
        pool = strat.pool()
        nxt = 0
        while nxt < min(in_flight, n_total):
            pool.submit(qs[nxt])
            nxt += 1
        while pool.active:
            for _, r in pool.pump():
                n_done += 1
                lat.append(r.latency_s * 1e3)
                if nxt < n_total:
                    pool.submit(qs[nxt])
                    nxt += 1
        mode = "closed loop: a finished question is replaced at the next engine step"
    else:
        from llm_consensus_amd.parallel.ensemble_parallel import LockstepDriver

        drv = LockstepDriver(ens)
        if ctx.rank != 0:
            drv.follow()
        else:
            sent, nxt = {}, 0
            while n_done < n_total:
                new = []
                while nxt < n_total and nxt - n_done < in_flight:
                    new.append((nxt, qs[nxt]))
                    sent[nxt] = time.perf_counter()
                    nxt += 1
                for t, _ in drv.round(new):
                    n_done += 1
                    lat.append((time.perf_counter() - sent.pop(t)) * 1e3)
            drv.round(stop=True)
        mode = "closed loop in lock-step rounds over all ranks: a finished question is replaced in the next round"
    ctx.sync()
    ctx.comm.barrier()
    el = ctx.comm.max_float(time.perf_counter() - t0)
    return {"mode": mode, "questions_in_flight": in_flight, "questions": n_total, "value": round(n_total / el, 3),
            "unit": "answers/s", "p50_latency_ms": pct(lat, 50), "p95_latency_ms": pct(lat, 95)}


# ------------------------------------------------------------------------------------ isolation
def _child_args(argv, letter: str, out_path: str):
    """The rank's own command line with the workload and output replaced."""
    keep, skip = [], False
    for tok in argv:
        if skip:
            skip = False
            continue
        if tok in ("--workload", "--out", "--placement"):
            skip = True
            continue
        if tok.startswith(("--workload=", "--out=", "--placement=")) or tok in ("--in-child", "--skip-stream"):
            continue
        keep.append(tok)
    return keep + ["--workload", letter, "--out", out_path, "--in-child"]


def run_isolated(ctx: Ctx, a, argv, letter: str, extra_args=(), pick=None):
    """Multi-GPU runs: a sharded workload (D: tensor-parallel judge, E: expert-parallel MoE) runs
    in child processes, one per rank, with a process group of their own and a time limit. Its
    collectives go through peer-to-peer kernels over IPC-mapped memory. A group whose first
    exchange fails its self-check switches to RCCL inside the child (parallel/p2p.py); should
    the child fail or stall all the same (a collective then times out and flags an error
    instead of hanging), the section is tried once more on eagerly issued RCCL collectives, and
    if that fails too the section reports the error.
    Either way the replica workloads measured before it keep their numbers and the line is
    printed. Children are new interpreters started as child processes; nothing is replaced."""
    import signal
    import tempfile

    lead = ctx.rank == 0
    errors = []
    # second attempt: the most conservative form, RCCL collectives issued eagerly
    attempts = [("p2p kernels over IPC-mapped memory", {}), ("rccl", {"LC_P2P": "0", "LC_RCCL_CAPTURE": "0"})]
    if getattr(ctx, "p2p_failed", False):
        attempts = attempts[1:]                    # an earlier section already found that path unusable here
    for label, extra in attempts:
        # rank 0's clock decides for everybody, so that all ranks take the same branch
        left = ctx.comm.broadcast_object(
            min(a.isolated_budget - getattr(ctx, "isolated_spent", 0.0), a.wall_budget - (time.perf_counter() - T_START))
            if lead else None, 0)
        if left < 60.0:
            errors.append([f"{label}: skipped, {left:.0f} s left of --isolated-budget / --wall-budget"])
            break
        t_attempt = time.perf_counter()
        limit = min(a.section_timeout, left)
        port, path = ctx.comm.broadcast_object(
            (_free_port(), os.path.join(tempfile.gettempdir(), f"bench_{letter}_{os.getpid()}_{len(errors)}.json"))
            if lead else None, 0)
        env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
        env.update(MASTER_PORT=str(port), MASTER_ADDR="127.0.0.1", **extra)
        log_path = path + f".rank{ctx.rank}.log"
        with open(log_path, "w") as log:
            p = subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + _child_args(argv, letter, path)
                                 + list(extra_args),
                                 env=env, stdout=subprocess.DEVNULL, stderr=log, start_new_session=True)
            try:
                rc = p.wait(limit)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)       # the child's own process group, nothing else
                p.wait()
                rc = "timeout"
        ctx.isolated_spent = getattr(ctx, "isolated_spent", 0.0) + time.perf_counter() - t_attempt
        if ctx.comm.all_true(rc == 0):
            sec = None
            if lead:
                full = json.loads(Path(path).read_text())
                sec = pick(full) if pick else full[{"D": "judge_consensus", "E": "moe_medoid"}[letter]]
                sec.setdefault("collectives", label)
                if errors:
                    sec["first_attempt"] = errors
            return ctx.comm.broadcast_object(sec, 0)
        tail = ""
        try:
            tail = Path(log_path).read_text()[-400:]
        except OSError:
            pass
        mine = f"{label}: rank {ctx.rank} exit {rc}" + (f" ({tail.strip().splitlines()[-1]})" if tail.strip() else "")
        errors.append([e for e in ctx.comm.all_gather_object(mine if rc != 0 else None) if e][:4])
        if not extra:
            ctx.p2p_failed = True
    return {"error": errors}


# ------------------------------------------------------------------------------------ main
def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    a = parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        return launch(a, argv)
    if env_world is not None and int(env_world) != a.gpus:
        print(f"bench.py: --gpus {a.gpus} disagrees with WORLD_SIZE={env_world}", file=sys.stderr)
        return 2
    ctx = Ctx(a)
    from llm_consensus_amd import workloads as W

    wl = a.workloads
    names = [n.strip() for n in a.models.split(",") if n.strip()]
    seq_len = a.prompt_len + a.new_tokens
    out = {
        "metric": "consensus answers/sec (node), 3x7-8B majority-vote ensemble",
        "value": None, "unit": "answers/s", "n_gpus": ctx.world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "baseline_note": "reference publishes no numbers; ratio is against its derived bound of 0.2 answers/s",
        "dtype": "bf16" if ctx.cuda else "fp32 activations on bf16 weights (CPU plumbing run)",
        "data": "synthetic prompts, random-init weights",
        "config": {
            "model": "+".join(names), "global_batch": None, "seq_len": seq_len, "prompt_len": a.prompt_len,
            "new_tokens": a.new_tokens, "sampling": {"temperature": a.temperature, "top_p": a.top_p},
            "reducer": "majority_vote (logprob-decided choices, device kernel)", "parallelism": None,
            "decode": "hipGraph" if (ctx.cuda and not a.no_graphs) else "eager",
        },
        "workloads": wl,
    }
    isolate = ctx.world > 1 and not a.in_child
    spread_extra = (isolate and "A" in wl and a.placement == "replica" and ctx.world >= len(names)
                    and not a.skip_spread)
    ctx.todo = [lab for lab, on in (
        ("A", "A" in wl), ("A.standing_queue", "A" in wl and a.placement == "replica" and not a.skip_stream),
        ("A.poisson_arrivals", "A" in wl and a.placement == "replica" and not a.skip_stream and a.poisson_load > 0),
        ("B", "B" in wl), ("C", "C" in wl and a.proto_questions > 0), ("A.one_model_per_gpu", spread_extra),
        ("D", "D" in wl), ("E", "E" in wl)) if on]
    if a.share_gpu:
        out["rehearsal"] = f"{ctx.world} ranks share ONE GPU (gloo process group, host-staged tensors): not a measurement"
    samp = W.sampling(a.temperature, a.top_p, a.new_tokens)
    if "A" in wl and a.placement == "spread":
        if ctx.world < len(names):
            print(f"bench.py: --placement spread needs {len(names)} GPUs per replica, have {ctx.world}", file=sys.stderr)
            return 2
        cfg = W.ensemble_vote_config(names, a.questions, a.prompt_len, a.new_tokens, samp, "spread", ctx.bs,
                                     a.prefill_tokens)
        sec = run_configured(ctx, cfg, a.questions, a.prompt_len, a.steps, a.warmup, out["metric"])
        out.update({k: sec[k] for k in ("value", "ms_per_step", "p50_latency_ms", "p95_latency_ms", "init_s")})
        rep = sec["placement"]["replicas"]
        out["config"].update({"global_batch": a.questions * rep, "placement": sec["placement"],
                              "parallelism": f"{rep} replica(s) x {len(names)} GPUs, one model per GPU"})
        wl_local = wl.replace("A", "")
        section_done(ctx, out, "A")
    else:
        wl_local = wl
    if any(w in wl_local for w in "ABC"):
        a.workloads = "".join(w for w in wl_local if w in "ABC")
        run_replica_workloads(ctx, out)
        a.workloads = wl
        ctx.free()
    if out["value"] is not None:
        out["vs_baseline"] = round(out["value"] / REFERENCE_BOUND_ANSWERS_PER_S, 1)
    aux_warm = min(a.warmup, 1)
    if spread_extra:
        # The same ensemble with ONE MODEL PER GPU (replicas of len(models) GPUs, the members of a
        # replica answer concurrently on their own devices and publish their results to each
        # other): lower latency per question, fewer GPUs busy than with a full ensemble per GPU.
        def pick(full):
            return {"metric": full["metric"] + ", one model per GPU", "value": full["value"], "unit": full["unit"],
                    "ms_per_step": full["ms_per_step"], "steps": full["steps"],
                    "p50_latency_ms": full.get("p50_latency_ms"), "p95_latency_ms": full.get("p95_latency_ms"),
                    "global_batch": full["config"]["global_batch"], "placement": full["config"].get("placement")}

        out["ensemble_one_model_per_gpu"] = run_isolated(ctx, a, argv, "A", ("--placement", "spread"), pick)
        section_done(ctx, out, "A.one_model_per_gpu")
    if "D" in wl:
        if isolate:
            out["judge_consensus"] = run_isolated(ctx, a, argv, "D")
        else:
            # D and E use all GPUs for ONE ensemble (the judge's TP group next to the proposers, the
            # MoE's experts over every GPU), so their batch grows with the GPU count: --judge-questions
            # and --moe-questions are per GPU.
            nq = a.judge_questions * ctx.world
            cfg = W.judge_config(ctx.world, nq, a.prompt_len, a.new_tokens, samp, a.proposer_model,
                                 a.judge_model, bs=ctx.bs, prefill_tokens=a.prefill_tokens)
            cfg.warm_judges = False if a.no_warm_judges else None      # None: when the judge's GPUs are its own
            out["judge_consensus"] = run_configured(
                ctx, cfg, nq, a.prompt_len, a.aux_steps, aux_warm,
                f"consensus answers/sec, 4x {a.proposer_model} proposers + {a.judge_model} judge (log-prob scoring)",
                judge_tp=cfg.models["judge"].tp, proposers=4)
        section_done(ctx, out, "D")
    if "E" in wl:
        if isolate:
            out["moe_medoid"] = run_isolated(ctx, a, argv, "E")
        else:
            nq = a.moe_questions * ctx.world
            cfg = W.moe_medoid_config(ctx.world, nq, a.prompt_len, a.new_tokens, samp, a.moe_samples,
                                      a.moe_model, a.embed_model, a.moe_experts, ctx.bs, a.prefill_tokens)
            out["moe_medoid"] = run_configured(
                ctx, cfg, nq, a.prompt_len, a.aux_steps, aux_warm,
                f"consensus answers/sec, {a.moe_model} proposer x{a.moe_samples} samples + {a.embed_model} embedding medoid",
                expert_parallel=cfg.models["moe"].ep, ep_mode=cfg.models["moe"].ep_mode,
                samples_per_question=a.moe_samples)
    emit(ctx, out)
    if ctx.dist is not None:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
