#!/usr/bin/env python3
"""Headline benchmark: consensus answers/sec and p50 latency on one MI355X node.

Workload A (the timed "step"): a heterogeneous 3-model ensemble (Llama-3-8B, Qwen2.5-7B,
Mistral-7B shapes, bf16, random-init weights) answers a batch of synthetic
multiple-choice questions; every member prefills the prompt, samples a fixed-length
answer with top-p/temperature, decides its choice by on-device log-probs at the last
position, and the votes are reduced by the majority-vote kernel. One step = one batch of
``--questions`` questions per GPU through the whole pipeline (tokenize -> 3x generate ->
vote -> cross-GPU gather of the results over RCCL).

Workload B (reported alongside, same run): Llama-3-8B self-consistency with K=16 sampled
chains per question forked from one prefill (shared-prefix paged KV), answer vote.

Scaling is weak: every GPU hosts a full ensemble replica (47 GB of weights fit easily in
288 GB) and serves its own shard of the questions; ranks only exchange results.

Launch: ``python bench.py`` (1 GPU) or
``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N``.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

REFERENCE_BOUND_ANSWERS_PER_S = 0.2  # derived upper bound for the reference (it publishes no numbers)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--models", default="llama-3-8b,qwen2.5-7b,mistral-7b")
    ap.add_argument("--questions", type=int, default=64, help="questions per GPU per step (workload A)")
    ap.add_argument("--prompt-len", type=int, default=512)
    ap.add_argument("--new-tokens", type=int, default=64)
    ap.add_argument("--k-chains", type=int, default=16)
    ap.add_argument("--sc-questions", type=int, default=16, help="questions per GPU per step (workload B)")
    ap.add_argument("--sc-steps", type=int, default=None, help="timed steps of workload B (default: --steps)")
    ap.add_argument("--skip-self-consistency", action="store_true")
    ap.add_argument("--skip-stream", action="store_true", help="skip the standing-queue runs of workload A")
    ap.add_argument("--proto-questions", type=int, default=8,
                    help="questions per GPU per step of the reference-protocol workload (0 = skip)")
    ap.add_argument("--proto-prompt-len", type=int, default=128)
    ap.add_argument("--temperature", type=float, default=0.8)
    ap.add_argument("--top-p", type=float, default=0.95)
    ap.add_argument("--prefill-tokens", type=int, default=8192, help="token budget of one prefill step")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--out", default=None, help="also write the JSON line to this file")
    return ap.parse_args()


def make_questions(rng: np.random.Generator, n: int, prompt_len: int):
    """Synthetic multiple-choice prompts of exactly prompt_len tokens under the byte-level
    tokenizer (BOS + prompt_len-1 printable ASCII bytes)."""
    tail = "\nA. first B. second C. third D. fourth\nAnswer:"
    body = prompt_len - 1 - len(tail)
    qs = []
    for _ in range(n):
        chars = rng.integers(97, 123, size=body)
        chars[rng.integers(0, body, size=body // 6)] = 32
        qs.append(bytes(chars.astype(np.uint8)).decode("ascii") + tail)
    return qs


def main() -> None:
    a = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist  # type: ignore

        dist.init_process_group("nccl", device_id=dev)

    from llm_consensus_amd._native import core
    from llm_consensus_amd.consensus.backend import GenParams
    from llm_consensus_amd.consensus.members import Member
    from llm_consensus_amd.consensus.strategies import MajorityVote, SelfConsistency
    from llm_consensus_amd.engine.backend import EngineBackend
    from llm_consensus_amd.engine.engine import Engine
    from llm_consensus_amd.models.config import get_preset
    from llm_consensus_amd.models.transformer import TransformerModel
    from llm_consensus_amd.parallel.comm import make_comm
    from llm_consensus_amd.parallel.replica import gather_winners
    from llm_consensus_amd.utils.tokenizer import ByteLevelTokenizer

    comm = make_comm(dist, dev)
    names = [n.strip() for n in a.models.split(",") if n.strip()]
    bs = 64
    seq_len = a.prompt_len + a.new_tokens
    blocks_per_seq = (seq_len + bs - 1) // bs + 1
    max_seqs = max(a.questions, a.sc_questions * a.k_chains)
    t_init = time.perf_counter()
    members = []
    personas = core.default_personas()
    for i, name in enumerate(names):
        cfg = get_preset(name)
        model = TransformerModel.random_init(cfg, dev, seed=1234 + i)
        n_blocks = (max_seqs + 8) * blocks_per_seq + 64
        eng = Engine(model, num_kv_blocks=n_blocks, block_size=bs, max_batch_seqs=max(256, max_seqs),
                     max_prefill_tokens=a.prefill_tokens, max_model_len=seq_len + 64, use_graphs=not a.no_graphs,
                     name=name)
        be = EngineBackend(eng, ByteLevelTokenizer(cfg.vocab_size), stop_at_eos=False)
        gp = GenParams(temperature=a.temperature, top_p=a.top_p, max_new_tokens=a.new_tokens, seed=17 * i)
        members.append(Member(personas[i % len(personas)], be, params=gp))
    torch.cuda.synchronize()
    init_s = time.perf_counter() - t_init

    vote = MajorityVote(members, choices="ABCD", decision_mode="logprob", device=dev)
    sc = SelfConsistency(members[0], k_chains=a.k_chains, choices="ABCD", decision_mode="logprob", device=dev)
    rng = np.random.default_rng(1000 + rank)

    def gather_answers(results):
        """Cross-GPU gather of the per-question winners (class ids) to every rank."""
        return gather_winners(comm, results, "ABCD", dev)

    def run_phase(strategy, n_q, steps, warmup, prompt_len=a.prompt_len, gather=gather_answers, keep=None):
        lat = []
        for _ in range(warmup):
            gather(strategy.ask_many(make_questions(rng, n_q, prompt_len)))
        comm.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_ans = 0
        for _ in range(steps):
            res = strategy.ask_many(make_questions(rng, n_q, prompt_len))
            allw = gather(res)
            n_ans += int(allw.numel())
            lat += [r.latency_s * 1e3 for r in res]
            if keep is not None:
                keep.extend(res)
        torch.cuda.synchronize()
        comm.barrier()
        el = comm.max_float(time.perf_counter() - t0)
        return el, n_ans, lat

    st0 = [m.backend.stats() for m in members]
    el_a, n_a, lat_a = run_phase(vote, a.questions, a.steps, a.warmup)
    st1 = [m.backend.stats() for m in members]
    lat_all = comm.gather_floats(lat_a)
    out = {
        "metric": "consensus answers/sec (node), 3x7-8B majority-vote ensemble",
        "value": round(n_a / el_a, 3),
        "unit": "answers/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(el_a / a.steps * 1e3, 2),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": round(n_a / el_a / REFERENCE_BOUND_ANSWERS_PER_S, 1),
        "baseline_note": "reference publishes no numbers; ratio is against its derived bound of 0.2 answers/s",
        "dtype": "bf16",
        "data": "synthetic prompts, random-init weights",
        "p50_latency_ms": round(float(np.percentile(lat_all, 50)), 1),
        "p95_latency_ms": round(float(np.percentile(lat_all, 95)), 1),
        "config": {
            "model": "+".join(names),
            "global_batch": a.questions * world,
            "seq_len": seq_len,
            "prompt_len": a.prompt_len,
            "new_tokens": a.new_tokens,
            "sampling": {"temperature": a.temperature, "top_p": a.top_p},
            "reducer": "majority_vote (logprob-decided choices, device kernel)",
            "parallelism": f"dp{world} (one full ensemble replica per GPU)",
            "decode": "hipGraph" if not a.no_graphs else "eager",
        },
        "init_s": round(init_s, 1),
        # host-side view of each member engine over warmup + timed steps (the members' steps overlap on
        # the GPU, so these wall-clock spans add up to more than the step time)
        "member_engines": {
            names[i]: {k: round(st1[i].get(k, 0.0) - st0[i].get(k, 0.0), 3)
                       for k in ("prefill.steps", "prefill.seconds", "decode.steps", "decode.seconds",
                                 "decode.wait_seconds")}
            for i in range(len(names))
        },
    }
    if not a.skip_stream:
        # The same ensemble under a standing queue: the questions of all timed steps are submitted
        # together instead of one step's worth at a time, first with prefill-only / decode-only
        # steps, then with mixed steps (decoding chains ride along in the steps that prefill later
        # prompts, so one pass over the weights serves both).
        def run_stream(n_total):
            comm.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = vote.ask_many(make_questions(rng, n_total, a.prompt_len))
            n = int(gather_answers(res).numel())
            torch.cuda.synchronize()
            comm.barrier()
            return comm.max_float(time.perf_counter() - t0), n, [r.latency_s * 1e3 for r in res]

        out["standing_queue"] = {"questions_in_flight_per_gpu": a.questions * a.steps}
        for mix in (False, True):
            for m in members:
                m.backend.engine.sched.set_mix_decode(mix)
            run_stream(a.questions * max(1, a.warmup))
            el_s, n_s, lat_s = run_stream(a.questions * a.steps)
            st = members[0].backend.stats()
            out["standing_queue"]["mixed_steps" if mix else "separate_steps"] = {
                "value": round(n_s / el_s, 3), "unit": "answers/s",
                "ms_per_64_questions": round(el_s / a.steps * 1e3, 2),
                "p50_latency_ms": round(float(np.percentile(comm.gather_floats(lat_s), 50)), 1),
            }
        out["standing_queue"]["rider_tokens_first_member"] = int(st.get("prefill.rider_tokens", 0))
        for m in members:
            m.backend.engine.sched.set_mix_decode(False)
    if not a.skip_self_consistency:
        steps_b = a.sc_steps or a.steps
        st_before = members[0].backend.stats()
        el_b, n_b, lat_b = run_phase(sc, a.sc_questions, steps_b, a.warmup)
        lat_b_all = comm.gather_floats(lat_b)
        st = members[0].backend.stats()

        def delta(key):
            return st.get(key, 0.0) - st_before.get(key, 0.0)

        out["self_consistency"] = {
            "metric": f"consensus answers/sec, {names[0]} K={a.k_chains} self-consistency",
            "value": round(n_b / el_b, 3),
            "ms_per_step": round(el_b / steps_b * 1e3, 2),
            "steps": steps_b,
            "questions_per_gpu": a.sc_questions,
            "p50_latency_ms": round(float(np.percentile(lat_b_all, 50)), 1),
            "p95_latency_ms": round(float(np.percentile(lat_b_all, 95)), 1),
            "chains_tokens_per_s": round(n_b * a.k_chains * a.new_tokens / el_b, 1),
            # the K chains of a question are forks of one prefilled prompt: they attend ONE copy of
            # its full KV blocks (shared-prefix decode kernel); the prefix cache is not involved
            "chains_per_question": a.k_chains,
            "prompt_kv_blocks_shared_by_the_chains": a.prompt_len // bs,
            "prompt_kv_reads_saved": round(1.0 - 1.0 / a.k_chains, 3),
            "cow_copies": int(delta("kv.cow_copies_total")),
            "decode_ms_per_step": round(1e3 * delta("decode.seconds") / max(1.0, delta("decode.steps")), 3),
            "decode_host_ms_per_step": round(1e3 * (delta("decode.seconds") - delta("decode.wait_seconds"))
                                             / max(1.0, delta("decode.steps")), 3),
            "decode_host_us_per_step": {k: round(1e6 * delta(f"host.{k}_seconds") / max(1.0, delta("decode.steps")), 1)
                                        for k in ("schedule", "decode_fill", "decode_launch", "decode_finish")},
        }
    if a.proto_questions > 0:
        # The reference's own topology and loop: four personas share ONE model, a random member
        # proposes, everyone (proposer included) judges, a random dissenter refines, at most five
        # evaluation rounds (/root/reference/src/main.rs:220-311, 359-426). Verdicts are the
        # on-device two-way log-prob decision because random-init weights never spell a verdict.
        from llm_consensus_amd.consensus.protocol import Coordinator

        model0 = members[0].backend.engine.model
        plen = 2048 + 2 * a.proto_prompt_len + 2 * a.new_tokens        # evaluate prompt: template + question + answer
        n_seq = a.proto_questions * 4
        eng_p = Engine(model0, num_kv_blocks=(n_seq + 4) * ((plen + bs - 1) // bs + 1) + 64, block_size=bs,
                       max_batch_seqs=max(64, n_seq), max_prefill_tokens=8192, max_model_len=plen,
                       use_graphs=not a.no_graphs, name=names[0] + ".panel")
        be_p = EngineBackend(eng_p, ByteLevelTokenizer(model0.cfg.vocab_size), stop_at_eos=False)
        gp_p = GenParams(temperature=a.temperature, top_p=a.top_p, max_new_tokens=a.new_tokens, seed=99)
        def count_answers(results):
            # Answers are free text here; ranks exchange one row per finished question.
            done = torch.tensor([[int(r.forced)] for r in results], dtype=torch.int32, device=dev)
            return comm.all_gather_rows(done)

        # Two prompt layouts: the reference's (domain named a third of the way into the evaluate
        # prompt) and the same instructions with the persona text moved to the end, which lets
        # the four evaluate prompts of a round share most of their prefill.
        # Random-init weights give all four personas the same verdict (always Good here), so the
        # loop ends after one round; the third run injects seeded dissent (a Good verdict becomes
        # NeedsRefinement with probability 0.25) to load the refinement path as well.
        for key, templates, dissent in (
                ("reference_protocol", core.PromptTemplates(), 0.0),
                ("reference_protocol_persona_last", core.PromptTemplates.persona_last(), 0.0),
                ("reference_protocol_persona_last_dissent25", core.PromptTemplates.persona_last(), 0.25)):
            panel = [Member(p, be_p, templates, gp_p, decision_mode="logprob", dissent_rate=dissent,
                            dissent_seed=1000 * rank + i) for i, p in enumerate(personas[:4])]
            sess = core.SessionConfig()
            sess.max_rounds, sess.seed = 5, 4242 + rank
            coord = Coordinator(panel, sess)
            kept = []
            st0 = be_p.stats()
            el_p, n_p, lat_p = run_phase(coord, a.proto_questions, a.steps, a.warmup, a.proto_prompt_len,
                                         count_answers, kept)
            lat_p_all = comm.gather_floats(lat_p)
            st = be_p.stats()
            hit = (st["kv.prefix_hit_tokens"] - st0["kv.prefix_hit_tokens"]) / max(1, st["kv.prompt_tokens"] - st0["kv.prompt_tokens"])
            out[key] = {
                "metric": f"consensus answers/sec, reference protocol (4 personas on {names[0]}, unanimity loop, <= 5 rounds)",
                "value": round(n_p / el_p, 3),
                "ms_per_step": round(el_p / a.steps * 1e3, 2),
                "steps": a.steps,
                "questions_per_gpu": a.proto_questions,
                "question_tokens": a.proto_prompt_len,
                "p50_latency_ms": round(float(np.percentile(lat_p_all, 50)), 1),
                "p95_latency_ms": round(float(np.percentile(lat_p_all, 95)), 1),
                "mean_rounds": round(float(np.mean([r.rounds for r in kept])), 2),
                "mean_member_calls": round(float(np.mean([r.member_calls for r in kept])), 2),
                "forced_fraction": round(float(np.mean([r.forced for r in kept])), 3),
                "injected_dissent_rate": dissent,
                "prefix_cache_hit_rate": round(float(hit), 3),
            }
        # Interactive use (the reference's REPL answers one question at a time): latency of a
        # single question through the whole protocol with nothing else in flight.
        panel = [Member(p, be_p, core.PromptTemplates(), gp_p, decision_mode="logprob") for p in personas[:4]]
        solo = Coordinator(panel, sess)
        lat1 = []
        for i in range(4):
            r1 = solo.ask(make_questions(rng, 1, a.proto_prompt_len)[0])
            torch.cuda.synchronize()
            if i:
                lat1.append(r1.latency_s * 1e3)
        out["reference_protocol"]["single_question_latency_ms"] = round(float(np.median(lat1)), 1)
    if rank == 0:
        line = json.dumps(out)
        print(line, flush=True)
        if a.out:
            Path(a.out).parent.mkdir(parents=True, exist_ok=True)
            Path(a.out).write_text(line + "\n")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
