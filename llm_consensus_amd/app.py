"""Assembles a running ensemble from a ConsensusConfig: models -> engines -> backends ->
members -> strategy object.

Counterpart of the reference's start-up code (/root/reference/src/main.rs:359-426), which
registers four hard-coded personas that all call the same remote model. Here the member
list, the models they run on, the placement of models on GPUs and the consensus strategy
come from configuration; with no configuration the result is the reference's topology
(four personas, one shared model, unanimity loop).
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from ._native import core
from .config import ConsensusConfig, ModelSpec
from .consensus.backend import Backend, GenParams
from .consensus.members import Member
from .consensus.protocol import Coordinator
from .consensus.strategies import EmbeddingMedoid, JudgeLogprob, MajorityVote, SelfConsistency
from .engine.backend import EngineBackend
from .engine.engine import Engine
from .models.config import from_hf_config, get_preset
from .models.loader import build_model
from .parallel.placement import plan_for_config
from .utils.log import get_logger
from .utils.tokenizer import make_tokenizer
from .utils.trace import instant, span

log = get_logger("consensus.app")


@dataclass
class Ensemble:
    config: ConsensusConfig
    members: List[Member]
    backends: Dict[str, Backend]
    strategy: object
    device: torch.device

    metrics: "core.Metrics" = field(default_factory=core.Metrics)

    def ask(self, question: str):
        return self.ask_many([question])[0]

    def ask_many(self, questions):
        """Answers a batch and records the ensemble-level counters: answers, batches, wall
        time, rounds per question, member calls, agreement, unanimity, per-question latency.
        The reference counts nothing (its only timer paces a poll loop, main.rs:449-454)."""
        qs = list(questions)
        t0 = time.perf_counter()
        with span("ask_many", "ensemble", questions=len(qs), strategy=self.config.strategy):
            results = self.strategy.ask_many(qs)
        self.metrics.inc("batches")
        self.metrics.inc("seconds", time.perf_counter() - t0)
        self.record(results)
        return results

    def record(self, results) -> None:
        """Ensemble-level counters for finished questions (also used by drivers that do not go
        through ask_many, such as the server's continuous loop)."""
        m = self.metrics
        m.inc("answers", len(results))
        for r in results:
            m.observe("latency_ms", r.latency_s * 1e3)
            m.observe("rounds", r.rounds)
            m.observe("agreement", r.agreement)
            m.inc("member_calls", r.member_calls)
            m.inc("unanimous", 1 if r.unanimous else 0)
            m.inc("forced", 1 if getattr(r, "forced", False) else 0)
            instant("session.done", "ensemble", rounds=r.rounds, calls=r.member_calls, unanimous=r.unanimous,
                    latency_ms=round(r.latency_s * 1e3, 2))

    def warm_up(self, **kw) -> Dict[str, Dict[str, float]]:
        """First-use costs of every member engine (kernel code objects, launch attributes, the
        decode graphs of every batch bucket and context class) are paid now instead of by the
        first questions. The reference's first question costs what every question costs
        (/root/reference/src/main.rs:428-445); so should ours."""
        t0 = time.perf_counter()
        out = {name: b.warm_up(**kw) for name, b in sorted(self.backends.items())}
        log.info("warm-up: %d decode graph(s) recorded in %.1f s",
                 sum(int(v.get("graphs", 0)) for v in out.values()), time.perf_counter() - t0)
        return out

    def stats(self) -> Dict[str, Dict[str, float]]:
        """Per-model engine statistics plus a ``consensus`` section (answers/s, latency
        percentiles, rounds, agreement) and a ``device`` section (HBM use)."""
        out: Dict[str, Dict[str, float]] = {name: b.stats() for name, b in self.backends.items()}
        c = dict(self.metrics.snapshot())
        if c.get("seconds", 0) > 0:
            c["answers_per_s"] = c.get("answers", 0) / c["seconds"]
        if c.get("answers", 0) > 0:
            c["unanimous_rate"] = c.get("unanimous", 0) / c["answers"]
        for name, st in out.items():
            if st.get("decode.seconds", 0) > 0:
                c[f"decode_tokens_per_s.{name}"] = st.get("decode.tokens", 0) / st["decode.seconds"]
        out["consensus"] = c
        if self.device.type == "cuda":
            free, total = torch.cuda.mem_get_info(self.device)
            out["device"] = {"hbm_total_bytes": float(total), "hbm_used_bytes": float(total - free),
                             "hbm_allocated_bytes": float(torch.cuda.memory_allocated(self.device))}
        return out


def default_device() -> torch.device:
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


def model_config_of(spec: ModelSpec):
    return from_hf_config(spec.path) if spec.path else get_preset(spec.preset or "tiny-llama")


def build_backend(spec: ModelSpec, device: torch.device, comm=None, tp_rank: int = 0) -> EngineBackend:
    """One engine for ``spec``. With tp/ep > 1, ``comm`` is the model's group and this
    process holds shard ``tp_rank`` (= comm.rank); a single process ignores tp/ep."""
    grouped = comm is not None and comm.world > 1
    tp_size = spec.tp if grouped else 1
    model = build_model(spec, device, tp_rank=tp_rank if tp_size > 1 else 0, tp_size=tp_size, comm=comm,
                        ep_size=spec.ep if grouped else 1, ep_mode=spec.ep_mode)
    max_len = min(model.cfg.max_position, spec.max_model_len) if spec.max_model_len > 0 else model.cfg.max_position
    eng = Engine(model, block_size=spec.kv_block_size if device.type == "cuda" else 16,
                 max_batch_seqs=spec.max_batch_seqs, max_prefill_tokens=spec.max_prefill_tokens,
                 max_model_len=max_len, kv_fraction=spec.kv_fraction,
                 graph_buckets=spec.graph_batch_buckets, name=spec.name, mix_decode=spec.mix_decode,
                 optimistic_admission=spec.admission == "optimistic",
                 num_kv_blocks=spec.kv_blocks or (None if device.type == "cuda" else 512))
    if grouped:
        # Ranks of a group finish loading at different times; none of them may launch a step
        # (its collectives wait for the peers with a timeout) before all of them are up.
        comm.barrier()
    tok = make_tokenizer(spec.tokenizer, model.cfg.vocab_size)
    return EngineBackend(eng, tok, name=spec.name, stop_at_eos=spec.tokenizer is not None,
                         chat_format=spec.chat_format)


def build_ensemble(cfg: ConsensusConfig, device: Optional[torch.device] = None,
                   backends: Optional[Dict[str, Backend]] = None) -> Ensemble:
    """``backends`` may pre-supply a Backend per model name (tests inject scripted ones)."""
    device = device or default_device()
    backends = dict(backends or {})
    used_models = {m.model for m in cfg.members}
    if device.type == "cuda":
        # this process holds every model whole on ONE device: that is what has to fit
        packed = plan_for_config(cfg, 1, single_process=True)
        if not packed.feasible:
            raise RuntimeError(f"placement failed: {packed.error}")
        want = max([max(m.tp, m.ep) for m in cfg.models.values()] + [1])
        plan = plan_for_config(cfg, max(torch.cuda.device_count(), want))
        log.debug("placement: %d GPU(s) per replica, %d replica(s)", plan.gpus_per_replica, len(plan.replicas))
        if plan.feasible and plan.gpus_per_replica > 1:
            # one process hosts every model on ONE device whatever the plan says: say so (the front
            # ends start one rank per planned GPU themselves, parallel/launch.py)
            log.warning("this ensemble is planned for %d GPUs per replica but is being built in one process: every "
                        "model goes to %s and tp / ep settings are ignored", plan.gpus_per_replica, device)
    for name in used_models:
        if name not in backends:
            backends[name] = build_backend(cfg.models[name], device)
    members = make_members(cfg, backends)
    strategy = make_strategy(cfg, members, device)
    return Ensemble(cfg, members, backends, strategy, device)


def make_members(cfg: ConsensusConfig, backends: Dict[str, Backend]) -> List[Member]:
    gp = GenParams(temperature=cfg.sampling.temperature, top_p=cfg.sampling.top_p, top_k=cfg.sampling.top_k,
                   max_new_tokens=cfg.sampling.max_new_tokens, seed=cfg.sampling.seed)
    templates = cfg.templates()
    return [
        Member(core.Persona(m.name, m.domain, m.tuning), backends[m.model], templates, gp, cfg.parse_mode_enum(),
               cfg.decision_mode, m.role, m.weight)
        for m in cfg.members
    ]


def make_strategy(cfg: ConsensusConfig, members: List[Member], device, judges_have_own_gpus: bool = False):
    """``judges_have_own_gpus``: what ``warm_judges = None`` resolves to. Prefilling the question
    part of the judge prompts during the proposer phase hides that work when the judges' GPUs
    would otherwise idle; on GPUs shared with the proposers it only moves the same work
    (measured on one MI355X: 4.46 answers/s with, 4.52 without)."""
    dev = device if str(device).startswith("cuda") else None
    voters = [m for m in members if m.role == "member"] or members
    judges = [m for m in members if m.role == "judge"]
    embedders = [m for m in members if m.role == "embedder"]
    if cfg.strategy == "reference_protocol":
        return Coordinator(members, cfg.session_config(), cfg.deadline_s, cfg.transcript_path)
    if cfg.strategy == "majority_vote":
        return MajorityVote(voters, cfg.choices, cfg.tie_break_enum(), cfg.decision_mode, device=dev)
    if cfg.strategy == "self_consistency":
        return SelfConsistency(voters[0], cfg.k_chains, cfg.choices, cfg.tie_break_enum(), cfg.decision_mode,
                               device=dev)
    if cfg.strategy == "judge_logprob":
        return JudgeLogprob(voters, judges or voters[-1:], device=dev, micro_batches=cfg.micro_batches,
                            min_micro_batch=cfg.min_micro_batch,
                            warm_judges=judges_have_own_gpus if cfg.warm_judges is None else cfg.warm_judges)
    if cfg.strategy == "embedding_medoid":
        return EmbeddingMedoid(voters, (embedders or voters)[0], samples_per_member=cfg.samples_per_member, device=dev,
                               micro_batches=cfg.micro_batches, min_micro_batch=cfg.min_micro_batch)
    raise ValueError(cfg.strategy)
