"""Assembles a running ensemble from a ConsensusConfig: models -> engines -> backends ->
members -> strategy object.

Counterpart of the reference's start-up code (/root/reference/src/main.rs:359-426), which
registers four hard-coded personas that all call the same remote model. Here the member
list, the models they run on, the placement of models on GPUs and the consensus strategy
come from configuration; with no configuration the result is the reference's topology
(four personas, one shared model, unanimity loop).
"""
from __future__ import annotations

from dataclasses import dataclass
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

log = get_logger("consensus.app")


@dataclass
class Ensemble:
    config: ConsensusConfig
    members: List[Member]
    backends: Dict[str, Backend]
    strategy: object
    device: torch.device

    def ask(self, question: str):
        return self.strategy.ask(question)

    def ask_many(self, questions):
        return self.strategy.ask_many(list(questions))

    def stats(self) -> Dict[str, Dict[str, float]]:
        return {name: b.stats() for name, b in self.backends.items()}


def default_device() -> torch.device:
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


def model_config_of(spec: ModelSpec):
    return from_hf_config(spec.path) if spec.path else get_preset(spec.preset or "tiny-llama")


def build_backend(spec: ModelSpec, device: torch.device, comm=None, tp_rank: int = 0) -> EngineBackend:
    """One engine for ``spec``. With tp/ep > 1, ``comm`` is the model's group and this
    process holds shard ``tp_rank`` (= comm.rank); a single process ignores tp/ep."""
    grouped = comm is not None and comm.world > 1
    model = build_model(spec, device, tp_rank=tp_rank, tp_size=spec.tp if grouped else 1, comm=comm,
                        ep_size=spec.ep if grouped else 1, ep_mode=spec.ep_mode)
    max_len = min(model.cfg.max_position, 4096)
    eng = Engine(model, block_size=spec.kv_block_size if device.type == "cuda" else 16,
                 max_batch_seqs=spec.max_batch_seqs, max_prefill_tokens=spec.max_prefill_tokens,
                 max_model_len=max_len, kv_fraction=spec.kv_fraction,
                 graph_buckets=spec.graph_batch_buckets, name=spec.name,
                 num_kv_blocks=None if device.type == "cuda" else 512)
    tok = make_tokenizer(spec.tokenizer, model.cfg.vocab_size)
    return EngineBackend(eng, tok, name=spec.name, stop_at_eos=spec.tokenizer is not None)


def build_ensemble(cfg: ConsensusConfig, device: Optional[torch.device] = None,
                   backends: Optional[Dict[str, Backend]] = None) -> Ensemble:
    """``backends`` may pre-supply a Backend per model name (tests inject scripted ones)."""
    device = device or default_device()
    backends = dict(backends or {})
    used_models = {m.model for m in cfg.members}
    if device.type == "cuda":
        plan = plan_for_config(cfg, torch.cuda.device_count())
        if not plan.feasible:
            raise RuntimeError(f"placement failed: {plan.error}")
        log.debug("placement: %d GPU(s) per replica, %d replica(s)", plan.gpus_per_replica, len(plan.replicas))
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


def make_strategy(cfg: ConsensusConfig, members: List[Member], device):
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
        return JudgeLogprob(voters, judges or voters[-1:], device=dev)
    if cfg.strategy == "embedding_medoid":
        return EmbeddingMedoid(voters, (embedders or voters)[0], device=dev)
    raise ValueError(cfg.strategy)
