"""Config -> placement plan glue over the native planner (csrc/core/placement.cpp).

The reference has nothing to place: its four personas share one remote model
(/root/reference/src/main.rs:82-86, 359-426). Here every distinct model of an ensemble
occupies HBM, so before anything is loaded the planner decides which GPUs of the node
form one ensemble replica and how many replicas fit; each rank (one process per GPU)
then loads the members whose device list names it and questions are sharded over
replicas.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List

from .._native import core
from ..models.config import from_hf_config, get_preset

HBM_BYTES_MI355X = 288 * 1000 ** 3


def member_specs(cfg, min_kv_tokens: int = 65536) -> List["core.MemberSpec"]:
    """One planner entry per configured member; members naming the same model share weights."""
    out = []
    for m in cfg.members:
        spec = cfg.models[m.model]
        mc = from_hf_config(spec.path) if spec.path else get_preset(spec.preset or "tiny-llama")
        ms = core.MemberSpec()
        ms.name, ms.model_key = m.name, spec.name
        ms.weight_bytes = mc.weight_bytes()
        ms.min_kv_bytes = mc.kv_bytes_per_token() * min_kv_tokens
        ms.tp, ms.ep = spec.tp, spec.ep
        out.append(ms)
    return out


def plan_for_config(cfg, num_gpus: int, hbm_bytes: int = HBM_BYTES_MI355X,
                    reserve_fraction: float = 0.08) -> "core.PlacementPlan":
    policy = core.PlacementPolicy.Spread if cfg.placement == "spread" else core.PlacementPolicy.Pack
    return core.plan_placement(member_specs(cfg), max(1, num_gpus), hbm_bytes, reserve_fraction, policy)


@dataclass
class RankAssignment:
    replica: int                   # index of the replica this rank belongs to (-1: idle)
    num_replicas: int
    local_index: int               # position of this rank inside its replica
    replica_ranks: List[int]       # ranks of the replica (for its process group)
    models: Dict[str, List[int]]   # model key -> ranks of its TP/EP group, for models this rank hosts


def assignment_for_rank(plan: "core.PlacementPlan", cfg, rank: int) -> RankAssignment:
    """What one process (rank == node-local GPU id) has to load under ``plan``."""
    for ri, rep in enumerate(plan.replicas):
        devs = list(rep.devices)
        if rank not in devs:
            continue
        models: Dict[str, List[int]] = {}
        by_name = {m.name: m for m in cfg.members}
        for mp in rep.members:
            if rank in mp.devices:
                models.setdefault(cfg.models[by_name[mp.name].model].name, list(mp.devices))
        return RankAssignment(ri, len(plan.replicas), devs.index(rank), devs, models)
    return RankAssignment(-1, len(plan.replicas), 0, [], {})
