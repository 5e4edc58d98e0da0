"""Config -> placement plan glue over the native planner (csrc/core/placement.cpp).

The reference has nothing to place: its four personas share one remote model
(/root/reference/src/main.rs:82-86, 359-426). Here every distinct model of an ensemble
occupies HBM, so before anything is loaded the planner decides which GPUs of the node
form one ensemble replica and how many replicas fit; each rank (one process per GPU)
then loads the members whose device list names it and questions are sharded over
replicas.
"""
from __future__ import annotations

from typing import List

from .._native import core
from ..models.config import from_hf_config, get_preset

HBM_BYTES_MI355X = 288 * 1000 ** 3


def member_specs(cfg, min_kv_tokens: int = 65536, unsharded: bool = False) -> List["core.MemberSpec"]:
    """One planner entry per configured member; members naming the same model share weights.
    ``unsharded`` plans every model whole (tp = ep = 1): what ONE process does with it."""
    out = []
    for m in cfg.members:
        spec = cfg.models[m.model]
        mc = from_hf_config(spec.path) if spec.path else get_preset(spec.preset or "tiny-llama")
        ms = core.MemberSpec()
        ms.name, ms.model_key = m.name, spec.name
        ms.weight_bytes = mc.weight_bytes()
        ms.min_kv_bytes = mc.kv_bytes_per_token() * min_kv_tokens
        ms.tp, ms.ep = (1, 1) if unsharded else (spec.tp, spec.ep)
        out.append(ms)
    return out


def plan_for_config(cfg, num_gpus: int, hbm_bytes: int = HBM_BYTES_MI355X,
                    reserve_fraction: float = 0.08, single_process: bool = False) -> "core.PlacementPlan":
    """``single_process``: the plan of one process that holds every model whole on one device
    (packed, tp / ep ignored); feasible iff the ensemble fits one GPU's HBM."""
    if single_process:
        return core.plan_placement(member_specs(cfg, unsharded=True), 1, hbm_bytes, reserve_fraction,
                                   core.PlacementPolicy.Pack)
    policy = core.PlacementPolicy.Spread if cfg.placement == "spread" else core.PlacementPolicy.Pack
    return core.plan_placement(member_specs(cfg), max(1, num_gpus), hbm_bytes, reserve_fraction, policy)
