"""The benchmark workloads as configurations.

Five ensembles are measured (synthetic multiple-choice prompts, random-init weights of the
named architectures):

    A  heterogeneous 3 x 7-8B ensemble, majority vote
    B  one 8B model, K sampled chains per question, answer vote (self-consistency)
    C  the reference's own loop: four personas on one model, propose -> evaluate -> refine
       until unanimous, at most five rounds (/root/reference/src/main.rs:220-314, 359-426)
    D  four 8B proposers + a 70B judge (tensor parallel over up to four GPUs) that scores
       every candidate by log P(Yes) - log P(No); the analogue of the reference's fan-in of
       verdicts (main.rs:259-291) with a dedicated judge
    E  a Mixtral-shape MoE proposer (expert parallel over up to eight GPUs) sampling several
       candidates per question + an embedding model; the cosine medoid of the candidates wins

``bench.py`` and the strategy benchmarks build their ensembles from these functions, so the
same description drives one process on one GPU (``app.build_ensemble``) and one process per
GPU (``parallel.ensemble_parallel.DistributedEnsemble``): the planner decides where members
live, and the tensor / expert parallel degree follows the number of GPUs.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np

from .config import ConsensusConfig, MemberConfig, ModelSpec, SamplingConfig


def make_questions(rng: np.random.Generator, n: int, prompt_len: int) -> List[str]:
    """Synthetic multiple-choice prompts of exactly ``prompt_len`` tokens under the byte-level
    tokenizer (BOS + prompt_len - 1 printable ASCII bytes)."""
    tail = "\nA. first B. second C. third D. fourth\nAnswer:"
    body = prompt_len - 1 - len(tail)
    if body < 8:
        raise ValueError(f"prompt_len {prompt_len} is too short for the answer options")
    qs = []
    for _ in range(n):
        chars = rng.integers(97, 123, size=body)
        chars[rng.integers(0, body, size=body // 6)] = 32
        qs.append(bytes(chars.astype(np.uint8)).decode("ascii") + tail)
    return qs


def _blocks(tokens: int, bs: int) -> int:
    return (tokens + bs - 1) // bs + 1


def _spec(name: str, preset: str, seqs: int, tokens_per_seq: int, bs: int, prefill_tokens: int, seed: int,
          max_len: int, **kw) -> ModelSpec:
    return ModelSpec(name=name, preset=preset, kv_block_size=bs, kv_blocks=(seqs + 8) * _blocks(tokens_per_seq, bs) + 64,
                     max_batch_seqs=max(256, seqs), max_prefill_tokens=prefill_tokens, max_model_len=max_len,
                     seed=seed, **kw)


def sampling(temperature: float, top_p: float, new_tokens: int, seed: int = 0) -> SamplingConfig:
    return SamplingConfig(temperature=temperature, top_p=top_p, max_new_tokens=new_tokens, seed=seed)


def ensemble_vote_config(models: Sequence[str], questions: int, prompt_len: int, new_tokens: int, samp: SamplingConfig,
                         placement: str = "pack", bs: int = 64, prefill_tokens: int = 8192) -> ConsensusConfig:
    """Workload A: one member per model, majority vote over log-prob decided choices.
    ``placement='spread'`` gives every model a GPU of its own (needs len(models) GPUs per
    replica); ``'pack'`` co-locates them and replicates the ensemble on every GPU."""
    seq = prompt_len + new_tokens
    specs: Dict[str, ModelSpec] = {}
    members = []
    for i, preset in enumerate(models):
        name = f"m{i}-{preset}"
        specs[name] = _spec(name, preset, questions, seq, bs, prefill_tokens, 1234 + i, seq + 64)
        members.append(MemberConfig(name=name, model=name))
    return ConsensusConfig(strategy="majority_vote", decision_mode="logprob", placement=placement, sampling=samp,
                           models=specs, members=members)


def judge_config(n_gpus: int, questions: int, prompt_len: int, new_tokens: int, samp: SamplingConfig,
                 proposer: str = "llama-3-8b", judge: str = "llama-3-70b", n_proposers: int = 4, bs: int = 64,
                 prefill_tokens: int = 8192, max_tp: int = 4) -> ConsensusConfig:
    """Workload D. On eight GPUs: every proposer on a GPU of its own, the judge tensor
    parallel over the other four. On fewer GPUs the judge's TP degree shrinks to what is
    there (1 GPU: unsharded, 141 GB of bf16 weights fit one 288 GB device) and members share
    GPUs as the planner packs them."""
    tp = 1
    while tp * 2 <= min(max_tp, n_gpus):
        tp *= 2
    seq = prompt_len + new_tokens
    judge_seq = prompt_len + 2 * new_tokens + 192          # question + candidate + the judge template
    specs: Dict[str, ModelSpec] = {}
    members = []
    for i in range(n_proposers):
        name = f"proposer{i}"
        specs[name] = _spec(name, proposer, questions, seq + 96, bs, prefill_tokens, 100 + i, seq + 160)
        members.append(MemberConfig(name=name, model=name))
    specs["judge"] = _spec("judge", judge, questions * n_proposers, judge_seq, bs, prefill_tokens, 99, judge_seq + 64,
                           tp=tp)
    members.append(MemberConfig(name="judge", model="judge", role="judge"))
    return ConsensusConfig(strategy="judge_logprob", placement="spread", sampling=samp, models=specs, members=members)


def moe_medoid_config(n_gpus: int, questions: int, prompt_len: int, new_tokens: int, samp: SamplingConfig,
                      samples: int = 4, proposer: str = "mixtral-8x7b", embedder: str = "qwen2.5-0.5b",
                      n_experts: int = 8, bs: int = 64, prefill_tokens: int = 8192) -> ConsensusConfig:
    """Workload E. The MoE proposer is expert parallel over min(n_gpus, experts) GPUs
    (all-to-all dispatch when sharded); the embedding model rides on one of them."""
    ep = 1
    while ep * 2 <= min(n_gpus, n_experts):
        ep *= 2
    seq = prompt_len + new_tokens
    specs = {
        "moe": _spec("moe", proposer, questions * samples, seq + 96, bs, prefill_tokens, 7, seq + 160, ep=ep,
                     ep_mode="a2a" if ep > 1 else "allreduce"),
        "embedder": _spec("embedder", embedder, questions * samples, new_tokens + 64, bs, prefill_tokens, 8,
                          new_tokens + 128),
    }
    members = [MemberConfig(name="moe", model="moe"),
               MemberConfig(name="embedder", model="embedder", role="embedder")]
    return ConsensusConfig(strategy="embedding_medoid", placement="pack", sampling=samp, models=specs, members=members,
                           samples_per_member=samples)
