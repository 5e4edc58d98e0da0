"""Configuration schema.

The reference has no configuration beyond two environment variables
(``GEMINI_API_KEY``, ``RUST_LOG``; /root/reference/src/main.rs:352-357); personas,
the round limit (main.rs:300), the poll period and the prompts are literals. Here
all of that is data. ``ConsensusConfig()`` with no arguments reproduces the
reference: four personas sharing one model, the unanimity loop, five evaluation
rounds, a random proposer.

Files may be TOML, JSON or YAML (chosen by extension).
"""
from __future__ import annotations

import dataclasses
import json
import os
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, List, Optional

from ._native import core

STRATEGIES = (
    "reference_protocol",
    "majority_vote",
    "self_consistency",
    "judge_logprob",
    "embedding_medoid",
)


@dataclass
class SamplingConfig:
    temperature: float = 0.7
    top_p: float = 0.95
    top_k: int = 0
    max_new_tokens: int = 256
    seed: int = 0


@dataclass
class ModelSpec:
    """One set of weights. ``path`` is a directory with HF ``config.json`` +
    ``*.safetensors``; ``preset`` names a built-in architecture for random-init runs."""

    name: str = "default"
    path: Optional[str] = None
    preset: Optional[str] = "tiny-llama"
    tp: int = 1
    ep: int = 1
    ep_mode: str = "allreduce"        # allreduce | a2a (see parallel/ep.py)
    kv_block_size: int = 64
    kv_fraction: float = 0.6          # share of free HBM given to the KV pool (when kv_blocks is 0)
    kv_blocks: int = 0                # explicit KV pool size in blocks; 0 = size it from kv_fraction
    max_batch_seqs: int = 256
    max_prefill_tokens: int = 8192
    mix_decode: bool = True           # decoding chains ride along in the steps that prefill new prompts
    admission: str = "reserve"        # reserve (worst-case blocks promised up front) | optimistic (preempt + recompute)
    max_model_len: int = 0            # longest prompt + answer served; 0 = the model's max_position
    graph_batch_buckets: List[int] = field(default_factory=lambda: [1, 2, 4, 8, 16, 32, 64, 128, 256])
    tokenizer: Optional[str] = None   # HF tokenizer dir; None => byte-level fallback
    chat_format: str = "none"         # none | llama3 | chatml | mistral: turn markers around each prompt
    seed: int = 0                     # random-init seed


@dataclass
class MemberConfig:
    name: str
    domain: str = ""
    tuning: str = ""
    model: str = "default"            # key into ConsensusConfig.models
    role: str = "member"              # member | judge | embedder
    weight: float = 1.0


@dataclass
class ConsensusConfig:
    strategy: str = "reference_protocol"
    max_rounds: int = 5
    proposer_policy: str = "random"   # random | round_robin | fixed
    fixed_proposer: int = 0
    failure_policy: str = "abstain"   # abstain | needs_refinement
    parse_mode: str = "strict"        # strict | tolerant
    decision_mode: str = "text"       # text | logprob (constrained two-way / choice decision on device) |
    #                                   tokens (vote strategies: choice read off the reply's token ids on device)
    forward_reasoning_to_refiner: bool = False
    k_chains: int = 16
    samples_per_member: int = 1       # embedding_medoid: candidates sampled per proposer and question
    micro_batches: int = 2            # judge_logprob / embedding_medoid: parts of a question batch pipelined
    #                                   through the two phases (1 = propose everything, then judge everything)
    min_micro_batch: int = 32         # ... and the fewest questions such a part may hold
    warm_judges: Optional[bool] = None  # judge_logprob: judges prefill the question part of their prompt while the proposers generate; None = when the judges have GPUs to themselves
    choices: str = "ABCD"
    tie_break: str = "lowest_class"
    deadline_s: Optional[float] = None
    seed: int = 0
    placement: str = "pack"           # pack | spread
    sampling: SamplingConfig = field(default_factory=SamplingConfig)
    models: Dict[str, ModelSpec] = field(default_factory=lambda: {"default": ModelSpec()})
    members: List[MemberConfig] = field(default_factory=list)
    prompt_layout: str = "reference"       # reference | persona_last (persona text at the end: shared between the personas of an answer) | persona_first (question and answer at the end: a persona's head is cached across questions)
    prompt_propose: Optional[str] = None   # template file paths
    prompt_evaluate: Optional[str] = None
    prompt_refine: Optional[str] = None
    transcript_path: Optional[str] = None  # JSONL dump of rounds/votes/reasoning

    def __post_init__(self) -> None:
        if not self.members:
            self.members = [
                MemberConfig(name=p.name, domain=p.domain, tuning=p.tuning) for p in core.default_personas()
            ]
        self.validate()

    # ------------------------------------------------------------------
    def validate(self) -> None:
        if self.strategy not in STRATEGIES:
            raise ValueError(f"unknown strategy {self.strategy!r}; expected one of {STRATEGIES}")
        if self.max_rounds < 1:
            raise ValueError("max_rounds must be >= 1")
        if self.proposer_policy not in ("random", "round_robin", "fixed"):
            raise ValueError(f"unknown proposer_policy {self.proposer_policy!r} (random | round_robin | fixed; "
                             "the vote strategies already ask every member)")
        for spec in self.models.values():
            if spec.admission not in ("reserve", "optimistic"):
                raise ValueError(f"model {spec.name!r}: unknown admission {spec.admission!r}")
        if self.failure_policy not in ("abstain", "needs_refinement"):
            raise ValueError(f"unknown failure_policy {self.failure_policy!r}")
        if self.parse_mode not in ("strict", "tolerant"):
            raise ValueError(f"unknown parse_mode {self.parse_mode!r}")
        if self.decision_mode not in ("text", "logprob", "tokens"):
            raise ValueError(f"unknown decision_mode {self.decision_mode!r}")
        if self.prompt_layout not in ("reference", "persona_last", "persona_first"):
            raise ValueError(f"unknown prompt_layout {self.prompt_layout!r}")
        names = [m.name for m in self.members]
        if len(set(names)) != len(names):
            raise ValueError("member names must be unique (they are registry keys)")
        for m in self.members:
            if m.model not in self.models:
                raise ValueError(f"member {m.name!r} refers to unknown model {m.model!r}")
            if m.role not in ("member", "judge", "embedder"):
                raise ValueError(f"member {m.name!r}: unknown role {m.role!r} (member | judge | embedder)")
            if not _is_number(m.weight) or m.weight < 0:
                raise ValueError(f"member {m.name!r}: weight must be a number >= 0, not {m.weight!r}")
        if self.tie_break not in ("lowest_class", "first_voter", "highest_weight"):
            raise ValueError(f"unknown tie_break {self.tie_break!r} (lowest_class | first_voter | highest_weight)")
        if self.placement not in ("pack", "spread"):
            raise ValueError(f"unknown placement {self.placement!r} (pack | spread)")
        if not isinstance(self.choices, str) or len(self.choices) < 2 or len(set(self.choices)) != len(self.choices):
            raise ValueError(f"choices must be a string of at least two distinct letters, not {self.choices!r}")
        for name, lo in (("k_chains", 1), ("samples_per_member", 1), ("micro_batches", 1), ("min_micro_batch", 1),
                         ("fixed_proposer", 0)):
            v = getattr(self, name)
            if not _is_int(v) or v < lo:
                raise ValueError(f"{name} must be an integer >= {lo}, not {v!r}")
        if self.deadline_s is not None and (not _is_number(self.deadline_s) or self.deadline_s <= 0):
            raise ValueError(f"deadline_s must be a positive number of seconds, not {self.deadline_s!r}")
        s = self.sampling
        if not _is_number(s.temperature) or s.temperature < 0:
            raise ValueError(f"sampling.temperature must be a number >= 0 (0 = greedy), not {s.temperature!r}")
        if not _is_number(s.top_p) or not 0 < s.top_p <= 1:
            raise ValueError(f"sampling.top_p must lie in (0, 1], not {s.top_p!r}")
        if not _is_int(s.top_k) or s.top_k < 0:
            raise ValueError(f"sampling.top_k must be an integer >= 0 (0 = off), not {s.top_k!r}")
        if not _is_int(s.max_new_tokens) or s.max_new_tokens < 1:
            raise ValueError(f"sampling.max_new_tokens must be an integer >= 1, not {s.max_new_tokens!r}")
        if not _is_int(s.seed):
            raise ValueError(f"sampling.seed must be an integer, not {s.seed!r}")
        if self.transcript_path:
            parent = Path(self.transcript_path).resolve().parent
            if not parent.is_dir() or not os.access(parent, os.W_OK):
                raise ValueError(f"transcript_path = {self.transcript_path}: {parent} is not a writable directory")
        if self.prompt_propose or self.prompt_evaluate or self.prompt_refine:
            self.templates()          # reads the files and checks their placeholders now, not after the models are loaded
        for spec in self.models.values():
            who = f"model {spec.name!r}"
            for name, lo in (("tp", 1), ("ep", 1), ("kv_blocks", 0), ("max_batch_seqs", 1), ("max_prefill_tokens", 1),
                             ("max_model_len", 0)):
                v = getattr(spec, name)
                if not _is_int(v) or v < lo:
                    raise ValueError(f"{who}: {name} must be an integer >= {lo}, not {v!r}")
            bs = spec.kv_block_size
            if not _is_int(bs) or bs < 8 or bs & (bs - 1):
                raise ValueError(f"{who}: kv_block_size must be a power of two >= 8 (the attention kernels read keys "
                                 f"in runs of 8), not {bs!r}")
            if not _is_number(spec.kv_fraction) or not 0 < spec.kv_fraction < 1:
                raise ValueError(f"{who}: kv_fraction must lie in (0, 1), not {spec.kv_fraction!r}")
            if spec.ep_mode not in ("allreduce", "a2a"):
                raise ValueError(f"{who}: unknown ep_mode {spec.ep_mode!r} (allreduce | a2a)")
            if spec.chat_format not in ("none", "llama3", "chatml", "mistral"):
                raise ValueError(f"{who}: unknown chat_format {spec.chat_format!r} (none | llama3 | chatml | mistral)")
            gb = spec.graph_batch_buckets
            if not isinstance(gb, (list, tuple)) or not gb or not all(_is_int(b) and b >= 1 for b in gb):
                raise ValueError(f"{who}: graph_batch_buckets must be a list of positive integers, not {gb!r}")
            if not spec.path and not spec.preset:
                raise ValueError(f"{who}: give a checkpoint `path` or a `preset`")

    def session_config(self) -> "core.SessionConfig":
        sc = core.SessionConfig()
        sc.max_rounds = self.max_rounds
        sc.proposer_policy = {
            "random": core.ProposerPolicy.Random,
            "round_robin": core.ProposerPolicy.RoundRobin,
            "fixed": core.ProposerPolicy.Fixed,
        }[self.proposer_policy]
        sc.fixed_proposer = self.fixed_proposer
        sc.failure_policy = (
            core.FailurePolicy.Abstain if self.failure_policy == "abstain" else core.FailurePolicy.NeedsRefinement
        )
        sc.forward_reasoning_to_refiner = self.forward_reasoning_to_refiner
        sc.seed = self.seed
        return sc

    def parse_mode_enum(self) -> "core.ParseMode":
        return core.ParseMode.Strict if self.parse_mode == "strict" else core.ParseMode.Tolerant

    def tie_break_enum(self) -> "core.TieBreak":
        return {
            "lowest_class": core.TieBreak.LowestClass,
            "first_voter": core.TieBreak.FirstVoter,
            "highest_weight": core.TieBreak.HighestWeight,
        }[self.tie_break]

    def templates(self) -> "core.PromptTemplates":
        t = {"persona_last": core.PromptTemplates.persona_last,
             "persona_first": core.PromptTemplates.persona_first}.get(self.prompt_layout, core.PromptTemplates)()
        needs = {"propose": ("{question}",), "evaluate": ("{question}", "{answer}"), "refine": ("{question}", "{answer}")}
        for attr, path in (("propose", self.prompt_propose), ("evaluate", self.prompt_evaluate),
                           ("refine", self.prompt_refine)):
            if path:
                text = Path(path).read_text()
                missing = [ph for ph in needs[attr] if ph not in text]
                if missing:
                    # (a template without its placeholders renders, and every member then answers a
                    # prompt that does not contain the question)
                    raise ValueError(f"prompt_{attr} = {path}: the template lacks {', '.join(missing)}")
                setattr(t, attr, text)
        return t

    def personas(self) -> List["core.Persona"]:
        return [core.Persona(m.name, m.domain, m.tuning) for m in self.members]

    def to_dict(self) -> Dict[str, Any]:
        return dataclasses.asdict(self)


def _is_int(v) -> bool:
    return isinstance(v, int) and not isinstance(v, bool)


def _is_number(v) -> bool:
    return isinstance(v, (int, float)) and not isinstance(v, bool)


def _build(cls, data: Dict[str, Any]):
    names = {f.name for f in dataclasses.fields(cls)}
    unknown = set(data) - names
    if unknown:
        raise ValueError(f"unknown {cls.__name__} keys: {sorted(unknown)}")
    return cls(**data)


def config_from_dict(d: Dict[str, Any]) -> ConsensusConfig:
    d = dict(d)
    if "sampling" in d:
        d["sampling"] = _build(SamplingConfig, d["sampling"])
    if "models" in d:
        models = d["models"]
        if isinstance(models, list):
            models = {m["name"]: m for m in models}
        d["models"] = {k: _build(ModelSpec, {"name": k, **v}) for k, v in models.items()}
    if "members" in d:
        d["members"] = [_build(MemberConfig, m) for m in d["members"]]
    return _build(ConsensusConfig, d)


def load_config_or_exit(path: Optional[str], prog: str) -> ConsensusConfig:
    """``load_config`` for a command line: a missing or malformed file, or a value the checks
    refuse, ends the command with one line on stderr and status 2 (no traceback, nothing loaded)."""
    import sys

    try:
        return load_config(path)
    except (OSError, ValueError, KeyError, TypeError) as e:      # TOML / JSON / YAML errors are ValueErrors
        print(f"{prog}: error: configuration {path or os.environ.get('CONSENSUS_CONFIG') or '(defaults)'}: {e}",
              file=sys.stderr, flush=True)
        raise SystemExit(2)


def load_config(path: Optional[str] = None) -> ConsensusConfig:
    """Loads a config file; ``None`` consults ``$CONSENSUS_CONFIG`` and falls back to
    the reference-equivalent defaults."""
    path = path or os.environ.get("CONSENSUS_CONFIG")
    if not path:
        return ConsensusConfig()
    p = Path(path)
    text = p.read_text()
    suffix = p.suffix.lower()
    if suffix == ".toml":
        try:
            import tomllib as toml_reader          # standard library from Python 3.11 on
        except ModuleNotFoundError:
            import tomli as toml_reader            # the same parser as a package (pyproject installs it below 3.11)

        data = toml_reader.loads(text)
    elif suffix in (".yaml", ".yml"):
        import yaml

        data = yaml.safe_load(text)
    else:
        data = json.loads(text)
    return config_from_dict(data or {})
