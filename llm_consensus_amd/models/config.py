"""Model architecture descriptions and built-in presets.

Presets carry the public Hugging Face shapes of the models the consensus workloads
name (Llama-3-8B/70B, Qwen2.5-0.5B/7B, Mistral-7B, Mixtral-8x7B) plus tiny variants
for tests. ``from_hf_config`` reads a checkpoint's ``config.json``. The reference has
no model description at all: its model is whatever the remote API serves
(/root/reference/src/main.rs:82-86).
"""
from __future__ import annotations

import json
from dataclasses import dataclass, replace
from pathlib import Path
from typing import Dict, Optional, Tuple

SUPPORTED_MODEL_TYPES = ("llama", "qwen2", "mistral", "mixtral")


@dataclass(frozen=True)
class ModelConfig:
    name: str
    vocab_size: int
    hidden_size: int
    num_layers: int
    num_heads: int
    num_kv_heads: int
    head_dim: int
    intermediate_size: int
    rms_eps: float = 1e-5
    rope_theta: float = 10000.0
    qkv_bias: bool = False
    tie_embeddings: bool = False
    sliding_window: int = 0          # 0 = full attention
    max_position: int = 8192
    num_experts: int = 0             # 0 = dense MLP
    experts_per_tok: int = 0
    model_type: str = "llama"
    # checkpoint's rope_scaling as sorted (key, value) pairs (hashable); see ops.reference.rope_inv_freq
    rope_scaling: Optional[Tuple[Tuple[str, object], ...]] = None

    @property
    def rope_scaling_dict(self) -> Optional[dict]:
        return dict(self.rope_scaling) if self.rope_scaling else None

    @property
    def is_moe(self) -> bool:
        return self.num_experts > 0

    @property
    def q_size(self) -> int:
        return self.num_heads * self.head_dim

    @property
    def kv_size(self) -> int:
        return self.num_kv_heads * self.head_dim

    def param_count(self) -> int:
        d, L = self.hidden_size, self.num_layers
        attn = d * (self.q_size + 2 * self.kv_size) + self.q_size * d
        if self.qkv_bias:
            attn += self.q_size + 2 * self.kv_size
        mlp = 3 * d * self.intermediate_size
        if self.is_moe:
            mlp = mlp * self.num_experts + d * self.num_experts
        emb = self.vocab_size * d * (1 if self.tie_embeddings else 2)
        return L * (attn + mlp + 2 * d) + emb + d

    def weight_bytes(self, bytes_per_param: int = 2) -> int:
        return self.param_count() * bytes_per_param

    def kv_bytes_per_token(self, bytes_per_elem: int = 2) -> int:
        return 2 * self.num_layers * self.kv_size * bytes_per_elem

    def validate(self) -> None:
        if self.head_dim not in (64, 128):
            raise ValueError(f"{self.name}: head_dim {self.head_dim} unsupported (attention kernels take 64 or 128)")
        if self.num_heads % self.num_kv_heads:
            raise ValueError(f"{self.name}: num_heads must be a multiple of num_kv_heads")
        if self.hidden_size % 64 or self.intermediate_size % 64:
            raise ValueError(f"{self.name}: hidden and intermediate sizes must be multiples of 64")


PRESETS: Dict[str, ModelConfig] = {
    "qwen2.5-0.5b": ModelConfig("qwen2.5-0.5b", 151936, 896, 24, 14, 2, 64, 4864, 1e-6, 1e6, True, True,
                                0, 32768, model_type="qwen2"),
    "llama-3-8b": ModelConfig("llama-3-8b", 128256, 4096, 32, 32, 8, 128, 14336, 1e-5, 5e5, max_position=8192),
    "qwen2.5-7b": ModelConfig("qwen2.5-7b", 152064, 3584, 28, 28, 4, 128, 18944, 1e-6, 1e6, True, False,
                              0, 32768, model_type="qwen2"),
    "mistral-7b": ModelConfig("mistral-7b", 32000, 4096, 32, 32, 8, 128, 14336, 1e-5, 1e4, False, False,
                              4096, 32768, model_type="mistral"),
    "mistral-7b-v0.3": ModelConfig("mistral-7b-v0.3", 32768, 4096, 32, 32, 8, 128, 14336, 1e-5, 1e6, False, False,
                                   0, 32768, model_type="mistral"),
    "llama-3-70b": ModelConfig("llama-3-70b", 128256, 8192, 80, 64, 8, 128, 28672, 1e-5, 5e5, max_position=8192),
    "mixtral-8x7b": ModelConfig("mixtral-8x7b", 32000, 4096, 32, 32, 8, 128, 14336, 1e-5, 1e6, False, False,
                                0, 32768, 8, 2, "mixtral"),
    # ---- small shapes for tests / CPU plumbing ----
    "tiny-llama": ModelConfig("tiny-llama", 512, 128, 2, 4, 2, 64, 256, 1e-5, 1e4, max_position=1024),
    "tiny-qwen": ModelConfig("tiny-qwen", 512, 128, 2, 4, 2, 64, 256, 1e-6, 1e6, True, True, 0, 1024,
                             model_type="qwen2"),
    "tiny-mistral": ModelConfig("tiny-mistral", 512, 128, 2, 4, 2, 64, 256, 1e-5, 1e4, False, False, 48, 1024,
                                model_type="mistral"),
    "tiny-mixtral": ModelConfig("tiny-mixtral", 512, 128, 2, 4, 2, 64, 256, 1e-5, 1e4, False, False, 0, 1024,
                                4, 2, "mixtral"),
    # four kv heads: the smallest shape a TP=4 group can shard (the judge of workload D on 4-8 ranks)
    "tiny-judge": ModelConfig("tiny-judge", 512, 512, 2, 8, 4, 64, 512, 1e-5, 1e4, max_position=1024),
    "small-llama": ModelConfig("small-llama", 4096, 512, 4, 8, 2, 128, 1536, 1e-5, 5e5, max_position=4096),
}


def get_preset(name: str, **overrides) -> ModelConfig:
    key = name.lower()
    if key not in PRESETS:
        raise KeyError(f"unknown model preset {name!r}; known: {sorted(PRESETS)}")
    cfg = PRESETS[key]
    return replace(cfg, **overrides) if overrides else cfg


def from_hf_config(path: str) -> ModelConfig:
    p = Path(path)
    if p.is_dir():
        p = p / "config.json"
    c = json.loads(p.read_text())
    mt = c.get("model_type", "llama")
    if mt not in SUPPORTED_MODEL_TYPES:
        raise ValueError(f"{p}: model_type {mt!r} is not supported (supported: {', '.join(SUPPORTED_MODEL_TYPES)})")
    scaling = c.get("rope_scaling") or None
    if scaling is None and isinstance(c.get("rope_parameters"), dict):
        rp = c["rope_parameters"]
        if rp.get("rope_type", "default") not in (None, "default"):
            scaling = {k: v for k, v in rp.items() if k != "rope_theta"}
    if scaling is not None:
        from ..ops.reference import rope_inv_freq

        rope_inv_freq(2, 10000.0, dict(scaling))      # raises on a rule the engine does not implement
        scaling = tuple(sorted((str(k), v) for k, v in scaling.items() if not isinstance(v, (list, dict))))
    heads = c["num_attention_heads"]
    head_dim = c.get("head_dim") or c["hidden_size"] // heads
    sw = c.get("sliding_window") or 0
    if mt == "qwen2" and not c.get("use_sliding_window", False):
        sw = 0
    cfg = ModelConfig(
        name=c.get("_name_or_path") or p.parent.name,
        vocab_size=c["vocab_size"],
        hidden_size=c["hidden_size"],
        num_layers=c["num_hidden_layers"],
        num_heads=heads,
        num_kv_heads=c.get("num_key_value_heads", heads),
        head_dim=head_dim,
        intermediate_size=c["intermediate_size"],
        rms_eps=c.get("rms_norm_eps", 1e-5),
        rope_theta=float(c.get("rope_theta") or (c.get("rope_parameters") or {}).get("rope_theta", 10000.0)),
        qkv_bias=True if mt == "qwen2" else bool(c.get("attention_bias", False)),
        tie_embeddings=bool(c.get("tie_word_embeddings", False)),
        sliding_window=int(sw),
        max_position=int(c.get("max_position_embeddings", 8192)),
        num_experts=int(c.get("num_local_experts", 0)),
        experts_per_tok=int(c.get("num_experts_per_tok", 0)),
        model_type=mt,
        rope_scaling=scaling,
    )
    cfg.validate()
    return cfg


def to_hf_config(cfg: ModelConfig) -> dict:
    arch = {"llama": "LlamaForCausalLM", "qwen2": "Qwen2ForCausalLM", "mistral": "MistralForCausalLM",
            "mixtral": "MixtralForCausalLM"}[cfg.model_type]
    d = {
        "architectures": [arch],
        "model_type": cfg.model_type,
        "vocab_size": cfg.vocab_size,
        "hidden_size": cfg.hidden_size,
        "num_hidden_layers": cfg.num_layers,
        "num_attention_heads": cfg.num_heads,
        "num_key_value_heads": cfg.num_kv_heads,
        "head_dim": cfg.head_dim,
        "intermediate_size": cfg.intermediate_size,
        "rms_norm_eps": cfg.rms_eps,
        "rope_theta": cfg.rope_theta,
        "tie_word_embeddings": cfg.tie_embeddings,
        "max_position_embeddings": cfg.max_position,
        "hidden_act": "silu",
        "torch_dtype": "bfloat16",
        "attention_bias": cfg.qkv_bias if cfg.model_type == "llama" else False,
        "sliding_window": cfg.sliding_window or None,
        "bos_token_id": 256 if cfg.vocab_size > 258 else 0,
        "eos_token_id": 257 if cfg.vocab_size > 258 else 1,
    }
    if cfg.is_moe:
        d["num_local_experts"] = cfg.num_experts
        d["num_experts_per_tok"] = cfg.experts_per_tok
    if cfg.model_type == "qwen2":
        d["use_sliding_window"] = False
    if cfg.rope_scaling:
        d["rope_scaling"] = dict(cfg.rope_scaling)
    return d
