"""Decoder-only transformer (Llama-3 / Mistral / Qwen2.5 dense, Mixtral MoE) over the
gfx950 op layer. One class covers the families: QKV bias and tied embeddings (Qwen),
sliding-window attention (Mistral v0.1) and block-sparse MoE MLPs (Mixtral) are config
switches. The same forward runs prefill chunks and decode steps; ``AttnMeta`` carries
the paged-KV metadata for the step.

Weights live in device layout (ops.LinearWeight): QKV fused into one matrix, gate/up
fused and tile-interleaved so SwiGLU runs in the GEMM epilogue, everything pre-shuffled
into MFMA fragment order. With tensor parallelism each rank holds Hq/tp query heads,
Hkv/tp KV heads, d_ff/tp MLP columns and V/tp vocabulary rows; the two per-layer
all-reduces and the logits all-gather go through ``parallel.comm.Comm`` (RCCL).
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from .. import ops
from ..ops import reference as ref
from ..parallel.ep import expert_range, moe_ffn_ep
from .config import ModelConfig


@dataclass
class AttnMeta:
    """Paged-attention metadata for one step (device tensors, int32)."""

    is_prefill: bool
    block_tables: torch.Tensor                 # [B, max_blocks]
    ctx_lens: torch.Tensor                     # [B]
    cu_q: Optional[torch.Tensor] = None        # [B+1]  (prefill)
    max_q_len: int = 0                         # (prefill)
    nsplit: int = 1                            # KV splits (decode)
    groups: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None  # shared-prefix groups (decode)
    kv_begin: Optional[torch.Tensor] = None    # [B] first private key of each row (decode, with groups)
    max_group_rows: int = 0
    prefix_splits: int = 1
    # Mixed step: the first ``prefill_tokens`` rows belong to the prefill sequences described
    # above; the remaining rows are one-token decode sequences described by ``riders``.
    riders: Optional["AttnMeta"] = None
    prefill_tokens: int = 0
    # Decode steps launched ahead of the previous step's results: (tokens, rows), see ops.embed.
    token_feed: Optional[Tuple[torch.Tensor, torch.Tensor]] = None


@dataclass
class LayerWeights:
    ln1: torch.Tensor
    ln2: torch.Tensor
    qkv: ops.LinearWeight
    o: ops.LinearWeight
    gate_up: Optional[ops.LinearWeight] = None
    down: Optional[ops.LinearWeight] = None
    router: Optional[torch.Tensor] = None            # [E, d] (MoE)
    experts_gate_up: Optional[ops.ExpertWeights] = None
    experts_down: Optional[ops.ExpertWeights] = None


class KVCache:
    """One flat bf16 pool: planes (layer, K|Vt) x blocks. K plane blocks are [Hkv, BS, Dh],
    Vt plane blocks are [Hkv, Dh, BS] (see attention.hip)."""

    def __init__(self, cfg: ModelConfig, num_blocks: int, block_size: int, device, kv_heads: Optional[int] = None,
                 alloc=None):
        """``alloc(numel) -> bf16 tensor`` supplies the pool's memory (an engine on a GPU hands in
        its runtime's KV arena); the default is a plain tensor. Either way the pool starts zeroed."""
        self.cfg, self.num_blocks, self.block_size = cfg, num_blocks, block_size
        self.kv_heads = kv_heads or cfg.num_kv_heads
        self.block_elems = self.kv_heads * block_size * cfg.head_dim
        self.n_planes = 2 * cfg.num_layers
        self.plane_elems = num_blocks * self.block_elems
        n = self.n_planes * self.plane_elems
        if alloc is None:
            self.pool = torch.zeros(n, dtype=torch.bfloat16, device=device)
        else:
            self.pool = alloc(n)
            self.pool.zero_()

    def k(self, layer: int) -> torch.Tensor:
        p = self.pool[(2 * layer) * self.plane_elems:(2 * layer + 1) * self.plane_elems]
        return p.view(self.num_blocks, self.kv_heads, self.block_size, self.cfg.head_dim)

    def vt(self, layer: int) -> torch.Tensor:
        p = self.pool[(2 * layer + 1) * self.plane_elems:(2 * layer + 2) * self.plane_elems]
        return p.view(self.num_blocks, self.kv_heads, self.cfg.head_dim, self.block_size)

    def copy_blocks(self, src: torch.Tensor, dst: torch.Tensor) -> None:
        ops.copy_blocks(self.pool, src, dst, self.block_elems, self.n_planes, self.plane_elems)

    @staticmethod
    def bytes_per_block(cfg: ModelConfig, block_size: int, kv_heads: Optional[int] = None) -> int:
        return 2 * cfg.num_layers * (kv_heads or cfg.num_kv_heads) * block_size * cfg.head_dim * 2


class TransformerModel:
    def __init__(self, cfg: ModelConfig, device, tp_rank: int = 0, tp_size: int = 1, comm=None,
                 ep_size: int = 1, ep_mode: str = "allreduce") -> None:
        """``comm`` is the communicator of this model's parallel group. Tensor parallelism
        (``tp_size`` = group size) shards heads, MLP columns and vocabulary; expert
        parallelism (``ep_size`` = group size, ``tp_size`` 1) keeps the dense parts replicated
        and shards Mixtral's experts (see parallel/ep.py)."""
        cfg.validate()
        if cfg.num_kv_heads % tp_size or cfg.intermediate_size % (16 * tp_size):
            raise ValueError(f"{cfg.name}: tp={tp_size} does not divide kv heads / d_ff")
        if ep_size > 1 and (tp_size > 1 or not cfg.is_moe):
            raise ValueError(f"{cfg.name}: expert parallelism needs an MoE model and tp=1")
        self.cfg, self.device = cfg, torch.device(device)
        self.tp_rank, self.tp_size, self.comm = tp_rank, tp_size, comm
        self.ep_size, self.ep_mode = ep_size, ep_mode
        self.ep_rank = comm.rank if (ep_size > 1 and comm is not None) else 0
        self.expert_lo, self.expert_hi = expert_range(cfg.num_experts, self.ep_rank, ep_size) if cfg.is_moe else (0, 0)
        self.hq = cfg.num_heads // tp_size
        self.hkv = cfg.num_kv_heads // tp_size
        self.dff = cfg.intermediate_size // tp_size
        self.embed: Optional[torch.Tensor] = None
        self.final_norm: Optional[torch.Tensor] = None
        self.lm_head: Optional[ops.LinearWeight] = None
        self.layers: List[LayerWeights] = []
        cos, sin = ref.rope_tables(cfg.head_dim, cfg.max_position, cfg.rope_theta, scaling=cfg.rope_scaling_dict)
        self.cos_t, self.sin_t = cos.to(self.device), sin.to(self.device)
        self.scale = cfg.head_dim ** -0.5

    # ------------------------------------------------------------------ construction
    def set_weights_from_hf(self, get) -> None:
        """``get(name) -> torch.Tensor`` (any dtype/device) with HF parameter names. Slices
        this rank's tensor-parallel shard and converts to device layout."""
        c, dev = self.cfg, self.device
        r, tp = self.tp_rank, self.tp_size
        Dh = c.head_dim
        bf = lambda t: t.to(torch.bfloat16)  # noqa: E731
        source = get

        def get(name: str):
            # A checkpoint whose tensors do not have the shapes its config.json implies (another
            # model's file, a truncated conversion) must fail here, by name: sliced and packed it
            # would load, and a GEMM would then read a weight of the wrong depth.
            t = source(name)
            want = tuple(hf_param_shape(c, name))
            if tuple(t.shape) != want:
                raise ValueError(f"{c.name}: tensor {name!r} has shape {tuple(t.shape)}, the configuration implies {want}")
            return t
        self.embed = bf(get("model.embed_tokens.weight")).to(dev).contiguous()
        self.final_norm = bf(get("model.norm.weight")).to(dev).contiguous()
        lm = get("model.embed_tokens.weight") if c.tie_embeddings else get("lm_head.weight")
        vshard = (c.vocab_size + tp - 1) // tp
        self.vocab_lo, self.vocab_hi = r * vshard, min(c.vocab_size, (r + 1) * vshard)
        self.lm_head = ops.LinearWeight.from_dense(bf(lm[self.vocab_lo:self.vocab_hi]), None, dev)
        self.layers = []
        for i in range(c.num_layers):
            p = f"model.layers.{i}."
            q = get(p + "self_attn.q_proj.weight")[r * self.hq * Dh:(r + 1) * self.hq * Dh]
            k = get(p + "self_attn.k_proj.weight")[r * self.hkv * Dh:(r + 1) * self.hkv * Dh]
            v = get(p + "self_attn.v_proj.weight")[r * self.hkv * Dh:(r + 1) * self.hkv * Dh]
            bias = None
            if c.qkv_bias:
                bias = torch.cat([
                    get(p + "self_attn.q_proj.bias")[r * self.hq * Dh:(r + 1) * self.hq * Dh],
                    get(p + "self_attn.k_proj.bias")[r * self.hkv * Dh:(r + 1) * self.hkv * Dh],
                    get(p + "self_attn.v_proj.bias")[r * self.hkv * Dh:(r + 1) * self.hkv * Dh]])
            qkv = ops.LinearWeight.from_dense(bf(torch.cat([q, k, v], 0)), bias, dev)
            o = get(p + "self_attn.o_proj.weight")[:, r * self.hq * Dh:(r + 1) * self.hq * Dh]
            lw = LayerWeights(
                ln1=bf(get(p + "input_layernorm.weight")).to(dev).contiguous(),
                ln2=bf(get(p + "post_attention_layernorm.weight")).to(dev).contiguous(),
                qkv=qkv,
                o=ops.LinearWeight.from_dense(bf(o).contiguous(), None, dev),
            )
            sl = slice(r * self.dff, (r + 1) * self.dff)
            if c.is_moe:
                m = p + "block_sparse_moe."
                lw.router = bf(get(m + "gate.weight")).to(dev).contiguous()
                gu = torch.stack([ops.interleave_gate_up(bf(get(f"{m}experts.{e}.w1.weight")[sl]),
                                                         bf(get(f"{m}experts.{e}.w3.weight")[sl]))
                                  for e in range(self.expert_lo, self.expert_hi)])
                dn = torch.stack([bf(get(f"{m}experts.{e}.w2.weight")[:, sl]).contiguous()
                                  for e in range(self.expert_lo, self.expert_hi)])
                lw.experts_gate_up = ops.ExpertWeights.from_dense(gu, dev, swiglu=True)
                lw.experts_down = ops.ExpertWeights.from_dense(dn, dev)
            else:
                gu = ops.interleave_gate_up(bf(get(p + "mlp.gate_proj.weight")[sl]),
                                            bf(get(p + "mlp.up_proj.weight")[sl]))
                lw.gate_up = ops.LinearWeight.from_dense(gu, None, dev, swiglu=True)
                lw.down = ops.LinearWeight.from_dense(bf(get(p + "mlp.down_proj.weight")[:, sl]).contiguous(), None, dev)
            self.layers.append(lw)

    def decode_capturable(self, rows: int) -> bool:
        """Whether a decode step of ``rows`` sequences can be recorded into a hipGraph: every
        model family's step is made of kernel launches with host-known shapes only (the
        expert-parallel exchange carries its row counts on the device, parallel/ep.py)."""
        return True

    @classmethod
    def random_init(cls, cfg: ModelConfig, device, seed: int = 0, tp_rank: int = 0, tp_size: int = 1, comm=None,
                    std: float = 0.02, ep_size: int = 1, ep_mode: str = "allreduce") -> "TransformerModel":
        """Random weights of the configured architecture generated directly on ``device``
        (no checkpoint files; used by benchmarks and smoke tests). Every rank draws the
        full tensors from the same seed and keeps its shard, so TP ranks are consistent."""
        m = cls(cfg, device, tp_rank, tp_size, comm, ep_size, ep_mode)
        dev = m.device
        gen = torch.Generator(device=dev)
        cache: Dict[str, torch.Tensor] = {}

        def get(name: str) -> torch.Tensor:
            shape = hf_param_shape(cfg, name)
            if name.endswith("norm.weight") or name.endswith("layernorm.weight"):
                return torch.ones(shape, dtype=torch.bfloat16, device=dev)
            if name == "model.embed_tokens.weight" and name in cache:
                return cache[name]
            # one stream per tensor name: a rank that skips tensors (non-resident experts)
            # still draws the same values for the ones it keeps
            gen.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFFFFFFFFFF)
            t = (torch.randn(shape, generator=gen, device=dev, dtype=torch.float32) * std).to(torch.bfloat16)
            if name == "model.embed_tokens.weight":
                cache[name] = t
            return t

        m.set_weights_from_hf(get)
        return m

    def weight_bytes(self) -> int:
        n = self.embed.numel() * 2 + self.lm_head.nbytes()
        for l in self.layers:
            n += l.qkv.nbytes() + l.o.nbytes()
            if l.gate_up is not None:
                n += l.gate_up.nbytes() + l.down.nbytes()
            if l.experts_gate_up is not None:
                n += l.experts_gate_up.nbytes() + l.experts_down.nbytes() + l.router.numel() * 2
        return n

    # ------------------------------------------------------------------ forward
    def _reduce_norm(self, h: torch.Tensor, delta, w: torch.Tensor) -> torch.Tensor:
        """RMSNorm(h += delta) * w, h updated in place. Under tensor parallelism ``delta`` is this
        rank's PARTIAL product (row-parallel O / down projection): it is summed over the group
        first. A communicator with ``allreduce_rmsnorm`` (parallel/p2p.py) does sum, residual
        add and norm in one launch, split-K slabs included; otherwise all-reduce, then the
        fused residual-norm kernel."""
        eps = self.cfg.rms_eps
        if self.tp_size > 1 and delta is not None:
            fused = getattr(self.comm, "allreduce_rmsnorm", None)
            if fused is not None and h.is_cuda:
                return fused(h, delta, w, eps)
            if isinstance(delta, ops.PartialSum):
                delta = delta.reduce()
            delta = self.comm.all_reduce(delta)
        return ops.rmsnorm(h, w, eps, delta)

    def forward(self, token_ids: torch.Tensor, positions: torch.Tensor, slot_mapping: torch.Tensor,
                meta: AttnMeta, kv: KVCache) -> torch.Tensor:
        """Returns the final-norm hidden states [T, d] for the step's tokens."""
        c = self.cfg
        h = ops.embed(token_ids, self.embed, meta.token_feed)
        delta = None
        for i, L in enumerate(self.layers):
            x = self._reduce_norm(h, delta, L.ln1)
            qkv = ops.linear(x, L.qkv, defer_reduce=True)
            kc, vc = kv.k(i), kv.vt(i)
            qkv = ops.rope_kv_write(qkv, positions, slot_mapping, self.cos_t, self.sin_t, kc, vc, self.hq, self.hkv,
                                    c.head_dim)
            if meta.is_prefill and meta.riders is not None:
                # Prompt chunks through the prefill kernel, the decoding rows behind them through
                # the decode kernel (which reads each KV head once for all of its query heads).
                Tp, r = meta.prefill_tokens, meta.riders
                a = torch.empty(qkv.shape[0], self.hq * c.head_dim, dtype=qkv.dtype, device=qkv.device)
                ops.attn_prefill(qkv[:Tp], kc, vc, meta.block_tables, meta.cu_q, meta.ctx_lens, meta.max_q_len,
                                 self.hq, self.hkv, c.head_dim, self.scale, c.sliding_window, out=a[:Tp])
                ops.attn_decode(qkv[Tp:], kc, vc, r.block_tables, r.ctx_lens, self.hq, self.hkv, c.head_dim,
                                self.scale, c.sliding_window, r.nsplit, out=a[Tp:])
            elif meta.is_prefill:
                a = ops.attn_prefill(qkv, kc, vc, meta.block_tables, meta.cu_q, meta.ctx_lens, meta.max_q_len,
                                     self.hq, self.hkv, c.head_dim, self.scale, c.sliding_window)
            else:
                a = ops.attn_decode(qkv, kc, vc, meta.block_tables, meta.ctx_lens, self.hq, self.hkv, c.head_dim,
                                    self.scale, c.sliding_window, meta.nsplit, meta.groups, meta.kv_begin,
                                    meta.max_group_rows, meta.prefix_splits)
            x2 = self._reduce_norm(h, ops.linear(a, L.o, defer_reduce=True), L.ln2)
            if c.is_moe and self.ep_size > 1:
                delta = moe_ffn_ep(x2, L.router, L.experts_gate_up, L.experts_down, c.experts_per_tok, self.comm,
                                   self.expert_lo, self.ep_mode)
            elif c.is_moe:
                delta = ops.moe_ffn(x2, L.router, L.experts_gate_up, L.experts_down, c.experts_per_tok)
            else:
                delta = ops.linear(ops.linear(x2, L.gate_up), L.down, defer_reduce=True)
        return self._reduce_norm(h, delta, self.final_norm)

    def logits(self, hidden_rows: torch.Tensor) -> torch.Tensor:
        """fp32 logits [R, V] for the given hidden rows (vocab-parallel + all-gather under TP)."""
        lg = ops.linear(hidden_rows, self.lm_head, out_f32=True)
        if self.tp_size > 1:
            vshard = (self.cfg.vocab_size + self.tp_size - 1) // self.tp_size
            if lg.shape[1] < vshard:  # ragged last shard: equal-sized parts for the all-gather
                lg = ops.pad_cols(lg, vshard, float("-inf"))
            lg = self.comm.all_gather_cols(lg, self.cfg.vocab_size)
        return lg

    def target_logprobs(self, hidden_rows: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
        """log softmax of the given hidden rows at the listed token ids only: fp32 [R, C] for
        ``ids`` [R, C]. Under tensor parallelism each rank reduces its own vocabulary shard to
        C target values plus one log-sum-exp per row and the ranks exchange those C + 1 numbers
        per row, instead of all-gathering the [R, V] logits (scoring a candidate list needs
        nothing else: the judge of the reference reads one verdict per answer,
        /root/reference/src/main.rs:259-291)."""
        lg = ops.linear(hidden_rows, self.lm_head, out_f32=True)
        if self.tp_size == 1:
            return ops.logprob_gather(lg, ids)
        vshard = (self.cfg.vocab_size + self.tp_size - 1) // self.tp_size
        packed = ops.logit_targets_shard(lg, ids, self.tp_rank * vshard)          # [R, Cp]: targets, lse, padding
        parts = self.comm.all_gather_rows(packed).view(self.tp_size, *packed.shape)
        return ops.logprob_merge(parts, ids.shape[-1])


def hf_param_shape(cfg: ModelConfig, name: str) -> Tuple[int, ...]:
    d, Dh = cfg.hidden_size, cfg.head_dim
    leaf = name.split(".")[-2] if name.endswith((".weight", ".bias")) else name
    is_bias = name.endswith(".bias")
    table = {
        "embed_tokens": (cfg.vocab_size, d), "lm_head": (cfg.vocab_size, d), "norm": (d,),
        "input_layernorm": (d,), "post_attention_layernorm": (d,),
        "q_proj": (cfg.num_heads * Dh, d), "k_proj": (cfg.num_kv_heads * Dh, d), "v_proj": (cfg.num_kv_heads * Dh, d),
        "o_proj": (d, cfg.num_heads * Dh),
        "gate_proj": (cfg.intermediate_size, d), "up_proj": (cfg.intermediate_size, d),
        "down_proj": (d, cfg.intermediate_size),
        "gate": (cfg.num_experts, d), "w1": (cfg.intermediate_size, d), "w3": (cfg.intermediate_size, d),
        "w2": (d, cfg.intermediate_size),
    }
    shape = table[leaf]
    return (shape[0],) if is_bias else shape


def hf_param_names(cfg: ModelConfig) -> List[str]:
    names = ["model.embed_tokens.weight", "model.norm.weight"]
    if not cfg.tie_embeddings:
        names.append("lm_head.weight")
    for i in range(cfg.num_layers):
        p = f"model.layers.{i}."
        names += [p + "input_layernorm.weight", p + "post_attention_layernorm.weight",
                  p + "self_attn.q_proj.weight", p + "self_attn.k_proj.weight", p + "self_attn.v_proj.weight",
                  p + "self_attn.o_proj.weight"]
        if cfg.qkv_bias:
            names += [p + f"self_attn.{x}_proj.bias" for x in "qkv"]
        if cfg.is_moe:
            names.append(p + "block_sparse_moe.gate.weight")
            for e in range(cfg.num_experts):
                names += [p + f"block_sparse_moe.experts.{e}.{w}.weight" for w in ("w1", "w2", "w3")]
        else:
            names += [p + f"mlp.{x}_proj.weight" for x in ("gate", "up", "down")]
    return names
