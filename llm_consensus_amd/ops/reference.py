"""Plain PyTorch (fp32 math) implementations of every op.

Two uses: (1) the numerics oracle each HIP kernel is tested against, (2) the CPU
plumbing backend, so the engine, the protocol drivers and the HTTP endpoint run end to
end on hosts without a GPU. They operate on the same data layouts as the kernels (paged
K / transposed-V cache, rotate-half RoPE tables) except that CPU weights stay row-major.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

MASK64 = (1 << 64) - 1


# ------------------------------------------------------------------ RNG (mirrors common.h)
def _mix64(z: int) -> int:
    z &= MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def rng_uniform(seed: int, step: int, counter: int = 0) -> float:
    z = _mix64((seed * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & MASK64)
    z = _mix64(z ^ (((step & 0xFFFFFFFF) << 32) | (counter & 0xFFFFFFFF)))
    return ((z >> 40) + 0.5) * (1.0 / 16777216.0)


# ------------------------------------------------------------------ elementwise / norm
def embed(ids: torch.Tensor, E: torch.Tensor) -> torch.Tensor:
    return E[ids.long().clamp(0, E.shape[0] - 1)]


def rmsnorm(h: torch.Tensor, w: torch.Tensor, eps: float, delta: Optional[torch.Tensor] = None) -> torch.Tensor:
    if delta is not None:
        h.copy_((h.float() + delta.float()).to(h.dtype))
    x = h.float()
    y = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w.float()
    return y.to(h.dtype)


def swiglu(gu: torch.Tensor) -> torch.Tensor:
    dff = gu.shape[-1] // 2
    g, u = gu[..., :dff].float(), gu[..., dff:].float()
    return (torch.nn.functional.silu(g) * u).to(gu.dtype)


def linear(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None, out_dtype=None) -> torch.Tensor:
    y = x.float() @ W.float().t()
    if bias is not None:
        y = y + bias.float()
    return y.to(out_dtype or x.dtype)


# ------------------------------------------------------------------ RoPE + KV write
def rope_inv_freq(head_dim: int, theta: float, scaling: Optional[dict] = None) -> torch.Tensor:
    """Rotary inverse frequencies (float64, [head_dim/2]). ``scaling`` is a checkpoint's
    ``rope_scaling`` dict: ``llama3`` (Llama-3.1/3.2: wavelengths beyond the original context
    divided by ``factor``, a smooth blend between ``low_freq_factor`` and ``high_freq_factor``),
    ``linear`` (every frequency divided by ``factor``) or none/``default``. Anything else raises:
    a silently ignored scaling rule changes the logits at every position."""
    import math

    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float64) / head_dim))
    if not scaling:
        return inv
    kind = scaling.get("rope_type", scaling.get("type", "default"))
    if kind in (None, "default"):
        return inv
    if kind == "linear":
        return inv / float(scaling["factor"])
    if kind == "llama3":
        factor = float(scaling["factor"])
        lo, hi = float(scaling.get("low_freq_factor", 1.0)), float(scaling.get("high_freq_factor", 4.0))
        old = float(scaling["original_max_position_embeddings"])
        wavelen = 2 * math.pi / inv
        smooth = ((old / wavelen - lo) / (hi - lo)).clamp(0.0, 1.0)
        blended = (1 - smooth) * inv / factor + smooth * inv
        out = torch.where(wavelen > old / lo, inv / factor, torch.where(wavelen < old / hi, inv, blended))
        return out
    raise ValueError(f"rope_scaling type {kind!r} is not supported (supported: default, linear, llama3)")


def rope_tables(head_dim: int, max_pos: int, theta: float, device="cpu",
                scaling: Optional[dict] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    inv = rope_inv_freq(head_dim, theta, scaling)
    ang = torch.arange(max_pos, dtype=torch.float64)[:, None] * inv[None, :]
    return ang.cos().float().to(device), ang.sin().float().to(device)


def rope_kv_write(qkv: torch.Tensor, positions: torch.Tensor, slot_mapping: torch.Tensor, cos_t: torch.Tensor,
                  sin_t: torch.Tensor, k_cache: torch.Tensor, vt_cache: torch.Tensor, Hq: int, Hkv: int,
                  Dh: int) -> None:
    """qkv [T, (Hq+2Hkv)*Dh]: Q rotated in place; K rotated -> k_cache[blk, h, off, :];
    V -> vt_cache[blk, h, :, off]."""
    T = qkv.shape[0]
    BS = k_cache.shape[2]
    x = qkv.view(T, Hq + 2 * Hkv, Dh)
    pos = positions.long().clamp(0, cos_t.shape[0] - 1)
    c, s = cos_t[pos][:, None, :], sin_t[pos][:, None, :]  # [T,1,Dh/2]
    half = Dh // 2
    qk = x[:, : Hq + Hkv].float()
    a, b = qk[..., :half], qk[..., half:]
    ra = (a * c - b * s).to(qkv.dtype)   # round like the kernel: each half separately
    rb = (b * c + a * s).to(qkv.dtype)
    rot = torch.cat([ra, rb], dim=-1)
    x[:, :Hq] = rot[:, :Hq]
    slots = slot_mapping.long()
    ok = slots >= 0
    blk, off = slots[ok] // BS, slots[ok] % BS
    k_cache[blk, :, off, :] = rot[ok][:, Hq:]
    vt_cache[blk, :, :, off] = x[ok][:, Hq + Hkv:]


# ------------------------------------------------------------------ attention
def _gather_kv(k_cache, vt_cache, table, n):
    BS = k_cache.shape[2]
    nb = (n + BS - 1) // BS
    blocks = table[:nb].long()
    K = k_cache[blocks].permute(1, 0, 2, 3).reshape(k_cache.shape[1], nb * BS, -1)[:, :n]          # [Hkv, n, Dh]
    V = vt_cache[blocks].permute(1, 2, 0, 3).reshape(vt_cache.shape[1], vt_cache.shape[2], nb * BS)[..., :n]  # [Hkv, Dh, n]
    return K.float(), V.float()


def attn_prefill(q: torch.Tensor, k_cache, vt_cache, block_tables, cu_q, ctx_lens, Hq, Hkv, Dh, scale,
                 window: int = 0) -> torch.Tensor:
    """q [T, Hq*Dh] (may be a strided view into qkv). Causal over the paged context."""
    T = q.shape[0]
    out = torch.zeros(T, Hq * Dh, dtype=q.dtype, device=q.device)
    G = Hq // Hkv
    for s in range(len(ctx_lens)):
        q0, q1 = int(cu_q[s]), int(cu_q[s + 1])
        n, ql = int(ctx_lens[s]), q1 - q0
        if ql == 0:
            continue
        K, V = _gather_kv(k_cache, vt_cache, block_tables[s], n)
        Q = q[q0:q1].float().view(ql, Hkv, G, Dh)
        sc = torch.einsum("qhgd,hkd->hgqk", Q, K) * scale
        pos = torch.arange(n - ql, n, device=q.device)[:, None]
        key = torch.arange(n, device=q.device)[None, :]
        mask = key <= pos
        if window > 0:
            mask &= key > pos - window
        sc = sc.masked_fill(~mask[None, None], float("-inf"))
        p = torch.softmax(sc, dim=-1)
        o = torch.einsum("hgqk,hdk->qhgd", p, V)
        out[q0:q1] = o.reshape(ql, Hq * Dh).to(q.dtype)
    return out


def attn_decode(q: torch.Tensor, k_cache, vt_cache, block_tables, ctx_lens, Hq, Hkv, Dh, scale,
                window: int = 0) -> torch.Tensor:
    B = q.shape[0]
    cu = torch.arange(B + 1, dtype=torch.int32, device=q.device)
    return attn_prefill(q, k_cache, vt_cache, block_tables, cu, ctx_lens, Hq, Hkv, Dh, scale, window)


# ------------------------------------------------------------------ sampling
TOP_P_SEARCH_RANGE = 30.0     # the threshold search starts this far below the row maximum (logit / T units)
TOP_P_SEARCH_BINS = 12        # thresholds tried per refinement
TOP_P_SEARCH_PASSES = 4       # refinements: resolution 30 / 12^4 = 1.4e-3


def top_p_threshold(z: torch.Tensor, keep: torch.Tensor, p: float, floor: float = float("-inf")) -> float:
    """The nucleus cut as sampling.hip finds it: the kept set is {z >= thr}, and thr is the
    LARGEST threshold on a grid, refined TOP_P_SEARCH_PASSES times by a factor of
    TOP_P_SEARCH_BINS inside [max - TOP_P_SEARCH_RANGE, max], whose kept mass still reaches
    p * mass(keep). Against the exact top-p (sort, cumulative sum) this keeps, besides the exact
    nucleus, at most the entries within 30 / 12^4 = 1.4e-3 of the exact cut; on the flat
    distributions of a random-init model that is a few entries of ~1e-5 of the mass each. The
    definition is what both the kernel and this oracle implement, so their draws agree token for
    token (the arithmetic of the grid is done in fp32 here as there)."""
    import numpy as np

    f32 = np.float32
    zz = z.to(torch.float32).numpy()
    live = keep.numpy()
    zmax = f32(zz[live].max()) if live.any() else f32(0.0)
    mass = np.where(live, np.exp((zz - zmax).astype(np.float64)), 0.0)
    target = p * float(mass.sum())
    lo, hi = f32(max(f32(floor), zmax - f32(TOP_P_SEARCH_RANGE))), zmax
    thr = f32(floor)
    above, hi_is_max = 0.0, True
    for _ in range(TOP_P_SEARCH_PASSES):
        w = f32((hi - lo) * f32(1.0 / TOP_P_SEARCH_BINS))
        inv_w = f32(1.0) / w if w > 0 else f32(0.0)
        cand = live & (zz >= lo) & (np.ones_like(live) if hi_is_max else (zz < hi))
        b = np.minimum(((zz - lo) * inv_w).astype(np.int64), TOP_P_SEARCH_BINS - 1)
        bins = np.bincount(b[cand], weights=mass[cand], minlength=TOP_P_SEARCH_BINS)
        cum, best, above_best = above, -1, above
        for j in range(TOP_P_SEARCH_BINS - 1, -1, -1):
            before = cum
            cum += float(bins[j])
            if cum >= target:
                best, above_best = j, before
                break
        if best < 0:
            break
        nlo = f32(lo + w * f32(best))
        if best < TOP_P_SEARCH_BINS - 1:
            hi = f32(lo + w * f32(best + 1))
            hi_is_max = False
            above = above_best
        lo = nlo
        thr = lo
    return float(thr)


def sample(logits: torch.Tensor, temperature, top_p, top_k, seeds, steps, row_index=None):
    """Same definition as sampling.hip: scale by 1/T, keep the k largest, then the nucleus
    {z >= thr} with thr from ``top_p_threshold`` (relative to the mass that survived top-k), draw
    by inverse CDF in *index* order with u = rng_uniform(seed, step)."""
    R = len(temperature)
    toks = torch.zeros(R, dtype=torch.int32)
    lps = torch.zeros(R, dtype=torch.float32)
    for r in range(R):
        row = logits[int(row_index[r]) if row_index is not None else r].double().cpu()
        T = float(temperature[r])
        if not T > 0:
            lp = torch.log_softmax(row, -1)
            t = int(torch.argmax(row))
            toks[r], lps[r] = t, float(lp[t])
            continue
        z32 = (logits[int(row_index[r]) if row_index is not None else r].float().cpu()
               * torch.tensor(1.0 / T, dtype=torch.float32))
        z = row / T
        w = torch.exp(z - z.max())
        order = torch.argsort(w, descending=True, stable=True)
        keep = torch.ones_like(w, dtype=torch.bool)
        k = int(top_k[r]) if top_k is not None else 0
        floor = float("-inf")
        if 0 < k < w.numel():
            kth = w[order[k - 1]]
            keep &= w >= kth
            floor = float(z32[order[k - 1]])
        p = float(top_p[r]) if top_p is not None else 1.0
        if p < 1.0:
            keep &= z32 >= top_p_threshold(z32, keep, p, floor)
        wk = torch.where(keep, w, torch.zeros_like(w))
        u = rng_uniform(int(seeds[r]), int(steps[r]) if steps is not None else 0)
        csum = torch.cumsum(wk, 0)
        target = u * float(csum[-1])
        t = int(torch.searchsorted(csum, torch.tensor(target, dtype=csum.dtype), right=True).item())
        t = min(t, w.numel() - 1)
        while t > 0 and not bool(keep[t]):
            t -= 1
        toks[r] = t
        lps[r] = float(torch.log(wk[t] / csum[-1]))
    return toks.to(logits.device), lps.to(logits.device)


def logprob_gather(logits: torch.Tensor, ids: torch.Tensor, row_index=None) -> torch.Tensor:
    rows = logits if row_index is None else logits[row_index.long()]
    lp = torch.log_softmax(rows.float(), -1)
    ids = ids.long()
    if ids.dim() == 1:
        ids = ids[None, :].expand(rows.shape[0], -1)
    return torch.gather(lp, 1, ids.clamp(0, rows.shape[1] - 1))


# ------------------------------------------------------------------ MoE
def moe_router(x: torch.Tensor, Wr: torch.Tensor, topk: int):
    logits = (x.float() @ Wr.float().t()).to(x.dtype).float()  # HF rounds router logits to the model dtype
    p = torch.softmax(logits, -1)
    w, ids = torch.topk(p, topk, dim=-1)
    w = w / w.sum(-1, keepdim=True)
    return ids.int(), w.float(), logits


def embed_pool(hidden: torch.Tensor, cu: torch.Tensor, mode: int = 0) -> torch.Tensor:
    outs = []
    for s in range(len(cu) - 1):
        r0, r1 = int(cu[s]), int(cu[s + 1])
        if r1 <= r0:
            outs.append(torch.zeros(hidden.shape[1]))
            continue
        v = hidden[r1 - 1].float() if mode == 1 else hidden[r0:r1].float().mean(0)
        n = v.norm()
        outs.append(v / n if n > 0 else v)
    return torch.stack(outs).to(hidden.device)
