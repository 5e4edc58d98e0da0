"""Functional op layer used by the models and the engine.

CUDA tensors (ROCm) go to the hand-written gfx950 kernels through ``ops.hip``; CPU
tensors go to the PyTorch oracles in ``ops.reference`` (plumbing/tests only). There is
no PyTorch fallback on the GPU path: if the kernel library is missing the call raises.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import reference as ref

EPI_BF16, EPI_F32, EPI_SWIGLU = 0, 1, 2

# M at or below which the decode kernels (one activation panel) are used instead of the tile kernels.
WS_MAX_M = int(os.environ.get("LC_WS_MAX_M", "256"))
# Fewest 256x256 tiles for which the 256x256 pipelined kernel is used. It beats the 128x128 kernel
# even when its tiles cover a third of the CUs (160 tiles: 171 vs 308 us, 80 tiles: 145 vs 155 us at
# K = 8192), and below half coverage its split-K form takes over, so the default is "always".
# -1 disables the kernel (128x128 tiles everywhere).
GEMM256_MIN_TILES = int(os.environ.get("LC_GEMM256_MIN_TILES", "1"))
# Reduction depth from which prefill-sized GEMMs take the four-wave 256 x 256 kernel (gemm256w.hip):
# its longer prologue pays off only over many K-tiles (0 = never).
GEMM256W_MIN_K = int(os.environ.get("LC_GEMM256W_MIN_K", "8192")) or (1 << 30)
# Decode GEMM for 9..256 rows: the panel kernel (gemm_panel.hip, deep LDS-DMA ring) and its split-K policy.
PANEL_MAX_BLOCKS = int(os.environ.get("LC_PANEL_MAX_BLOCKS", "256"))
PANEL_MIN_KTILES = int(os.environ.get("LC_PANEL_MIN_KTILES", "4"))
# Ring depth of the panel kernel for M <= 128: deep (all of a CU's LDS, one workgroup per CU, fastest
# alone) or three stages (two workgroups per CU, for member engines that share the GPU on
# concurrent streams). Chosen when a decode graph is captured; see set_concurrent_engines().
PANEL_DEEP = os.environ.get("LC_PANEL_DEEP", "1") == "1"
PANEL_WIDE = os.environ.get("LC_PANEL_WIDE", "1") == "1"   # 160-column workgroups where they save a round (see linear())
PANEL_WIDE_MAX_M = int(os.environ.get("LC_PANEL_WIDE_MAX_M", "256"))


# Decode attention: how many of the KV splits the waves of one workgroup may take (1, 2 or 4).
ATTN_DECODE_MAX_WAVES = int(os.environ.get("LC_ATTN_WAVES", "4"))


def set_concurrent_engines(n: int) -> None:
    """Tells the op layer how many engines will issue decode steps on concurrent streams."""
    global PANEL_DEEP
    if "LC_PANEL_DEEP" not in os.environ:
        PANEL_DEEP = n <= 1
# In-workgroup K-split decode kernel (kgemm.hip): widest output it is used for, and a test override
# (-1 = automatic, 0 = never, 1/2/4 = force that many n-tiles per workgroup).
KGEMM_MAX_N = int(os.environ.get("LC_KGEMM_MAX_N", "8192"))
# Most rows the in-workgroup K-split kernel is used for. Measured against the panel kernel (with
# non-temporal weight loads) on HBM-resident weights: at 1 row it wins on the narrow projections
# (6144x4096 10.9 vs 13.0 us, 4096x4096 7.9 vs 9.3) and ties on the wide ones; at 8 rows it still wins
# those (11.9 vs 13.5, 8.4 vs 9.4) but loses where K is long (3584x18944: 26.5 vs 23.9); at 16 rows it
# loses everywhere (4096x14336: 28.3 vs 21.4, 3584x18944: 36.3 vs 24.2).
KGEMM_MAX_M = int(os.environ.get("LC_KGEMM_MAX_M", "8"))
KGEMM_LONG_K = 8192          # beyond this K the kernel is only used up to KGEMM_LONG_K_MAX_M rows
KGEMM_LONG_K_MAX_M = 4
KGEMM_FORCE_NTW = int(os.environ.get("LC_KGEMM_NTW", "-1"))


def _hip():
    from . import hip

    return hip


# ---------------------------------------------------------------------------- weights
def padded_rows(N: int) -> int:
    """Output rows are padded with zeros to whole column tiles: 256 (so the 256x256 prefill
    kernel and its split-K / grouped forms apply to every projection), 128 for tiny outputs."""
    return (N + 127) // 128 * 128 if N <= 128 else (N + 255) // 256 * 256


def pack_weight(W: torch.Tensor) -> torch.Tensor:
    """[N, K] row-major -> MFMA-fragment order [n_pad/16, K/32, 64, 8] (see gemm.hip):
    packed[nt, kt, lane, j] = W[nt*16 + (lane & 15), kt*32 + 8*(lane >> 4) + j]."""
    N, K = W.shape
    if K % 64:
        raise ValueError(f"K={K} must be a multiple of 64")
    n_pad = padded_rows(N)
    if n_pad != N:
        W = torch.cat([W, W.new_zeros(n_pad - N, K)], 0)
    return W.view(n_pad // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(n_pad // 16, K // 32, 64, 8)


def unpack_weight(Wp: torch.Tensor, N: int) -> torch.Tensor:
    nt, kt = Wp.shape[0], Wp.shape[1]
    return Wp.view(nt, kt, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(nt * 16, kt * 32)[:N].contiguous()


def interleave_gate_up(Wg: torch.Tensor, Wu: torch.Tensor) -> torch.Tensor:
    """Row order for the fused SwiGLU epilogue: 16-row tiles alternate gate / up."""
    dff, K = Wg.shape
    if dff % 16:
        raise ValueError("d_ff must be a multiple of 16")
    return torch.stack([Wg.view(dff // 16, 16, K), Wu.view(dff // 16, 16, K)], 1).reshape(2 * dff, K)


@dataclass
class LinearWeight:
    """A linear layer's weights in device layout. ``data`` is packed on GPU and row-major
    [N, K] on CPU. ``swiglu`` marks gate/up interleaving (output width N/2)."""

    data: torch.Tensor
    n: int
    k: int
    bias: Optional[torch.Tensor] = None
    swiglu: bool = False

    @property
    def n_pad(self) -> int:
        return padded_rows(self.n)

    @property
    def is_packed(self) -> bool:
        return self.data.dim() == 4

    @classmethod
    def from_dense(cls, W: torch.Tensor, bias: Optional[torch.Tensor] = None, device=None, swiglu: bool = False):
        """W row-major [N, K] in final row order (already interleaved when swiglu)."""
        device = torch.device(device) if device is not None else W.device
        N, K = W.shape
        W = W.to(device=device, dtype=torch.bfloat16)
        b = bias.to(device=device, dtype=torch.bfloat16).contiguous() if bias is not None else None
        data = pack_weight(W) if device.type == "cuda" else W.contiguous()
        return cls(data, N, K, b, swiglu)

    def dense(self) -> torch.Tensor:
        return unpack_weight(self.data, self.n) if self.is_packed else self.data

    def nbytes(self) -> int:
        return self.data.numel() * self.data.element_size() + (self.bias.numel() * 2 if self.bias is not None else 0)


def _kgemm_ntw(n_pad: int, M: int, swiglu: bool, K: int = 0) -> int:
    """n-tiles per workgroup for the in-workgroup K-split kernel, or 0 to use another kernel.
    It re-reads the activations once per workgroup from L2, so it is limited to M <= 64 and
    to outputs narrow enough that this stays below the weight traffic."""
    if KGEMM_FORCE_NTW >= 0:
        return KGEMM_FORCE_NTW if M <= 64 else 0
    if M > KGEMM_MAX_M or n_pad > KGEMM_MAX_N or (K > KGEMM_LONG_K and M > KGEMM_LONG_K_MAX_M):
        return 0
    for ntw in (4, 2, 1):
        if swiglu and ntw < 2:
            break
        if n_pad // (16 * ntw) >= 256:
            return ntw
    return 2 if swiglu else 1


def _choose_panel_splits(n_pad: int, K: int) -> int:
    """K splits of the panel kernel (gemm_panel.hip). Its LDS ring admits one workgroup per CU,
    so the aim is as many workgroups as possible WITHOUT exceeding the 256 CUs (a 257th would
    run alone in a second round); every split is a whole number of 64-deep K-tiles and at
    least PANEL_MIN_KTILES of them."""
    blocks = n_pad // 128
    nk = K // 64
    best = 1
    for s in (2, 3, 4, 5, 6, 7, 8, 12, 14, 16):
        if blocks * s > PANEL_MAX_BLOCKS:
            break
        if nk % s == 0 and nk // s >= PANEL_MIN_KTILES:
            best = s
    return best


def _choose_gemm256_splits(tiles: int, K: int) -> int:
    """K splits for the 256x256 kernel when its tiles alone cover less than half of the CUs:
    the most splits that keep tiles * splits within the 256 CUs, in whole 64-deep K-tiles, at
    least 8 of them each (below that the pipeline's fill and the slab traffic dominate)."""
    if tiles * 2 > 256 or GEMM256_MAX_SPLITS <= 1:
        return 1
    nk = K // 64
    best = 1
    for s in (2, 3, 4, 6, 7, 8):
        if tiles * s > 256 or s > GEMM256_MAX_SPLITS:
            break
        if nk % s == 0 and nk // s >= 8:
            best = s
    return best


@dataclass
class PartialSum:
    """Un-reduced output of a split-K GEMM: ``part`` holds S fp32 slabs [S, M, N] whose sum
    (+ bias) is the product. Consumers that read the product exactly once (the residual-add
    RMSNorm, the RoPE/KV-write kernel) sum the slabs themselves, which removes one kernel
    launch per GEMM from the decode step; anything else calls ``reduce()``."""

    part: torch.Tensor
    bias: Optional[torch.Tensor] = None

    @property
    def shape(self):
        return self.part.shape[1:]

    def reduce(self, out_f32: bool = False) -> torch.Tensor:
        S, M, N = self.part.shape
        y = torch.empty(M, N, dtype=torch.float32 if out_f32 else torch.bfloat16, device=self.part.device)
        _hip().call("lc_splitk_reduce", self.part.data_ptr(), _hip().ptr(self.bias), y.data_ptr(), S, M, N, N,
                    1 if out_f32 else 0)
        return y


def use_four_wave_gemm(k: int, out_f32: bool = False) -> bool:
    """Whether a prefill-sized GEMM (256 x 256 tiles) of reduction depth ``k`` takes the four-wave
    kernel (gemm256w.hip) instead of the eight-wave one: from GEMM256W_MIN_K on (the down
    projections of the served models: K = 14336 / 18944 / 28672; QKV, O and gate/up have K = d_model).
    The four-wave kernel has bf16 and SwiGLU epilogues only and needs at least four K-tiles."""
    return not out_f32 and k >= max(GEMM256W_MIN_K, 256)


def use_wide_panel(M: int, n_pad: int, k: int) -> bool:
    """Whether a decode GEMM of ``M`` rows against ``n_pad`` columns takes the 160-column workgroups:
    its 128-column tiles would need a second round on the 256 CUs (and K is not split), while
    ten-n-tile workgroups cover it in one."""
    return (PANEL_WIDE and M <= PANEL_WIDE_MAX_M and n_pad // 128 > 256
            and (n_pad // 16 + 9) // 10 <= 256 and _choose_panel_splits(n_pad, k) == 1)


def linear(x: torch.Tensor, w: LinearWeight, out_f32: bool = False, defer_reduce: bool = False):
    """y = x @ W^T (+bias). With ``w.swiglu`` returns silu(gate)*up of width N/2. With
    ``defer_reduce`` a split-K product may come back as a ``PartialSum``."""
    M = x.shape[0]
    out_n = w.n // 2 if w.swiglu else w.n
    if not x.is_cuda:
        Wd = w.dense()
        y = ref.linear(x, Wd, w.bias, torch.float32)
        if w.swiglu:
            y = y.view(M, w.n // 32, 2, 16)
            y = (torch.nn.functional.silu(y[:, :, 0]) * y[:, :, 1]).reshape(M, out_n)
        return y if out_f32 else y.to(x.dtype)
    h = _hip()
    if not w.is_packed:
        raise RuntimeError("GPU linear needs packed weights (LinearWeight.from_dense on a cuda device)")
    x = x.contiguous()
    epi = EPI_SWIGLU if w.swiglu else (EPI_F32 if out_f32 else EPI_BF16)
    y = torch.empty(M, out_n, dtype=torch.float32 if (out_f32 and not w.swiglu) else torch.bfloat16, device=x.device)
    if M == 0:
        return y
    if M > WS_MAX_M:
        tiles256 = ((M + 255) // 256) * (w.n_pad // 256)
        splits = _choose_gemm256_splits(tiles256, w.k) if (w.n_pad % 256 == 0 and not w.swiglu) else 1
        if splits > 1 and GEMM256_MIN_TILES >= 0:
            # Too few 256x256 tiles to fill the chip (a few hundred rows, narrow output): split K.
            part = torch.empty(splits, M, w.n, dtype=torch.float32, device=x.device)
            h.call("lc_gemm256_splitk", w.data.data_ptr(), x.data_ptr(), part.data_ptr(), M, w.n, w.k, w.n_pad, w.n,
                   splits)
            if defer_reduce and not out_f32:
                return PartialSum(part, w.bias)
            h.call("lc_splitk_reduce", part.data_ptr(), h.ptr(w.bias), y.data_ptr(), splits, M, w.n, out_n,
                   1 if out_f32 else 0)
            return y
        if GEMM256_MIN_TILES >= 0 and w.n_pad % 256 == 0 and tiles256 >= GEMM256_MIN_TILES:
            # long K (down projections): four waves of 128 x 128 with the accumulators in AGPRs
            kern = "lc_gemm256w" if use_four_wave_gemm(w.k, epi == EPI_F32) else "lc_gemm256"
            h.call(kern, w.data.data_ptr(), x.data_ptr(), y.data_ptr(), h.ptr(w.bias), M, w.n, w.k, w.n_pad,
                   out_n, epi)
            return y
        h.call("lc_gemm_tile", w.data.data_ptr(), x.data_ptr(), y.data_ptr(), h.ptr(w.bias), M, w.n, w.k, w.n_pad,
               out_n, epi)
        return y
    # ---- decode shapes: one activation panel of at most 256 rows ----
    ntw = _kgemm_ntw(w.n_pad, M, w.swiglu, w.k)
    if ntw:   # a handful of rows, narrow output: K split over the waves of one workgroup
        h.call("lc_kgemm", w.data.data_ptr(), x.data_ptr(), y.data_ptr(), h.ptr(w.bias), M, w.n, w.k, w.n_pad,
               out_n, epi, ntw)
        return y
    # 257..320 column tiles of 128 (Qwen2.5-7B gate/up: 296): a second round would keep 40 CUs busy
    # as long as the first kept 256; workgroups of 160 columns cover the matrix in one round
    # (at 256 rows 85-88 us against 100 for the 256x256 tiles below on that matrix)
    wide = epi != EPI_F32 and use_wide_panel(M, w.n_pad, w.k)
    # 129..256 rows against more than 256 column tiles (vocabulary projection, very wide gate/up):
    # a second round of 128-column tiles would start; 256x256 tiles (half as many) do better.
    if M > 128 and not wide and w.n_pad // 128 > 256 and w.n_pad % 256 == 0 and GEMM256_MIN_TILES >= 0:
        h.call("lc_gemm256", w.data.data_ptr(), x.data_ptr(), y.data_ptr(), h.ptr(w.bias), M, w.n, w.k, w.n_pad,
               out_n, epi)
        return y
    kernel, splits, extra = "lc_gemm_panel", _choose_panel_splits(w.n_pad, w.k), (1 if PANEL_DEEP else 0,)
    if wide:
        extra = (2,)
    if splits == 1:
        h.call(kernel, w.data.data_ptr(), x.data_ptr(), y.data_ptr(), h.ptr(w.bias), M, w.n, w.k, w.n_pad, out_n, epi, 1,
               *extra)
        return y
    part = torch.empty(splits, M, w.n, dtype=torch.float32, device=x.device)
    h.call(kernel, w.data.data_ptr(), x.data_ptr(), part.data_ptr(), None, M, w.n, w.k, w.n_pad, w.n, EPI_F32, splits,
           *extra)
    if defer_reduce and not w.swiglu and not out_f32:
        return PartialSum(part, w.bias)
    mode = 2 if w.swiglu else (1 if out_f32 else 0)
    h.call("lc_splitk_reduce", part.data_ptr(), h.ptr(w.bias), y.data_ptr(), splits, M, w.n, out_n, mode)
    return y


# ---------------------------------------------------------------------------- elementwise
def embed(ids: torch.Tensor, E: torch.Tensor, feed: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> torch.Tensor:
    """x[t] = E[ids[t]]. ``feed`` = (tokens int32 [*], rows int32 [T]): row t with rows[t] >= 0
    takes the id tokens[rows[t]] instead (the token sampled by a row of the previous decode step,
    read on the device by a step that was launched before that step's results came back)."""
    if not ids.is_cuda:
        if feed is not None:
            tok, rows = feed
            ids = torch.where(rows >= 0, tok[rows.clamp_min(0).long()].to(ids.dtype), ids)
        return ref.embed(ids, E)
    h = _hip()
    T, d = ids.numel(), E.shape[1]
    x = torch.empty(T, d, dtype=E.dtype, device=E.device)
    h.call("lc_embed_gather", ids.data_ptr(), E.data_ptr(), x.data_ptr(), T, d, E.shape[0],
           feed[0].data_ptr() if feed is not None else None, feed[1].data_ptr() if feed is not None else None)
    return x


def rmsnorm(hstate: torch.Tensor, w: torch.Tensor, eps: float, delta=None) -> torch.Tensor:
    """y = RMSNorm(hstate [+ delta]) * w. When ``delta`` (a tensor or a ``PartialSum``) is
    given, ``hstate`` (the residual stream) is updated in place to hstate + delta first."""
    if not hstate.is_cuda:
        return ref.rmsnorm(hstate, w, eps, delta)
    h = _hip()
    T, d = hstate.shape
    y = torch.empty_like(hstate)
    if isinstance(delta, PartialSum) and delta.bias is not None:
        delta = delta.reduce()
    if isinstance(delta, PartialSum):
        h.call("lc_rmsnorm", hstate.data_ptr(), None, w.data_ptr(), y.data_ptr(), T, d, float(eps),
               delta.part.data_ptr(), delta.part.shape[0])
    else:
        h.call("lc_rmsnorm", hstate.data_ptr(), h.ptr(delta), w.data_ptr(), y.data_ptr(), T, d, float(eps), None, 0)
    return y


def embed_pool(hidden: torch.Tensor, cu: torch.Tensor, mode: int = 0) -> torch.Tensor:
    """Pooled, L2-normalised embeddings of the sequences whose rows are [cu[s], cu[s+1])
    (mode 0: mean, 1: last row)."""
    return embed_pool_ranges(hidden, cu[:-1], cu[1:], mode)


def embed_pool_ranges(hidden: torch.Tensor, begin: torch.Tensor, end: torch.Tensor, mode: int = 0) -> torch.Tensor:
    """As ``embed_pool`` over explicit row ranges; mode 2 returns the plain fp32 row sums
    (chunked prompts add their chunk sums and normalise at the end with ``l2_normalize``)."""
    if not hidden.is_cuda:
        if mode == 2:
            return torch.stack([hidden[int(b):int(e)].float().sum(0) for b, e in zip(begin, end)])
        return torch.stack([ref.embed_pool(hidden, torch.tensor([int(b), int(e)]), mode)[0] for b, e in zip(begin, end)])
    h = _hip()
    n = begin.numel()
    begin, end = begin.contiguous(), end.contiguous()
    out = torch.empty(n, hidden.shape[1], dtype=torch.float32, device=hidden.device)
    h.call("lc_embed_pool", hidden.data_ptr(), begin.data_ptr(), end.data_ptr(), out.data_ptr(), n, hidden.shape[1], mode)
    return out


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    """Rows of fp32 ``x`` scaled to unit length (zero rows stay zero)."""
    if not x.is_cuda:
        return x / x.norm(dim=-1, keepdim=True).clamp_min(1e-30)
    h = _hip()
    x = x.contiguous()
    y = torch.empty_like(x)
    h.call("lc_l2_normalize", x.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1])
    return y


# ---------------------------------------------------------------------------- rope + cache
def rope_kv_write(qkv, positions, slot_mapping, cos_t, sin_t, k_cache, vt_cache, Hq, Hkv, Dh) -> torch.Tensor:
    """Rotates Q in place and writes rotated K / V into the paged cache. ``qkv`` may be a
    ``PartialSum``; returns the bf16 [T, (Hq+2Hkv)*Dh] buffer whose Q section attention reads."""
    if isinstance(qkv, PartialSum):
        h = _hip()
        S, T, N = qkv.part.shape
        out = torch.empty(T, N, dtype=torch.bfloat16, device=qkv.part.device)
        h.call("lc_rope_kv_write", out.data_ptr(), positions.data_ptr(), slot_mapping.data_ptr(), cos_t.data_ptr(),
               sin_t.data_ptr(), k_cache.data_ptr(), vt_cache.data_ptr(), T, Hq, Hkv, Dh, k_cache.shape[2],
               cos_t.shape[0], qkv.part.data_ptr(), S, h.ptr(qkv.bias))
        return out
    if not qkv.is_cuda:
        ref.rope_kv_write(qkv, positions, slot_mapping, cos_t, sin_t, k_cache, vt_cache, Hq, Hkv, Dh)
        return qkv
    h = _hip()
    h.call("lc_rope_kv_write", qkv.data_ptr(), positions.data_ptr(), slot_mapping.data_ptr(), cos_t.data_ptr(),
           sin_t.data_ptr(), k_cache.data_ptr(), vt_cache.data_ptr(), qkv.shape[0], Hq, Hkv, Dh, k_cache.shape[2],
           cos_t.shape[0], None, 0, None)
    return qkv


def copy_blocks(kv_base: torch.Tensor, src: torch.Tensor, dst: torch.Tensor, block_elems: int, n_planes: int,
                plane_elems: int) -> None:
    """kv_base is the flat KV pool; planes (layer x {K,V}) are plane_elems apart."""
    if not kv_base.is_cuda:
        flat = kv_base.view(n_planes, plane_elems // block_elems, block_elems)
        flat[:, dst.long()] = flat[:, src.long()]
        return
    h = _hip()
    es = kv_base.element_size()
    h.call("lc_copy_blocks", kv_base.data_ptr(), src.data_ptr(), dst.data_ptr(), src.numel(), block_elems * es,
           plane_elems * es, n_planes)


# ---------------------------------------------------------------------------- attention
def attn_prefill(qkv, k_cache, vt_cache, block_tables, cu_q, ctx_lens, max_q_len, Hq, Hkv, Dh, scale,
                 window: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """qkv [T, (Hq+2Hkv)*Dh] after rope_kv_write; returns [T, Hq*Dh] (written into ``out``, a
    contiguous row range of a larger tensor, when given)."""
    T = qkv.shape[0]
    if not qkv.is_cuda:
        r = ref.attn_prefill(qkv[:, : Hq * Dh], k_cache, vt_cache, block_tables, cu_q, ctx_lens, Hq, Hkv, Dh,
                             scale, window)
        return r if out is None else out.copy_(r)
    h = _hip()
    if out is None:
        out = torch.empty(T, Hq * Dh, dtype=qkv.dtype, device=qkv.device)
    h.call("lc_attn_prefill", qkv.data_ptr(), qkv.stride(0), k_cache.data_ptr(), vt_cache.data_ptr(),
           block_tables.data_ptr(), block_tables.shape[1], cu_q.data_ptr(), ctx_lens.data_ptr(), out.data_ptr(),
           ctx_lens.numel(), int(max_q_len), Hq, Hkv, Dh, k_cache.shape[2], float(scale), int(window))
    return out


MOE_PANEL_MAX_ROWS = int(os.environ.get("LC_MOE_PANEL_ROWS", "640"))   # routed rows (tokens x top-k) up to which the MoE GEMMs use the panel kernel
MOE_GEMM256_MIN_ROWS = int(os.environ.get("LC_MOE_G256_ROWS", "641"))   # routed rows from which the 256x256 grouped GEMM is used
GEMM256_MAX_SPLITS = int(os.environ.get("LC_G256_SPLITS", "8"))
ATTN_PREFIX_WAVES = 4   # 1 = one wave per prefix piece and one partial slot per piece
# The per-row pass folds the shared-prefix partials in and writes the final rows (LC_ATTN_FOLD_PREFIX=0:
# the three-launch form with a separate merge, kept for comparison runs).
ATTN_FOLD_PREFIX = os.environ.get("LC_ATTN_FOLD_PREFIX", "1") == "1"


def choose_kv_splits(B: int, Hkv: int, max_ctx: int) -> int:
    """KV splits per (sequence, kv head): about 1024 waves in total (4 per CU) without slicing
    below 128 keys. Measured with an HBM-resident cache (benchmarks/attn_decode_bench.py): fewer
    waves leave the memory pipe short of requests, more waves shorten every wave's key loop
    until its fixed cost (query load, block-table walk, merge) dominates: 64 sequences x 8 kv
    heads x 544 keys run at 3.0 / 4.2 / 3.7 / 2.7 TB/s with 1 / 2 / 4 / 8 splits."""
    waves = max(1, B * Hkv)
    want = max(1, 1024 // waves)
    cap = max(1, max_ctx // 128)
    return int(max(1, min(want, cap, 16)))


def attn_decode(qkv, k_cache, vt_cache, block_tables, ctx_lens, Hq, Hkv, Dh, scale, window: int = 0,
                nsplit: int = 1, groups: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None,
                kv_begin: Optional[torch.Tensor] = None, max_group_rows: int = 0,
                prefix_splits: int = 1, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One query token per row of qkv [B, (Hq+2Hkv)*Dh]. ``groups`` = (first, count,
    shared_blocks) int32 device tensors enables the shared-prefix pass: rows of a group
    attend to their common leading blocks in one MFMA pass and ``kv_begin`` [B] tells the
    per-row pass where each row's private suffix starts."""
    B = qkv.shape[0]
    if not qkv.is_cuda:
        r = ref.attn_decode(qkv[:, : Hq * Dh], k_cache, vt_cache, block_tables, ctx_lens, Hq, Hkv, Dh, scale,
                            window)
        return r if out is None else out.copy_(r)
    h = _hip()
    if out is None:
        out = torch.empty(B, Hq * Dh, dtype=qkv.dtype, device=qkv.device)
    use_prefix = groups is not None and max_group_rows > 1
    # Up to four of the requested KV splits are taken by the waves of one workgroup, which
    # merges them on chip; only the remaining factor produces partial slots in HBM.
    waves = (4 if nsplit >= 4 else (2 if nsplit >= 2 else 1)) if Hq // Hkv <= 8 else 1
    waves = min(waves, ATTN_DECODE_MAX_WAVES)
    nsplit = max(1, nsplit // waves)
    if nsplit == 1 and not use_prefix:
        h.call("lc_attn_decode", qkv.data_ptr(), qkv.stride(0), k_cache.data_ptr(), vt_cache.data_ptr(),
               block_tables.data_ptr(), block_tables.shape[1], ctx_lens.data_ptr(), None, out.data_ptr(), None, None,
               1, 1, B, Hq, Hkv, Dh, k_cache.shape[2], float(scale), int(window), waves, 0)
        return out
    # The prefix pass merges four pieces of its key range on chip the same way.
    pwaves = 4 if (use_prefix and prefix_splits >= 4 and ATTN_PREFIX_WAVES >= 4) else 1
    prefix_splits = max(1, prefix_splits // pwaves)
    np_total = nsplit + (prefix_splits if use_prefix else 0)
    part_o = torch.empty(B, Hq, np_total, Dh, dtype=torch.float32, device=qkv.device)
    part_lse = torch.empty(B, Hq, np_total, dtype=torch.float32, device=qkv.device)
    # With one private split per row the per-row pass runs AFTER the shared-prefix pass and folds
    # the prefix slots into its own result: two launches and no partial slot for the private range
    # (16 x 16-chain step of Llama-3-8B: 7.70 -> 7.61 ms; both passes as one launch of independent
    # workgroups + merge measured 7.72: they do not overlap, the per-row pass is bound by its ~30 MB of private keys).
    fold = use_prefix and nsplit == 1 and waves == 1 and kv_begin is not None and ATTN_FOLD_PREFIX
    if use_prefix and fold:
        gf, gc, gs = groups
        h.call("lc_attn_prefix", qkv.data_ptr(), qkv.stride(0), k_cache.data_ptr(), vt_cache.data_ptr(),
               block_tables.data_ptr(), block_tables.shape[1], gf.data_ptr(), gc.data_ptr(), gs.data_ptr(),
               gf.numel(), int(max_group_rows), int(prefix_splits), part_o.data_ptr(), part_lse.data_ptr(), np_total,
               nsplit, Hq, Hkv, Dh, k_cache.shape[2], float(scale), pwaves)
    h.call("lc_attn_decode", qkv.data_ptr(), qkv.stride(0), k_cache.data_ptr(), vt_cache.data_ptr(),
           block_tables.data_ptr(), block_tables.shape[1], ctx_lens.data_ptr(),
           kv_begin.data_ptr() if (use_prefix and kv_begin is not None) else None, out.data_ptr(), part_o.data_ptr(),
           part_lse.data_ptr(), np_total, nsplit, B, Hq, Hkv, Dh, k_cache.shape[2], float(scale), int(window), waves,
           1 if fold else 0)
    if fold:
        return out
    if use_prefix:
        gf, gc, gs = groups
        h.call("lc_attn_prefix", qkv.data_ptr(), qkv.stride(0), k_cache.data_ptr(), vt_cache.data_ptr(),
               block_tables.data_ptr(), block_tables.shape[1], gf.data_ptr(), gc.data_ptr(), gs.data_ptr(),
               gf.numel(), int(max_group_rows), int(prefix_splits), part_o.data_ptr(), part_lse.data_ptr(), np_total,
               nsplit, Hq, Hkv, Dh, k_cache.shape[2], float(scale), pwaves)
    h.call("lc_attn_merge", part_o.data_ptr(), part_lse.data_ptr(), out.data_ptr(), np_total, B * Hq, Dh)
    return out


# ---------------------------------------------------------------------------- sampling
def sample(logits, temperature, top_p, top_k, seeds, steps, row_index=None, out_tok: Optional[torch.Tensor] = None):
    """logits fp32 [*, V]; per-row params are device tensors (float32/float32/int32/int64/int32).
    Returns (tokens int32 [R], logprobs float32 [R]); the tokens are written into ``out_tok``
    (int32, at least R long) when given."""
    if not logits.is_cuda:
        tok, lp = ref.sample(logits, temperature, top_p, top_k, seeds, steps, row_index)
        if out_tok is not None:
            out_tok[: tok.numel()].copy_(tok)
            tok = out_tok[: tok.numel()]
        return tok, lp
    h = _hip()
    R = temperature.numel()
    tok = out_tok[:R] if out_tok is not None else torch.empty(R, dtype=torch.int32, device=logits.device)
    lp = torch.empty(R, dtype=torch.float32, device=logits.device)
    h.call("lc_sample", logits.data_ptr(), logits.shape[1], logits.stride(0), h.ptr(row_index),
           temperature.data_ptr(), h.ptr(top_p), h.ptr(top_k), seeds.data_ptr(), h.ptr(steps), tok.data_ptr(),
           lp.data_ptr(), R)
    return tok, lp


def logprob_gather(logits, ids, row_index=None) -> torch.Tensor:
    """log_softmax(logits[row])[ids]. ids int32 [C] (shared) or [R, C]."""
    if not logits.is_cuda:
        return ref.logprob_gather(logits, ids, row_index)
    h = _hip()
    R = row_index.numel() if row_index is not None else logits.shape[0]
    C = ids.shape[-1]
    out = torch.empty(R, C, dtype=torch.float32, device=logits.device)
    h.call("lc_logprob_gather", logits.data_ptr(), logits.shape[1], logits.stride(0), h.ptr(row_index),
           ids.data_ptr(), C if ids.dim() == 2 else 0, C, out.data_ptr(), R)
    return out


def logit_targets_shard(logits, ids, vocab_lo: int, row_index=None) -> torch.Tensor:
    """One vocabulary shard's share of ``logprob_gather``: fp32 [R, Cp] with the raw logits of the
    targets that fall into columns [vocab_lo, vocab_lo + V_local) (0 elsewhere) in columns
    [0, C), the shard's log-sum-exp in column C and zero padding up to Cp = C + 1 rounded up to
    four floats (whole 16-byte rows for the exchange). ``logprob_merge`` combines the shards."""
    C = ids.shape[-1]
    Cp = (C + 1 + 3) // 4 * 4
    R = row_index.numel() if row_index is not None else logits.shape[0]
    if not logits.is_cuda:
        rows = logits[row_index.long()] if row_index is not None else logits
        local = ids.long() - vocab_lo
        local = local.expand(R, C) if local.dim() == 2 else local.view(1, C).expand(R, C)
        mine = (local >= 0) & (local < rows.shape[1])
        out = torch.zeros(R, Cp, dtype=torch.float32)
        out[:, :C] = torch.where(mine, rows.gather(1, local.clamp(0, rows.shape[1] - 1)), torch.zeros(()))
        out[:, C] = torch.logsumexp(rows.float(), dim=-1)
        return out
    h = _hip()
    out = torch.empty(R, Cp, dtype=torch.float32, device=logits.device)
    h.call("lc_logit_targets_shard", logits.data_ptr(), logits.shape[1], logits.stride(0), h.ptr(row_index),
           ids.data_ptr(), C if ids.dim() == 2 else 0, C, Cp, int(vocab_lo), out.data_ptr(), R)
    return out


def logprob_merge(parts: torch.Tensor, C: int) -> torch.Tensor:
    """parts fp32 [W, R, Cp] (``logit_targets_shard`` of W vocabulary shards) -> log-softmax values
    of the targets, fp32 [R, C]."""
    W, R, Cp = parts.shape
    if not parts.is_cuda:
        return parts[:, :, :C].sum(0) - torch.logsumexp(parts[:, :, C], dim=0).unsqueeze(1)
    h = _hip()
    parts = parts.contiguous()
    out = torch.empty(R, C, dtype=torch.float32, device=parts.device)
    h.call("lc_logprob_merge", parts.data_ptr(), W, R, C, Cp, out.data_ptr())
    return out


def zero_(t: torch.Tensor) -> torch.Tensor:
    """Fills a contiguous tensor whose size is a multiple of 4 bytes with zero bits (in place)."""
    if not t.is_cuda:
        return t.zero_()
    nbytes = t.numel() * t.element_size()
    if nbytes % 4 or not t.is_contiguous():
        raise ValueError("zero_ needs a contiguous tensor of a multiple of 4 bytes")
    if nbytes:
        _hip().call("lc_fill_f32", t.data_ptr(), 0.0, nbytes // 4)
    return t


def accumulate_(y: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """y += x for fp32 tensors of one shape (in place)."""
    if not y.is_cuda:
        return y.add_(x)
    h = _hip()
    h.call("lc_accumulate_f32", y.data_ptr(), x.contiguous().data_ptr(), y.numel())
    return y


def pad_cols(x: torch.Tensor, n_pad: int, fill: float) -> torch.Tensor:
    """fp32 [R, n] -> [R, n_pad] with ``fill`` in the new columns."""
    R, n = x.shape
    if n == n_pad:
        return x
    if not x.is_cuda:
        return torch.nn.functional.pad(x, (0, n_pad - n), value=fill)
    h = _hip()
    out = torch.empty(R, n_pad, dtype=torch.float32, device=x.device)
    h.call("lc_pad_cols_f32", x.contiguous().data_ptr(), out.data_ptr(), R, n, n_pad, float(fill))
    return out


def concat_cols(parts: torch.Tensor, total_cols: Optional[int] = None) -> torch.Tensor:
    """[W, R, c] -> [R, W * c] (the W blocks side by side), cut to ``total_cols`` columns."""
    W, R, c = parts.shape
    total = W * c if total_cols is None else int(total_cols)
    if not parts.is_cuda:
        return parts.permute(1, 0, 2).reshape(R, W * c)[:, :total]
    es = parts.element_size()
    if es % 2:
        raise ValueError("concat_cols needs elements of 2 or 4 bytes")
    h = _hip()
    parts = parts.contiguous()
    out = torch.empty(R, total, dtype=parts.dtype, device=parts.device)
    u = es // 2
    h.call("lc_concat_cols", parts.data_ptr(), W, R, c * u, total * u, out.data_ptr())
    return out


def gather_rows(x: torch.Tensor, rows: torch.Tensor) -> torch.Tensor:
    """x[rows] for a bf16 matrix and int32 row numbers (the embedding-gather kernel: a row
    gather is what it is)."""
    if not x.is_cuda:
        return x[rows.long()]
    return embed(rows, x)


# ---------------------------------------------------------------------------- MoE
@dataclass
class ExpertWeights:
    """Stacked per-expert weights. GPU: packed [E, n_pad/16, K/32, 64, 8]; CPU: [E, N, K]."""

    data: torch.Tensor
    n: int
    k: int
    swiglu: bool = False

    @property
    def n_pad(self) -> int:
        return padded_rows(self.n)

    @classmethod
    def from_dense(cls, W: torch.Tensor, device=None, swiglu: bool = False):
        device = torch.device(device) if device is not None else W.device
        E, N, K = W.shape
        W = W.to(device=device, dtype=torch.bfloat16)
        data = torch.stack([pack_weight(W[e]) for e in range(E)]) if device.type == "cuda" else W.contiguous()
        return cls(data, N, K, swiglu)

    def nbytes(self) -> int:
        return self.data.numel() * self.data.element_size()


def moe_route(x: torch.Tensor, Wr: torch.Tensor, topk: int):
    """Top-k routing with softmax over the selected logits: (ids int32 [T,k], weights fp32 [T,k])."""
    if not x.is_cuda:
        ids, wts, _ = ref.moe_router(x, Wr, topk)
        return ids.to(torch.int32), wts
    h = _hip()
    T, d = x.shape
    ids = torch.empty(T, topk, dtype=torch.int32, device=x.device)
    wts = torch.empty(T, topk, dtype=torch.float32, device=x.device)
    h.call("lc_moe_router", x.data_ptr(), Wr.data_ptr(), T, d, Wr.shape[0], topk, ids.data_ptr(), wts.data_ptr(), None)
    return ids, wts


def _expert_mlp_cpu(xe: torch.Tensor, w_gate_up: "ExpertWeights", w_down: "ExpertWeights", e: int) -> torch.Tensor:
    gu = ref.linear(xe, w_gate_up.data[e], None, torch.float32)
    gu = gu.view(-1, w_gate_up.n // 32, 2, 16)
    act = (torch.nn.functional.silu(gu[:, :, 0]) * gu[:, :, 1]).reshape(xe.shape[0], -1).to(xe.dtype)
    return ref.linear(act, w_down.data[e], None, xe.dtype).float()


def moe_permute(ids: torch.Tensor, E: int):
    """Sorts the (token, k) pairs of ``ids`` [T, k] by expert, keeping token order inside an
    expert. Returns (offsets int32 [E+1], sorted_src int32 [T*k]: token row of every sorted
    position, slot_of int32 [T*k]: sorted position of pair t*k + j)."""
    T, topk = ids.shape
    A = T * topk
    dev = ids.device
    if not ids.is_cuda:
        flat = ids.reshape(-1).long()
        order = torch.argsort(flat, stable=True)
        counts = torch.bincount(flat, minlength=E)
        offsets = torch.zeros(E + 1, dtype=torch.int32)
        offsets[1:] = torch.cumsum(counts, 0).to(torch.int32)
        slot_of = torch.empty(A, dtype=torch.int32)
        slot_of[order] = torch.arange(A, dtype=torch.int32)
        return offsets, (order // topk).to(torch.int32), slot_of
    h = _hip()
    offsets = torch.empty(E + 1, dtype=torch.int32, device=dev)
    sorted_src = torch.empty(A, dtype=torch.int32, device=dev)
    slot_of = torch.empty(A, dtype=torch.int32, device=dev)
    h.call("lc_moe_permute", ids.data_ptr(), T, topk, E, offsets.data_ptr(), sorted_src.data_ptr(), slot_of.data_ptr())
    return offsets, sorted_src, slot_of


def experts_on_sorted(x: torch.Tensor, row_index: Optional[torch.Tensor], offsets: torch.Tensor, expert0: int,
                      w_gate_up: "ExpertWeights", w_down: "ExpertWeights", rows_bound: int,
                      zero_unwritten: bool = False) -> torch.Tensor:
    """The resident experts' SwiGLU MLPs over rows in expert-sorted order. Sorted position p
    in [offsets[expert0 + e], offsets[expert0 + e + 1]) belongs to resident expert e and reads
    ``x[row_index[p]]``. Returns y [rows_bound, d]; positions outside the resident experts'
    ranges are left unwritten (zero with ``zero_unwritten``). ``rows_bound`` (a host integer)
    bounds every offset, so nothing here waits for the device."""
    E_local = w_gate_up.data.shape[0]
    d = w_down.n
    dff = w_gate_up.n // 2
    dev = x.device
    if not x.is_cuda:
        y = torch.zeros(rows_bound, d, dtype=x.dtype)
        for e in range(E_local):
            lo, hi = int(offsets[expert0 + e]), int(offsets[expert0 + e + 1])
            if hi > lo:
                rows = x[row_index[lo:hi].long()] if row_index is not None else x[lo:hi]
                y[lo:hi] = _expert_mlp_cpu(rows, w_gate_up, w_down, e).to(x.dtype)
        return y
    h = _hip()
    off_ptr = offsets.data_ptr() + 4 * expert0          # the kernels index rows by absolute offsets
    act = torch.empty(rows_bound, dff, dtype=x.dtype, device=dev)
    y = torch.empty(rows_bound, d, dtype=x.dtype, device=dev)
    if zero_unwritten:
        zero_(y)
    # Decode steps (a few dozen routed rows in total): every expert's weights are streamed once
    # against one small panel. Prefill-sized batches use the tiled grouped kernels.
    gg = "lc_gemm_panel_grouped" if rows_bound <= MOE_PANEL_MAX_ROWS else "lc_gemm_grouped"
    if (gg == "lc_gemm_grouped" and rows_bound >= MOE_GEMM256_MIN_ROWS and w_gate_up.n_pad % 256 == 0
            and w_down.n_pad % 256 == 0 and w_gate_up.k % 64 == 0 and w_down.k % 64 == 0):
        # Prefill-sized batches: the 256x256 pipelined tile per expert. It reads contiguous rows,
        # so the rows are first gathered into expert order (a copy of rows_bound rows, a fraction
        # of a percent of the GEMMs' time).
        xs = embed(row_index, x) if row_index is not None else x
        h.call("lc_gemm256_grouped", w_gate_up.data.data_ptr(), w_gate_up.data[0].numel(), xs.data_ptr(), off_ptr,
               E_local, act.data_ptr(), rows_bound, w_gate_up.n, w_gate_up.k, w_gate_up.n_pad, dff, EPI_SWIGLU)
        h.call("lc_gemm256_grouped", w_down.data.data_ptr(), w_down.data[0].numel(), act.data_ptr(), off_ptr,
               E_local, y.data_ptr(), rows_bound, w_down.n, w_down.k, w_down.n_pad, d, EPI_BF16)
        return y
    h.call(gg, w_gate_up.data.data_ptr(), w_gate_up.data[0].numel(), x.data_ptr(), h.ptr(row_index), off_ptr, E_local,
           act.data_ptr(), rows_bound, w_gate_up.n, w_gate_up.k, w_gate_up.n_pad, dff, EPI_SWIGLU)
    h.call(gg, w_down.data.data_ptr(), w_down.data[0].numel(), act.data_ptr(), None, off_ptr, E_local, y.data_ptr(),
           rows_bound, w_down.n, w_down.k, w_down.n_pad, d, EPI_BF16)
    return y


def moe_combine(y: torch.Tensor, slot_of: torch.Tensor, wts: torch.Tensor, T: int,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[t] = sum_k wts[t, k] * y[slot_of[t*k + j]] (fp32 accumulation, activation dtype out;
    written into ``out``, T contiguous rows, when given)."""
    topk = wts.shape[1]
    d = y.shape[1]
    if not y.is_cuda:
        rows = y[slot_of.long()].float().view(T, topk, d)
        res = (rows * wts.float()[:, :, None]).sum(1).to(y.dtype)
        return res if out is None else out.copy_(res)
    h = _hip()
    if out is None:
        out = torch.empty(T, d, dtype=y.dtype, device=y.device)
    h.call("lc_moe_combine", y.data_ptr(), slot_of.data_ptr(), wts.data_ptr(), out.data_ptr(), T, d, topk)
    return out


def moe_experts(x: torch.Tensor, ids: torch.Tensor, wts: torch.Tensor, w_gate_up: "ExpertWeights",
                w_down: "ExpertWeights", expert_lo: int = 0, n_experts: Optional[int] = None) -> torch.Tensor:
    """sum_k wts[t,k] * MLP_{ids[t,k]}(x[t]) over the experts held in ``w_*`` (global ids
    [expert_lo, expert_lo + E_local)); (token, k) pairs routed elsewhere contribute zero, so
    expert-parallel ranks can sum their results. ``n_experts`` is the global expert count."""
    T, d = x.shape
    topk = ids.shape[1]
    E_local = w_gate_up.data.shape[0]
    E = n_experts or (expert_lo + E_local)
    offsets, sorted_src, slot_of = moe_permute(ids, E)
    y = experts_on_sorted(x, sorted_src, offsets, expert_lo, w_gate_up, w_down, T * topk, zero_unwritten=E_local != E)
    return moe_combine(y, slot_of, wts, T)


# ---- expert-parallel exchange pieces (parallel/ep.py strings them together) ----
def ep_pack(xs: torch.Tensor, sorted_src: torch.Tensor, offsets: torch.Tensor, slot_of: torch.Tensor, per: int,
            ep: int, cap: int):
    """Send layout of one rank: ``send`` [ep*cap, d] (rows j < count of destination r at
    r*cap + j, in expert-sorted order), ``send_counts`` int32 [ep*per] (pairs per expert) and
    ``slot_back`` int32 [A]: row of the returned [ep*cap, d] buffer holding pair a's output."""
    A = slot_of.numel()
    d = xs.shape[1]
    dev = xs.device
    if not xs.is_cuda:
        send = torch.zeros(ep * cap, d, dtype=xs.dtype)
        off = offsets.long()
        counts = (off[1:] - off[:-1]).to(torch.int32)
        slot_back = torch.empty(A, dtype=torch.int32)
        for r in range(ep):
            lo, hi = int(off[r * per]), int(off[(r + 1) * per])
            send[r * cap: r * cap + hi - lo] = xs[sorted_src[lo:hi].long()]
            m = (slot_of >= lo) & (slot_of < hi)
            slot_back[m] = (r * cap + slot_of[m] - lo).to(torch.int32)
        return send, counts, slot_back
    h = _hip()
    send = torch.empty(ep * cap, d, dtype=xs.dtype, device=dev)
    counts = torch.empty(ep * per, dtype=torch.int32, device=dev)
    slot_back = torch.empty(A, dtype=torch.int32, device=dev)
    h.call("lc_ep_pack", xs.data_ptr(), sorted_src.data_ptr(), offsets.data_ptr(), slot_of.data_ptr(), per, ep, cap, A,
           d, send.data_ptr(), counts.data_ptr(), slot_back.data_ptr())
    return send, counts, slot_back


def ep_recv_index(recv_counts: torch.Tensor, ep: int, per: int, cap: int):
    """From the received per-(source, expert) counts: (row_index int32 [ep*cap]: receive-buffer
    row of every expert-sorted position, offsets_local int32 [per+1])."""
    dev = recv_counts.device
    if not recv_counts.is_cuda:
        c = recv_counts.view(ep, per).long()
        row_index = torch.zeros(ep * cap, dtype=torch.int32)
        offsets = torch.zeros(per + 1, dtype=torch.int32)
        pos = 0
        for e in range(per):
            offsets[e] = pos
            for s in range(ep):
                n, start = int(c[s, e]), int(c[s, :e].sum())
                row_index[pos:pos + n] = torch.arange(s * cap + start, s * cap + start + n, dtype=torch.int32)
                pos += n
        offsets[per] = pos
        return row_index, offsets
    h = _hip()
    row_index = torch.empty(ep * cap, dtype=torch.int32, device=dev)
    offsets = torch.empty(per + 1, dtype=torch.int32, device=dev)
    h.call("lc_ep_recv_index", recv_counts.data_ptr(), ep, per, cap, row_index.data_ptr(), offsets.data_ptr())
    return row_index, offsets


def scatter_rows(y: torch.Tensor, row_index: torch.Tensor, total: torch.Tensor) -> torch.Tensor:
    """out[row_index[p]] = y[p] for p < total[0] (a device integer); other rows are unspecified."""
    if not y.is_cuda:
        out = torch.zeros_like(y)
        n = int(total[0])
        out[row_index[:n].long()] = y[:n]
        return out
    h = _hip()
    out = torch.empty_like(y)
    h.call("lc_scatter_rows", y.data_ptr(), row_index.data_ptr(), total.data_ptr(), y.shape[0], y.shape[1], out.data_ptr())
    return out


def moe_ffn(x: torch.Tensor, Wr: torch.Tensor, w_gate_up: ExpertWeights, w_down: ExpertWeights, topk: int,
            expert_lo: int = 0) -> torch.Tensor:
    """Mixtral block-sparse FFN: route, run the selected experts' SwiGLU MLP, combine. With a
    subset of the experts resident (``expert_lo``, stacked weights shorter than the router)
    the result is this rank's partial sum."""
    ids, wts = moe_route(x, Wr, topk)
    return moe_experts(x, ids, wts, w_gate_up, w_down, expert_lo, Wr.shape[0])
