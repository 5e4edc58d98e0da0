"""Prefill attention alone at the benchmark shape (16 x 512 tokens, Llama-3-8B heads), for counter passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_consensus_amd import ops  # noqa: E402

dev = "cuda"
Hq, Hkv, Dh, BS = 32, 8, 128, 64
S, L = 16, 512
nblk = S * (L // BS)
kc = (torch.randn(nblk, Hkv, BS, Dh, device=dev) * 0.5).bfloat16()
vt = (torch.randn(nblk, Hkv, Dh, BS, device=dev) * 0.5).bfloat16()
bt = torch.arange(nblk, dtype=torch.int32, device=dev).view(S, L // BS)
qkv = (torch.randn(S * L, (Hq + 2 * Hkv) * Dh, device=dev) * 0.5).bfloat16()
cu = torch.arange(0, S * L + 1, L, dtype=torch.int32, device=dev)
ctx = torch.full((S,), L, dtype=torch.int32, device=dev)
for _ in range(3):
    ops.attn_prefill(qkv, kc, vt, bt, cu, ctx, L, Hq, Hkv, Dh, Dh ** -0.5, 0)
torch.cuda.synchronize()
