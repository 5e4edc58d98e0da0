"""Prefill attention alone (Llama-3-8B heads, causal): us per call and TFLOP/s (causal FLOPs) at the
ensemble benchmark's chunk shape and at longer prompts.   python benchmarks/attn_prefill_bench.py [out.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_consensus_amd import ops  # noqa: E402

DEV = "cuda"
Hq, Hkv, Dh, BS = 32, 8, 128, 64


def run(nseq, L, iters=20):
    nb = (L + BS - 1) // BS
    kc = (torch.randn(nseq * nb, Hkv, BS, Dh, device=DEV) * 0.5).bfloat16()
    vc = (torch.randn(nseq * nb, Hkv, Dh, BS, device=DEV) * 0.5).bfloat16()
    tables = torch.arange(nseq * nb, dtype=torch.int32, device=DEV).view(nseq, nb)
    cu = torch.arange(nseq + 1, dtype=torch.int32, device=DEV) * L
    ctx_lens = torch.full((nseq,), L, dtype=torch.int32, device=DEV)
    qkv = (torch.randn(nseq * L, (Hq + 2 * Hkv) * Dh, device=DEV) * 0.5).bfloat16()
    fn = lambda: ops.attn_prefill(qkv, kc, vc, tables, cu, ctx_lens, L, Hq, Hkv, Dh, Dh ** -0.5)  # noqa: E731
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    flops = 4 * nseq * Hq * Dh * L * L / 2
    return {"nseq": nseq, "L": L, "us": round(us, 1), "tflops": round(flops / us / 1e6, 1)}


def run_ragged(lens, iters=20):
    """Prompts of different lengths in one call (block table sized for the longest)."""
    nseq, Lmax = len(lens), max(lens)
    nb = (Lmax + BS - 1) // BS
    kc = (torch.randn(nseq * nb, Hkv, BS, Dh, device=DEV) * 0.5).bfloat16()
    vc = (torch.randn(nseq * nb, Hkv, Dh, BS, device=DEV) * 0.5).bfloat16()
    tables = torch.arange(nseq * nb, dtype=torch.int32, device=DEV).view(nseq, nb)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=DEV)
    ctx_lens = torch.tensor(lens, dtype=torch.int32, device=DEV)
    qkv = (torch.randn(sum(lens), (Hq + 2 * Hkv) * Dh, device=DEV) * 0.5).bfloat16()
    fn = lambda: ops.attn_prefill(qkv, kc, vc, tables, cu, ctx_lens, Lmax, Hq, Hkv, Dh, Dh ** -0.5)  # noqa: E731
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    flops = sum(4 * Hq * Dh * L * L / 2 for L in lens)
    return {"lens": f"{lens[0]} + {len(lens) - 1} x {lens[1]}", "us": round(us, 1), "tflops": round(flops / us / 1e6, 1)}


def main():
    rows = [run(128, 64), run(64, 128), run(32, 256), run(16, 512), run(8, 1024), run(4, 2048), run(1, 8192),
            run(2, 16384, iters=5), run_ragged([4096] + [288] * 15), run_ragged([8192] + [32] * 31)]
    for r in rows:
        print(r, flush=True)
    if len(sys.argv) > 1:
        json.dump(rows, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
