"""Decode attention at the ensemble benchmark's shape with an HBM-resident KV cache: several
layers' worth of distinct caches (>= 1.5 GB) are walked inside one hipGraph so every call
misses the last-level cache, as a real model's 32 layers do. Reports us per call and the
KV-stream bandwidth for 1-, 2- and 4-wave workgroups.

  python benchmarks/attn_decode_bench.py [B] [ctx] [nsplit]      (default 64 544 auto)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_consensus_amd import ops  # noqa: E402

DEV = "cuda"


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    ctx_len = int(sys.argv[2]) if len(sys.argv) > 2 else 544
    Hq, Hkv, Dh, BS = 32, 8, 128, 64
    nb_seq = (ctx_len + BS - 1) // BS + 1
    nblk = B * nb_seq
    kv_bytes = B * ctx_len * Hkv * Dh * 2 * 2
    layers = max(2, int(1.5e9 // kv_bytes))
    caches = [((torch.randn(nblk, Hkv, BS, Dh, device=DEV) * 0.5).bfloat16(),
               (torch.randn(nblk, Hkv, Dh, BS, device=DEV) * 0.5).bfloat16()) for _ in range(layers)]
    bt = torch.arange(nblk, dtype=torch.int32, device=DEV).view(B, nb_seq)
    ctx = torch.full((B,), ctx_len, dtype=torch.int32, device=DEV)
    qkv = (torch.randn(B, (Hq + 2 * Hkv) * Dh, device=DEV) * 0.5).bfloat16()
    nsplit = int(sys.argv[3]) if len(sys.argv) > 3 else ops.choose_kv_splits(B, Hkv, 1024)
    print(f"B={B} ctx={ctx_len} layers={layers} nsplit={nsplit} kv={kv_bytes / 1e6:.0f} MB/call")
    ref = None
    for waves in (1, 2, 4):
        ops.ATTN_DECODE_MAX_WAVES = waves

        def run():
            out = None
            for kc, vt in caches:
                out = ops.attn_decode(qkv, kc, vt, bt, ctx, Hq, Hkv, Dh, Dh ** -0.5, 0, nsplit)
            return out

        out = run()
        torch.cuda.synchronize()
        if ref is None:
            ref = out.float()
        err = (out.float() - ref).abs().max().item()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(3):
                run()
        g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / (3 * layers)
        print(f"waves={waves}: {us:7.1f} us  {kv_bytes / us / 1e6:5.2f} TB/s  max|diff vs waves=1| {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
