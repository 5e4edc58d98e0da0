"""Decode attention: us per call for every (batch, context, requested KV splits) of a small grid,
HBM-resident caches walked inside one hipGraph (see attn_decode_bench.py). The table is what
ops.choose_kv_splits is fitted to.   python benchmarks/attn_decode_sweep.py [out.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_consensus_amd import ops  # noqa: E402

DEV = "cuda"
Hq, Hkv, Dh, BS = 32, 8, 128, 64


def measure(B, ctx_len, nsplit):
    nb_seq = (ctx_len + BS - 1) // BS + 1
    nblk = B * nb_seq
    kv_bytes = B * ctx_len * Hkv * Dh * 2 * 2
    layers = max(2, min(24, int(1.2e9 // kv_bytes)))
    caches = [((torch.randn(nblk, Hkv, BS, Dh, device=DEV) * 0.5).bfloat16(),
               (torch.randn(nblk, Hkv, Dh, BS, device=DEV) * 0.5).bfloat16()) for _ in range(layers)]
    bt = torch.arange(nblk, dtype=torch.int32, device=DEV).view(B, nb_seq)
    ctx = torch.full((B,), ctx_len, dtype=torch.int32, device=DEV)
    qkv = (torch.randn(B, (Hq + 2 * Hkv) * Dh, device=DEV) * 0.5).bfloat16()

    def run():
        for kc, vt in caches:
            ops.attn_decode(qkv, kc, vt, bt, ctx, Hq, Hkv, Dh, Dh ** -0.5, 0, nsplit)

    run()
    torch.cuda.synchronize()
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
    return us, kv_bytes / us / 1e6


def main():
    rows = []
    for B in (4, 16, 64, 256):
        for ctx_len in (544, 2048, 8192):
            if B * ctx_len > 256 * 2048:
                continue
            line = []
            for nsplit in (1, 2, 4, 8, 16):
                if nsplit > 1 and ctx_len // nsplit < 64:
                    continue
                us, tbs = measure(B, ctx_len, nsplit)
                rows.append({"B": B, "ctx": ctx_len, "nsplit": nsplit, "us": round(us, 1), "TBps": round(tbs, 2)})
                line.append(f"{nsplit}:{us:6.1f}us/{tbs:4.2f}")
            print(f"B={B:3d} ctx={ctx_len:5d}  " + "  ".join(line), flush=True)
    if len(sys.argv) > 1:
        json.dump(rows, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
