"""Mixtral-8x7B block-sparse FFN (8 experts, top-2, d 4096, d_ff 14336) at prefill and decode
sizes: time per call, achieved FLOP/s on the routed rows and weight-streaming bandwidth.

  python benchmarks/moe_ffn_bench.py [T ...]      (default 16 64 1024 8192)
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_consensus_amd import ops  # noqa: E402

DEV = "cuda"


def main():
    Ts = [int(a) for a in sys.argv[1:]] or [16, 64, 1024, 8192]
    d, dff, E, topk, copies = 4096, 14336, 8, 2, 3
    g = torch.Generator(device=DEV).manual_seed(0)
    layers = []
    for _ in range(copies):                       # distinct copies: weights come from HBM, not the last-level cache
        gu = torch.stack([(torch.randn(2 * dff, d, device=DEV, generator=g) * 0.02).bfloat16() for _ in range(E)])
        dn = torch.stack([(torch.randn(d, dff, device=DEV, generator=g) * 0.02).bfloat16() for _ in range(E)])
        layers.append((ops.ExpertWeights.from_dense(gu, DEV, swiglu=True), ops.ExpertWeights.from_dense(dn, DEV)))
        del gu, dn
    Wr = (torch.randn(E, d, device=DEV, generator=g) * 0.02).bfloat16()
    wbytes = E * 3 * dff * d * 2
    out = []
    for T in Ts:
        x = (torch.randn(T, d, device=DEV, generator=g) * 0.5).bfloat16()
        fns = [lambda w=w: ops.moe_ffn(x, Wr, w[0], w[1], topk) for w in layers]
        for f in fns:
            f()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for f in fns:
                f()
        gr.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        gr.replay()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / len(fns) * 1e3
        flops = 2.0 * T * topk * 3 * dff * d
        row = {"T": T, "us": round(us, 1), "PFLOPs": round(flops / us / 1e9, 3), "weight_TBs": round(wbytes / us / 1e6, 2)}
        out.append(row)
        print(row, flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/moe_ffn_bench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
