"""Decode-shape GEMM benchmark with HBM-resident weights.

A 7-8B model streams ~15 GB of weights per decode step, so every GEMM reads its matrix from
HBM. Timing one matrix in a loop would serve it from the 256 MB last-level cache instead;
here each shape gets enough distinct copies (>= 1.5 GB) that every call misses, and all
copies run inside one hipGraph (no host launch overhead). Reports us per GEMM and the
weight-stream bandwidth; torch.matmul (hipBLASLt) on the same copies is the yardstick.

  python benchmarks/decode_gemm_bench.py [M ...]      (default M = 64 256)
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_consensus_amd import ops  # noqa: E402

DEV = "cuda"
SHAPES = [("qkv", 6144, 4096, False), ("o", 4096, 4096, False), ("gate_up", 28672, 4096, True),
          ("down", 4096, 14336, False), ("qwen_gate_up", 37888, 3584, True), ("qwen_down", 3584, 18944, False)]


def graph_time(fns, reps=3):
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            for f in fns:
                f()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (reps * len(fns))


def main():
    Ms = [int(a) for a in sys.argv[1:]] or [64, 256]
    variants = os.environ.get("LC_VARIANTS", "policy,reduce,torch").split(",")
    out = []
    for name, N, K, swiglu in SHAPES:
        nbytes = N * K * 2
        copies = max(2, int(1.5e9 // nbytes))
        Ws = [(torch.randn(N, K, device=DEV) * 0.05).bfloat16() for _ in range(copies)]
        lws = [ops.LinearWeight.from_dense(W, None, DEV, swiglu=swiglu) for W in Ws]
        for M in Ms:
            X = (torch.randn(M, K, device=DEV) * 0.5).bfloat16()
            row = {"shape": name, "M": M, "N": N, "K": K, "copies": copies}
            for v in variants:
                if v == "policy":      # what the model runs: deferred reduction where a consumer fuses it
                    fns = [lambda lw=lw: ops.linear(X, lw, defer_reduce=True) for lw in lws]
                elif v == "reduce":    # with the stand-alone split-K reduction
                    fns = [lambda lw=lw: ops.linear(X, lw) for lw in lws]
                elif v == "torch":
                    fns = [lambda W=W: X @ W.t() for W in Ws]
                elif v == "tile":
                    ops.WS_MAX_M, ops.GEMM256_MIN_TILES = 0, -1
                    fns = [lambda lw=lw: ops.linear(X, lw) for lw in lws]
                elif v == "panel":           # the decode kernel with its split-K slabs left unreduced
                    fns = [lambda lw=lw: ops.linear(X, lw, defer_reduce=True) for lw in lws]
                elif v == "g256":
                    if M < 256 or lws[0].n_pad % 256:
                        continue
                    ops.WS_MAX_M, ops.GEMM256_MIN_TILES = 0, 0
                    fns = [lambda lw=lw: ops.linear(X, lw) for lw in lws]
                else:
                    continue
                try:
                    us = graph_time(fns)
                    row[v + "_us"] = round(us, 1)
                    row[v + "_TBs"] = round(nbytes / us / 1e6, 2)
                except Exception as ex:  # noqa: BLE001
                    row[v + "_err"] = str(ex)[:80]
                ops.WS_MAX_M, ops.GEMM256_MIN_TILES = 256, 1
            out.append(row)
            print(row, flush=True)
        del Ws, lws
        torch.cuda.empty_cache()
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/decode_gemm_bench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
