"""Prefill-shape GEMMs (M = one 8192-token chunk) of a 7-8B model: the 256x256 kernel against
torch.matmul (hipBLASLt) as a yardstick. Each shape cycles through distinct weight copies so
the weights come from HBM as they do in a 32-layer model.

  python benchmarks/prefill_gemm_bench.py [M]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_consensus_amd import ops  # noqa: E402

DEV = "cuda"


def graph_time(fns, reps=3):
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            for f in fns:
                f()
    # the first replays run while the clocks are still coming up from idle (the first shape of a
    # run measured ~15 % slow with a single untimed replay): warm for ~50 ms, then take the median
    for _ in range(8):
        g.replay()
    torch.cuda.synchronize()
    times = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        times.append(s.elapsed_time(e))
    return sorted(times)[len(times) // 2] / (reps * len(fns)) * 1e3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    shapes = [("qkv", 6144, 4096, False), ("o", 4096, 4096, False), ("gate_up", 28672, 4096, True),
              ("gate_up_plain", 28672, 4096, False), ("down", 4096, 14336, False)]
    out = []
    for name, N, K, swiglu in shapes:
        copies = 4
        Ws = [(torch.randn(N, K, device=DEV) * 0.05).bfloat16() for _ in range(copies)]
        lws = [ops.LinearWeight.from_dense(W, None, DEV, swiglu=swiglu) for W in Ws]
        X = (torch.randn(M, K, device=DEV) * 0.5).bfloat16()
        flops = 2.0 * M * N * K
        ours = [lambda lw=lw: ops.linear(X, lw) for lw in lws]
        ref = [lambda W=W: X @ W.t() for W in Ws]
        row = {"shape": name, "M": M, "N": N, "K": K}
        us = graph_time(ours)
        row["ours_us"], row["ours_PF"] = round(us, 1), round(flops / us / 1e9, 3)
        us = graph_time(ref)
        row["torch_us"], row["torch_PF"] = round(us, 1), round(flops / us / 1e9, 3)
        out.append(row)
        print(row, flush=True)
        del Ws, lws
        torch.cuda.empty_cache()
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/prefill_gemm_bench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
