"""Ablations of the 256x256 prefill GEMM main loop (LC_ABL bits: 1 no LDS fragment reads after the
first K-tile, 2 no LDS-DMA requests in the loop, 4 no barriers in the loop, 8 no MFMAs). Results are
wrong by construction; only the time matters. One process per variant (the switch is read once).

  python benchmarks/micro/gemm256_ablate.py          # runs every variant in child processes
"""
import json
import os
import subprocess
import sys

VARIANTS = [0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 12]


def child():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from llm_consensus_amd import ops
    M, N, K = 8192, 28672, 4096
    Ws = [(torch.randn(N, K, device="cuda") * 0.05).bfloat16() for _ in range(3)]
    lws = [ops.LinearWeight.from_dense(W, None, "cuda") for W in Ws]
    X = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    for lw in lws:
        ops.linear(X, lw)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(2):
            for lw in lws:
                ops.linear(X, lw)
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 6 * 1e3)
    print(json.dumps({"abl": int(os.environ.get("LC_ABL", "0")), "us": round(best, 1),
                      "PF": round(2.0 * M * N * K / best / 1e9, 3)}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for v in VARIANTS:
            env = dict(os.environ, LC_ABL=str(v))
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False, timeout=120)
