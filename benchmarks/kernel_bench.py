"""Micro-benchmarks of the hot kernels on one MI355X; writes gpurun_out/kernel_bench.json.
torch.matmul (hipBLASLt) is timed on the same random data purely as a yardstick."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_consensus_amd import ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=20, warmup=3, graph=True):
    """Time per call in us. With graph=True the calls are captured into one hipGraph so host
    launch overhead (ctypes + allocator, ~15 us per op) does not mask short kernels."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters * 1e3
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


MS = tuple(int(x) for x in os.environ.get("LC_BENCH_MS", "1,16,64,128,256,1024,8192").split(","))


def bench_gemm(results):
    shapes = [(4096, 4096), (6144, 4096), (28672, 4096), (4096, 14336), (128256, 4096)]
    for N, K in shapes:
        W = (torch.randn(N, K, device=DEV) * 0.05).bfloat16()
        lw = ops.LinearWeight.from_dense(W, None, DEV)
        for M in MS:
            if N > 100000 and M > 256:
                continue
            X = (torch.randn(M, K, device=DEV) * 0.5).bfloat16()
            row = {"op": "linear", "M": M, "N": N, "K": K}
            for name, ws_max, g256 in (("panel", 256, -1), ("k1", 256, -1), ("k2", 256, -1), ("k4", 256, -1),
                                       ("tile", 0, -1), ("g256", 0, 0)):
                if name in ("panel", "k1", "k2", "k4") and M > 256:
                    continue
                if name[0] == "k" and (M > 64 or N > 40960):
                    continue
                ops.KGEMM_FORCE_NTW = int(name[1]) if name[0] == "k" else 0
                if name == "g256" and (M < 256 or lw.n_pad % 256):
                    continue
                ops.WS_MAX_M, ops.GEMM256_MIN_TILES = ws_max, g256
                try:
                    us = timeit(lambda: ops.linear(X, lw))
                except Exception as ex:  # noqa: BLE001
                    row[name + "_err"] = str(ex)
                    continue
                row[name + "_us"] = round(us, 2)
                row[name + "_tflops"] = round(2 * M * N * K / us / 1e6, 1)
                row[name + "_gbs"] = round((N * K * 2 + M * K * 2 + M * N * 2) / us / 1e3, 1)
            ops.WS_MAX_M, ops.GEMM256_MIN_TILES, ops.KGEMM_FORCE_NTW = 256, 1, -1
            us = timeit(lambda: X @ W.t())
            row["torch_us"] = round(us, 2)
            row["torch_tflops"] = round(2 * M * N * K / us / 1e6, 1)
            results.append(row)
            print(row, flush=True)
        del W, lw


def bench_attention(results):
    Hq, Hkv, Dh, BS = 32, 8, 128, 64
    scale = Dh ** -0.5
    for B, ctx in ((128, 640), (16, 640), (128, 2048)):
        nb = (ctx + BS - 1) // BS
        kc = (torch.randn(B * nb, Hkv, BS, Dh, device=DEV) * 0.5).bfloat16()
        vc = (torch.randn(B * nb, Hkv, Dh, BS, device=DEV) * 0.5).bfloat16()
        tables = torch.arange(B * nb, dtype=torch.int32, device=DEV).view(B, nb)
        ctx_lens = torch.full((B,), ctx, dtype=torch.int32, device=DEV)
        qkv = (torch.randn(B, (Hq + 2 * Hkv) * Dh, device=DEV) * 0.5).bfloat16()
        for nsplit in (1, 2, 4, 8):
            us = timeit(lambda: ops.attn_decode(qkv, kc, vc, tables, ctx_lens, Hq, Hkv, Dh, scale, nsplit=nsplit))
            kv_bytes = B * ctx * Hkv * Dh * 2 * 2
            row = {"op": "attn_decode", "B": B, "ctx": ctx, "nsplit": nsplit, "us": round(us, 2),
                   "gbs": round(kv_bytes / us / 1e3, 1)}
            results.append(row)
            print(row, flush=True)
        # K=16 chains sharing the whole context except the last block
        if B % 16 == 0:
            G = B // 16
            t2 = tables.clone()
            for g in range(G):
                t2[g * 16:(g + 1) * 16, : nb - 1] = tables[g * 16, : nb - 1]
            groups = (torch.arange(G, dtype=torch.int32, device=DEV) * 16,
                      torch.full((G,), 16, dtype=torch.int32, device=DEV),
                      torch.full((G,), nb - 1, dtype=torch.int32, device=DEV))
            kvb = torch.full((B,), (nb - 1) * BS, dtype=torch.int32, device=DEV)
            for ps in (1, 2, 4):
                us = timeit(lambda: ops.attn_decode(qkv, kc, vc, t2, ctx_lens, Hq, Hkv, Dh, scale, nsplit=1,
                                                    groups=groups, kv_begin=kvb, max_group_rows=16, prefix_splits=ps))
                row = {"op": "attn_decode_shared_prefix", "B": B, "ctx": ctx, "prefix_splits": ps, "us": round(us, 2)}
                results.append(row)
                print(row, flush=True)
    # prefill: 16 sequences x 512 tokens
    for nseq, L in ((16, 512), (4, 2048)):
        nb = (L + BS - 1) // BS
        kc = (torch.randn(nseq * nb, Hkv, BS, Dh, device=DEV) * 0.5).bfloat16()
        vc = (torch.randn(nseq * nb, Hkv, Dh, BS, device=DEV) * 0.5).bfloat16()
        tables = torch.arange(nseq * nb, dtype=torch.int32, device=DEV).view(nseq, nb)
        cu = torch.arange(nseq + 1, dtype=torch.int32, device=DEV) * L
        ctx_lens = torch.full((nseq,), L, dtype=torch.int32, device=DEV)
        qkv = (torch.randn(nseq * L, (Hq + 2 * Hkv) * Dh, device=DEV) * 0.5).bfloat16()
        us = timeit(lambda: ops.attn_prefill(qkv, kc, vc, tables, cu, ctx_lens, L, Hq, Hkv, Dh, scale))
        flops = 4 * nseq * Hq * Dh * L * L / 2
        row = {"op": "attn_prefill", "nseq": nseq, "L": L, "us": round(us, 2), "tflops": round(flops / us / 1e6, 1)}
        results.append(row)
        print(row, flush=True)


def bench_misc(results):
    T, d = 8192, 4096
    h = torch.randn(T, d, device=DEV).bfloat16()
    delta = torch.randn(T, d, device=DEV).bfloat16()
    w = torch.ones(d, device=DEV).bfloat16()
    us = timeit(lambda: ops.rmsnorm(h, w, 1e-5, delta))
    results.append({"op": "add_rmsnorm", "T": T, "d": d, "us": round(us, 2), "gbs": round(T * d * 2 * 4 / us / 1e3, 1)})
    print(results[-1], flush=True)
    V, R = 128256, 128
    logits = torch.randn(R, V, device=DEV)
    temp = torch.full((R,), 0.8, device=DEV)
    tp = torch.full((R,), 0.95, device=DEV)
    seeds = torch.arange(R, dtype=torch.int64, device=DEV)
    us = timeit(lambda: ops.sample(logits, temp, tp, None, seeds, None), iters=5)
    results.append({"op": "sample_top_p", "R": R, "V": V, "us": round(us, 2)})
    print(results[-1], flush=True)
    us = timeit(lambda: ops.sample(logits, temp * 0, None, None, seeds, None), iters=5)
    results.append({"op": "sample_greedy", "R": R, "V": V, "us": round(us, 2)})
    print(results[-1], flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "attn", "misc"]
    results = []
    if "gemm" in which:
        bench_gemm(results)
    if "attn" in which:
        bench_attention(results)
    if "misc" in which:
        bench_misc(results)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/kernel_bench.json", "w") as f:
        json.dump(results, f, indent=1)
