"""Runs every hot kernel a few times at the benchmark's shapes, for hardware-counter passes:

    rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES ... --kernel-trace --output-format csv -d out -- python3 benchmarks/pmc_kernels.py

and, with ``--summarise DIR OUT.json``, folds the counter CSV of such a pass into one row per
kernel (mean over its dispatches). Short on purpose: counter collection serialises kernels.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch

    from llm_consensus_amd import ops
    from llm_consensus_amd.ops import reference as ref

    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)

    def rnd(*shape, scale=0.05):
        return (torch.randn(*shape, device=dev, generator=g) * scale).bfloat16()

    K = 4096
    w_gu = ops.LinearWeight.from_dense(rnd(28672, K), None, dev, swiglu=True)
    w_o = ops.LinearWeight.from_dense(rnd(4096, K), None, dev)
    w_dn = ops.LinearWeight.from_dense(rnd(4096, 14336), None, dev)
    x8k, x256, x64 = rnd(8192, K, scale=0.5), rnd(256, K, scale=0.5), rnd(64, K, scale=0.5)
    a64 = rnd(64, 14336, scale=0.5)
    a8k = rnd(8192, 14336, scale=0.5)
    for _ in range(3):
        ops.linear(a8k, w_dn)          # gemm256w (four waves, K >= 8192)
        ops.linear(x8k, w_gu)          # gemm256 (SwiGLU epilogue)
        ops.linear(x8k, w_o)           # gemm256
        ops.linear(x256, w_gu)         # gemm_panel, 256-row panel
        ops.linear(x64, w_gu)          # gemm_panel, 64-row panel (SwiGLU epilogue)
        ops.linear(x64, w_o, defer_reduce=True)
        ops.linear(a64, w_dn, defer_reduce=True)
    # ---- attention at the benchmark's shapes (Llama-3-8B: 32 q heads, 8 kv heads, Dh 128)
    Hq, Hkv, Dh, BS = 32, 8, 128, 64
    nblk = 64 * 10
    kc = rnd(nblk, Hkv, BS, Dh, scale=0.5)
    vt = rnd(nblk, Hkv, Dh, BS, scale=0.5)
    bt = torch.arange(nblk, dtype=torch.int32, device=dev).view(64, 10)
    ctx = torch.full((64,), 544, dtype=torch.int32, device=dev)
    qkv = rnd(64, (Hq + 2 * Hkv) * Dh, scale=0.5)
    for _ in range(3):
        ops.attn_decode(qkv, kc, vt, bt, ctx, Hq, Hkv, Dh, Dh ** -0.5, 0, 4)
    # ---- self-consistency step: 16 prompts x 16 chains, 512 shared prompt tokens + private suffix
    bt_sc = torch.empty(256, 10, dtype=torch.int32, device=dev)
    rows = torch.arange(256, device=dev)
    bt_sc[:, :8] = ((rows // 16) * 8).view(-1, 1).int() + torch.arange(8, device=dev, dtype=torch.int32)
    bt_sc[:, 8] = (128 + rows).int()
    bt_sc[:, 9] = 0
    ctx_sc = torch.full((256,), 544, dtype=torch.int32, device=dev)
    begin_sc = torch.full((256,), 512, dtype=torch.int32, device=dev)
    groups = (torch.arange(0, 256, 16, dtype=torch.int32, device=dev), torch.full((16,), 16, dtype=torch.int32, device=dev),
              torch.full((16,), 8, dtype=torch.int32, device=dev))
    qkv_sc = rnd(256, (Hq + 2 * Hkv) * Dh, scale=0.5)
    for _ in range(3):
        ops.attn_decode(qkv_sc, kc, vt, bt_sc, ctx_sc, Hq, Hkv, Dh, Dh ** -0.5, 0, 1, groups, begin_sc, 16, 4)
    T = 16 * 512
    qkv_p = rnd(T, (Hq + 2 * Hkv) * Dh, scale=0.5)
    cu = torch.arange(0, T + 1, 512, dtype=torch.int32, device=dev)
    ctx_p = torch.full((16,), 512, dtype=torch.int32, device=dev)
    for _ in range(2):
        ops.attn_prefill(qkv_p, kc, vt, bt[:16], cu, ctx_p, 512, Hq, Hkv, Dh, Dh ** -0.5, 0)
    # ---- row kernels
    cos, sin = ref.rope_tables(Dh, 1024, 5e5)
    cos, sin = cos.to(dev), sin.to(dev)
    pos = torch.full((64,), 543, dtype=torch.int32, device=dev)
    slots = (bt[:, 8] * BS + 31).to(torch.int32)
    wn = rnd(K, scale=1.0)
    h64, h8k = rnd(64, K, scale=1.0), rnd(8192, K, scale=1.0)
    logits = torch.randn(64, 128256, device=dev, generator=g)
    temp = torch.full((64,), 0.8, device=dev)
    top_p = torch.full((64,), 0.95, device=dev)
    top_k = torch.zeros(64, dtype=torch.int32, device=dev)
    seeds = torch.arange(64, dtype=torch.int64, device=dev)
    steps = torch.zeros(64, dtype=torch.int32, device=dev)
    for _ in range(3):
        ops.rope_kv_write(qkv.clone(), pos, slots, cos, sin, kc, vt, Hq, Hkv, Dh)
        ops.rmsnorm(h64, wn, 1e-5, x64)
        ops.rmsnorm(h8k, wn, 1e-5, x8k)
        ops.sample(logits, temp, top_p, top_k, seeds, steps)
    torch.cuda.synchronize()


def summarise(d, out):
    files = glob.glob(os.path.join(d, "*", "*counter_collection.csv")) + glob.glob(os.path.join(d, "*counter_collection.csv"))
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            m = re.search(r"(\w+_kernel\w*)(<[^>]*>)?", r["Kernel_Name"])
            if not m:
                continue
            acc[m.group(0)[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
    for k, cs in sorted(rows.items()):
        print(k, {c: round(v, 1) for c, v in cs.items()})
    prev = json.load(open(out)) if os.path.exists(out) else {}
    for k, cs in rows.items():
        prev.setdefault(k, {}).update(cs)
    json.dump(prev, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2], sys.argv[3])
    else:
        run()
