"""RoPE + KV write at a prefill-sized step (8192 tokens, Llama-3-8B heads): us per call."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llm_consensus_amd import ops
from llm_consensus_amd.ops import reference as ref
T, Hq, Hkv, Dh, BS = 8192, 32, 8, 128, 64
dev = "cuda"
cos, sin = ref.rope_tables(Dh, 4096, 5e5)
cos, sin = cos.to(dev), sin.to(dev)
nb = T // BS
pos = (torch.arange(T, dtype=torch.int32) % 512).to(dev)
slots = torch.arange(T, dtype=torch.int32).to(dev)
layers = 8
qkvs = [(torch.randn(T, (Hq + 2 * Hkv) * Dh, device=dev) * 0.5).bfloat16() for _ in range(layers)]
caches = [(torch.zeros(nb, Hkv, BS, Dh, dtype=torch.bfloat16, device=dev), torch.zeros(nb, Hkv, Dh, BS, dtype=torch.bfloat16, device=dev)) for _ in range(layers)]
def run():
    for q, (k, v) in zip(qkvs, caches):
        ops.rope_kv_write(q, pos, slots, cos, sin, k, v, Hq, Hkv, Dh)
run(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5): run()
e.record(); torch.cuda.synchronize()
print("rope_kv_write T=8192:", round(s.elapsed_time(e) * 1e3 / (5 * layers), 1), "us per call")
