"""Where the sampler's time goes: greedy (one sweep), temperature only (sweep + draw), top-p (adds the
threshold search), top-k + top-p. 64 rows x 128256 logits."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llm_consensus_amd import ops
R, V = 64, 128256
dev = "cuda"
logits = torch.randn(R, V, device=dev) * 2.0
seeds = torch.arange(R, dtype=torch.int64, device=dev)
steps = torch.zeros(R, dtype=torch.int32, device=dev)
def t(temp, p, k):
    T = torch.full((R,), temp, device=dev); P = torch.full((R,), p, device=dev)
    K = torch.full((R,), k, dtype=torch.int32, device=dev)
    for _ in range(3): ops.sample(logits, T, P, K, seeds, steps)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ops.sample(logits, T, P, K, seeds, steps)
    e.record(); torch.cuda.synchronize()
    return round(s.elapsed_time(e) * 1e3 / 20, 1)
print("greedy", t(0.0, 1.0, 0), "us")
print("temperature only", t(0.8, 1.0, 0), "us")
print("top-p 0.95", t(0.8, 0.95, 0), "us")
print("top-k 50 + top-p 0.95", t(0.8, 0.95, 50), "us")
