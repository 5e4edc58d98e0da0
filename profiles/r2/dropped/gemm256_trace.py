"""Phase timeline of the 256x256 GEMM main loop (LC_G2_TRACE build switch): cycle stamps of wave 0
(leading group) and wave 4 (lagging group) of workgroup 0 at the eight barrier points of K-tiles 8-11."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["LC_G2_TRACE"] = "1"
from llm_consensus_amd import ops  # noqa: E402
from llm_consensus_amd.ops import hip  # noqa: E402

M, N, K = 8192, 28672, 4096
W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
lw = ops.LinearWeight.from_dense(W, None, "cuda")
X = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
for _ in range(3):
    ops.linear(X, lw)
torch.cuda.synchronize()
out = np.zeros(64, dtype=np.uint32)
fn = hip.lib().lc_gemm256_trace_read
fn.argtypes = [ctypes.c_void_p]
fn(out.ctypes.data)
names = ["R1 end", "M1 start", "P2 start", "M2 start", "P3 start", "M3 start", "P4 start", "M4 start"]
for g, label in ((0, "lead wave 0"), (1, "lag wave 4")):
    st = out[g * 32:(g + 1) * 32].astype(np.int64)
    d = np.diff(st) & 0xffffffff
    print(label, "stamps relative:", (st - st[0]).tolist())
    print(label, "deltas:", d.tolist())
    per = d[:24].reshape(3, 8)
    print(label, "mean delta by position (after stamp k):", per.mean(0).round(0).tolist(), "sum", per.mean(0).sum())
