"""Runs the prefill GEMM shapes through torch.matmul so that a kernel trace shows which hipBLASLt
kernels (macro tile, MFMA shape, prefetch settings are encoded in the name) serve them."""
import torch
for N, K in ((6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)):
    X = (torch.randn(8192, K, device="cuda") * 0.5).bfloat16()
    W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    for _ in range(3):
        Y = X @ W.t()
    torch.cuda.synchronize()
