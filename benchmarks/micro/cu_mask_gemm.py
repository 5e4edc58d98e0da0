"""Prefill GEMM (8192 x 28672 x 4096, fused SwiGLU) on a stream restricted to a subset of the CUs:
how much does the MFMA-bound kernel lose when a quarter / half of the CUs are left to other work?"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llm_consensus_amd import ops

hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(keep):
    """keep(bit) -> bool for each of 256 CU bits."""
    words = (ctypes.c_uint32 * 8)()
    for b in range(256):
        if keep(b):
            words[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)

torch.cuda.init(); torch.zeros(1, device="cuda")
M, N, K = 8192, 28672, 4096
lws = [ops.LinearWeight.from_dense((torch.randn(N, K, device="cuda") * 0.05).bfloat16(), None, "cuda", swiglu=True) for _ in range(4)]
X = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
def timed(stream):
    with torch.cuda.stream(stream):
        for lw in lws: ops.linear(X, lw)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(stream)
        for _ in range(3):
            for lw in lws: ops.linear(X, lw)
        e.record(stream); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / 12
which = sys.argv[1] if len(sys.argv) > 1 else "q"
print("all CUs            ", round(timed(torch.cuda.Stream()), 1), "us", flush=True)
if which == "q":
    print("3/4 (bit//8 %4 !=3)", round(timed(masked_stream(lambda b: (b // 8) % 4 != 3)), 1), "us", flush=True)
elif which == "i":
    print("3/4 (bit %4 != 3)  ", round(timed(masked_stream(lambda b: b % 4 != 3)), 1), "us", flush=True)
else:
    print("1/2 (bit//8 %2 ==0)", round(timed(masked_stream(lambda b: (b // 8) % 2 == 0)), 1), "us", flush=True)
