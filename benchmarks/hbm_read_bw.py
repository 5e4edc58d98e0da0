"""Read-bandwidth ceiling of the device as seen by simple kernels (yardstick for the
weight-streaming GEMMs): torch reductions and copies over buffers far larger than the
last-level cache."""
import torch

x = torch.empty(4 * 1024 ** 3 // 2, dtype=torch.bfloat16, device="cuda").normal_()
y = torch.empty_like(x)


def t(fn, n=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


gb = x.numel() * 2 / 1e12
print("sum      %.2f TB/s" % (gb / t(lambda: x.sum())))
print("max      %.2f TB/s" % (gb / t(lambda: x.max())))
print("copy     %.2f TB/s (read+write)" % (2 * gb / t(lambda: y.copy_(x))))
xf = x.view(torch.float32)
print("sum f32  %.2f TB/s" % (gb / t(lambda: xf.sum())))
