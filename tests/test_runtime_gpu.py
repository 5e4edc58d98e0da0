"""Native device runtime (csrc/runtime/device_runtime.hip through llm_consensus_amd/runtime.py) on a
real GPU: arenas, pinned memory with asynchronous copies, streams and events, and the graph cache."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def test_arena_hands_out_aligned_disjoint_memory_and_reports_exhaustion():
    from llm_consensus_amd import runtime as rt

    a = rt.Arena(1 << 20, device=DEV)
    x = a.tensor((1000,), torch.float32)
    y = a.tensor((3, 5), torch.bfloat16)
    z = a.tensor((7,), torch.int64)
    for t in (x, y, z):
        assert t.is_cuda and t.data_ptr() % 256 == 0
    spans = sorted((t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()) for t in (x, y, z))
    assert all(a1 <= b0 for (_, a1), (b0, _) in zip(spans, spans[1:]))
    x.fill_(3.0)
    y.fill_(1.5)
    z.copy_(torch.arange(7, device=DEV))
    torch.cuda.synchronize()
    assert float(x.sum()) == 3000.0 and float(y.float().sum()) == 22.5 and z.tolist() == list(range(7))
    st = a.stats()
    assert st["capacity"] == 1 << 20 and st["allocations"] == 3 and st["used"] >= 4000 + 30 + 56
    m = a.mark()
    a.tensor((1024,), torch.float32)
    a.release(m)
    assert a.stats()["used"] == m
    with pytest.raises(rt.DeviceRuntimeError):
        a.tensor((1 << 20,), torch.float32)
    ch = rt.ChunkedArena(1 << 16, device=DEV)
    ts = [ch.tensor((10000,), torch.float32) for _ in range(5)]          # 40 KB each: two chunks and more
    big = ch.tensor((1 << 18,), torch.uint8)                             # larger than a chunk: its own
    for i, t in enumerate(ts):
        t.fill_(float(i))
    big.fill_(7)
    torch.cuda.synchronize()
    assert [float(t[0]) for t in ts] == [0.0, 1.0, 2.0, 3.0, 4.0] and int(big[-1]) == 7
    assert ch.stats()["chunks"] >= 4


def test_pinned_arena_copies_are_asynchronous_and_ordered_by_events():
    from llm_consensus_amd import runtime as rt

    pin = rt.ChunkedArena(1 << 20, pinned=True)
    dev = rt.ChunkedArena(1 << 20, device=DEV)
    h_in = pin.array((4096,), np.int32)
    h_out = pin.array((4096,), np.int32)
    d = dev.tensor((4096,), torch.int32)
    s = rt.Stream(DEV)
    h_in[:] = np.arange(4096, dtype=np.int32)
    h_out[:] = -1
    with torch.cuda.stream(s.torch):
        rt.copy_async(d.data_ptr(), h_in.ctypes.data, h_in.nbytes, rt.H2D)
        d.mul_(2)                                   # a torch kernel on the runtime's stream
        rt.copy_async(h_out.ctypes.data, d.data_ptr(), h_out.nbytes, rt.D2H)
        done = rt.Event().record()
    done.synchronize()
    assert done.query() and (h_out == 2 * np.arange(4096)).all()
    t0, t1 = rt.Event(timing=True), rt.Event(timing=True)
    with torch.cuda.stream(s.torch):
        t0.record()
        for _ in range(20):
            d.add_(1)
        t1.record()
    s.synchronize()
    assert t0.elapsed_ms(t1) > 0.0
    free, total = rt.mem_info()
    assert 0 < free <= total


def test_graph_cache_records_and_replays_with_allocations_inside():
    from llm_consensus_amd import ops, runtime as rt

    s = rt.Stream(DEV)
    gc = rt.GraphCache(DEV)
    x = torch.zeros(64, 512, dtype=torch.bfloat16, device=DEV)
    w = torch.ones(512, dtype=torch.bfloat16, device=DEV)
    out = {}

    def body():
        y = ops.rmsnorm(x, w, 1e-5)                 # own kernel, allocates its output
        z = (y.float() * 2.0 + 1.0)                 # torch ops, allocate temporaries
        return z.sum(dim=1)

    with torch.cuda.stream(s.torch):
        x.copy_(torch.randn(64, 512, device=DEV))
        body()                                      # warm-up outside the capture
        with gc.capture(11):
            out["r"] = body()
        assert len(gc) == 1 and gc.nodes[11] >= 3
        for seed in (1, 2, 3):
            x.copy_(torch.randn(64, 512, device=DEV, generator=torch.Generator(device=DEV).manual_seed(seed)))
            gc.launch(11)
            s.synchronize()
            want = body()
            s.synchronize()
            assert torch.allclose(out["r"], want, atol=1e-3)
        # memory the graph writes must not be handed to later allocations
        junk = [torch.full((64, 512), 9.0, device=DEV) for _ in range(8)]
        gc.launch(11)
        s.synchronize()
        assert torch.allclose(out["r"], want, atol=1e-3) and all(float(j[0, 0]) == 9.0 for j in junk)
        # a failing body leaves no capture open
        with pytest.raises(ValueError):
            with gc.capture(12):
                raise ValueError("boom")
        torch.ones(4, device=DEV).sum().item()      # the stream is usable again
        assert gc.launches == 4 and gc.erase(11) and len(gc) == 0
        with pytest.raises(rt.DeviceRuntimeError):
            gc.launch(11)
