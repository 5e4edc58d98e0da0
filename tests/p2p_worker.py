"""One rank of tests/test_p2p_gpu.py: W processes on ONE GPU exchange HIP IPC handles and run
the peer-to-peer collectives of csrc/kernels/p2p.hip against fp32 references (same code path as
W GPUs; the peers' regions are simply in the same HBM). Usage: p2p_worker.py <out_prefix>;
RANK / WORLD_SIZE / MASTER_* come from the environment. Writes <out_prefix>.<rank> on success."""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from llm_consensus_amd import ops  # noqa: E402
from llm_consensus_amd._native import core  # noqa: E402
from llm_consensus_amd.engine.engine import Engine  # noqa: E402
from llm_consensus_amd.models.config import get_preset  # noqa: E402
from llm_consensus_amd.models.transformer import TransformerModel  # noqa: E402
from llm_consensus_amd.parallel import p2p  # noqa: E402
from llm_consensus_amd.parallel.comm import TorchDistComm  # noqa: E402

DEV = torch.device("cuda", 0)
_LOG = None


def progress(msg: str) -> None:
    """Progress lines go to LC_P2P_LOG_DIR/p2p_worker.<rank>.log when set (a run that hangs
    still leaves a trace of how far every rank came)."""
    global _LOG
    d = os.environ.get("LC_P2P_LOG_DIR")
    if not d:
        return
    if _LOG is None:
        Path(d).mkdir(parents=True, exist_ok=True)
        _LOG = open(Path(d) / f"p2p_worker.{os.environ.get('RANK', '0')}.log", "a")
    _LOG.write(f"{time.strftime('%H:%M:%S')} {msg}\n")
    _LOG.flush()


def bf(x):
    return x.to(torch.bfloat16)


def rand(shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def reference(parts_bf16, h, w, eps):
    """parts_bf16: every rank's staged (bf16) partial. Returns (h_new bf16, y fp32)."""
    total = torch.zeros_like(parts_bf16[0], dtype=torch.float32)
    for p in parts_bf16:                       # rank order, fp32
        total = total + p.float()
    if h is None:
        return None, bf(total).float()
    hn = bf(h.float() + bf(total).float())
    x = hn.float()
    y = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w.float()
    return hn, y


def rel_err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def same_on_all_ranks(base, t):
    digest = bytes(t.cpu().contiguous().view(torch.uint8).numpy().tobytes())
    import hashlib

    hs = base.all_gather_object(hashlib.sha1(digest).hexdigest())
    assert len(set(hs)) == 1, "ranks hold different results"


def check_allreduce(comm, base, W, rank, T, d, S, tag):
    eps = 1e-5
    w = bf(rand((d,), 7, 0.5) + 1.0)
    h0 = bf(rand((T, d), 11))
    if S:
        slabs = [rand((S, T, d), 100 + r, 0.5) for r in range(W)]
        staged = [bf(s.sum(0)) if S == 1 else bf(sum(s[i] for i in range(1, S)) + 0 * s[0] + s[0]) for s in slabs]
        # the kernel sums the slabs in order s = 0..S-1 in fp32
        staged = []
        for s in slabs:
            acc = torch.zeros(T, d)
            for i in range(S):
                acc = acc + s[i]
            staged.append(bf(acc))
        mine = ops.PartialSum(slabs[rank].to(DEV).contiguous())
    else:
        xs = [bf(rand((T, d), 200 + r, 0.5)) for r in range(W)]
        staged = xs
        mine = xs[rank].to(DEV)
    hn, y_ref = reference(staged, h0, w, eps)
    h = h0.to(DEV).clone()
    y = comm.allreduce_rmsnorm(h, mine, w.to(DEV), eps)
    torch.cuda.synchronize()
    comm.check()
    assert torch.equal(h.cpu(), hn), f"{tag}: residual stream differs (max {float((h.cpu().float() - hn.float()).abs().max())})"
    assert rel_err(y, y_ref) < 4e-3, (tag, rel_err(y, y_ref))
    same_on_all_ranks(base, y)


def main() -> None:
    out_prefix = sys.argv[1]
    progress("start")
    torch.cuda.set_device(DEV)
    dist.init_process_group("gloo")
    progress("gloo up")
    rank, W = dist.get_rank(), dist.get_world_size()
    base = TorchDistComm(dist, torch.device("cpu"))
    comm = p2p.P2PComm(base, DEV, stage_bytes=16 << 20, timeout_s=5, grid_cap=32)   # the ranks share one GPU
    report = {"world": W}
    progress("regions mapped")

    # ---- fused all-reduce + residual + RMSNorm, one-shot and two-shot, slabs and bf16 inputs
    shapes = [(1, 8192, 0), (64, 8192, 3), (7, 512, 2), (300, 4096, 0), (33, 128, 1), (5, 16384, 0)]
    for two_shot in (False, True):
        p2p.TWO_SHOT_BYTES = 0 if two_shot else 1 << 40
        for T, d, S in shapes:
            check_allreduce(comm, base, W, rank, T, d, S, f"two_shot={two_shot} T={T} d={d} S={S}")
            progress(f"allreduce two_shot={two_shot} T={T} d={d} S={S} ok")
    p2p.TWO_SHOT_BYTES = 1 << 20

    progress("plain all-reduce")
    # ---- plain all-reduce
    xs = [bf(rand((40, 1024), 300 + r)) for r in range(W)]
    t = xs[rank].to(DEV).clone()
    comm.all_reduce(t)
    _, want = reference(xs, None, None, 0.0)
    assert torch.equal(t.cpu().float(), want)

    progress("byte exchanges")
    # ---- byte exchanges
    g = comm.all_gather_rows(torch.full((3, 8), float(rank), dtype=torch.float32, device=DEV))
    assert g.shape == (3 * W, 8) and g[:, 0].cpu().tolist() == [float(r) for r in range(W) for _ in range(3)]
    c = comm.all_gather_cols(torch.arange(2 * 4, dtype=torch.float32, device=DEV).view(2, 4) + 100 * rank, total_cols=4 * W - 1)
    assert c.shape == (2, 4 * W - 1) and c[1].cpu().tolist() == [100.0 * (j // 4) + 4 + j % 4 for j in range(4 * W - 1)]
    send = (torch.arange(W * 2 * 8, dtype=torch.bfloat16, device=DEV).view(W * 2, 8) % 64) + 64 * rank
    got = comm.all_to_all_equal(send)
    for s in range(W):
        want_blk = ((torch.arange(W * 2 * 8).view(W * 2, 8) % 64) + 64 * s)[rank * 2:(rank + 1) * 2]
        assert torch.equal(got[s * 2:(s + 1) * 2].cpu().float(), want_blk.float()), (s, got)
    odd = comm.all_gather_rows(torch.full((1, 3), float(rank), device="cpu"))     # not a device tensor: bootstrap path
    assert odd.shape == (W, 3)
    tiny = comm.all_to_all_equal(torch.arange(W * 3, dtype=torch.int32, device=DEV) + 100 * rank)   # 12-byte blocks: padded
    assert tiny.cpu().tolist() == [100 * s + 3 * rank + j for s in range(W) for j in range(3)]
    one = comm.all_gather_rows(torch.tensor([[rank]], dtype=torch.int64, device=DEV))
    assert one.view(-1).cpu().tolist() == list(range(W))

    # ---- counted all-to-all: only the rows that exist move, the counts travel on the device
    cap, per, dd = 6, 2, 64
    rows = torch.zeros(W * cap, dd, dtype=torch.bfloat16)
    counts = torch.zeros(W * per, dtype=torch.int32)
    for t in range(W):
        n_t = (rank + 2 * t) % (cap + 1)                       # rows this rank sends to rank t
        counts[t * per], counts[t * per + 1] = n_t // 2, n_t - n_t // 2
        rows[t * cap: t * cap + n_t] = (torch.arange(n_t).view(-1, 1) + 10 * t + 100 * rank).to(torch.bfloat16)
    sentinel = torch.full((W * cap, dd), -7.0, dtype=torch.bfloat16, device=DEV)
    got, got_counts = comm.all_to_all_counted(rows.to(DEV), counts.to(DEV), per)
    assert got.shape == (W * cap, dd) and got_counts.shape == (W * per,)
    for s_ in range(W):
        n_s = (s_ + 2 * rank) % (cap + 1)                      # rows rank s_ sent here
        assert got_counts[s_ * per: (s_ + 1) * per].cpu().tolist() == [n_s // 2, n_s - n_s // 2]
        want_rows = (torch.arange(n_s).view(-1, 1) + 10 * rank + 100 * s_).to(torch.bfloat16).expand(n_s, dd)
        assert torch.equal(got[s_ * cap: s_ * cap + n_s].cpu(), want_rows), (s_, n_s)
    del sentinel
    odd7 = comm.all_gather_rows(torch.arange(7, dtype=torch.uint8, device=DEV).view(1, 7) + rank)   # 7-byte blocks
    assert odd7.cpu().tolist() == [[(j + s_) for j in range(7)] for s_ in range(W)]

    progress("messages larger")
    # ---- messages larger than the staging slot are cut into row ranges
    small = p2p.P2PComm(base, DEV, stage_bytes=64 << 10, timeout_s=5, grid_cap=32)
    check_allreduce(small, base, W, rank, 100, 1024, 0, "chunked bf16")
    check_allreduce(small, base, W, rank, 100, 1024, 2, "chunked slabs")
    small.close()

    progress("inside a captured graph")
    # ---- inside a captured graph, replayed with new inputs
    T, d = 64, 4096
    w = bf(rand((d,), 8, 0.3) + 1.0).to(DEV)
    x_static = torch.zeros(T, d, dtype=torch.bfloat16, device=DEV)
    h_static = torch.zeros(T, d, dtype=torch.bfloat16, device=DEV)

    def body():
        ys = []
        for _ in range(3):
            ys.append(comm.allreduce_rmsnorm(h_static, x_static, w, 1e-5))
        return ys

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        outs = body()
    for it in range(4):
        xs = [bf(rand((T, d), 1000 + 10 * it + r, 0.5)) for r in range(W)]
        h0 = bf(rand((T, d), 2000 + it))
        x_static.copy_(xs[rank].to(DEV))
        h_static.copy_(h0.to(DEV))
        graph.replay()
        torch.cuda.synchronize()
        hn = h0
        for k in range(3):
            hn, y_ref = reference(xs, hn, w.cpu(), 1e-5)
            assert rel_err(outs[k], y_ref) < 4e-3, (it, k)
        assert torch.equal(h_static.cpu(), hn)
    comm.check()

    progress("latency of the fused call")
    # ---- latency of the fused call from a graph (all ranks share one GPU here, so this is the
    # protocol's cost with HBM in place of xGMI)
    for T in (1, 64):
        x1 = torch.zeros(T, 8192, dtype=torch.bfloat16, device=DEV)
        h1 = torch.zeros(T, 8192, dtype=torch.bfloat16, device=DEV)
        w1 = torch.ones(8192, dtype=torch.bfloat16, device=DEV)
        for _ in range(3):
            comm.allreduce_rmsnorm(h1, x1, w1, 1e-5)
        torch.cuda.synchronize()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, capture_error_mode="thread_local"):
            for _ in range(50):
                comm.allreduce_rmsnorm(h1, x1, w1, 1e-5)
        base.barrier()
        g2.replay()
        torch.cuda.synchronize()
        base.barrier()
        t0 = time.perf_counter()
        for _ in range(4):
            g2.replay()
        torch.cuda.synchronize()
        report[f"fused_allreduce_us_T{T}_d8192"] = round((time.perf_counter() - t0) / 200 * 1e6, 2)
        progress(f"fused all-reduce, T={T} d=8192, {W} ranks on one GPU: {report[f'fused_allreduce_us_T{T}_d8192']} us per call")
    comm.check()

    progress("a tensor-parallel model")
    # ---- a tensor-parallel model whose decode step is ONE captured graph
    cfg = get_preset("small-llama")
    if cfg.num_kv_heads % W == 0:
        prompt, targets = list(range(3, 90)), [1, 65, 66, 300]
        sp = core.SamplingParams()
        sp.temperature, sp.max_new_tokens, sp.ignore_eos = 0.0, 6, True

        def run(model, graphs):
            eng = Engine(model, num_kv_blocks=64, block_size=64, use_graphs=graphs)
            r1 = eng.submit(prompt, core.RequestKind.Score, score_ids=targets)
            r2 = eng.submit(prompt[:40], core.RequestKind.Generate, 1, sp)
            eng.run_until_done()
            return np.array(eng.pop_result(r1).scores), eng.pop_result(r2).token_ids[0], eng

        want_scores, want_toks, _ = run(TransformerModel.random_init(cfg, DEV, seed=6, std=0.05), True)
        tp = TransformerModel.random_init(cfg, DEV, seed=6, std=0.05, tp_rank=rank, tp_size=W, comm=comm)
        got_scores, got_toks, eng = run(tp, None)
        assert eng.use_graphs and any(g.graph for g in eng._graphs.values()), "TP decode was not captured"
        assert np.allclose(got_scores, want_scores, atol=0.06), (got_scores, want_scores)
        toks = base.all_gather_object(list(got_toks))
        assert all(t == toks[0] for t in toks), "TP ranks sampled different tokens"
        report["tp_tokens_match_unsharded"] = int(sum(a == b for a, b in zip(got_toks, want_toks)))
        # a step that only scores exchanges target log-probs, not logits rows
        r = eng.submit(prompt, core.RequestKind.Score, score_ids=targets)
        eng.run_until_done()
        assert eng.metrics.counter("prefill.target_only_score_rows") == 1
        only = np.array(eng.pop_result(r).scores)
        assert np.allclose(only, want_scores, atol=0.06), (only, want_scores)
        # eager steps of the sharded model (prefill + score exchange, prefill + sample, decode with
        # the fused all-reduce, the logits all-gather): PyTorch launches no kernel of its own
        from llm_consensus_amd.utils.audit import AtenAudit

        eager = Engine(tp, num_kv_blocks=64, block_size=64, use_graphs=False)
        for _ in range(2):                      # the second pass is the audited one (lazy initialisation is over)
            with AtenAudit() as audit:
                r1 = eager.submit(prompt, core.RequestKind.Score, score_ids=targets)
                r2 = eager.submit(prompt[:40], core.RequestKind.Generate, 1, sp)
                eager.run_until_done()
                torch.cuda.synchronize()
            eager.pop_result(r1), eager.pop_result(r2)
        assert audit.kernels == {}, f"TP steps launched PyTorch kernels: {audit.kernels}"
        # the bootstrap communicator alone cannot be captured: asking for graphs is an error, not a crash
        plain = TransformerModel.random_init(cfg, DEV, seed=6, std=0.05, tp_rank=rank, tp_size=W, comm=base)
        try:
            Engine(plain, num_kv_blocks=64, block_size=64, use_graphs=True)
            raise AssertionError("use_graphs=True with a library communicator must be refused")
        except ValueError:
            pass
        comm.check()

    progress("expert parallel")
    # ---- expert parallel (all-to-all dispatch) under a captured graph
    mcfg = get_preset("tiny-mixtral")
    if mcfg.num_experts % W == 0:
        prompt, targets = list(range(3, 60)), [1, 65, 66, 300]
        sp = core.SamplingParams()
        sp.temperature, sp.max_new_tokens, sp.ignore_eos = 0.0, 5, True

        def run_moe(model):
            eng = Engine(model, num_kv_blocks=64, block_size=64)
            r1 = eng.submit(prompt, core.RequestKind.Score, score_ids=targets)
            r2 = eng.submit(prompt[:30], core.RequestKind.Generate, 1, sp)
            eng.run_until_done()
            return np.array(eng.pop_result(r1).scores), eng.pop_result(r2).token_ids[0], eng

        want_scores, _, _ = run_moe(TransformerModel.random_init(mcfg, DEV, seed=5, std=0.08))
        for mode in ("a2a", "allreduce"):
            m = TransformerModel.random_init(mcfg, DEV, seed=5, std=0.08, comm=comm, ep_size=W, ep_mode=mode)
            got_scores, got_toks, eng = run_moe(m)
            assert eng.use_graphs and any(g.graph for g in eng._graphs.values()), mode
            assert np.allclose(got_scores, want_scores, atol=0.06), (mode, got_scores, want_scores)
            toks = base.all_gather_object(list(got_toks))
            assert all(t == toks[0] for t in toks), mode
            from llm_consensus_amd.utils.audit import AtenAudit

            eager = Engine(m, num_kv_blocks=64, block_size=64, use_graphs=False)
            for _ in range(2):
                with AtenAudit() as audit:
                    r1 = eager.submit(prompt, core.RequestKind.Score, score_ids=targets)
                    r2 = eager.submit(prompt[:30], core.RequestKind.Generate, 1, sp)
                    eager.run_until_done()
                    torch.cuda.synchronize()
                eager.pop_result(r1), eager.pop_result(r2)
            assert audit.kernels == {}, f"EP {mode} steps launched PyTorch kernels: {audit.kernels}"
        comm.check()
        # a slice longer than one exchange carries: several counted exchanges per layer, same result
        from llm_consensus_amd.parallel import ep as ep_mod

        ep_mod.EP_CHUNK_ROWS = 8
        m = TransformerModel.random_init(mcfg, DEV, seed=5, std=0.08, comm=comm, ep_size=W, ep_mode="a2a")
        got_scores, _, _ = run_moe(m)
        assert np.allclose(got_scores, want_scores, atol=0.06), ("chunked a2a", got_scores, want_scores)
        ep_mod.EP_CHUNK_ROWS = 1024
        comm.check()

    progress("all checks passed")
    base.barrier()
    comm.close()
    Path(f"{out_prefix}.{rank}").write_text(json.dumps(report))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
