"""Decode step of one rank of a tensor-parallel model with REAL group collectives inside ONE
captured hipGraph, on a single GPU.

W processes share the GPU. Rank 0 holds a full 1/W shard of the model (default Llama-3-70B at
TP=4: 35 GB) and runs the production path: Engine + P2PComm, every decode step one graph whose
2L+1 reductions are the fused all-reduce + residual + RMSNorm kernel of csrc/kernels/p2p.hip.
The other ranks are hollow: same engine, same requests, same sequence of collective calls with
the same shapes, but no GEMMs and no attention, so they take the collectives' peer role without
competing for HBM bandwidth. What is measured is therefore one rank's step with the full
protocol cost (flags, staging, peer reads) and HBM in place of xGMI.

  python benchmarks/tp_group_decode_bench.py [preset] [tp] [batch] [ctx] [out.json]
"""
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def worker():
    import torch
    import torch.distributed as dist

    from llm_consensus_amd._native import core
    from llm_consensus_amd.engine.engine import Engine
    from llm_consensus_amd.models.config import get_preset
    from llm_consensus_amd.models.transformer import LayerWeights, TransformerModel
    from llm_consensus_amd.parallel import p2p
    from llm_consensus_amd.parallel.comm import TorchDistComm

    preset, tp, B, ctx, out_path = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    new = 24
    log = open(f"{out_path}.rank{os.environ['RANK']}.log", "w")

    def progress(msg):
        log.write(f"{time.strftime('%H:%M:%S')} {msg}\n")
        log.flush()

    progress("start")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    base = TorchDistComm(dist, torch.device("cpu"))
    progress("process group up")
    comm = p2p.P2PComm(base, dev, timeout_s=60)
    progress("exchange regions mapped")
    # Hollow peers publish and leave (no waiting blocks that would take CUs, registers and LDS
    # from the rank under test, which shares this GPU with them). All ranks use the one-shot
    # form: a peer that does not consume cannot take part in a reduce-scatter.
    p2p.TWO_SHOT_BYTES = 1 << 60
    comm.publish_only = rank != 0
    cfg = get_preset(preset)

    class Hollow(TransformerModel):
        """Peer role only: the collectives of a real rank's step, nothing else."""

        def __init__(self):
            super().__init__(cfg, dev, tp_rank=rank, tp_size=tp, comm=comm)
            ones = torch.ones(cfg.hidden_size, dtype=torch.bfloat16, device=dev)
            self.final_norm = ones
            self.layers = [LayerWeights(ones, ones, None, None) for _ in range(cfg.num_layers)]
            self.vshard = (cfg.vocab_size + tp - 1) // tp

        def forward(self, token_ids, positions, slot_mapping, meta, kv):
            T = token_ids.numel()
            h = torch.zeros(T, cfg.hidden_size, dtype=torch.bfloat16, device=dev)
            z = torch.zeros_like(h)
            for L in self.layers:
                self._reduce_norm(h, z, L.ln2)
                self._reduce_norm(h, z, L.ln1)
            return h

        def logits(self, rows):
            lg = torch.zeros(rows.shape[0], self.vshard, dtype=torch.float32, device=dev)
            return self.comm.all_gather_cols(lg, cfg.vocab_size)

    # A real forward makes 2L reductions (after O and after down of every layer; the last one
    # is fused into the final norm), the hollow one above makes the same 2L.
    if rank == 0:
        model = TransformerModel.random_init(cfg, dev, seed=1, tp_rank=0, tp_size=tp, comm=comm)
    else:
        model = Hollow()
    eng = Engine(model, num_kv_blocks=B * ((ctx + new) // 64 + 2) + 16, block_size=64, max_batch_seqs=max(64, B),
                 max_model_len=ctx + new + 64, name=f"rank{rank}", mix_decode=False)
    assert eng.use_graphs
    progress("model and engine ready")
    base.barrier()            # the shard takes far longer to build than a hollow peer
    progress("all ranks ready")
    p = core.SamplingParams()
    p.temperature, p.top_p, p.max_new_tokens, p.ignore_eos = 0.8, 0.95, new, True
    g = torch.Generator().manual_seed(0)
    res = {}
    for rep in range(2):                       # the first pass captures, the second is timed
        for b in range(B):
            p.seed = b
            eng.submit(torch.randint(5, 1000, (ctx,), generator=g).tolist(), params=p)
        t_dec = n_dec = 0
        while eng.has_work():
            is_decode = not eng.sched.num_waiting and eng.metrics.counter("prefill.tokens") >= (rep + 1) * B * ctx
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.step()
            torch.cuda.synchronize()
            progress(f"rep {rep} step done decode={is_decode} {1e3 * (time.perf_counter() - t0):.2f} ms")
            if is_decode:
                t_dec += time.perf_counter() - t0
                n_dec += 1
        res = {"decode_ms_per_step": round(t_dec / max(1, n_dec) * 1e3, 3), "decode_steps": n_dec}
    comm.check()
    base.barrier()
    if rank == 0:
        out = {"model": preset, "tp": tp, "batch": B, "ctx": ctx, "weight_GB_per_rank": round(model.weight_bytes() / 1e9, 1),
               "mode": "one hipGraph per decode step, fused all-reduce + residual + RMSNorm over IPC-mapped peer regions",
               "collective_calls": comm.calls, **res}
        print(json.dumps(out), flush=True)
        os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
        json.dump(out, open(out_path, "w"), indent=1)
    comm.close()
    dist.destroy_process_group()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        return worker()
    preset = sys.argv[1] if len(sys.argv) > 1 else "llama-3-70b"
    tp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    B = sys.argv[3] if len(sys.argv) > 3 else "64"
    ctx = sys.argv[4] if len(sys.argv) > 4 else "512"
    out = sys.argv[5] if len(sys.argv) > 5 else "gpurun_out/tp_group_decode_bench.json"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(tp):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(tp), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", preset, str(tp), B, ctx, out],
                                      env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        try:
            rc = p.wait(timeout=400) or rc
        except subprocess.TimeoutExpired:
            p.kill()
            rc = 124
    sys.exit(rc)


if __name__ == "__main__":
    main()
