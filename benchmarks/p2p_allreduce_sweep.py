"""Message-size sweep of the peer-to-peer all-reduce (csrc/kernels/p2p.hip): one-shot against
two-shot, plain and fused with residual + RMSNorm, for W ranks that are W PROCESSES ON ONE GPU
(the only multi-rank form a one-GPU machine offers: same kernels, same flags and IPC mappings,
the peers' regions in the same HBM instead of behind xGMI, and the ranks compete for the same
CUs, so the numbers are latency floors and an upper bound on per-byte cost, not link rates).

    python benchmarks/p2p_allreduce_sweep.py W out.json          (launcher: starts W ranks)
"""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
D = 8192


def worker(out_path: str) -> None:
    import torch
    import torch.distributed as dist

    from llm_consensus_amd.parallel import p2p
    from llm_consensus_amd.parallel.comm import TorchDistComm

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")
    rank, W = dist.get_rank(), dist.get_world_size()
    base = TorchDistComm(dist, torch.device("cpu"))
    comm = p2p.P2PComm(base, dev, stage_bytes=64 << 20, timeout_s=20, grid_cap=32)
    log = open(f"{out_path}.rank{rank}.log", "w")
    w = torch.ones(D, dtype=torch.bfloat16, device=dev)
    rows = []
    for T in (1, 8, 64, 256, 1024, 4096):
        x = (torch.randn(T, D, device=dev) * 0.1).bfloat16()
        h = torch.zeros(T, D, dtype=torch.bfloat16, device=dev)
        nbytes = T * D * 2
        for two_shot in (False, True):
            p2p.TWO_SHOT_BYTES = 0 if two_shot else 1 << 60
            for fused in (False, True):
                def call():
                    if fused:
                        comm.allreduce_rmsnorm(h, x, w, 1e-5)
                    else:
                        comm.all_reduce(x)
                iters = 30 if T <= 256 else 8
                for _ in range(3):
                    call()
                torch.cuda.synchronize()
                base.barrier()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(iters):
                    call()
                e.record()
                torch.cuda.synchronize()
                comm.check()
                us = s.elapsed_time(e) * 1e3 / iters
                rows.append({"bytes_per_rank": nbytes, "two_shot": two_shot, "fused_residual_rmsnorm": fused,
                             "us": round(us, 1)})
                log.write(f"{time.strftime('%H:%M:%S')} T={T} two_shot={two_shot} fused={fused} {us:.1f} us\n")
                log.flush()
                x.mul_(0.5)       # keep the values finite over repeated in-place sums
    base.barrier()
    if rank == 0:
        Path(out_path).write_text(json.dumps({"world": W, "d": D, "ranks_share_one_gpu": True, "rows": rows}, indent=1))
        for r in rows:
            print(r, flush=True)
    comm.close()
    dist.destroy_process_group()


def main() -> int:
    if os.environ.get("LC_SWEEP_WORKER") == "1":
        worker(sys.argv[2])
        return 0
    W, out = int(sys.argv[1]), sys.argv[2]
    port = 20000 + os.getpid() % 20000
    procs = []
    for r in range(W):                          # children start before this process touches a GPU
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(W), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LC_SWEEP_WORKER="1")
        procs.append(subprocess.Popen([sys.executable, "-u", __file__, str(W), out], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        try:
            rc = p.wait(timeout=240) or rc
        except subprocess.TimeoutExpired:
            p.kill()
            rc = 124
    return rc


if __name__ == "__main__":
    sys.exit(main())
