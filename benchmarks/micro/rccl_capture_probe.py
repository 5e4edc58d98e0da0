"""What happens when a torch.distributed (RCCL) all-reduce is recorded into a hipGraph on this
torch / ROCm pair? Each variant runs in a child process (an abort must not take the probe down):
world size 1, one bf16 all-reduce between two elementwise kernels, captured under one of torch's
capture error modes, replayed three times and checked. Variants: the three capture error modes;
the process group's watchdog thread disabled; ``cold`` = the communicator's first collective is
the captured one (no eager call before, so RCCL creates its channels inside the capture);
``subgroup`` = the collective runs on a dist.new_group communicator; ``slow`` = the capture stays
open for 2.5 s in global mode (time for the watchdog and heartbeat threads to make a HIP call
while it is open). The parent records exit status and the tail of stderr per variant:
    python benchmarks/micro/rccl_capture_probe.py [out.json]"""
import json
import os
import socket
import subprocess
import sys

CHILD = r'''
import os, sys, torch, torch.distributed as dist
mode, flags = sys.argv[1], sys.argv[3].split(",")
import time
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=0, world_size=1, device_id=dev)
grp = dist.new_group([0]) if "subgroup" in flags else None
x = torch.ones(64, 8192, dtype=torch.bfloat16, device=dev)
if "cold" not in flags:
    dist.all_reduce(x, group=grp)      # communicator and channels exist before the capture
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        y = x * 2
        dist.all_reduce(y, group=grp)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    x.fill_(1)
print("eager ok", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode=mode):
    y = x * 2
    dist.all_reduce(y, group=grp)
    if "slow" in flags:
        time.sleep(2.5)
    z = y + 1
print("captured", flush=True)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
assert float(z[0, 0]) == 3.0, float(z[0, 0])
print("replayed ok", flush=True)
dist.destroy_process_group()
print("done", flush=True)
'''


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main():
    out = {}
    no_watchdog = {"TORCH_NCCL_ASYNC_ERROR_HANDLING": "0", "TORCH_NCCL_ENABLE_MONITORING": "0"}
    for mode, flags, extra in (("thread_local", "", {}), ("global", "", {}), ("relaxed", "", {}),
                               ("thread_local", "", no_watchdog), ("thread_local", "cold", {}),
                               ("global", "cold", {}), ("thread_local", "subgroup", {}),
                               ("thread_local", "subgroup,cold", {}), ("global", "slow", {}),
                               ("thread_local", "slow", {})):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
        try:
            p = subprocess.run([sys.executable, "-c", CHILD, mode, str(free_port()), flags or "-"], env=env,
                               capture_output=True,
                               text=True, timeout=60)
            rc, so, se = p.returncode, p.stdout, p.stderr
        except subprocess.TimeoutExpired as e:
            rc, so, se = "timeout", (e.stdout or b"").decode(errors="replace"), (e.stderr or b"").decode(errors="replace")
        key = mode + ("+" + flags if flags else "") + ("+no_watchdog" if extra else "")
        out[key] = {"exit": rc, "stdout": so.strip().splitlines(), "stderr_tail": se.strip().splitlines()[-12:]}
        print(key, rc, so.strip().splitlines()[-1:] , flush=True)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
