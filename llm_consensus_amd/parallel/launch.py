"""Starting the ranks of a multi-GPU front end from one command.

The reference's ``main()`` brings up every member itself (/root/reference/src/main.rs:350-426);
so do ``llm-consensus`` and ``llm-consensus-serve``: when the placement plan of the configured
ensemble wants more than one GPU (or ``--gpus N`` says so), the command starts N child
processes, one per GPU, before it has made any GPU call of its own, waits for them and returns
their status. Rank 0 inherits the terminal (stdin / stdout) or opens the socket; the other
ranks follow it (parallel/ensemble_parallel.py). Nothing is replaced: the children are new
interpreters, the parent only supervises.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import List, Optional, Sequence

from ..utils.log import get_logger

log = get_logger("consensus.launch")


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def under_launcher() -> bool:
    """True in a process that is already one rank of a group (ours or torch.distributed.run's)."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def visible_gpus() -> int:
    import torch

    return torch.cuda.device_count()          # counts devices without creating a context


def ranks_for(cfg, requested: Optional[int], device: Optional[str], replicate: bool) -> int:
    """How many ranks a front end should run as. ``requested`` (--gpus) wins. Otherwise, on a
    machine with several GPUs, what the placement plan needs: one replica's GPUs for an
    interactive front end, every replica that fits for a server (``replicate``). One rank when
    the device is the CPU, one GPU is visible, or no plan fits."""
    if requested is not None:
        return max(1, requested)
    if device is not None and str(device).startswith("cpu"):
        return 1
    n = visible_gpus()
    if n <= 1:
        return 1
    from .placement import plan_for_config

    plan = plan_for_config(cfg, n)
    if not plan.feasible:
        return 1
    return plan.gpus_per_replica * (len(plan.replicas) if replicate else 1)


def warn_if_packed(cfg, ranks: int, device: Optional[str]) -> Optional[str]:
    """One line for a single process whose ensemble is configured for several GPUs (tensor /
    expert parallel members, or ``placement = "spread"``): everything is packed onto one device
    and tp / ep are ignored. Returns the line (also logged), or None."""
    if ranks > 1 or (device is not None and str(device).startswith("cpu")):
        return None
    from .placement import plan_for_config

    want = max([max(m.tp, m.ep) for m in cfg.models.values()] + [1])
    n = max(visible_gpus(), want, len({m.model for m in cfg.members}) if cfg.placement == "spread" else 1)
    plan = plan_for_config(cfg, n)
    if plan.feasible and plan.gpus_per_replica > 1:
        msg = (f"warning: this ensemble is planned for {plan.gpus_per_replica} GPUs per replica but runs as ONE "
               f"process: every model is packed onto one device and tp / ep settings are ignored "
               f"(start it with --gpus {plan.gpus_per_replica})")
        log.warning(msg)
        return msg
    return None


def strip_flag(argv: Sequence[str], flag: str, has_value: bool = True) -> List[str]:
    out, skip = [], False
    for tok in argv:
        if skip:
            skip = False
            continue
        if tok == flag:
            skip = has_value
            continue
        if has_value and tok.startswith(flag + "="):
            continue
        out.append(tok)
    return out


def launch_ranks(module: str, argv: Sequence[str], n: int, extra_env: Optional[dict] = None,
                 grace_s: float = 40.0) -> int:
    """Runs ``python -m module *argv`` as ``n`` ranks (rendezvous on 127.0.0.1, RANK / LOCAL_RANK
    / WORLD_SIZE set) and returns the worst exit status. Rank 0 inherits stdin and stdout; the
    others read nothing and write to stderr only. If a rank dies the others are terminated:
    they would wait for it in their next collective for ever."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, "-m", module] + list(argv), env=env,
                                      stdin=None if r == 0 else subprocess.DEVNULL,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = set(range(n))

    def on_term(signum, frame):          # a supervisor that is told to stop takes its ranks with it
        raise KeyboardInterrupt

    import signal

    old = signal.signal(signal.SIGTERM, on_term) if hasattr(signal, "SIGTERM") else None
    try:
        while live:
            for r in list(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0:
                    rc = rc or code
                    for o in live:
                        procs[o].terminate()
            time.sleep(0.05)
    except KeyboardInterrupt:
        # Rank 0 is told first and given time: it finishes what it has admitted (a server drains),
        # closes the ensemble, and with that lets the other ranks leave their loops on their own.
        # Only what is still there afterwards is terminated.
        if 0 in live:
            procs[0].terminate()
            deadline = time.time() + grace_s
            while time.time() < deadline and any(procs[r].poll() is None for r in live):
                time.sleep(0.05)
        for r in live:
            if procs[r].poll() is None:
                procs[r].terminate()
        for r in live:
            try:
                procs[r].wait(10)
            except subprocess.TimeoutExpired:
                procs[r].kill()
        rc = rc or 130
    finally:
        if old is not None:
            signal.signal(signal.SIGTERM, old)
    return rc
