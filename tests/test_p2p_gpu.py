"""Peer-to-peer collectives (csrc/kernels/p2p.hip) with 2 and 4 ranks as separate processes on
one GPU: HIP IPC mapping of the peers' exchange regions, the flag protocol, the fused
all-reduce + residual + RMSNorm against an fp32 reference, graph capture and replay, and
tensor / expert parallel engines whose decode step is a single captured graph. One device is
what a test box has; the code path is the one W GPUs run, with HBM in place of xGMI."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 4])
def test_p2p_collectives_across_processes(world, tmp_path):
    port = _free_port()
    prefix = tmp_path / "ok"
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / "p2p_worker.py"), str(prefix)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=240)[0])
    finally:
        for p in procs:                      # exactly the processes started here
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-4000:]}"
    reports = [json.loads(Path(f"{prefix}.{r}").read_text()) for r in range(world)]
    assert all(rep["world"] == world for rep in reports)
    out = os.environ.get("LC_P2P_REPORT")
    if out:
        Path(out).parent.mkdir(parents=True, exist_ok=True)
        with open(out, "a") as f:
            f.write(json.dumps(reports[0]) + "\n")


def _run_ranks(script, args, world, timeout=240, extra_env=None):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / script)] + list(args), env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:                      # exactly the processes started here
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-4000:]}"


def test_first_contact_check_and_in_process_fallback(tmp_path):
    """A new model group checks one tiny all-reduce (one- and two-shot), the fused form, an
    all-gather and an all-to-all before it serves. With a peer that never shows up (stall) or
    contributes wrong values (corrupt) every rank notices within seconds, all agree, and the
    group serves from its bootstrap communicator in the same processes: same tokens."""
    prefix = tmp_path / "fc"
    seen = {}
    for mode in ("ok", "stall", "corrupt"):
        _run_ranks("p2p_firstcontact_worker.py", [str(prefix), mode], 2,
                   extra_env={"LC_P2P_SELF_CHECK_TIMEOUT_S": "3"})
        seen[mode] = [json.loads(Path(f"{prefix}.{mode}.{r}").read_text()) for r in range(2)]
    assert seen["ok"][0]["collectives"].startswith("p2p") and seen["ok"][0]["reason"] is None
    for mode in ("stall", "corrupt"):
        for rep in seen[mode]:
            assert rep["reason"] and not rep["collectives"].startswith("p2p")
            assert rep["tokens"] == seen["ok"][0]["tokens"]
    assert "timed out" in " ".join(r["reason"] for r in seen["stall"])
    assert "wrong" in " ".join(r["reason"] for r in seen["corrupt"])
