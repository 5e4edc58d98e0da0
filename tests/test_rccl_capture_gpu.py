"""A torch.distributed (RCCL) all-reduce recorded into a hipGraph under the thread-local capture
mode the engine uses: records, replays and gives the right sum even when the capture stays open
for seconds (the case that ends the process under the global mode, where the process group's
watchdog thread trips over the open capture). Runs in a child process."""
import importlib.util
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.gpu
def test_rccl_all_reduce_is_capturable_in_thread_local_mode():
    spec = importlib.util.spec_from_file_location("rccl_capture_probe", ROOT / "benchmarks/micro/rccl_capture_probe.py")
    probe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(probe)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", probe.CHILD, "thread_local", str(probe.free_port()), "slow,subgroup"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "replayed ok" in p.stdout
