"""Builds the native core's stress driver (csrc/core/selftest.cpp) with host sanitizers and
runs it: AddressSanitizer + UBSan, then ThreadSanitizer. Host code only. The reference has
no concurrency to check (single-threaded actor system, /root/reference/src/main.rs:350)."""
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_native_core_selftest_under_sanitizer(san):
    if shutil.which("g++") is None:
        pytest.skip("no host compiler")
    import build_native

    try:
        exe = build_native.build_selftest(san)
    except RuntimeError as e:
        pytest.skip(f"sanitizer runtime unavailable: {e}")
    r = subprocess.run([str(exe), "4", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "selftest ok" in r.stdout
