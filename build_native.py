#!/usr/bin/env python3
"""Builds the two in-tree native libraries.

* ``llm_consensus_amd/_native/_core<ext>``      -- C++17 consensus core (g++ + pybind11);
  no GPU toolchain involved, so CPU-only hosts can build and test it.
* ``llm_consensus_amd/_native/liblc_kernels.so`` -- gfx950 HIP kernels behind a C ABI
  (hipcc --offload-arch=gfx950); cross-compiles on hosts without a GPU.

Objects are cached under ``build/`` and rebuilt only when a source or header is newer.
Usage: ``python build_native.py [core] [kernels] [selftest --sanitize=thread] [--force] [-j N]``.
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
NATIVE_DIR = ROOT / "llm_consensus_amd" / "_native"
BUILD_DIR = ROOT / "build"
GFX_ARCH = "gfx950"


def _newer_than(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def _run(cmd, **kw):
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, **kw)
    if proc.returncode != 0:
        sys.stderr.write(" ".join(map(str, cmd)) + "\n" + proc.stdout + "\n")
        raise RuntimeError(f"command failed ({proc.returncode}): {cmd[0]} ... {cmd[-1]}")
    if proc.stdout.strip():
        sys.stderr.write(proc.stdout)


def core_path() -> Path:
    return NATIVE_DIR / ("_core" + sysconfig.get_config_var("EXT_SUFFIX"))


def kernels_path() -> Path:
    return NATIVE_DIR / "liblc_kernels.so"


def build_core(force: bool = False, jobs: int = 8) -> Path:
    import pybind11

    src_dir = ROOT / "csrc" / "core"
    srcs = [s for s in sorted(src_dir.glob("*.cpp")) if s.name != "selftest.cpp"]
    hdrs = sorted(src_dir.glob("*.hpp"))
    obj_dir = BUILD_DIR / "core"
    obj_dir.mkdir(parents=True, exist_ok=True)
    inc = [f"-I{sysconfig.get_paths()['include']}", f"-I{pybind11.get_include()}", f"-I{src_dir}"]
    flags = ["-O2", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wextra", "-Wno-unused-parameter"]
    todo = []
    objs = []
    for s in srcs:
        o = obj_dir / (s.stem + ".o")
        objs.append(o)
        if force or _newer_than(o, [s, *hdrs]):
            todo.append(["g++", *flags, *inc, "-c", str(s), "-o", str(o)])
    with ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        list(ex.map(_run, todo))
    out = core_path()
    if force or todo or _newer_than(out, objs):
        _run(["g++", "-shared", "-o", str(out), *map(str, objs)])
    return out


def build_selftest(sanitize: str = "address,undefined") -> Path:
    """Stand-alone stress driver of the host core (csrc/core/selftest.cpp) under a host
    sanitizer: ``address,undefined`` or ``thread``. Host code only; GPU code is never
    built with sanitizers."""
    src_dir = ROOT / "csrc" / "core"
    srcs = [s for s in sorted(src_dir.glob("*.cpp")) if s.name != "bindings.cpp"]
    out = BUILD_DIR / ("selftest_" + sanitize.replace(",", "_"))
    BUILD_DIR.mkdir(parents=True, exist_ok=True)
    static = "-static-libtsan" if sanitize == "thread" else "-static-libasan"
    _run(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", f"-fsanitize={sanitize}", static,
          "-fno-sanitize-recover=all", f"-I{src_dir}", *map(str, srcs), "-o", str(out), "-lpthread"])
    return out


def build_kernels(force: bool = False, jobs: int = 8) -> Path:
    src_dir = ROOT / "csrc" / "kernels"
    # the device runtime (host code over the HIP runtime API) lives in the same library
    srcs = sorted(src_dir.glob("*.hip")) + sorted((ROOT / "csrc" / "runtime").glob("*.hip"))
    hdrs = sorted(src_dir.glob("*.h")) + sorted(src_dir.glob("*.hpp")) + sorted(src_dir.glob("*.cuh"))
    obj_dir = BUILD_DIR / "kernels"
    obj_dir.mkdir(parents=True, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = [f"--offload-arch={GFX_ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-math-errno", "-ffp-contract=fast",
             "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", f"-I{src_dir}"]
    todo = []
    objs = []
    for s in srcs:
        o = obj_dir / (s.stem + ".o")
        objs.append(o)
        if force or _newer_than(o, [s, *hdrs]):
            todo.append([hipcc, *flags, "-c", str(s), "-o", str(o)])
    with ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        list(ex.map(_run, todo))
    out = kernels_path()
    if force or todo or _newer_than(out, objs):
        _run([hipcc, f"--offload-arch={GFX_ARCH}", "-shared", "-fPIC", "-o", str(out), *map(str, objs)])
    return out


def main() -> None:
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("targets", nargs="*", default=["core", "kernels"])
    ap.add_argument("--force", action="store_true")
    ap.add_argument("-j", "--jobs", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--sanitize", default="address,undefined", help="sanitizer set of the 'selftest' target")
    a = ap.parse_args()
    if "selftest" in a.targets:
        print("built", build_selftest(a.sanitize))
    if "core" in a.targets:
        print("built", build_core(a.force, a.jobs))
    if "kernels" in a.targets:
        print("built", build_kernels(a.force, a.jobs))


if __name__ == "__main__":
    main()
