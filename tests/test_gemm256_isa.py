"""Build-time check of the 256x256 GEMM's main loop (csrc/kernels/gemm256.hip, SCHED 2).

The loop loads its weight fragments with inline-assembly ``global_load_dwordx4`` and waits for them
with hand-counted ``s_waitcnt vmcnt(N)``. That is only sound if, between a load and the MFMAs that
consume it, nothing else touches the destination registers, a wait that covers the load comes before
the first of those MFMAs, and no memory instruction the counts do not know about is issued: a register copy made before the data has landed would copy stale bits, a
spill (scratch store / load) would shift every count. The compiler gives no such guarantee, so the
generated code is inspected: hipcc cross-compiles gfx950 without a GPU."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HIPCC = "/opt/rocm/bin/hipcc"


def _regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


@pytest.mark.skipif(not shutil.which(HIPCC), reason="hipcc not installed")
def test_weights_in_registers_loop_has_no_copies_spills_or_uncounted_memory_ops(tmp_path):
    out = tmp_path / "gemm256.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-math-errno", "-ffp-contract=fast",
                    "-fno-gpu-rdc", f"-I{ROOT / 'csrc' / 'kernels'}", "--cuda-device-only", "-S",
                    str(ROOT / "csrc" / "kernels" / "gemm256.hip"), "-o", str(out)], check=True, timeout=600)
    text = out.read_text()
    kernels = re.findall(r"^(_ZN\S*gemm256_kernelILi\dELb[01]ELi2EE[^:\s]*):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M)
    assert len(kernels) >= 5, "expected the SCHED 2 instantiations (3 epilogues + 2 grouped)"
    for name, body in kernels:
        raw = [l.strip() for l in body.splitlines() if l.strip() and not l.strip().startswith(";")]
        assert not any(l.startswith("scratch_") for l in raw), f"{name}: register spills"
        # the steady-state loop: from a label to the backward branch that targets it, 128 MFMAs inside
        # (two K-tiles of 64)
        loop = None
        for i, l in enumerate(raw):
            m = re.match(r"(\.LBB\d+_\d+):", l)
            if not m:
                continue
            for j in range(i + 1, len(raw)):
                if re.match(r"\.LBB\d+_\d+:", raw[j]):
                    break
                if raw[j].startswith("s_cbranch") and raw[j].split()[-1] == m.group(1):
                    body_ij = raw[i + 1: j]
                    if sum(x.startswith("v_mfma") for x in body_ij) == 128:
                        loop = body_ij
                    break
            if loop:
                break
        assert loop is not None, f"{name}: steady-state loop of 128 MFMAs not found"
        loaded = set()
        for l in loop:
            if l.startswith("global_load_dwordx4"):
                loaded |= _regs(l.split()[1].rstrip(","))
        assert len(loaded) == 64, f"{name}: expected 16 weight fragments x 4 registers, found {len(loaded)}"
        # a register with a load in flight belongs to that load until an MFMA has read it: going
        # round the loop from each load, the next instruction that names the register is an MFMA
        # (a temporary living there between the MFMA and the next load is fine)
        touched = []
        for l in loop:
            op, *rest = l.replace(",", " ").split()
            t = set()
            for tok in rest:
                t |= _regs(tok)
            touched.append((op, t))
        n = len(loop)
        for i, (op, t) in enumerate(touched):
            if op != "global_load_dwordx4":
                continue
            dst = _regs(loop[i].split()[1].rstrip(","))
            for r in dst:
                # ... and by then a wait has let at most as many younger memory operations stay in
                # flight as were issued after this load (vmcnt counts loads and LDS-DMA in issue order)
                younger, covered = 0, False
                for k in range(1, n + 1):
                    op2, t2 = touched[(i + k) % n]
                    line2 = loop[(i + k) % n]
                    if op2.startswith(("global_load", "buffer_load", "global_store", "buffer_store")):
                        younger += 1
                    elif op2 == "s_waitcnt":
                        m = re.search(r"vmcnt\((\d+)\)", line2)
                        covered = covered or (m is not None and int(m.group(1)) <= younger)
                    if r in t2:
                        assert op2.startswith("v_mfma"), (
                            f"{name}: v{r} is loaded by `{loop[i]}` and next touched by `{line2}`")
                        assert covered, f"{name}: `{line2}` reads v{r} before a wait covers `{loop[i]}`"
                        break
        # memory operations of the loop: only the counted ones, and only the hand-written waits
        vmem = [l.split()[0] for l in loop if l.startswith(("global_", "buffer_", "flat_", "scratch_"))]
        assert set(vmem) <= {"global_load_dwordx4", "global_load_lds_dwordx4"}, (name, set(vmem))
        assert vmem.count("global_load_dwordx4") == 16 and vmem.count("global_load_lds_dwordx4") == 8, name
        waits = [l for l in loop if l.startswith("s_waitcnt") and "vmcnt" in l]
        assert len(waits) == 4 and all(re.search(r"vmcnt\((12|14)\)", w) for w in waits), (name, waits)


@pytest.mark.skipif(not shutil.which(HIPCC), reason="hipcc not installed")
def test_four_wave_loop_keeps_accumulators_in_place_and_counts_its_requests(tmp_path):
    """csrc/kernels/gemm256w.hip: one wave per SIMD, 64 accumulator tiles in AGPRs, inline-assembly
    MFMAs and weight loads, hand-counted waits two K-tiles deep. Checked in the generated code: the
    steady-state loop (two K-tiles, 256 MFMAs) accumulates in place in AGPRs (no accumulator copies),
    spills nothing, issues exactly the 32 weight loads and 16 LDS-DMA the counts assume, and every
    register a load targets is next read by an MFMA after a wait that covers that load."""
    out = tmp_path / "gemm256w.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-math-errno", "-ffp-contract=fast",
                    "-fno-gpu-rdc", f"-I{ROOT / 'csrc' / 'kernels'}", "--cuda-device-only", "-S",
                    str(ROOT / "csrc" / "kernels" / "gemm256w.hip"), "-o", str(out)], check=True, timeout=600)
    text = out.read_text()
    kernels = re.findall(r"^(_ZN\S*gemm256w_kernelILi\dEE[^:\s]*):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M)
    assert len(kernels) == 2, "expected the bf16 and SwiGLU instantiations"
    for name, body in kernels:
        raw = [l.strip() for l in body.splitlines() if l.strip() and not l.strip().startswith(";")]
        loop = None
        for i, l in enumerate(raw):
            m = re.match(r"(\.LBB\d+_\d+):", l)
            if not m:
                continue
            for j in range(i + 1, len(raw)):
                if re.match(r"\.LBB\d+_\d+:", raw[j]):
                    break
                if raw[j].startswith("s_cbranch") and raw[j].split()[-1] == m.group(1):
                    body_ij = raw[i + 1: j]
                    if sum(x.startswith("v_mfma") for x in body_ij) == 256:
                        loop = body_ij
                    break
            if loop:
                break
        assert loop is not None, f"{name}: steady-state loop of 256 MFMAs not found"
        assert not any(l.startswith(("scratch_", "v_accvgpr")) for l in loop), f"{name}: spills or accumulator copies"
        accs = set()
        for l in loop:
            if l.startswith("v_mfma"):
                m = re.match(r"v_mfma_f32_16x16x32_bf16 (a\[\d+:\d+\]), v\[\d+:\d+\], v\[\d+:\d+\], (a\[\d+:\d+\])$", l)
                assert m and m.group(1) == m.group(2), f"{name}: not an in-place AGPR accumulate: {l}"
                accs.add(m.group(1))
        assert len(accs) == 64, f"{name}: {len(accs)} accumulator tiles"
        touched = []
        for l in loop:
            op, *rest = l.replace(",", " ").split()
            t = set()
            for tok in rest:
                t |= _regs(tok)
            touched.append((op, t))
        n = len(loop)
        loaded = set()
        for i, (op, t) in enumerate(touched):
            if op != "global_load_dwordx4":
                continue
            dst = _regs(loop[i].split()[1].rstrip(","))
            loaded |= dst
            for r in dst:
                younger, covered = 0, False
                for k in range(1, n + 1):
                    op2, t2 = touched[(i + k) % n]
                    line2 = loop[(i + k) % n]
                    if op2.startswith(("global_load", "buffer_load", "global_store", "buffer_store")):
                        younger += 1
                    elif op2 == "s_waitcnt":
                        m = re.search(r"vmcnt\((\d+)\)", line2)
                        covered = covered or (m is not None and int(m.group(1)) <= younger)
                    if r in t2:
                        assert op2.startswith("v_mfma"), f"{name}: v{r} is loaded by `{loop[i]}` and next touched by `{line2}`"
                        assert covered, f"{name}: `{line2}` reads v{r} before a wait covers `{loop[i]}`"
                        break
        assert len(loaded) == 128, f"{name}: expected 32 weight fragments x 4 registers, found {len(loaded)}"
        vmem = [l.split()[0] for l in loop if l.startswith(("global_", "buffer_", "flat_", "scratch_"))]
        assert vmem.count("global_load_dwordx4") == 32 and vmem.count("global_load_lds_dwordx4") == 16, name
        assert set(vmem) == {"global_load_dwordx4", "global_load_lds_dwordx4"}, (name, set(vmem))
        waits = [l for l in loop if l.startswith("s_waitcnt") and "vmcnt" in l]
        assert len(waits) == 4 and all(re.search(r"vmcnt\((32|40)\)", w) for w in waits), (name, waits)
