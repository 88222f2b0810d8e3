"""k_dft_analysis_sq_h2 issues its field loads from inline asm and waits for them with a hand-counted s_waitcnt
(DESIGN.md section 3.1): the compiler does not know that the destination registers are pending, so nothing may read,
copy OR REUSE them between the load and the wait that covers it -- on any path.  (Round 3: waves without a duty
branched around the wait; the compiler, for which the registers were filled and dead, used them for the next address
computation, and a load that landed late overwrote the address: a memory fault in training at 96^2 that no parity test
showed.)  This test compiles the kernel to gfx950 assembly (no GPU needed) and follows the set of pending registers
through the code in layout order, merging it into the target of every forward branch: a load adds its destination, the
kernel's wait statements name the registers they cover (`; landed v[..] ...`), `vmcnt(0)` clears everything, and no
vector / LDS / memory instruction may touch a pending register."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "resolution-pde_amd", "csrc")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    return None


def _regs(line):
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]", line):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", line):
        out.add(int(m.group(1)))
    return out


def _pending_violations(body):
    """(instruction, pending load) pairs; see the module docstring"""
    label_at = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"\.LBB\d+_\d+:", l)}
    merged, pending, bad, nloads, nlanded = {}, {}, [], 0, 0
    for i, line in enumerate(body):
        ins = line.strip()
        m = re.match(r"(\.LBB\d+_\d+):", ins)
        if m:
            for r, src in merged.pop(m.group(1), {}).items():
                pending.setdefault(r, src)
            continue
        if "; landed" in ins:                                        # the end of a wait statement of the kernel
            nlanded += 1
            for r in _regs(ins.split("; landed")[1]):
                pending.pop(r, None)
            continue
        if ins.startswith("s_waitcnt"):
            if re.search(r"vmcnt\(0\)", ins):
                pending.clear()
            continue
        m = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", ins)
        if m:
            if label_at.get(m.group(1), -1) > i:                     # forward: the target inherits what is pending here
                merged.setdefault(m.group(1), {}).update(pending)
            if ins.startswith("s_branch"):
                pending = {}                                         # (what follows is reached through its label only)
            continue
        m = re.match(r"global_load_dwordx4 v\[(\d+):(\d+)\], (v\[\d+:\d+\])", ins)
        if m:
            hit = _regs(m.group(3)) & set(pending)
            if hit:
                bad.append((ins, pending[min(hit)]))
            nloads += 1
            for r in range(int(m.group(1)), int(m.group(2)) + 1):
                pending[r] = ins
            continue
        if ins.startswith(("v_", "ds_", "global_", "buffer_", "scratch_", "flat_")):
            hit = _regs(ins) & set(pending)
            if hit:
                bad.append((ins, pending[min(hit)]))
    return bad, nloads, nlanded


def test_the_checker_sees_a_skipped_wait():
    """the round-3 fault in miniature: the wait sits in a block that a branch skips, the register is reused behind it"""
    ok = ["global_load_dwordx4 v[4:7], v[2:3], off", "s_waitcnt vmcnt(0) ; landed v[4:7] v8 v9 v10 v11 v12 v13 v14",
          "s_cbranch_execz .LBB0_2", "v_add_f32_e32 v1, v4, v5", ".LBB0_2:", "v_mov_b32_e32 v4, 0"]
    skipped = ["global_load_dwordx4 v[4:7], v[2:3], off", "s_cbranch_execz .LBB0_2",
               "s_waitcnt vmcnt(0) ; landed v[4:7] v8 v9 v10 v11 v12 v13 v14", "v_add_f32_e32 v1, v4, v5", ".LBB0_2:",
               "v_mov_b32_e32 v4, 0"]
    assert _pending_violations(ok)[0] == []
    bad = _pending_violations(skipped)[0]
    assert len(bad) == 1 and bad[0][0].startswith("v_mov_b32_e32 v4")


@pytest.mark.skipif(_hipcc() is None, reason="hipcc not available")
def test_no_use_of_asm_loaded_registers_before_the_counted_wait(tmp_path):
    asm = tmp_path / "fused_spectral.s"
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-I" + os.path.join(REPO, "include"),
           "-I" + CSRC, "-S", "--cuda-device-only", os.path.join(CSRC, "fused_spectral.hip"), "-o", str(asm)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    text = asm.read_text().split("\n")
    for name in ("k_dft_analysis_sq_h2ILi1E", "k_dft_analysis_sq_h2ILi2E", "k_dft_analysis_sq_h2ILi3E"):
        start = next(i for i, l in enumerate(text) if l.startswith("_ZN4rpde20" + name) and ":" in l)      # the label line
        end = next(i for i in range(start, len(text)) if ".Lfunc_end" in text[i])
        bad, nloads, nlanded = _pending_violations(text[start:end])
        assert nloads >= 32, (name, nloads)              # 8 per axis, prologue + loop (+ the table copy)
        assert nlanded >= 2, (name, nlanded)
        assert not bad, (name, bad[:3])
