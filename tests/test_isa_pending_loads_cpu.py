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
    merged, pending, bad, nloads, nlanded, in_asm = {}, {}, [], 0, 0, False
    for i, line in enumerate(body):
        ins = line.strip()
        if ins.startswith(";;#ASM"):                                 # only loads from inline asm are followed: the
            in_asm = ins.startswith(";;#ASMSTART")                   # compiler waits for its own
            continue
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
        m = re.match(r"global_load_dwordx4 v\[(\d+):(\d+)\], (v\[\d+:\d+\])", ins) if in_asm else None
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
    ok = [";;#ASMSTART", "global_load_dwordx4 v[4:7], v[2:3], off", ";;#ASMEND",
          "s_waitcnt vmcnt(0) ; landed v[4:7] v8 v9 v10 v11 v12 v13 v14",
          "s_cbranch_execz .LBB0_2", "v_add_f32_e32 v1, v4, v5", ".LBB0_2:", "v_mov_b32_e32 v4, 0"]
    skipped = [";;#ASMSTART", "global_load_dwordx4 v[4:7], v[2:3], off", ";;#ASMEND", "s_cbranch_execz .LBB0_2",
               "s_waitcnt vmcnt(0) ; landed v[4:7] v8 v9 v10 v11 v12 v13 v14", "v_add_f32_e32 v1, v4, v5", ".LBB0_2:",
               "v_mov_b32_e32 v4, 0"]
    assert _pending_violations(ok)[0] == []
    bad = _pending_violations(skipped)[0]
    assert len(bad) == 1 and bad[0][0].startswith("v_mov_b32_e32 v4")


def _kernel_bodies(tmp_path, source, names):
    asm = tmp_path / (source + ".s")
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-I" + os.path.join(REPO, "include"),
           "-I" + CSRC, "-S", "--cuda-device-only", os.path.join(CSRC, source + ".hip"), "-o", str(asm)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    text = asm.read_text().split("\n")
    for name in names:
        start = next(i for i, l in enumerate(text) if l.startswith("_ZN4rpde" + name) and ":" in l)      # the label line
        end = next(i for i in range(start, len(text)) if ".Lfunc_end" in text[i])
        yield name, text[start:end]


@pytest.mark.skipif(_hipcc() is None, reason="hipcc not available")
def test_no_use_of_asm_loaded_registers_before_the_counted_wait(tmp_path):
    for name, body in _kernel_bodies(tmp_path, "fused_spectral", ("20k_dft_analysis_sq_h2ILi1E", "20k_dft_analysis_sq_h2ILi2E",
                                                                  "20k_dft_analysis_sq_h2ILi3E", "20k_dft_analysis_rr_h2ILi1E",
                                                                  "20k_dft_analysis_rr_h2ILi2E", "20k_dft_analysis_rr_h2ILi3E")):
        bad, nloads, nlanded = _pending_violations(body)
        assert nloads >= 32, (name, nloads)              # 8 per axis, prologue + loop
        assert nlanded >= 2, (name, nlanded)
        assert not bad, (name, bad[:3])


@pytest.mark.skipif(_hipcc() is None, reason="hipcc not available")
def test_feedforward_backward_kernel_keeps_its_asm_loads_untouched_until_the_wait(tmp_path):
    """k_ff3_bwd_h2 fetches g and z3 a tile ahead with asm loads (csrc/ff_fused.hip: fb_gload) and waits with a counted
    vmcnt; the same rule applies.  Its tile loop must also be free of compiler-inserted `vmcnt(0)` (each one is a wait
    for the LDS-DMA requested a moment earlier: the default instance lost 10 % to five of them) and of spill reloads."""
    for name, body in _kernel_bodies(tmp_path, "ff_fused", ("12k_ff3_bwd_h2ILb0E", "12k_ff3_bwd_h2ILb1E")):
        bad, nloads, nlanded = _pending_violations(body)
        assert nlanded >= 1 and nloads >= 4, (name, nloads, nlanded)
        assert not bad, (name, bad[:3])
        if name.endswith("ILb0E"):                        # the default (stash) instance; the recompute instance spills
            depth, drains = 0, []
            for i, l in enumerate(body):
                m = re.match(r"\.LBB\d+_\d+:(.*)", l)
                if m:
                    d = re.search(r"Depth=(\d+)", m.group(1))
                    depth = int(d.group(1)) if d else 0
                if depth > 0 and (re.search(r"s_waitcnt.*vmcnt\(0\)", l) or "scratch_load" in l):
                    drains.append((i, l.strip()))
            assert not drains, drains[:4]
