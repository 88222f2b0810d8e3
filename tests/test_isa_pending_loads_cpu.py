"""k_dft_analysis_sq_h2 issues its field loads from inline asm and waits for them with a hand-counted s_waitcnt
(DESIGN.md section 3.1): the compiler does not know that the destination registers are pending, so nothing may read or
copy them between the load and the wait.  This test compiles the kernel to gfx950 assembly (no GPU needed) and checks,
in layout order, that no vector / LDS / store instruction touches a loaded register before the next s_waitcnt vmcnt."""
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


@pytest.mark.skipif(_hipcc() is None, reason="hipcc not available")
def test_no_use_of_asm_loaded_registers_before_the_counted_wait(tmp_path):
    asm = tmp_path / "fused_spectral.s"
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-I" + os.path.join(REPO, "include"),
           "-I" + CSRC, "-S", "--cuda-device-only", os.path.join(CSRC, "fused_spectral.hip"), "-o", str(asm)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    text = asm.read_text().split("\n")
    checked = 0
    for name in ("k_dft_analysis_sq_h2ILi1E", "k_dft_analysis_sq_h2ILi2E", "k_dft_analysis_sq_h2ILi3E"):
        start = next(i for i, l in enumerate(text) if l.startswith("_ZN4rpde20" + name) and ":" in l)      # the label line
        end = next(i for i in range(start, len(text)) if ".Lfunc_end" in text[i])
        body = text[start:end]
        loads = [(i, m) for i, l in enumerate(body) for m in [re.search(r"global_load_dwordx4 v\[(\d+):(\d+)\]", l)] if m]
        assert len(loads) >= 16, (name, len(loads))          # 8 per axis, prologue + loop
        for i, m in loads:
            dest = set(range(int(m.group(1)), int(m.group(2)) + 1))
            for j in range(i + 1, len(body)):
                ins = body[j].strip()
                if ins.startswith("s_waitcnt") and "vmcnt" in ins:
                    break
                if ins.startswith(("v_", "ds_", "global_store", "buffer_", "scratch_")) and (_regs(ins) & dest):
                    raise AssertionError(f"{name}: '{ins}' reads a register of the pending load '{body[i].strip()}'")
            checked += 1
    assert checked >= 48
