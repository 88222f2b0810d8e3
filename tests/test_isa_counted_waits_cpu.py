"""Counted `s_waitcnt vmcnt(N)` sites (csrc/cw.h): on the compiled gfx950 assembly, over every path, at least N
vector-memory instructions that are CERTAIN to issue lie between the mark behind the data (`; cw_mark T`) and the wait
that names it (`; cw_wait T N`).  If fewer were issued the wait would return before the data has landed -- a silent race
of the class of the round-3 memory fault; a toolchain that emits one store fewer, spills, or moves a predicated store
behind a branch changes the count without any parity test noticing.  No GPU needed: hipcc cross-compiles to assembly.

What counts as certain: global_/buffer_/scratch_ loads, stores, atomics and LDS-DMA outside any exec-masked region (an
instruction whose lanes are all off is not issued and not counted by the hardware).  The analysis is a forward
data-flow over the kernel's basic blocks (minimum over paths, loops to a fixed point), so it proves `>= N`; it cannot
prove counts that depend on run-time values (k_dft_analysis_sq_h2 picks vmcnt(8..11) by a wave-uniform store count: its
floor of 8 is what is checked there)."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "resolution-pde_amd", "csrc")
CAP = 64                               # counts saturate here (vmcnt is 6 bits)
VMEM = re.compile(r"(global_|buffer_|scratch_)(load|store|atomic)")
LABEL = re.compile(r"(\.LBB\d+_\d+):")
BRANCH = re.compile(r"s_(c?branch)\S*\s+(\.LBB\d+_\d+)")
# exec-masked regions as the compiler lowers structured control flow: `s_and_saveexec` (if) / `s_mov_b64 exec, mask` /
# `s_and(n2)_b64 exec, ..` open one; `s_or_saveexec` + `s_xor_b64 exec, exec, ..` switch to the else part at the same
# depth; `s_or_b64 exec, exec, saved` closes it
# (`s_andn2_b64 exec, exec, ..` inside a loop retires lanes of an already masked region: no new region)
EXEC_MASK = re.compile(r"s_(and|andn2|andn1)_saveexec_b64|s_and_b64 exec,|s_mov_b64 exec, (s|vcc)")
EXEC_ELSE = re.compile(r"s_or_saveexec_b64")
EXEC_RESTORE = re.compile(r"s_or_b64 exec, exec,|s_mov_b64 exec, -1")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    return None


def check_counted_waits(body, assume_live=False):
    """-> (violations, n_marks, n_waits, n_vmem); a violation = (line number, text, proven count).
    assume_live: every exec-masked region runs with at least one lane on -- lane-predicated stores are counted, and an
    `s_cbranch_execz` edge around them only serves to reach code that nothing else reaches (a wave outside `w < 2` does
    skip that region: it then neither marks nor waits).  That is an assumption about the DATA, stated by the kernel that
    uses it (the fused FeedForward: a tile that has a successor is full)."""
    # ---- instructions with the exec-mask depth in layout order (structured control flow: regions nest) ----
    ins, depth = [], 0
    for raw in body:
        t = raw.strip()
        if not t or t.startswith((";;", "//", ".p2align", ".loc", ".cfi", ".file")):
            continue
        if EXEC_RESTORE.search(t):
            depth = max(0, depth - 1)
        if EXEC_ELSE.search(t):
            depth = max(1, depth)
        ins.append((t, depth))
        if EXEC_MASK.search(t):
            depth += 1
    # ---- basic blocks ----
    starts = {0}
    for i, (t, _) in enumerate(ins):
        if LABEL.match(t):
            starts.add(i)
        if BRANCH.match(t) or t.startswith("s_endpgm"):
            starts.add(i + 1)
    starts = sorted(x for x in starts if x < len(ins))
    block_of = {}
    for b, s0 in enumerate(starts):
        m = LABEL.match(ins[s0][0])
        if m:
            block_of[m.group(1)] = b
    ends = starts[1:] + [len(ins)]
    state_in = [None] * len(starts)            # None: not reached yet; dict tag -> proven count (missing: CAP)
    state_in[0] = {}
    violations, marks, waits, nvmem = {}, set(), set(), 0

    def merge(dst, st):
        if state_in[dst] is None:
            state_in[dst] = dict(st)
            return True
        cur, changed = state_in[dst], False
        for tag in set(cur) | set(st):
            v = min(cur.get(tag, CAP), st.get(tag, CAP))
            if v != cur.get(tag, CAP):
                cur[tag] = v
                changed = True
        return changed

    weak = []                                   # assume_live: (target block, state) of the execz edges seen in a pass
    changed = True
    while changed:
        changed = False
        if weak:                                 # seed ONE block that only an execz edge reaches, then go on as before
            for tgt, st0 in weak:
                if state_in[tgt] is None:
                    state_in[tgt] = dict(st0)
                    break
            weak = []
        for b, (s0, e0) in enumerate(zip(starts, ends)):
            if state_in[b] is None:
                continue
            st = dict(state_in[b])
            fall = True
            for i in range(s0, e0):
                t, d = ins[i]
                m = re.search(r"; cw_mark (\d+)", t)
                if m:
                    st[int(m.group(1))] = 0
                    marks.add(i)
                m = re.search(r"; cw_wait (\d+) (\d+)", t)
                if m:
                    tag, n = int(m.group(1)), int(m.group(2))
                    waits.add(i)
                    have = st.get(tag, CAP)
                    if have < n:
                        violations[i] = (i, t, have)
                    # everything at least n instructions old is done
                    for k in list(st):
                        if st[k] >= n:
                            del st[k]
                    continue
                if t.startswith("s_waitcnt") and re.search(r"vmcnt\(0\)", t):
                    st = {}
                    continue
                if VMEM.match(t) and (d == 0 or assume_live):
                    for k in st:
                        st[k] = min(CAP, st[k] + 1)
                m = BRANCH.match(t)
                if m and assume_live and t.startswith("s_cbranch_execz"):
                    tgt = block_of.get(m.group(2))
                    if tgt is not None and state_in[tgt] is None:
                        weak.append((tgt, dict(st)))
                    m = None
                if m:
                    tgt = block_of.get(m.group(2))
                    if tgt is not None and merge(tgt, st):
                        changed = True
                    if m.group(1) == "branch":
                        fall = False
                if t.startswith("s_endpgm"):
                    fall = False
            if fall and b + 1 < len(starts) and merge(b + 1, st):
                changed = True
        if not changed and any(state_in[tgt] is None for tgt, _ in weak):
            changed = True
    nvmem = sum(1 for t, d in ins if VMEM.match(t))
    return sorted(violations.values()), len(marks), len(waits), nvmem


# ---------------------------------------------------------------------------------------------------------------
# the checker on hand-written miniatures
# ---------------------------------------------------------------------------------------------------------------
def test_checker_counts_over_paths_and_loops():
    ok = ["global_load_lds_dwordx4 v1, s[0:1]", "; cw_mark 0", "global_store_dwordx4 v[2:3], v[4:7], off",
          "global_store_dwordx4 v[2:3], v[4:7], off", "s_waitcnt vmcnt(2) ; cw_wait 0 2", "s_endpgm"]
    assert check_counted_waits(ok)[0] == []
    # one store sits behind a uniform branch: on the path that skips it only one instruction is younger than the data
    skipped = ["global_load_lds_dwordx4 v1, s[0:1]", "; cw_mark 0", "global_store_dwordx4 v[2:3], v[4:7], off",
               "s_cbranch_scc1 .LBB0_2", "global_store_dwordx4 v[2:3], v[4:7], off", ".LBB0_2:",
               "s_waitcnt vmcnt(2) ; cw_wait 0 2", "s_endpgm"]
    bad = check_counted_waits(skipped)[0]
    assert len(bad) == 1 and bad[0][2] == 1
    # a store under an exec mask may not issue at all (all lanes off): it does not count
    masked = ["global_load_lds_dwordx4 v1, s[0:1]", "; cw_mark 0", "global_store_dwordx4 v[2:3], v[4:7], off",
              "s_and_saveexec_b64 s[4:5], vcc", "global_store_dwordx4 v[2:3], v[4:7], off", "s_or_b64 exec, exec, s[4:5]",
              "s_waitcnt vmcnt(2) ; cw_wait 0 2", "s_endpgm"]
    assert len(check_counted_waits(masked)[0]) == 1
    # a loop: the mark of iteration i is waited for in iteration i + 1, behind the stores at the loop's end
    loop = ["global_load_lds_dwordx4 v1, s[0:1]", "; cw_mark 0", ".LBB0_1:", "s_waitcnt vmcnt(2) ; cw_wait 0 2",
            "global_load_lds_dwordx4 v1, s[0:1]", "; cw_mark 0", "global_store_dwordx4 v[2:3], v[4:7], off",
            "global_store_dwordx4 v[2:3], v[4:7], off", "s_cbranch_scc1 .LBB0_1", "s_endpgm"]
    bad = check_counted_waits(loop)[0]
    assert len(bad) == 1 and bad[0][2] == 0            # first iteration: nothing behind the prologue's request
    loop[3] = "s_waitcnt vmcnt(0)"
    loop.insert(3, "s_nop 0")
    assert check_counted_waits(loop)[0] == []


def _kernel_bodies(tmp_path, source, prefixes):
    asm = tmp_path / (source + ".s")
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-I" + os.path.join(REPO, "include"),
           "-I" + CSRC, "-S", "--cuda-device-only", os.path.join(CSRC, source + ".hip"), "-o", str(asm)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    text = asm.read_text().split("\n")
    found = []
    for i, line in enumerate(text):
        m = re.match(r"(_ZN4rpde\w+):", line)
        if m and any(line.startswith("_ZN4rpde" + p) for p in prefixes):
            end = next(j for j in range(i, len(text)) if ".Lfunc_end" in text[j])
            found.append((m.group(1), text[i + 1:end]))
    return found


CASES = [
    # source, mangled-name prefixes, minimum (marks, waits) per instance, assume_live
    # (k_dft_synthesis3_h2: the instances with the skip gradient -- <.., true>, mangled ..Lb1E -- are the ones dispatched by
    #  default; the forward instances' counts depend on a flag the path-insensitive analysis cannot follow)
    ("fused_spectral", ("19k_dft_synthesis3_h2ILi0ELi1ELb1E", "19k_dft_synthesis3_h2ILi0ELi2ELb1E",
                        "19k_dft_synthesis3_h2ILi0ELi3ELb1E", "19k_dft_synthesis3_h2ILi1ELi0ELb1E",
                        "19k_dft_synthesis3_h2ILi1ELi1ELb1E", "19k_dft_synthesis4_h2"), 8, 8, False),
    # (the instances without FULL drain the queue instead -- a predicated store may not issue: no counted wait in them)
    ("conv_syn_h2", ("13k_conv_syn_h2",), 1, 0, False),
    # the training forward: the input DMA of the next tile against the saved-tensor stores of this one (lane-predicated
    # by `point < P`; a tile with a successor is full).  k_ff3_bwd_h2's counts (8 / 4 or 10 / 10 or 2) depend on `first`,
    # `has_next` and on how many points of the last tile are live -- run-time facts: it is covered by
    # tests/test_isa_pending_loads_cpu.py (registers in flight) and tests/test_gpu_counted_waits.py (bitwise soak).
    ("ff_fused", ("12k_ff3_fwd_h2ILi1E", "12k_ff3_fwd_h2ILi2E"), 1, 1, True),
]


@pytest.mark.skipif(_hipcc() is None, reason="hipcc not available")
@pytest.mark.parametrize("source,prefixes,min_marks,min_waits,assume_live", CASES, ids=[c[0] for c in CASES])
def test_counted_waits_hold_on_the_compiled_kernels(tmp_path, source, prefixes, min_marks, min_waits, assume_live):
    bodies = _kernel_bodies(tmp_path, source, prefixes)
    assert bodies, (source, prefixes)
    seen, total_waits = set(), 0
    for name, body in bodies:
        bad, nm, nw, nv = check_counted_waits(body, assume_live)
        seen.add(next(p for p in prefixes if name.startswith("_ZN4rpde" + p)))
        assert nm >= min_marks and nw >= min_waits, (name, nm, nw)
        total_waits += nw
        assert not bad, (name, bad[:4])
        # no spill traffic in kernels that count their memory instructions: a scratch access is a vector-memory
        # instruction the source does not show
        assert not any("scratch_" in l for l in body), name
    assert seen == set(prefixes), (seen, prefixes)
    assert total_waits >= 2, total_waits
