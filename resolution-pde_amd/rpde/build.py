"""Build librpde_hip.so (gfx950) in-tree with hipcc.

    python -m rpde.build          (from resolution-pde_amd/)
    python resolution-pde_amd/rpde/build.py [--force]

hipcc cross-compiles without a GPU; objects go to <repo>/build/, the shared
library to resolution-pde_amd/rpde/lib/librpde_hip.so (git-ignored, shipped to
the GPU box with the tree).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)                       # resolution-pde_amd/
REPO = os.path.dirname(ROOT)
CSRC = os.path.join(ROOT, "csrc")
INCLUDE = os.path.join(REPO, "include")
OBJDIR = os.path.join(REPO, "build", "rpde")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "librpde_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I" + INCLUDE, "-I" + CSRC,
         "-Wno-unused-result",
         # packed fp32 VALU does not overlap with the matrix pipe on gfx950 (profiles/ubench/overlap.hip):
         # keep the compiler from pairing scalar fp32 adds / multiplies into v_pk_* instructions
         "-fno-slp-vectorize"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    return "hipcc"


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newest_header() -> float:
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return max(os.path.getmtime(h) for h in hs)


def build(force: bool = False, verbose: bool = True, stamps: bool = False, define: str = "") -> str:
    """stamps=True: debug variant with in-kernel phase timestamps (RPDE_STAMPS) -> lib/librpde_hip_stamps.so,
    loaded instead of the product library when RPDE_LIB points at it (profiles/stamps.py).
    define="X": experiment variant compiled with -DX -> lib/librpde_hip_X.so (same mechanism; profiles/ff_bench.py)."""
    global OBJDIR, LIB
    if stamps:
        OBJDIR, LIB = os.path.join(REPO, "build", "rpde_stamps"), os.path.join(LIBDIR, "librpde_hip_stamps.so")
        if "-DRPDE_STAMPS" not in FLAGS:
            FLAGS.append("-DRPDE_STAMPS")
    elif define:
        OBJDIR, LIB = os.path.join(REPO, "build", "rpde_" + define), os.path.join(LIBDIR, f"librpde_hip_{define}.so")
        FLAGS.extend("-D" + d for d in define.split(","))
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hdr = _newest_header()
    jobs, objs = [], []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src[:-4] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [_hipcc(), *FLAGS, "-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr[-4000:]}")
        return s

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for s in ex.map(compile_one, jobs):
                if verbose:
                    print(f"[rpde.build] compiled {os.path.basename(s)}", flush=True)
    if jobs or force or not os.path.exists(LIB):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-Wl,--no-undefined", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[rpde.build] linked {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    _def = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--define=")]
    build(force="--force" in sys.argv, stamps="--stamps" in sys.argv, define=_def[0] if _def else "")
