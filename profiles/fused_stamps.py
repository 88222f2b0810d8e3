#!/usr/bin/env python3
"""s_memtime phase timeline of the fused spectral kernels (debug build):
    python resolution-pde_amd/rpde/build.py --stamps
    RPDE_LIB=resolution-pde_amd/rpde/lib/librpde_hip_stamps.so python profiles/fused_stamps.py"""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "resolution-pde_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from rpde import _lib, ops  # noqa: E402

lib = _lib.load()
x = torch.randn(32, 256, 256, 64, device="cuda")
wy = torch.randn(64, 64, 20, 2, device="cuda") * 0.1
wx = torch.randn(64, 64, 20, 2, device="cuda") * 0.1
with torch.no_grad():
    for _ in range(3):
        ops.fspectral2d(x, wy, wx, 20)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (2 * 64 * 32))()
lib.rpde_debug_fused_stamps.argtypes = [C.c_void_p]
_lib.check(lib.rpde_debug_fused_stamps(buf), "stamps")
t = np.array(buf, dtype=np.uint64).reshape(2, 64, 32).astype(np.int64)
rr = os.environ.get("RPDE_ANA_RR", "1") != "0"
for k, name in ((0, "analysis_rr, rounds 9..12 of one wave: [top, barrier passed, first landed, first processed, (issue +) second landed, "
                    "second processed, second issued] x 4" if rr else
                    "analysis_sq, units 9..11 of one wave: [top, y landed, y done, B1 passed, partials written, B2 passed, "
                    "reduced+stored, B3 passed, x landed, x done] x 3"), (1, "synthesis")):
    a = t[k]
    a = a[a[:, 0] > 0]
    rel = a - a[:, :1]
    med = np.median(rel, axis=0)
    print(name, f"({len(a)} waves sampled), s_memtime ticks (100 MHz constant clock? no: shader cycles), median")
    print("  ", [int(v) for v in med[:30]])
    print("   deltas", [int(med[i + 1] - med[i]) for i in range(29)])
