#!/usr/bin/env python3
"""s_memtime phase timeline of the fused FeedForward kernel (debug build, see profiles/fused_stamps.py)"""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "resolution-pde_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from rpde import _lib  # noqa: E402
from models.custom_layer import FeedForward  # noqa: E402

lib = _lib.load()
ff = FeedForward(64, 4, n_layers=3, layer_norm=True, dropout=0.1).to("cuda")
x = torch.randn(32, 256, 256, 64, device="cuda")
res = torch.randn_like(x)
mode = sys.argv[1] if len(sys.argv) > 1 else "train"
if mode == "eval":
    ff.eval()
    with torch.no_grad():
        for _ in range(2):
            ff(x, residual=res)
elif mode == "bwd":
    xg = x.requires_grad_(True)
    g = torch.randn_like(x)
    for _ in range(2):
        ff(xg, residual=res).backward(g)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (3 * 64))()
    lib.rpde_debug_ffb_stamps.argtypes = [C.c_void_p]
    _lib.check(lib.rpde_debug_ffb_stamps(buf), "stamps")
    t = np.array(buf, dtype=np.uint64).reshape(3, 8, 8).astype(np.int64)
    names = ["top", "A passed", "B passed", "C passed", "d2 landed", "du2 done", "D passed", "gemm2 + d1 landed"]
    for wi, wn in enumerate(("wave 0", "wave 3", "wave 6")):
        print(mode, wn)
        for ti in range(4):
            row = t[wi, ti]
            print("   tile", ti, " ".join(f"{names[i]}:+{int(row[i] - row[0])}" for i in range(1, 8)),
                  f" | tile period {int(t[wi, ti + 1, 0] - row[0])}")
    sys.exit(0)
else:
    xg = x.requires_grad_(True)
    for _ in range(2):
        ff(xg, residual=res)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (3 * 64))()
lib.rpde_debug_ff_stamps.argtypes = [C.c_void_p]
_lib.check(lib.rpde_debug_ff_stamps(buf), "stamps")
t = np.array(buf, dtype=np.uint64).reshape(3, 8, 8).astype(np.int64)
names = ["top", "B0 passed", "phase1 done", "B1 passed", "phase2+3 done", "B2 passed", "phase4/convert done"]
for wi, wn in enumerate(("wave 0 (converter)", "wave 3", "wave 6 (epilogue)")):
    print(mode, wn)
    for ti in range(3):
        row = t[wi, ti]
        print("   tile", ti, " ".join(f"{names[i]}:+{int(row[i] - row[0])}" for i in range(1, 7)),
              f" | tile period {int(t[wi, ti + 1, 0] - row[0])}")
