#!/usr/bin/env python3
"""Phase timeline of the dominant split-bf16 GEMM from in-kernel s_memtime stamps.
    python resolution-pde_amd/rpde/build.py --stamps
    RPDE_LIB=resolution-pde_amd/rpde/lib/librpde_hip_stamps.so python profiles/stamps.py [B]
Wave 0 of 64 workgroups from the middle of the launch records: start, first loads issued, per k-tile
(LDS stores done / barrier passed / MFMAs + second barrier done), epilogue start, end."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "resolution-pde_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from rpde import _lib  # noqa: E402


def report(lib, names, order):
    buf = (C.c_ulonglong * (64 * 32))()
    lib.rpde_debug_stamps.argtypes = [C.c_void_p]
    _lib.check(lib.rpde_debug_stamps(buf), "stamps")
    t = np.array(buf, dtype=np.uint64).reshape(64, 32).astype(np.int64)
    t = t[t[:, 0] > 0]
    rel = t - t[:, :1]
    med = np.median(rel, axis=0)
    # slots 31 / 25 hold the constant 100 MHz clock at start / loop end: shader frequency over the main loop
    ghz = np.median((t[:, 26] - t[:, 0]) / np.maximum(1, (t[:, 25] - t[:, 31])) * 0.1)
    print(f"shader clock over the main loop: {ghz:.2f} GHz (s_memtime ticks per s_memrealtime tick x 100 MHz)")
    prev = 0
    print(f"{len(t)} workgroups sampled; s_memtime ticks, median over workgroups")
    for i in order:
        print(f"{names[i]:22s} t={med[i]:9.0f}  (+{med[i] - prev:7.0f})")
        prev = med[i]


NAMES = ["start", "loads issued"] + sum([[f"k{k} stored", f"k{k} barrier1", f"k{k} mfma+barrier2"] for k in range(8)], []) + \
        ["loop end", "epilogue end", "slab0 in LDS", "slab0 rows done", "slab1 in LDS"]


def tn(lib, dev, P, K, N):
    """weight gradient: [256,P] x [P,256], 192 K-slabs; the stamps cover the first 8 of each workgroup's k-tiles"""
    g = torch.randn(P, N, device=dev)
    h = torch.randn(P, K, device=dev)
    S = 192
    slabs = torch.empty(S * N * K, device=dev)
    d = _lib.GemmDesc()
    d.A, d.B, d.C = g.data_ptr(), h.data_ptr(), slabs.data_ptr()
    d.M, d.N, d.K, d.a_kmajor, d.b_kmajor = N, K, P, 0, 0
    d.lda, d.ldb, d.ldc, d.batch, d.zdiv, d.ksplit, d.alpha, d.sCk = N, K, K, 1, 1, S, 1.0, N * K
    for _ in range(3):
        _lib.check(lib.rpde_gemm_f32(C.byref(d), _lib.stream_ptr()), "gemm")
    torch.cuda.synchronize()
    report(lib, NAMES, list(range(25)))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    mode = sys.argv[2] if len(sys.argv) > 2 else "nt"
    lib = _lib.load()
    dev = "cuda:0"
    P, K, N = B * 65536, 256, 256
    if mode == "tn":
        return tn(lib, dev, P, K, N)
    h1 = torch.randn(P, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.06
    b = torch.randn(N, device=dev)
    h2, d2 = torch.empty(P, N, device=dev), torch.empty(P, N, device=dev)
    img = torch.empty(lib.rpde_split_weights_bytes(N, K), dtype=torch.uint8, device=dev)
    _lib.check(lib.rpde_split_weights(w.data_ptr(), 1, K, N, K, img.data_ptr(), _lib.stream_ptr()), "split")
    d = _lib.GemmDesc()
    d.A, d.B, d.C = h1.data_ptr(), w.data_ptr(), h2.data_ptr()
    d.M, d.N, d.K, d.a_kmajor, d.b_kmajor = P, N, K, 1, 1
    d.lda, d.ldb, d.ldc, d.batch, d.zdiv, d.ksplit, d.alpha = K, K, N, 1, 1, 1, 1.0
    d.bias, d.bias_mode, d.write_act, d.aux_out = b.data_ptr(), 1, 1, d2.data_ptr()
    d.drop_p, d.drop_seed, d.drop_ld, d.drop_where = 0.1, 12345, N, 4
    d.b_split = img.data_ptr()
    for _ in range(3):
        _lib.check(lib.rpde_gemm_f32(C.byref(d), _lib.stream_ptr()), "gemm")
    torch.cuda.synchronize()
    report(lib, NAMES, list(range(25)) + [26, 28, 29, 30, 27])


if __name__ == "__main__":
    main()
