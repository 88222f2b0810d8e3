#!/usr/bin/env python3
"""Micro-benchmark of the GEMM variants the FeedForward uses (P = B*65536 points):
separates main-loop efficiency from the staged-activation / epilogue cost.
    python profiles/kernel_bench.py [B]"""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "resolution-pde_amd"))
import torch  # noqa: E402
from rpde import _lib  # noqa: E402


def run(name, M, N, K, a_k, b_k, iters=10, **kw):
    lib = _lib.load()
    dev = "cuda:0"
    A = torch.randn(M * K if True else 0, device=dev)
    Bm = torch.randn(N * K, device=dev) * 0.05
    ks = kw.pop("ksplit", 1)
    Cm = torch.empty(ks * M * N, device=dev)
    d = _lib.GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), Bm.data_ptr(), Cm.data_ptr()
    d.M, d.N, d.K, d.a_kmajor, d.b_kmajor = M, N, K, a_k, b_k
    d.lda = K if a_k else M
    d.ldb = K if b_k else N
    d.ldc, d.batch, d.zdiv, d.ksplit, d.alpha, d.sCk = N, 1, 1, ks, 1.0, M * N
    keep = []
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            keep.append(v)
            v = v.data_ptr()
        setattr(d, k, v)
    st = _lib.stream_ptr()
    for _ in range(2):
        _lib.check(lib.rpde_gemm_f32(C.byref(d), st), name)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        _lib.check(lib.rpde_gemm_f32(C.byref(d), st), name)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
    print(f"{name:46s} M={M:8d} N={N:4d} K={K:8d}  {ms:8.3f} ms  {tf:6.1f} TF  ({100 * tf / 157.3:4.1f}% of fp32 MFMA peak)",
          flush=True)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    P = B * 65536
    dev = "cuda:0"
    bias = torch.randn(256, device=dev)
    run("fwd NT 256->256 plain", P, 256, 256, 1, 1)
    run("fwd NT 256->256 +bias", P, 256, 256, 1, 1, bias=bias, bias_mode=1)
    run("fwd NT 256->256 gelu(A)", P, 256, 256, 1, 1, bias=bias, bias_mode=1, act_a=1)
    run("fwd NT 256->256 gelu+drop(A)", P, 256, 256, 1, 1, bias=bias, bias_mode=1, act_a=1, drop_p=0.1, drop_seed=7,
        drop_ld=256, drop_where=1)
    hout = torch.empty(P * 256, device=dev)
    run("fwd NT 256->256 -> h=gelu(drop z), d=gelu'", P, 256, 256, 1, 1, bias=bias, bias_mode=1, write_act=1, aux_out=hout,
        drop_p=0.1, drop_seed=7, drop_ld=256, drop_where=4)
    run("fwd NT 64->256 -> h, d", P, 256, 64, 1, 1, bias=bias, bias_mode=1, write_act=1, aux_out=hout, drop_p=0.1,
        drop_seed=7, drop_ld=256, drop_where=4)
    run("fwd NT 64->256 -> h only (eval)", P, 256, 64, 1, 1, bias=bias, bias_mode=1, write_act=1)
    del hout
    run("fwd NT 64->256 plain", P, 256, 64, 1, 1, bias=bias, bias_mode=1)
    run("fwd NT 256->64 plain", P, 64, 256, 1, 1)
    run("fwd NT 256->64 gelu+drop(A)", P, 64, 256, 1, 1, act_a=1, drop_p=0.1, drop_seed=7, drop_ld=256, drop_where=1)
    aux = torch.randn(P * 256, device=dev)
    run("dgrad NN 256->256 plain", P, 256, 256, 1, 0)
    run("dgrad NN 256->256 gelu'(aux)", P, 256, 256, 1, 0, epi_dact=1, aux=aux, ldaux=256)
    run("dgrad NN 256->256 gelu'+drop+colsum", P, 256, 256, 1, 0, epi_dact=1, aux=aux, ldaux=256, drop_p=0.1, drop_seed=7,
        drop_ld=256, drop_where=4, colsum=torch.empty((P // 128) * 256, device=dev))
    run("dgrad NN 256->256 * stored d + colsum", P, 256, 256, 1, 0, epi_dact=100, aux=aux, ldaux=256,
        colsum=torch.empty((P // 128) * 256, device=dev))
    run("dgrad NN 64->256 * stored d + colsum", P, 256, 64, 1, 0, epi_dact=100, aux=aux, ldaux=256,
        colsum=torch.empty((P // 128) * 256, device=dev))
    run("dgrad NN 64->256 gelu'+drop", P, 256, 64, 1, 0, epi_dact=1, aux=aux, ldaux=256, drop_p=0.1, drop_seed=7,
        drop_ld=256, drop_where=4)
    del aux
    run("wgrad TN 256x256 split128 plain", 256, 256, P, 0, 0, ksplit=128)
    run("wgrad TN 256x256 split128 gelu+drop(B)", 256, 256, P, 0, 0, ksplit=128, act_b=1, drop_p=0.1, drop_seed=7,
        drop_ld=256, drop_where=2)
    run("wgrad TN 64x256 split256 gelu+drop(B)", 64, 256, P, 0, 0, ksplit=256, act_b=1, drop_p=0.1, drop_seed=7,
        drop_ld=256, drop_where=2)
    run("wgrad TN 256x64 split256 plain", 256, 64, P, 0, 0, ksplit=256)


if __name__ == "__main__":
    main()
