import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/resolution-pde_amd')
import torch, bench
from rpde import _lib
import ctypes as C
dev = torch.device('cuda', 0)
B = 32
P, K, N = B * 65536, 256, 256
lib = _lib.load()
for kind in ("randn", "zeros", "const"):
    if kind == "randn": h1 = torch.randn(P, K, device=dev)
    elif kind == "zeros": h1 = torch.zeros(P, K, device=dev)
    else: h1 = torch.full((P, K), 0.5, device=dev)
    w = torch.randn(N, K, device=dev) * 0.06
    b = torch.randn(N, device=dev)
    h2 = torch.empty(P, N, device=dev); d2 = torch.empty(P, N, device=dev)
    img = torch.empty(lib.rpde_split_weights_bytes(N, K), dtype=torch.uint8, device=dev)
    _lib.check(lib.rpde_split_weights(w.data_ptr(), 1, K, N, K, img.data_ptr(), _lib.stream_ptr()), "s")
    d = _lib.GemmDesc()
    d.batch, d.zdiv, d.ksplit, d.alpha = 1, 1, 1, 1.0
    d.A, d.B, d.C = h1.data_ptr(), w.data_ptr(), h2.data_ptr()
    d.M, d.N, d.K, d.a_kmajor, d.b_kmajor = P, N, K, 1, 1
    d.lda, d.ldb, d.ldc = K, K, N
    d.bias, d.bias_mode, d.write_act, d.aux_out = b.data_ptr(), 1, 1, d2.data_ptr()
    d.drop_p, d.drop_seed, d.drop_ld, d.drop_where = 0.1, 12345, N, 4
    d.b_split = img.data_ptr()
    print(kind, round(bench._time_gemm(d, 20), 4), "ms", flush=True)
    del h1, h2, d2
