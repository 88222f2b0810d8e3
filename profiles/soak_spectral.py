#!/usr/bin/env python3
"""Soak of the fused spectral layer (k_dft_analysis_sq_h2 waits for its asm loads with hand-counted vmcnt): forward +
backward on grids of 64 .. 256 incl. those that leave waves without a duty, every result computed twice -- once with a
copy stream hammering HBM beside it -- and compared bitwise (all reductions run in fixed order), and against the
per-GEMM path.
    python profiles/soak_spectral.py [rounds]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "resolution-pde_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from rpde import ops  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = "cuda:0"
side = torch.cuda.Stream()
junk = torch.empty(64 << 20, device=dev)
K = 20


def run(x, wy, wx, g, hammer, fused="1"):
    os.environ["RPDE_FUSED_SPECTRAL"] = fused
    xs, a, b = x.clone().requires_grad_(True), wy.clone().requires_grad_(True), wx.clone().requires_grad_(True)
    if hammer:
        with torch.cuda.stream(side):
            for _ in range(20):
                junk.copy_(junk.flip(0))
    out, skip = ops.fspectral2d(xs, a, b, K, with_skip=True)
    (out * g + skip * (0.5 * g)).sum().backward()
    torch.cuda.synchronize()
    del os.environ["RPDE_FUSED_SPECTRAL"]
    return out.detach(), xs.grad, a.grad, b.grad


worst = 0.0
for r in range(rounds):
    for B, n in ((32, 256), (8, 192), (8, 160), (16, 128), (8, 96), (32, 64), (3, 224)):
        torch.manual_seed(100 * r + n)
        x = torch.randn(B, n, n, 64, device=dev)
        wy = torch.randn(64, 64, K, 2, device=dev) * 0.1
        wx = torch.randn(64, 64, K, 2, device=dev) * 0.1
        g = torch.randn_like(x)
        a, b, c = run(x, wy, wx, g, False), run(x, wy, wx, g, True), run(x, wy, wx, g, False, "0")
        same = all(torch.equal(u, v) for u, v in zip(a, b))
        rel = max(float((u - v).norm() / v.norm()) for u, v in zip(a, c))
        worst = max(worst, rel)
        print(f"round {r} grid {n:3d} B={B:2d}: repeat bitwise equal {same}; vs per-GEMM path {rel:.2e}", flush=True)
        assert same and rel < 3e-6, (same, rel)
print(f"soak OK: worst difference to the per-GEMM path {worst:.2e}")
