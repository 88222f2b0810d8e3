#!/usr/bin/env python3
"""Soak of the fused FeedForward backward at the headline size: N rounds of forward + backward on fresh data, every
result computed twice (once with a copy stream hammering HBM beside it) and compared bitwise -- the fused kernels
reduce in fixed order -- and with the per-GEMM path (RPDE_FUSED_FF=0) on the same data and dropout seed.  The fused kernels wait for
their prefetched operands with hand-counted vmcnt; a wrong count would show here as a sporadic mismatch.
    python profiles/soak_ff_bwd.py [rounds] [batch]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "resolution-pde_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from models.custom_layer import FeedForward  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = "cuda:0"
torch.manual_seed(0)
ff = FeedForward(64, 4, n_layers=3, layer_norm=True, dropout=0.1).to(dev).train()
side = torch.cuda.Stream()
junk = torch.empty(64 << 20, device=dev)
worst = 0.0
names = ["dx"] + [n for n, _ in ff.named_parameters()]


def run(fused, r, hammer):
    os.environ["RPDE_FUSED_FF"] = fused
    torch.manual_seed(1000 + r)                           # the same dropout masks on every path
    xs = x.clone().requires_grad_(True)
    for p_ in ff.parameters():
        p_.grad = None
    if hammer:                                            # a copy stream hammering HBM beside the kernels
        with torch.cuda.stream(side):
            for _ in range(40):
                junk.copy_(junk.flip(0))
    ff(xs, residual=res).backward(g)
    torch.cuda.synchronize()
    del os.environ["RPDE_FUSED_FF"]
    return [xs.grad.clone()] + [p_.grad.clone() for p_ in ff.parameters()]


for r in range(rounds):
    x = torch.randn(B, 256, 256, 64, device=dev) * (10.0 ** ((r % 5) - 2))
    res = torch.randn_like(x)
    g = torch.randn_like(x)
    a, b, c = run("1", r, False), run("1", r, True), run("0", r, False)
    # the fused kernels reduce in fixed order: two runs on the same data are bitwise equal unless a wait was too short
    same = all(torch.equal(u, v) for u, v in zip(a, b))
    rels = {n: float((u - v).norm() / v.norm().clamp_min(1e-30)) for n, u, v in zip(names, a, c)}
    rel = max(rels.values())
    worst = max(worst, rel)
    print(f"round {r:2d}: repeat bitwise equal {same}; vs per-GEMM worst {rel:.2e} ({max(rels, key=rels.get)})", flush=True)
    assert same and rel < 1e-4 and all(torch.isfinite(t).all() for t in a), (same, rels)
print(f"soak OK: {rounds} rounds, worst difference to the per-GEMM path {worst:.2e} (sums over 2 M points in another order)")
