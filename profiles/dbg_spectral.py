import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "resolution-pde_amd")):
    sys.path.insert(0, p)
import torch
from rpde import ops
sys.path.insert(0, os.path.join(REPO, "profiles"))
from spectral_bench import ref_forward

dev = "cuda:0"
torch.manual_seed(0)
for (M, N, K) in [(64, 64, 4), (64, 64, 8), (64, 64, 16), (64, 64, 20), (64, 32, 12), (256, 256, 20)]:
    C = 64
    x = torch.randn(2, M, N, C, device=dev)
    for which in ("both", "y", "x"):
        wy = torch.randn(C, C, K, 2, device=dev) * 0.1
        wx = torch.randn(C, C, K, 2, device=dev) * 0.1
        if which == "y": wx.zero_()
        if which == "x": wy.zero_()
        with torch.no_grad():
            out = ops.fspectral2d(x, wy, wx, K)
            ref = ref_forward(x.cpu(), wy.cpu(), wx.cpu(), K)
        e = float((out.cpu().double() - ref).norm() / ref.norm())
        # per-column-block / spatial error structure
        d = (out.cpu().double() - ref)[0]
        em = d.pow(2).sum(dim=(1, 2)).sqrt()[:20]
        print(f"M{M} N{N} K{K} {which}: rel {e:.3e}   row-err[:8] {[round(float(v),3) for v in em[:8]]}", flush=True)
