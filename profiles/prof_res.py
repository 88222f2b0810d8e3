"""FFNO2D cfg3 (the headline model) training steps at another grid, for rocprofv3 --kernel-trace:
    python profiles/prof_res.py [res] [B] [steps]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "resolution-pde_amd")):
    sys.path.insert(0, p)
import torch
import bench
from models.ffno import FFNO2D
from rpde.optim import FlatAdamW
from utils.loss import RelativeL2Loss
dev = "cuda:0"
res = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
torch.manual_seed(0)
m = FFNO2D(**bench.CFG3).to(dev).train()
opt = FlatAdamW(m.parameters(), lr=1e-3)
loss_fn = RelativeL2Loss()
x = torch.randn(B, 1, res, res, device=dev); y = torch.randn_like(x)
for i in range(steps):
    opt.zero_grad()
    loss_fn(m(x), y).backward()
    opt.step()
torch.cuda.synchronize()
