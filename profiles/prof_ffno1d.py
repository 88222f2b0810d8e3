"""FFNO1D (BASELINE config 2 shape: width 128, 4 layers, 64 modes, 3-layer FeedForward with LayerNorm) train steps at
B = 16, n = 512, for rocprofv3 --kernel-trace: which kernels make up the ~1.5 ms step"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "resolution-pde_amd"))
import torch
from models.ffno import FFNO1D
from rpde.optim import FlatAdamW
from utils.loss import RelativeL2Loss
dev = "cuda:0"
torch.manual_seed(0)
m = FFNO1D(1, 1, width=128, n_layers=4, n_modes=64, factor=4, ff_weight_norm=True, n_ff_layers=3, layer_norm=True, dropout=0.0).to(dev).train()
opt = FlatAdamW(m.parameters(), lr=1e-3)
loss_fn = RelativeL2Loss()
x = torch.randn(16, 1, 512, device=dev); y = torch.randn_like(x)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    opt.zero_grad()
    loss_fn(m(x), y).backward()
    opt.step()
torch.cuda.synchronize()
