"""FFNO2D with the reference's shipped yaml (conf/model/ffno_2d/ffno_2d.yaml: width 64, 4 layers, n_modes 64, 3-layer
FeedForward with LayerNorm, dropout 0.1, weight norm) at 256^2: train steps for rocprofv3 --kernel-trace
    python profiles/prof_yaml2d.py [B] [steps]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "resolution-pde_amd"))
import torch
from models.ffno import FFNO2D
from rpde.optim import FlatAdamW
from utils.loss import RelativeL2Loss
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
torch.manual_seed(0)
m = FFNO2D(1, 1, width=64, n_layers=4, n_modes=64, factor=4, ff_weight_norm=True, n_ff_layers=3, layer_norm=True,
           dropout=0.1).to(dev).train()
opt = FlatAdamW(m.parameters(), lr=1e-3)
loss_fn = RelativeL2Loss()
x = torch.randn(B, 1, 256, 256, device=dev); y = torch.randn_like(x)
for i in range(steps):
    opt.zero_grad()
    loss_fn(m(x), y).backward()
    opt.step()
torch.cuda.synchronize()
