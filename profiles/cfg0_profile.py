import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "resolution-pde_amd"))
import torch
from models.fno import FNO1d
from models.ffno import FFNO1D
from utils.loss import RelativeL2Loss
dev = "cuda:0"
which = sys.argv[1] if len(sys.argv) > 1 else "fno1d"
torch.manual_seed(0)
if which == "fno1d":
    m = FNO1d(1, 1, modes=16, width=64).to(dev).train(); x = torch.randn(16, 1, 1024, device=dev)
else:
    m = FFNO1D(1, 1, width=128, n_layers=4, n_modes=64, factor=4, ff_weight_norm=True, n_ff_layers=2, layer_norm=False, dropout=0.0).to(dev).train()
    x = torch.randn(16, 1, 512, device=dev)
y = torch.randn_like(x)
opt = torch.optim.AdamW(m.parameters(), lr=1e-3); lf = RelativeL2Loss()
for _ in range(12):
    opt.zero_grad(set_to_none=True); lf(m(x), y).backward(); opt.step()
torch.cuda.synchronize()
