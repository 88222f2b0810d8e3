import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "resolution-pde_amd")):
    sys.path.insert(0, p)
import torch
from models.ffno import FFNO2D
from utils.loss import RelativeL2Loss
from utils.synthetic import advance, random_fields
from rpde import ops
CFG = dict(in_channels=1, out_channels=1, width=64, n_layers=2, n_modes=8, factor=4, ff_weight_norm=True, n_ff_layers=3,
           layer_norm=True, dropout=0.0)
dev = "cuda:0"
torch.manual_seed(0)
model = FFNO2D(**CFG).to(dev).train()
x = random_fields(4, 64, 2, seed=7).to(dev); y = advance(x.cpu(), 2).to(dev)
with torch.no_grad():
    full = model(x)
    halves = torch.cat([model(x[:2]), model(x[2:])])
print("forward full vs halves", float((full - halves).norm() / full.norm()))
def grads(xb, yb):
    for p in model.parameters(): p.grad = None
    RelativeL2Loss()(model(xb), yb).backward()
    return torch.cat([torch.view_as_real(p.grad).flatten() if p.grad.is_complex() else p.grad.flatten() for p in model.parameters()])
gf = grads(x, y)
gh = 0.5 * (grads(x[:2], y[:2]) + grads(x[2:], y[2:]))
print("grad full vs mean of halves", float((gf - gh).norm() / gf.norm()))
names = [n for n, _ in model.named_parameters()]
off = 0
for n, p in model.named_parameters():
    k = p.numel()
    a, b = gf[off:off + k], gh[off:off + k]
    e = float((a - b).norm() / a.norm().clamp_min(1e-30))
    if e > 1e-5: print("  ", n, e)
    off += k
# spectral layer alone
xs = torch.randn(4, 64, 64, 64, device=dev)
wy = torch.randn(64, 64, 8, 2, device=dev) * 0.1; wx = torch.randn(64, 64, 8, 2, device=dev) * 0.1
with torch.no_grad():
    a = ops.fspectral2d(xs, wy, wx, 8); b = torch.cat([ops.fspectral2d(xs[:2], wy, wx, 8), ops.fspectral2d(xs[2:], wy, wx, 8)])
print("spectral fwd full vs halves", float((a - b).norm() / a.norm()))
