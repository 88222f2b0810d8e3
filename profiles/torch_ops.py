"""Which torch (ATen) ops still run inside a training step -- the native kernels between the HIP entry points.
    python profiles/torch_ops.py [ffno1d|fno1d|ffno2d]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "resolution-pde_amd"))
import torch  # noqa: E402
from models.ffno import FFNO1D, FFNO2D  # noqa: E402
from models.fno import FNO1d  # noqa: E402
from rpde.optim import FlatAdamW  # noqa: E402
from utils.loss import RelativeL2Loss  # noqa: E402

dev = "cuda:0"
which = sys.argv[1] if len(sys.argv) > 1 else "ffno1d"
torch.manual_seed(0)
if which == "fno1d":
    m = FNO1d(1, 1, modes=16, width=64).to(dev).train(); x = torch.randn(16, 1, 1024, device=dev)
elif which == "ffno1d":
    m = FFNO1D(1, 1, width=128, n_layers=4, n_modes=64, factor=4, ff_weight_norm=True, n_ff_layers=2, layer_norm=False,
               dropout=0.0).to(dev).train()
    x = torch.randn(16, 1, 512, device=dev)
else:
    m = FFNO2D(1, 1, width=64, n_layers=4, n_modes=20, factor=4, ff_weight_norm=True, n_ff_layers=3, layer_norm=True,
               dropout=0.1).to(dev).train()
    x = torch.randn(4, 1, 256, 256, device=dev)
y = torch.randn_like(x)
opt = FlatAdamW(m.parameters(), lr=1e-3)
lf = RelativeL2Loss()


def step():
    opt.zero_grad()
    lf(m(x), y).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA],
                            record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=48))
