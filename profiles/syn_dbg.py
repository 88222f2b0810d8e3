import os, sys, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "resolution-pde_amd")):
    sys.path.insert(0, p)
import torch
from rpde import ops
x = torch.randn(32, 256, 256, 64, device="cuda")
wy = torch.randn(64, 64, 20, 2, device="cuda") * 0.1
wx = torch.randn(64, 64, 20, 2, device="cuda") * 0.1
with torch.no_grad():
    for _ in range(3): ops.fspectral2d(x, wy, wx, 20)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.fspectral2d(x, wy, wx, 20)
    e1.record(); torch.cuda.synchronize()
print(os.environ.get("RPDE_SYN_DBG"), "fwd ms", e0.elapsed_time(e1) / 20, flush=True)
