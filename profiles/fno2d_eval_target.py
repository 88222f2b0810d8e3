import sys
sys.path.insert(0, '/root/repo/resolution-pde_amd')
import torch
from models.fno import FNO2d
dev = torch.device('cuda', 0)
torch.manual_seed(0)
m = FNO2d(1, 1, modes1=12, modes2=12, width=32).to(dev).eval()
x = torch.randn(8, 1, 512, 512, device=dev)
with torch.no_grad():
    for _ in range(5):
        m(x)
torch.cuda.synchronize()
