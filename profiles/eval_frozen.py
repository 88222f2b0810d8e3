import sys, time, torch
sys.path.insert(0, 'resolution-pde_amd')
from models.ffno import FFNO2D
from rpde import ops
from utils.synthetic import random_fields
m = FFNO2D(in_channels=1, out_channels=1, width=64, n_layers=4, n_modes=20, factor=4, ff_weight_norm=True, n_ff_layers=3, layer_norm=True, dropout=0.0).cuda().eval()
for B, R in ((32, 256), (1, 256), (4, 64)):
    x = random_fields(B, R, 2, seed=1).cuda()
    def t(n=30):
        for _ in range(5): m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): m(x)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    with torch.no_grad():
        a = t()
        with ops.frozen_weights():
            b = t()
    print(f"eval forward B={B} R={R}: plain {a:.3f} ms  frozen {b:.3f} ms")
