"""scratch probe: host time inside FlatAdamW.step for FFNO1D variants created one after the other in one process"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "resolution-pde_amd"))
import torch
from rpde.launch import limit_host_threads
limit_host_threads()
from models.ffno import FFNO1D
from rpde.optim import FlatAdamW
from utils.loss import RelativeL2Loss
dev = "cuda:0"
order = [(0.0, 2, False), (0.0, 3, True), (0.2, 3, True)]
if len(sys.argv) > 1:
    order = order[::-1]
for drop, nff, ln in order:
    torch.manual_seed(0)
    m = FFNO1D(1, 1, width=128, n_layers=4, n_modes=64, factor=4, ff_weight_norm=True, n_ff_layers=nff, layer_norm=ln, dropout=drop).to(dev).train()
    for cap in (False, True):
        opt = FlatAdamW(m.parameters(), lr=1e-3, capturable=cap)
        tg = [0.0]
        og = opt.bucket.gather
        def timed_gather(og=og):
            t = time.perf_counter(); og(); tg[0] += time.perf_counter() - t
        opt.bucket.gather = timed_gather
        loss_fn = RelativeL2Loss()
        x = torch.randn(16, 1, 512, device=dev); y = torch.randn_like(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gpu_opt = 0.0
        for i in range(25):
            if i == 5:
                torch.cuda.synchronize(); tg[0] = 0.0; to = 0.0; t_all = time.perf_counter()
            opt.zero_grad()
            loss_fn(m(x), y).backward()
            t2 = time.perf_counter()
            if i == 24: e0.record()
            opt.step()
            if i == 24: e1.record()
            if i >= 5: to += time.perf_counter() - t2
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t_all) / 20 * 1e3
        print(f"dropout {drop} n_ff {nff} ln {ln} capturable {cap}: wall {wall:.3f} ms/step; opt.step host {to/20*1e3:.3f} (gather {tg[0]/20*1e3:.3f}); GPU time of the last opt.step {e0.elapsed_time(e1):.3f} ms", flush=True)
