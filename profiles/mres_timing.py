#!/usr/bin/env python3
"""FFNO2D cfg3 train step at the three resolutions of the multi-resolution configuration (BASELINE configs[3])."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "resolution-pde_amd"))
import torch  # noqa: E402
from rpde.launch import limit_host_threads  # noqa: E402
limit_host_threads()                      # the eager legs are host-bound: no oversubscribed thread pool beside them
import bench  # noqa: E402
from models.ffno import FFNO2D  # noqa: E402
from utils.loss import RelativeL2Loss  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = FFNO2D(**bench.CFG3).to(dev).train()
from rpde.optim import FlatAdamW  # noqa: E402
opt = FlatAdamW(model.parameters(), lr=1e-3)
loss_fn = RelativeL2Loss(size_average=True)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
RES = [int(r) for r in sys.argv[2].split(',')] if len(sys.argv) > 2 else [64, 128, 256]
for res in RES:
    x, y = bench.synth_batch(B, res, 7, dev)

    def step():
        opt.zero_grad()
        loss_fn(model(x), y).backward()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"res {res:3d}^2  B={B}: {ms:8.3f} ms/step  {B / ms * 1e3:9.1f} samples/s  {B * res * res / ms * 1e3 / 1e6:8.2f} Mpoints/s", flush=True)
