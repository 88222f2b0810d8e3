#!/usr/bin/env python3
"""Timing of the other BASELINE configs (parity-test cases, not bench lines): train step of FNO1d (cfg0) and
FFNO1D (cfg1), eval forward + 5-step rollout of FNO2d at 512^2 (cfg4).  GPU events, synthetic data."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "resolution-pde_amd"))
import torch  # noqa: E402
from models.ffno import FFNO1D  # noqa: E402
from models.fno import FNO1d, FNO2d  # noqa: E402
from utils.loss import RelativeL2Loss  # noqa: E402

dev = torch.device("cuda", 0)
FLAT = os.environ.get("RPDE_FLAT_ADAMW", "1") != "0"        # rpde.optim.FlatAdamW (one kernel per step) or torch.optim.AdamW
from rpde.optim import FlatAdamW  # noqa: E402


def timed(fn, iters=30, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def train_step(model, x, y):
    opt = FlatAdamW(model.parameters(), lr=1e-3) if FLAT else torch.optim.AdamW(model.parameters(), lr=1e-3)
    loss_fn = RelativeL2Loss(size_average=True)

    def step():
        opt.zero_grad(set_to_none=not FLAT)
        loss_fn(model(x), y).backward()
        opt.step()
    return step


def graphed(model, x, y):
    from rpde.graph import GraphedTrainStep
    opt = (FlatAdamW(model.parameters(), lr=1e-3, capturable=True) if FLAT
           else torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True))
    step = GraphedTrainStep(model, RelativeL2Loss(size_average=True), opt, x, y)
    return lambda: step(x, y)


torch.manual_seed(0)
m = FNO1d(1, 1, modes=16, width=64).to(dev).train()
for B in (16, 256):
    x = torch.randn(B, 1, 1024, device=dev)
    ms = timed(train_step(m, x, torch.randn_like(x)))
    mg = timed(graphed(m, x, torch.randn_like(x)))
    print(f"cfg0 FNO1d 1024 train  B={B:4d}: {ms:8.3f} ms/step  {B / ms * 1e3:10.0f} samples/s | hipGraph {mg:8.3f} ms/step "
          f"{B / mg * 1e3:10.0f} samples/s", flush=True)
m = FFNO1D(1, 1, width=128, n_layers=4, n_modes=64, factor=4, ff_weight_norm=True, n_ff_layers=2, layer_norm=False,
           dropout=0.0).to(dev).train()
for B in (16, 256):
    x = torch.randn(B, 1, 512, device=dev)
    ms = timed(train_step(m, x, torch.randn_like(x)))
    mg = timed(graphed(m, x, torch.randn_like(x)))
    print(f"cfg1 FFNO1D 512 train  B={B:4d}: {ms:8.3f} ms/step  {B / ms * 1e3:10.0f} samples/s | hipGraph {mg:8.3f} ms/step "
          f"{B / mg * 1e3:10.0f} samples/s", flush=True)
m = FNO2d(1, 1, modes1=12, modes2=12, width=32).to(dev).eval()
for B in (4, 16):
    x = torch.randn(B, 1, 512, 512, device=dev)
    with torch.no_grad():
        ms = timed(lambda: m(x))
        print(f"cfg4 FNO2d 512^2 eval  B={B:4d}: {ms:8.3f} ms/fwd   {B / ms * 1e3:10.0f} samples/s", flush=True)

        def roll():
            s = x
            for _ in range(5):
                s = m(s)
        ms = timed(roll, iters=5)
        print(f"cfg4 FNO2d 512^2 5-step rollout B={B}: {ms:8.3f} ms", flush=True)
