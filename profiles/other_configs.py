#!/usr/bin/env python3
"""Timing of the other BASELINE configs (parity-test cases, not bench lines): train step of FNO1d (cfg0) and
FFNO1D (cfg1), eval forward + 5-step rollout of FNO2d at 512^2 (cfg4); round 4: the reference's SHIPPED FFNO2D yaml
(conf/model/ffno_2d/ffno_2d.yaml: n_modes 64 -- off the fused spectral path) beside the stock-ATen
run of the same op sequence on this GPU.  GPU events, synthetic data.
    python profiles/other_configs.py [yaml]        (yaml: only the shipped-yaml section)"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "resolution-pde_amd"))
import torch  # noqa: E402
from rpde.launch import limit_host_threads  # noqa: E402
limit_host_threads()                      # the eager legs are host-bound: no oversubscribed thread pool beside them
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




def shipped_yaml():
    """FFNO2D(width 64, 4 layers, n_modes 64, 3 FeedForward layers, LayerNorm, dropout 0.1) at 256^2: the training step and
    one layer's forward_fourier, this library against the oracle's op sequence on stock ATen (rocFFT + hipBLASLt)"""
    sys.path.insert(0, REPO)
    from models.ffno import FFNO2D
    from oracle import reference_path as R          # baseline leg only (as bench.py's gpu_aten_baseline)
    from rpde import ops
    cfg = dict(in_channels=1, out_channels=1, width=64, n_layers=4, n_modes=64, factor=4, ff_weight_norm=True,
               n_ff_layers=3, layer_norm=True, dropout=0.1)
    K = 64
    for B in (8, 32):
        x = torch.randn(B, 256, 256, 64, device=dev)
        wy = torch.randn(64, 64, K, 2, device=dev) * 0.1
        wx = torch.randn(64, 64, K, 2, device=dev) * 0.1
        alg = 4.0 * B * 256 * 256 * 2 * 64 + 2 * 8.0 * 64 * 64 * K
        with torch.no_grad():
            ms = timed(lambda: ops.fspectral2d(x, wy, wx, K), iters=10, warm=3)
            ma = timed(lambda: R.fspectral2d_fourier(x, wy, wx, K), iters=5, warm=2)
        print(f"yaml FSpectralConv2d.forward_fourier K=64 256^2 B={B:3d}: {ms:8.3f} ms = {alg / ms / 1e6:7.0f} GB/s algorithmic "
              f"(frac of 8 TB/s {alg / ms / 1e6 / 8000:.3f}) | stock ATen {ma:8.3f} ms  x{ma / ms:.1f}", flush=True)
        del x
        torch.manual_seed(0)
        m = FFNO2D(**cfg).to(dev).train()
        xb = torch.randn(B, 1, 256, 256, device=dev)
        yb = torch.randn_like(xb)
        ms = timed(train_step(m, xb, yb), iters=10, warm=3)
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        params = R.make_params(sd)
        opt = torch.optim.AdamW(list(params.values()), lr=1e-3)

        def astep():
            opt.zero_grad()
            R.relative_l2(R.ffno2d_forward(params, xb, 4, K, 3, True, 0.1, training=True), yb).backward()
            opt.step()
        ma = timed(astep, iters=3, warm=2)
        print(f"yaml FFNO2D(n_modes 64) 256^2 train step  B={B:3d}: {ms:8.3f} ms/step {B / ms * 1e3:8.1f} samples/s | stock ATen "
              f"{ma:8.3f} ms/step {B / ma * 1e3:8.1f} samples/s  x{ma / ms:.1f}", flush=True)
        del m, params, opt


if len(sys.argv) > 1 and sys.argv[1] == "yaml":
    shipped_yaml()
    raise SystemExit(0)
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
# the same model with the reference's yaml settings (conf/model/ffno_1d/ffno_1d.yaml: 3 FeedForward layers, LayerNorm, dropout 0.2)
m = FFNO1D(1, 1, width=128, n_layers=4, n_modes=64, factor=4, ff_weight_norm=True, n_ff_layers=3, layer_norm=True,
           dropout=0.2).to(dev).train()
for B in (16,):
    x = torch.randn(B, 1, 512, device=dev)
    ms = timed(train_step(m, x, torch.randn_like(x)))
    mg = timed(graphed(m, x, torch.randn_like(x)))
    print(f"cfg1 FFNO1D 512 yaml   B={B:4d}: {ms:8.3f} ms/step  {B / ms * 1e3:10.0f} samples/s | hipGraph {mg:8.3f} ms/step "
          f"{B / mg * 1e3:10.0f} samples/s   (ffno_1d.yaml: 3 FeedForward layers, LayerNorm, dropout 0.2)", flush=True)
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
del m, x
shipped_yaml()
