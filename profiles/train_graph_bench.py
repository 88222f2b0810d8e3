"""train() (the reference's loop, train/training.py) eager against training.graph=true: seconds per epoch of
BASELINE configs[0] / [1] models at batch 16 on resident synthetic data (200 batches per epoch, 3 epochs, the last timed)
    python profiles/train_graph_bench.py"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "resolution-pde_amd"))
import torch  # noqa: E402
from rpde.launch import freeze_setup_garbage, limit_host_threads  # noqa: E402
limit_host_threads()
from models.ffno import FFNO1D  # noqa: E402
from models.fno import FNO1d  # noqa: E402
from rpde.optim import FlatAdamW  # noqa: E402
from train import training  # noqa: E402

dev = torch.device("cuda:0")
cases = [("cfg0 FNO1d(modes 16, width 64) grid 1024", lambda: FNO1d(1, 1, modes=16, width=64), 1024),
         ("cfg1 FFNO1D(ffno_1d.yaml) grid 512", lambda: FFNO1D(1, 1, width=128, n_layers=4, n_modes=64, factor=4, ff_weight_norm=True,
                                                               n_ff_layers=3, layer_norm=True, dropout=0.2), 512)]
for name, make, n in cases:
    batches = [(torch.randn(16, 1, n, device=dev), torch.randn(16, 1, n, device=dev)) for _ in range(200)]
    for graph in (False, True):
        torch.manual_seed(0)
        m = make().to(dev)
        opt = FlatAdamW(m.parameters(), lr=1e-3, weight_decay=1e-4, capturable=graph)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=100, eta_min=1e-5)
        freeze_setup_garbage()
        real_train = training.train
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hist = real_train(m, batches, [], opt, sched, epochs=3, device=dev, graph=graph)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name}  graph={graph!s:5}: {dt / 600 * 1e3:7.3f} ms per step over 3 epochs x 200 batches "
              f"({16 * 600 / dt:9.0f} samples/s); train loss {hist[0][0]:.4f} -> {hist[0][-1]:.4f}", flush=True)
