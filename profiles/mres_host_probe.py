"""Where the host time of the mixed-resolution bench leg goes (bench.py:mres_leg): per step, the time to draw the next
batch, to start its copies, to launch the step, and the number of device-memory segments the caching allocator
had to get from / give back to the driver meanwhile."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "resolution-pde_amd")):
    sys.path.insert(0, p)
import torch
import bench
from models.ffno import FFNO2D
from rpde.optim import FlatAdamW
from rpde.parallel import FlatGradBucket
from utils.loss import RelativeL2Loss
from train.mres_training import ResolutionGroupedDataLoader, SimpleDataset
from utils.synthetic import markov_pairs

dev = torch.device("cuda:0")
torch.manual_seed(0)
B = 32
model = FFNO2D(**bench.CFG3).to(dev).train()
opt = FlatAdamW(model.parameters(), lr=1e-3)
bucket = opt.bucket
loss_fn = RelativeL2Loss()
per_res = 4 * B
samples = markov_pairs({64: per_res, 128: per_res, 256: per_res}, 2, 4321)
loader = ResolutionGroupedDataLoader(SimpleDataset(samples), B, shuffle=True, seed=0, verbose=False, pin_memory=True)
copy_stream = torch.cuda.Stream(device=dev)
main_stream = torch.cuda.current_stream(dev)


import gc
_gc_log, _gc_t = [], [0.0]


def _gc_cb(phase, info):
    if phase == "start":
        _gc_t[0] = time.perf_counter()
    else:
        _gc_log.append((info["generation"], (time.perf_counter() - _gc_t[0]) * 1e3, info["collected"]))


gc.callbacks.append(_gc_cb)
if os.environ.get("PROBE_GC_FREEZE") == "1":
    gc.collect(); gc.freeze()


def stats():
    s = torch.cuda.memory_stats(dev)
    return s["num_device_alloc"], s["num_device_free"], s["reserved_bytes.all.current"] >> 20


for ep in range(3):
    it = iter(loader)
    torch.cuda.synchronize()
    rows = []
    t_ep = time.perf_counter()
    nxt = None
    while True:
        a0 = stats()
        t0 = time.perf_counter()
        try:
            xb, yb = next(it)
        except StopIteration:
            break
        t1 = time.perf_counter()
        with torch.cuda.stream(copy_stream):
            xd, yd = xb.to(dev, non_blocking=True), yb.to(dev, non_blocking=True)
            ready = torch.cuda.Event(); ready.record(copy_stream)
        t2 = time.perf_counter()
        main_stream.wait_event(ready)
        xd.record_stream(main_stream); yd.record_stream(main_stream)
        bucket.zero()
        loss_fn(model(xd), yd).backward()
        bucket.all_reduce_mean()
        opt.step()
        t3 = time.perf_counter()
        a1 = stats()
        rows.append((int(xb.shape[-1]), (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, a1[0] - a0[0], a1[1] - a0[1], a1[2]))
    torch.cuda.synchronize()
    print(f"epoch {ep}: {(time.perf_counter() - t_ep) * 1e3:.1f} ms; collections (generation, ms, collected): "
          + ", ".join("(%d, %.1f, %d)" % g for g in _gc_log if g[1] > 1.0))
    _gc_log.clear()
    for r in rows:
        print("  res %4d  draw %7.2f ms  copy-start %7.2f ms  launch %7.2f ms  segments +%d -%d  reserved %d MiB" % r)
