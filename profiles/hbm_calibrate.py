#!/usr/bin/env python3
"""HBM calibration on the box: pure write (fill), copy (read+write), pure read (sum) of 1 GiB fp32 tensors."""
import torch
dev = "cuda:0"
n = 1 << 28
x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
def t(fn, iters=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
gb = n * 4 / 1e9
ms = t(lambda: x.fill_(1.0)); print(f"fill  (write {gb:.2f} GB): {ms:.3f} ms  {gb / ms:.2f} TB/s")
ms = t(lambda: y.copy_(x)); print(f"copy  (r+w {2 * gb:.2f} GB): {ms:.3f} ms  {2 * gb / ms:.2f} TB/s")
ms = t(lambda: x.sum()); print(f"sum   (read {gb:.2f} GB): {ms:.3f} ms  {gb / ms:.2f} TB/s")
ms = t(lambda: torch.add(x, y, out=y)); print(f"add   (2r+1w {3 * gb:.2f} GB): {ms:.3f} ms  {3 * gb / ms:.2f} TB/s")
