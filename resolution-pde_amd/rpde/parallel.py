"""Data-parallel gradient exchange: one process per GPU, one flat fp32 bucket,
one all-reduce per step (torch.distributed 'nccl' = RCCL over xGMI on ROCm;
'gloo' in the CPU tests).

Every op of the hot path is per-sample; the only cross-sample coupling is the
batch mean of the loss, so ranks own disjoint samples and exchange nothing but
the gradient sum (6.8 MB for FFNO2D-m20-w64).  The parameters' ``.grad`` are
views into the bucket, so there is no pack / unpack copy around the
collective.  The reference's only multi-GPU mode is single-process
nn.DataParallel (main_2d.py:147-149); this is its one-process-per-GPU
counterpart with the same semantics for equal local batches (mean over the
concatenated batch).
"""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist


class FlatGradBucket:
    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        sizes = [p.numel() * (2 if p.is_complex() else 1) for p in self.params]
        padded = [(n + 3) // 4 * 4 for n in sizes]          # 16-byte aligned chunks (complex views need even offsets)
        # layout: all real parameters first, then the complex ones, each group in parameter order.  (One buffer for
        # the collective; rpde.optim.FlatAdamW mirrors the two regions in separate storages, because torch.save
        # refuses float and complex views of one storage in a state_dict.)
        order = [i for i, p in enumerate(self.params) if not p.is_complex()] + \
                [i for i, p in enumerate(self.params) if p.is_complex()]
        self.offsets: List[int] = [0] * len(self.params)
        off = 0
        for i in order:
            self.offsets[i] = off
            off += padded[i]
        self.n_real = sum(padded[i] for i, p in enumerate(self.params) if not p.is_complex())     # complex region: [n_real, end)
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self._views: List[torch.Tensor] = []
        self._touched = set()
        for i, (p, n) in enumerate(zip(self.params, sizes)):
            chunk = self.flat[self.offsets[i]:self.offsets[i] + n]
            if p.is_complex():
                view = torch.view_as_complex(chunk.view(*p.shape, 2))
            else:
                if p.dtype != torch.float32:
                    raise TypeError(f"expected fp32 / complex64 parameters, got {p.dtype}")
                view = chunk.view(p.shape)
            self._views.append(view)
            p.grad = view
            p.register_post_accumulate_grad_hook(lambda _p, i=i: self._touched.add(i))

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4

    def zero(self) -> None:
        """replaces optimizer.zero_grad(): zeroes the bucket and (re)attaches every .grad view"""
        self.flat.zero_()
        self._touched.clear()
        for p, v in zip(self.params, self._views):
            p.grad = v

    def detach_untouched(self) -> None:
        """call between backward (+ all-reduce) and optimizer.step(): a parameter that took no part in this step's
        graph (fourier_weight in mode='low-pass', an unused forecast_ff ...) gets grad None, which is what the
        reference's zero_grad() leaves it with, so AdamW applies neither moments nor weight decay to it.  Every
        rank runs the same graph, so the sets agree; the next zero() re-attaches the views."""
        for i, p in enumerate(self.params):
            if i not in self._touched:
                p.grad = None

    def all_reduce_mean(self) -> None:
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.mul_(1.0 / dist.get_world_size())


def shard_indices(n_items: int, world: int, rank: int) -> range:
    """contiguous, equal-sized shard of [0, n_items) (tail dropped so that all
    ranks run the same number of identical-shape steps)"""
    per = n_items // world
    return range(rank * per, (rank + 1) * per)
