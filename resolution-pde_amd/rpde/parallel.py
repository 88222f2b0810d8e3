"""Data-parallel gradient exchange: one process per GPU, one flat fp32 bucket,
one all-reduce per step (torch.distributed 'nccl' = RCCL over xGMI on ROCm;
'gloo' in the CPU tests).

Every op of the hot path is per-sample; the only cross-sample coupling is the
batch mean of the loss, so ranks own disjoint samples and exchange nothing but
the gradient sum (6.8 MB for FFNO2D-m20-w64).  After backward one multi-tensor
copy brings the gradients into the bucket and the parameters' ``.grad`` become
views into it, so the collective and the optimizer work on one flat buffer.  The reference's only multi-GPU mode is single-process
nn.DataParallel (main_2d.py:147-149); this is its one-process-per-GPU
counterpart with the same semantics for equal local batches (mean over the
concatenated batch).
"""
from __future__ import annotations

import os
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
        self._views: List[torch.Tensor] = []          # parameter-shaped (complex where the parameter is)
        self._chunks: List[torch.Tensor] = []         # the same memory as 1-D float32
        for i, (p, n) in enumerate(zip(self.params, sizes)):
            chunk = self.flat[self.offsets[i]:self.offsets[i] + n]
            if p.is_complex():
                view = torch.view_as_complex(chunk.view(*p.shape, 2))
            else:
                if p.dtype != torch.float32:
                    raise TypeError(f"expected fp32 / complex64 parameters, got {p.dtype}")
                view = chunk.view(p.shape)
            self._views.append(view)
            self._chunks.append(chunk)
            p.grad = None
        self._gathered = False

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4

    def zero(self) -> None:
        """replaces optimizer.zero_grad(): every .grad becomes None, so that backward ASSIGNS the first gradient of
        each parameter instead of launching one `grad += new` kernel per parameter into a pre-zeroed view"""
        for p in self.params:
            p.grad = None
        self._gathered = False

    def gather(self) -> None:
        """after backward: bring the gradients into the flat buffer with one multi-tensor copy and point every
        .grad at its view.  A parameter that took no part in this step's graph (fourier_weight in mode='low-pass', an
        unused forecast_ff ...) keeps grad None -- what the reference's zero_grad() leaves it with, so AdamW applies
        neither moments nor weight decay to it -- and its slot is zeroed (it still rides in the all-reduce; every
        rank runs the same graph, so the sets agree).  Idempotent."""
        if self._gathered:
            return
        src, dst, idle = [], [], []
        for p, v, c in zip(self.params, self._views, self._chunks):
            g = p.grad
            if g is None:
                idle.append(c)
            elif g.data_ptr() != v.data_ptr():
                g = g.contiguous()
                src.append((torch.view_as_real(g) if g.is_complex() else g).reshape(-1))
                dst.append(c)
        with torch.no_grad():
            if dst:
                torch._foreach_copy_(dst, src)
            if idle:
                torch._foreach_zero_(idle)
        for p, v in zip(self.params, self._views):
            if p.grad is not None:
                p.grad = v
        self._gathered = True

    def detach_untouched(self) -> None:
        """kept for callers of the first interface: gather() already leaves untouched parameters with grad None"""
        self.gather()

    def all_reduce_mean(self) -> None:
        self.gather()
        # (a single rank skips the collective unless RPDE_FORCE_DIST=1 asks for it: the 1-GPU RCCL rehearsal)
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or
                                                              os.environ.get("RPDE_FORCE_DIST") == "1"):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.mul_(1.0 / dist.get_world_size())


def shard_indices(n_items: int, world: int, rank: int) -> range:
    """contiguous, equal-sized shard of [0, n_items) (tail dropped so that all
    ranks run the same number of identical-shape steps)"""
    per = n_items // world
    return range(rank * per, (rank + 1) * per)
