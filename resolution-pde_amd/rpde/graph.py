"""hipGraph capture of a whole training step (forward + loss + backward + optimizer).

The 1-D configurations (FNO1d / FFNO1D at batch 16) run ~100-170 small kernels per step.  Captured once and
replayed, the step costs one graph launch on the host.  Measured on MI355X (profiles/other_configs.py): in round 1,
with ~250 kernels of 8+ us each and torch's capturable AdamW (a dozen extra tiny kernels), no gain (1.54 ms eager
vs 1.71 ms replayed); with the round-2 kernels and rpde.optim.FlatAdamW(capturable=True) the GPU side is short
enough that host launches show: FNO1d 1024, B=16: 1.05 ms eager vs 0.72 ms replayed; FFNO1D 512, B=16: 1.62 vs
1.50 ms.
Everything the step does already is stream-ordered and allocation-free at the HIP level (workspaces come
from torch's caching allocator, DFT plans are created on first use), so the capture needs no changes in the
library -- only: plans and autotuned state must exist before capture (warm-up steps), the optimizer must be
capturable, and nothing in the step may depend on a host-side random draw.  FeedForward dropout draws its seed
on the host, which a replay would repeat; the dropout kernels therefore also mix a DEVICE counter into the seed
(rpde.ops.drop_epoch, rpde_ff_params.seed_epoch) and the captured step advances it first thing, so every
replay draws new masks and forward / backward of one step agree.  torch.nn.Dropout modules (none in the FNO /
FFNO families) would still freeze and are refused.

Round 4: the headline step (FFNO2D cfg3, dropout 0.1, B = 32, 256^2: ~250 launches per 31 ms) captured together
with FlatGradBucket.all_reduce_mean -- bench.py reports eager vs replayed; with 8 ranks on one host the launches
of a step are issued by 8 processes, which is the case the graph is for.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch


def _has_torch_dropout(model: torch.nn.Module) -> bool:
    """a torch.nn.Dropout that would really run (the FeedForward keeps nn.Dropout children for the state_dict layout
    only: its masks come from the fused kernels)"""
    from models.custom_layer import FeedForward
    owned = set()
    for m in model.modules():
        if isinstance(m, FeedForward):
            owned.update(id(c) for c in m.modules())
    return any(isinstance(m, torch.nn.Dropout) and m.p > 0.0 and m.training and id(m) not in owned for m in model.modules())


class GraphedTrainStep:
    """step = GraphedTrainStep(model, loss_fn, optimizer, x_example, y_example); loss = step(x, y)

    x / y of later calls must have the example's shape and dtype (one graph per shape; build one instance per
    resolution for multi-resolution training).  `after_backward` (e.g. FlatGradBucket.all_reduce_mean) runs
    inside the captured region between backward and the optimizer step.

    warmup = 0: the caller has already run eager steps of this very shape (plans, allocator pools and optimizer state
    exist) -- the construction then applies NO optimizer step of its own, which is what a training loop needs
    (train/training.py: the first steps of a shape run eagerly, the graph is captured in between two real steps).
    With warmup = 0 the caller must not keep a loss / output tensor of an earlier eager step alive across the
    construction: its autograd graph pins the parameters' gradient accumulators to the stream it ran on, and the
    captured backward would then synchronise with that stream -- an illegal dependency inside a capture.
    A learning-rate schedule: with rpde.optim.FlatAdamW(capturable=True) the values live on the device and are
    refreshed before a replay when a scheduler has moved them; other optimizers bake them into the graph, and the
    replay is refused once they change."""

    def __init__(self, model, loss_fn, optimizer, x: torch.Tensor, y: torch.Tensor, warmup: int = 3,
                 after_backward: Optional[Callable[[], None]] = None):
        if not x.is_cuda:
            raise ValueError("GraphedTrainStep needs HIP tensors")
        if _has_torch_dropout(model):
            raise ValueError("GraphedTrainStep: a torch.nn.Dropout in training mode draws from the host-side generator; "
                             "a captured graph would replay one mask.  Use eager steps for this model.")
        for group in optimizer.param_groups:
            if "capturable" in group and not group["capturable"]:
                raise ValueError("GraphedTrainStep: build the optimizer with capturable=True")
        self.model, self.loss_fn, self.optimizer = model, loss_fn, optimizer
        self.after_backward = after_backward
        self.x, self.y = x.clone(), y.clone()
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side):
            for _ in range(max(0, warmup)):      # creates DFT plans, sizes the allocator pools, primes optimizer state
                self._eager()
        torch.cuda.current_stream(x.device).wait_stream(side)
        torch.cuda.synchronize(x.device)
        self.graph = torch.cuda.CUDAGraph()
        self._captured_hyper = self._hyper()
        with torch.cuda.graph(self.graph):
            self.loss = self._eager()

    def _hyper(self):
        """learning rate / weight decay reach the optimizer kernel as launch arguments: a replay repeats the captured ones"""
        return [(float(g.get("lr", 0.0)), float(g.get("weight_decay", 0.0))) for g in self.optimizer.param_groups]

    def _eager(self) -> torch.Tensor:
        from . import ops
        ops.drop_epoch(self.x.device).add_(1)        # new dropout masks at every replay (device-side counter)
        self.optimizer.zero_grad(set_to_none=False)
        loss = self.loss_fn(self.model(self.x), self.y)
        loss.backward()
        if self.after_backward is not None:
            self.after_backward()
        self.optimizer.step()
        return loss.detach()

    def __call__(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        if x.shape != self.x.shape or y.shape != self.y.shape:
            raise ValueError(f"GraphedTrainStep was captured for {tuple(self.x.shape)} / {tuple(self.y.shape)}")
        if self._hyper() != self._captured_hyper and hasattr(self.optimizer, "sync_hyper_to_device"):
            self.optimizer.sync_hyper_to_device()          # device-side lr / weight decay: the graph reads them
            self._captured_hyper = self._hyper()
        if self._hyper() != self._captured_hyper:
            raise RuntimeError(f"GraphedTrainStep: lr / weight_decay changed after capture ({self._captured_hyper} -> "
                               f"{self._hyper()}); the graph would keep training at the captured values -- build a new "
                               "GraphedTrainStep (one per learning-rate plateau) or step eagerly")
        self.x.copy_(x, non_blocking=True)
        self.y.copy_(y, non_blocking=True)
        self.graph.replay()
        return self.loss
