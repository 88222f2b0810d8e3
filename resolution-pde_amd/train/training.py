"""train() / evaluate() with the call shape of the reference's loops
(reference: train/training.py:19-88, 93-175): zero_grad -> model(x) ->
RelativeL2Loss -> backward -> optimizer.step per batch, validation pass,
scheduler step per epoch.

Differences, all on the host side of the hot path:
  * the loss is accumulated on the device and read back once per epoch (the
    reference calls ``loss.item()`` every step, a device sync that caps
    multi-GPU scaling);
  * wandb is optional: metrics go to ``log`` (a callable taking a dict) and
    to stdout as JSON lines;
  * with ``torch.distributed`` initialised, gradients are averaged through a
    flat bucket (rpde.parallel) -- one all-reduce per step over RCCL/xGMI.
"""
from __future__ import annotations

import json
import os
from typing import Callable, Optional

import torch
import torch.distributed as dist

from rpde.ops import frozen_weights
from rpde.parallel import FlatGradBucket
from utils.loss import RelativeL2Loss


def _dist_on() -> bool:
    # (RPDE_FORCE_DIST=1: a single rank still runs its collectives -- the 1-GPU RCCL rehearsal, rpde/entry.py)
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or
                                                              os.environ.get("RPDE_FORCE_DIST") == "1")


def _mean_over_ranks(total: torch.Tensor, count: int) -> float:
    """sum(total) / sum(count) over the ranks; NaN when nothing was counted anywhere (an empty loader is an error
    of the caller's data split, never a perfect score)"""
    t = torch.stack([total.detach().double().reshape(()), torch.tensor(float(count), dtype=torch.float64,
                                                                       device=total.device)])
    if _dist_on():
        dist.all_reduce(t)
    if float(t[1]) == 0.0:
        return float("nan")
    return float(t[0] / t[1])


GRAPH_AFTER = 2          # eager steps of a batch shape before its step is captured (plans, allocator pools, moments)


def train(model, train_loader, val_loader, optimizer, scheduler, y_normalizer=None, use_normalizer=False, time=1,
          model_type="ffno", epochs=100, device="cuda", log: Optional[Callable[[dict], None]] = None,
          graph: Optional[bool] = None):
    """graph (default: environment RPDE_TRAIN_GRAPH=1): replay the step -- zero_grad, forward, loss, backward, gradient
    all-reduce, optimizer -- as one hipGraph per batch shape (rpde.graph.GraphedTrainStep) once GRAPH_AFTER eager
    steps of that shape have run.  The small 1-D configurations are bound by the host's launches (FNO1d-1024 at
    batch 16: 1.1-1.3 ms per eager step, 0.69 ms replayed); same arithmetic, same order.  Needs
    rpde.optim.FlatAdamW(capturable=True) (its learning rate lives on the device, so the per-epoch scheduler step is
    followed without a new capture); silently stays eager otherwise."""
    loss_fn = RelativeL2Loss(size_average=True)
    # rpde.optim.FlatAdamW brings its own bucket (its gradients, parameters and moments share one flat layout)
    bucket = getattr(optimizer, "bucket", None) or FlatGradBucket(model.parameters())
    if graph is None:
        graph = os.environ.get("RPDE_TRAIN_GRAPH") == "1"
    graph = bool(graph) and hasattr(optimizer, "sync_hyper_to_device") and getattr(optimizer, "_step_dev", None) is not None
    if graph and _dist_on() and dist.get_backend() != "nccl":
        graph = False            # only RCCL's all-reduce is a stream operation a capture can record (gloo runs on the host)
    decode = use_normalizer and y_normalizer is not None
    if decode:
        # the reference's normalisers keep their statistics where they were computed and move them in every decode() call
        # (models/custom_layer.py:31): a shallow copy with device-resident tensors makes those moves no-ops -- two
        # host-to-device copies less per step, and nothing a hipGraph capture would have to refuse
        import copy
        resident = copy.copy(y_normalizer)
        for k, v in vars(y_normalizer).items():
            if torch.is_tensor(v):
                setattr(resident, k, v.to(device))
        y_normalizer = resident

    def step_loss(pred_y, batch_y):
        if decode:
            pred_y = y_normalizer.decode(pred_y, device=device)
            batch_y = y_normalizer.decode(batch_y, device=device)
        return loss_fn(pred_y, batch_y)
    seen: dict = {}          # batch shape -> eager steps run so far, then its GraphedTrainStep
    loss_history, val_loss_history = [], []
    for epoch in range(epochs):
        model.train()
        running = torch.zeros((), device=device)
        n_batches = 0
        for batch_x, batch_y in train_loader:
            batch_x = batch_x.to(device, non_blocking=True)
            batch_y = batch_y.to(device, non_blocking=True)
            if graph:
                key = (tuple(batch_x.shape), tuple(batch_y.shape))
                state = seen.get(key, 0)
                if isinstance(state, int) and state >= GRAPH_AFTER:
                    from rpde.graph import GraphedTrainStep
                    try:
                        state = GraphedTrainStep(model, step_loss, optimizer, batch_x, batch_y, warmup=0,
                                                 after_backward=bucket.all_reduce_mean)
                    except ValueError:             # a model the graph refuses (host-side randomness): this shape stays eager
                        state = -1
                    seen[key] = state
                if not isinstance(state, int):
                    running += state(batch_x, batch_y)
                    n_batches += 1
                    continue
                if state >= 0:
                    seen[key] = state + 1
            bucket.zero()
            pred_y = model(batch_x)
            if use_normalizer and y_normalizer is not None:
                pred_y = y_normalizer.decode(pred_y, device=device)
                batch_y = y_normalizer.decode(batch_y, device=device)
            loss = loss_fn(pred_y, batch_y)
            loss.backward()
            bucket.all_reduce_mean()
            bucket.detach_untouched()          # parameters outside the graph keep grad None, as after zero_grad()
            optimizer.step()
            running += loss.detach()
            n_batches += 1
            # no autograd graph of an eager step may outlive it when a capture can follow: the parameters' gradient
            # accumulators stay bound to THIS stream while any graph holds them, and a captured backward that meets
            # them drags the default stream into the capture (hipStreamEndCapture then faults)
            del loss, pred_y
        avg_train = _mean_over_ranks(running, n_batches)
        loss_history.append(avg_train)

        model.eval()
        vrun = torch.zeros((), device=device)
        vn = 0
        with torch.no_grad(), frozen_weights():      # the weights rest until the next epoch: prepare them once
            for val_x, val_y in val_loader:
                val_x, val_y = val_x.to(device), val_y.to(device)
                val_pred = model(val_x)
                if use_normalizer and y_normalizer is not None:
                    val_pred = y_normalizer.decode(val_pred, device=device)
                    val_y = y_normalizer.decode(val_y, device=device)
                vrun += loss_fn(val_pred, val_y) * val_x.shape[0]     # sample-weighted: ranks may hold ragged shares
                vn += val_x.shape[0]
        avg_val = _mean_over_ranks(vrun, vn)
        val_loss_history.append(avg_val)

        if "ReduceLROnPlateau" in type(scheduler).__name__:
            scheduler.step(avg_val)
        elif scheduler is not None:
            scheduler.step()
        rec = {"epoch": epoch, "train_loss": avg_train, "val_loss": avg_val}
        if log is not None:
            log(rec)
        if epoch % 10 == 0 and (not _dist_on() or dist.get_rank() == 0):
            print(json.dumps(rec), flush=True)
    return loss_history, val_loss_history


def denormalize_data(data, min_val, max_val):
    return data * (max_val - min_val) + min_val


def evaluate(model, test_loader,
             normalization_type='minmax',
             min_data=None, max_data=None, min_model=None, max_model=None,
             y_normalizer=None,
             time=1, model_type='ffno', device='cuda'):
    """mean relative L2 over the test samples on de-normalised fields; signature, defaults and the ValueError of
    the reference (train/training.py:93-146).  'minmax': affine from (min_model, max_model); 'simple':
    y_normalizer.decode.  The mean is sample-weighted (the reference averages batch means: identical for equal
    batches) and reduced over the ranks."""
    loss_fn = RelativeL2Loss(size_average=True)
    model.eval()
    total = torch.zeros((), device=device)
    n = 0
    warned = False
    with torch.no_grad(), frozen_weights():
        for x, y in test_loader:
            x, y = x.to(device), y.to(device)
            pred = model(x)
            if normalization_type == 'minmax':
                if min_model is not None and max_model is not None:
                    pred = denormalize_data(pred, min_model, max_model)
                    y = denormalize_data(y, min_model, max_model)
                elif not warned:
                    print("Warning: min_model/max_model not provided for minmax normalization")
                    warned = True
            elif normalization_type == 'simple':
                if y_normalizer is not None:
                    pred = y_normalizer.decode(pred, device=device)
                    y = y_normalizer.decode(y, device=device)
                elif not warned:
                    print("Warning: y_normalizer not provided for simple normalization")
                    warned = True
            else:
                raise ValueError(f"Invalid normalization_type: {normalization_type}. Must be 'minmax' or 'simple'")
            total += loss_fn(pred, y) * x.shape[0]
            n += x.shape[0]
    avg = _mean_over_ranks(total, n)
    if not _dist_on() or dist.get_rank() == 0:
        print(f"Test L2 Loss: {avg:.6f}")
    return avg
