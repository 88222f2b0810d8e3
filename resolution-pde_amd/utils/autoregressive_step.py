"""Autoregressive rollout (reference: utils/autoregressive_step.py:284-309
``perform_rollout_1d`` and the per-step metric :190-197).  The reference has no
2-D rollout; ``perform_rollout_2d`` is its analogue for
BASELINE config 5, specified by analogy ("parity unpinned" beyond the pinned
single-step forward)."""
from __future__ import annotations

from typing import Optional

import torch

from utils.loss import RelativeL2Loss


@torch.no_grad()
def _rollout(model, state, steps, x_normalizer, y_normalizer, device):
    preds = []
    for _ in range(steps):
        nxt = model(state.unsqueeze(1))                        # [B,1,*S] in, [B,1,*S] out
        if nxt.shape[1] == 1:
            nxt = nxt.squeeze(1)
        preds.append(nxt.unsqueeze(1))
        if x_normalizer is not None and y_normalizer is not None:   # decode with y-stats, re-encode with x-stats
            phys = y_normalizer.decode(nxt.unsqueeze(1), device=device).squeeze(1)
            state = x_normalizer.encode(phys.unsqueeze(1)).squeeze(1)
        else:
            state = nxt
    return torch.cat(preds, dim=1)


def perform_rollout_1d(model, initial_condition, rollout_steps, model_type="ffno1d", time_val=None, device="cuda",
                       x_normalizer=None, y_normalizer=None):
    """initial_condition [B,n] (normalised) -> predictions [B,steps,n] (normalised)"""
    return _rollout(model, initial_condition, rollout_steps, x_normalizer, y_normalizer, device)


def perform_rollout_2d(model, initial_condition, rollout_steps, device="cuda", x_normalizer=None, y_normalizer=None):
    """initial_condition [B,M,N] -> predictions [B,steps,M,N]"""
    return _rollout(model, initial_condition, rollout_steps, x_normalizer, y_normalizer, device)


def rollout_loss(predictions: torch.Tensor, trajectory: torch.Tensor) -> float:
    """(1/T) sum_t mean_b |pred_t - target_t|_2 / |target_t|_2 with target_t = trajectory[:, t+1]"""
    loss_fn = RelativeL2Loss(size_average=True)
    steps = predictions.shape[1]
    return sum(float(loss_fn(predictions[:, t], trajectory[:, t + 1])) for t in range(steps)) / steps
