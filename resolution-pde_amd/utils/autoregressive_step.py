"""Autoregressive rollout (reference: utils/autoregressive_step.py:284-309
``perform_rollout_1d`` and the per-step metric :190-197).  The reference has no
2-D rollout; ``perform_rollout_2d`` is its analogue for
BASELINE config 5, specified by analogy ("parity unpinned" beyond the pinned
single-step forward)."""
from __future__ import annotations

from typing import Optional

import torch

from rpde.ops import frozen_weights
from utils.loss import RelativeL2Loss


def _scalar_stats(norm):
    """(mean, std + eps) when the normaliser is one global affine map (SimpleNormalizer), else None"""
    try:
        m, s, e = norm.mean, norm.std, getattr(norm, "eps", 0.0)
        if isinstance(m, (int, float)) and isinstance(s, (int, float)):
            return float(m), float(s) + float(e)
    except AttributeError:
        pass
    return None


@torch.no_grad()
def _rollout(model, state, steps, x_normalizer, y_normalizer, device):
    """Between two steps the reference decodes the prediction with the y-statistics and re-encodes it with the
    x-statistics (autoregressive_step.py:296-303).  With global statistics both maps are affine, so the pair is ONE
    multiply-add  state = pred * (s_y / s_x) + (m_y - m_x) / s_x  -- a single pass over the field instead of four;
    point-wise normalisers keep the two calls."""
    fused = None
    if x_normalizer is not None and y_normalizer is not None:
        sx, sy = _scalar_stats(x_normalizer), _scalar_stats(y_normalizer)
        if sx is not None and sy is not None:
            fused = (sy[1] / sx[1], torch.full((), (sy[0] - sx[0]) / sx[1], device=state.device, dtype=state.dtype))
    with frozen_weights():                                     # one weight preparation for all steps
        return _rollout_steps(model, state, steps, x_normalizer, y_normalizer, device, fused)


def _rollout_steps(model, state, steps, x_normalizer, y_normalizer, device, fused):
    preds = []
    for _ in range(steps):
        nxt = model(state.unsqueeze(1))                        # [B,1,*S] in, [B,1,*S] out
        if nxt.shape[1] == 1:
            nxt = nxt.squeeze(1)
        preds.append(nxt.unsqueeze(1))
        if fused is not None:
            state = torch.add(fused[1], nxt, alpha=fused[0])
        elif x_normalizer is not None and y_normalizer is not None:   # decode with y-stats, re-encode with x-stats
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
    # all steps in one launch and one host read: the mean over [B * T] (sample, step) pairs of equal weight is the mean
    # over steps of the per-step batch means
    steps = predictions.shape[1]
    B = predictions.shape[0]
    pred = predictions.reshape(B * steps, -1)
    tgt = trajectory[:, 1:steps + 1].reshape(B * steps, -1)
    return float(RelativeL2Loss(size_average=True)(pred, tgt))
