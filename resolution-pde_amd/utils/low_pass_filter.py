"""Fourier low-pass filters of the data layer (reference: utils/low_pass_filter.py:3-40 and :42-101).

Host-side preprocessing (runs once per dataset on whatever device the tensor lives on).  Both keep the grid
size: they zero modes, they do not decimate -- the reference's loaders rely on that (quirk: "low-pass
downsampling" leaves the resolution unchanged, dataloaders/ns_naive_markov.py:227-242)."""
from __future__ import annotations

import torch


def lowpass_filter_1d(data: torch.Tensor, cutoff_ratio: float = 0.25) -> torch.Tensor:
    """data [B,T,C,n] or [B,T,n]: rfft bins [int(n_bins * cutoff_ratio):] are dropped (so cutoff 1.0 keeps all)."""
    squeeze = data.ndim == 3
    if squeeze:
        data = data.unsqueeze(2)
    spec = torch.fft.rfft(data, dim=-1)
    spec[..., int(spec.size(-1) * cutoff_ratio):] = 0
    out = torch.fft.irfft(spec, n=data.shape[-1], dim=-1)
    return out.squeeze(2) if squeeze else out


def lowpass_filter_2d(data: torch.Tensor, cutoff_ratio: float = 0.25) -> torch.Tensor:
    """data [B,T,C,S,S] or [B,T,S,S] (square grids): keeps |f_y| <= cutoff/2 and |f_x| <= cutoff/2 cycles per
    sample (rectangular mask, Nyquist = 0.5)."""
    squeeze = data.ndim == 4
    if squeeze:
        data = data.unsqueeze(2)
    s = data.shape[-1]
    spec = torch.fft.rfft2(data, dim=(-2, -1))
    cutoff = cutoff_ratio * 0.5
    keep_y = torch.fft.fftfreq(s, device=data.device).abs() <= cutoff
    keep_x = torch.fft.rfftfreq(s, device=data.device).abs() <= cutoff
    spec = spec * (keep_y.view(-1, 1) * keep_x.view(1, -1)).view(1, 1, 1, s, spec.shape[-1])
    out = torch.fft.irfft2(spec, s=(s, s), dim=(-2, -1))
    return out.squeeze(2) if squeeze else out
