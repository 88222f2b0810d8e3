"""Synthetic stand-ins for the reference's HDF5 data sets (not redistributable,
absent offline): periodic band-limited Gaussian random fields with the
spectrum of the reference's NS initial conditions, (4 pi^2 |k|^2 + tau^2)^(-alpha/2)
with alpha = 2.5, tau = 7 (data_generation/random_fields.py), standardised like
its SimpleNormalizer output.  The target is the input advanced by a fixed
linear spectral filter (diffusion + advection phase), so the relative-L2 loss
is learnable and decreases."""
from __future__ import annotations

import math
from typing import List, Tuple

import torch


def random_fields(n: int, res: int, dims: int, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    shape = (n, 1) + (res,) * dims
    white = torch.randn(shape, generator=g)
    if dims == 1:
        k2 = torch.fft.rfftfreq(res, 1.0 / res) ** 2
        f = torch.fft.irfft(torch.fft.rfft(white) * (4 * math.pi ** 2 * k2 + 49.0) ** (-1.25), n=res)
    else:
        kx = torch.fft.fftfreq(res, 1.0 / res)[:, None]
        ky = torch.fft.rfftfreq(res, 1.0 / res)[None, :]
        f = torch.fft.irfft2(torch.fft.rfft2(white) * (4 * math.pi ** 2 * (kx ** 2 + ky ** 2) + 49.0) ** (-1.25),
                             s=(res, res))
    return (f - f.mean()) / f.std()


def advance(x: torch.Tensor, dims: int, nu: float = 2e-3, shift: float = 0.02) -> torch.Tensor:
    """one 'time step': spectral diffusion exp(-nu |k|^2) and a translation by ``shift`` of the domain"""
    res = x.shape[-1]
    if dims == 1:
        k = torch.fft.rfftfreq(res, 1.0 / res)
        mult = torch.exp(-nu * k ** 2) * torch.exp(-2j * math.pi * k * shift)
        return torch.fft.irfft(torch.fft.rfft(x) * mult, n=res)
    kx = torch.fft.fftfreq(res, 1.0 / res)[:, None]
    ky = torch.fft.rfftfreq(res, 1.0 / res)[None, :]
    mult = torch.exp(-nu * (kx ** 2 + ky ** 2)) * torch.exp(-2j * math.pi * (kx + ky) * shift)
    return torch.fft.irfft2(torch.fft.rfft2(x) * mult, s=(res, res))


def markov_pairs(resolutions, dims: int, seed: int) -> List[Tuple[torch.Tensor, torch.Tensor]]:
    """[(x [1,*res], y [1,*res])] over all requested resolutions (mixed shapes allowed)"""
    out = []
    for i, (res, n) in enumerate(sorted(dict(resolutions).items())):
        x = random_fields(int(n), int(res), dims, seed + 1000 * i)
        y = advance(x, dims)
        out += [(x[j], y[j]) for j in range(x.shape[0])]
    return out
