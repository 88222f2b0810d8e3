"""Truncated real-DFT matrices in float64 (TEST INFRASTRUCTURE ONLY).

The HIP product path never materialises a full spectrum: with K retained
modes the R2C transform restricted to bins [0,K) is the real matrix
``analysis(n,K)`` of shape [2K, n] and the C2R transform of a spectrum that is
zero outside [0,K) is ``synthesis(n,K)`` of shape [n, 2K].  This file restates
those matrices independently (numpy, float64) so tests can prove, on CPU, that
the matrix formulation equals ``torch.fft.rfft/irfft`` as the reference calls
them (models/spectral_convolution.py:41,54,165,198,265,284,289,308) including
the C2R rule that Im(DC) and Im(Nyquist) are ignored.

Row/column order of the "2K" axis is interleaved: index 2k is Re(bin k),
2k+1 is Im(bin k).
"""
from __future__ import annotations

import numpy as np


def _scales(n: int, norm: str):
    if norm == "ortho":
        return 1.0 / np.sqrt(n), 1.0 / np.sqrt(n)
    if norm == "backward":
        return 1.0, 1.0 / n
    if norm == "forward":
        return 1.0 / n, 1.0
    raise ValueError(norm)


def _angles(n: int, k: int) -> np.ndarray:
    kk = np.arange(k, dtype=np.int64)[:, None]
    yy = np.arange(n, dtype=np.int64)[None, :]
    return 2.0 * np.pi * ((kk * yy) % n).astype(np.float64) / n   # [k, n]


def analysis(n: int, k: int, norm: str) -> np.ndarray:
    """[2k, n]: rfft(x, norm)[:k] as (Re,Im)-interleaved rows."""
    sf, _ = _scales(n, norm)
    ang = _angles(n, k)
    out = np.empty((2 * k, n), dtype=np.float64)
    out[0::2] = sf * np.cos(ang)
    out[1::2] = -sf * np.sin(ang)
    return out


def hermitian_weight(n: int, k: int) -> np.ndarray:
    """c_k of the C2R sum: 1 for DC and (n even) Nyquist, else 2."""
    c = np.full(k, 2.0)
    c[0] = 1.0
    if n % 2 == 0 and k > n // 2:
        c[n // 2] = 1.0
    return c


def synthesis(n: int, k: int, norm: str) -> np.ndarray:
    """[n, 2k]: irfft(spec zero beyond k, n, norm) as a matrix acting on
    (Re,Im)-interleaved coefficients.  sin(0)=sin(pi*y)=0 drops Im(DC/Nyquist)."""
    _, si = _scales(n, norm)
    ang = _angles(n, k).T            # [n, k]
    c = hermitian_weight(n, k)[None, :]
    out = np.empty((n, 2 * k), dtype=np.float64)
    out[:, 0::2] = si * c * np.cos(ang)
    out[:, 1::2] = -si * c * np.sin(ang)
    return out


def complex_analysis(m: int, rows: np.ndarray, norm: str) -> np.ndarray:
    """Full complex forward DFT along an axis of length m restricted to the
    output bins ``rows`` (used for the first transformed dim of rfft2):
    returns [len(rows), m] complex128."""
    sf, _ = _scales(m, norm)
    kk = np.asarray(rows, dtype=np.int64)[:, None]
    yy = np.arange(m, dtype=np.int64)[None, :]
    ang = 2.0 * np.pi * ((kk * yy) % m).astype(np.float64) / m
    return sf * (np.cos(ang) - 1j * np.sin(ang))


def complex_synthesis(m: int, rows: np.ndarray, norm: str) -> np.ndarray:
    """Inverse complex DFT along an axis of length m from the bins ``rows``
    only: [m, len(rows)] complex128."""
    _, si = _scales(m, norm)
    kk = np.asarray(rows, dtype=np.int64)[None, :]
    yy = np.arange(m, dtype=np.int64)[:, None]
    ang = 2.0 * np.pi * ((kk * yy) % m).astype(np.float64) / m
    return si * (np.cos(ang) + 1j * np.sin(ang))
