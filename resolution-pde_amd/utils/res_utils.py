"""Spectral resize used by the resize-mode evaluators (reference:
utils/res_utils.py:29-50 ``resize``, :93-125 ``resize_1d``): rfft -> copy the
bins both sizes share -> irfft at the new size, scaled by out/in -- on the
MI355X through the same truncated-DFT plans as the spectral layers (analysis
at the source size, synthesis at the target size; nothing is zero-padded)."""
from __future__ import annotations

from rpde import ops


def resize(x, out_size, permute=False):
    """x [B,C,M,N] (or [B,M,N,C] with ``permute``) -> spatial size ``out_size``"""
    if permute:
        x = x.permute(0, 3, 1, 2)
    y = ops.resize2d(x.contiguous(), out_size)
    return y.permute(0, 2, 3, 1) if permute else y


def resize_1d(x, out_size):
    """x [..., n] -> [..., out_size]"""
    return ops.resize1d(x, int(out_size))
