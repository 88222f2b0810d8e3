"""SpectralConv1d / SpectralConv2d / FSpectralConv1d / FSpectralConv2d with the
reference's constructor signatures, parameter names, dtypes and initial
distributions (reference: models/spectral_convolution.py:24-318), computed on
the MI355X by librpde_hip.so as truncated-DFT fp32-MFMA GEMMs:

    rfft -> keep K modes -> per-mode complex channel mixing -> irfft

never materialises the full spectrum (only K of n/2+1 bins are computed) and
never transposes channels-last tensors to channels-first.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.nn import functional as F

from models.custom_layer import FeedForward
from rpde import ops

act_registry = {"gelu": F.gelu, "identity": nn.Identity(), "relu": F.relu}


class SpectralConv1d(nn.Module):
    """x [B,Cin,n] -> [B,Cout,n]; ``weights1`` complex64 [Cin,Cout,modes1] drawn
    as scale * U[0,1) on both parts, scale = 1/(Cin*Cout); norm='backward'.
    modes1 > n//2+1 raises, as the reference's einsum does."""

    def __init__(self, in_channels, out_channels, modes1):
        super().__init__()
        self.in_channels, self.out_channels, self.modes1 = in_channels, out_channels, modes1
        self.scale = 1 / (in_channels * out_channels)
        self.weights1 = nn.Parameter(self.scale * torch.rand(in_channels, out_channels, modes1, dtype=torch.cfloat))

    def forward(self, x, act_in="identity"):
        """``act_in`` lets an FNO block feed the previous pre-activation and have
        the activation applied while x is staged (no extra HBM pass)."""
        return ops.spectral1d(x, self.weights1, act_in)


class SpectralConv2d(nn.Module):
    """x [B,Cin,M,N] -> [B,Cout,M,N]; ``weights1`` acts on rows [0,modes1),
    ``weights2`` on rows [M-modes1,M) (and wins where they overlap, quirk Q6),
    columns [0,modes2) of the half spectrum."""

    def __init__(self, in_channels, out_channels, modes1, modes2):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.modes1, self.modes2 = modes1, modes2
        self.scale = 1 / (in_channels * out_channels)
        shape = (in_channels, out_channels, modes1, modes2)
        self.weights1 = nn.Parameter(self.scale * torch.rand(*shape, dtype=torch.cfloat))
        self.weights2 = nn.Parameter(self.scale * torch.rand(*shape, dtype=torch.cfloat))

    def forward(self, x, act_in="identity"):
        return ops.spectral2d(x, self.weights1, self.weights2, act_in)


def _fourier_weights(d_model, modes, count):
    plist = nn.ParameterList()
    for _ in range(count):
        p = nn.Parameter(torch.empty(d_model, d_model, modes, 2))
        nn.init.xavier_normal_(p)          # fans see (d_model, d_model, modes*2), quirk Q9
        plist.append(p)
    return plist


class FSpectralConv1d(nn.Module):
    """Factorised spectral layer on channels-last [B,n,C]: forward_fourier, then
    the FeedForward tail, then ``activation``; returns (b, None)."""

    def __init__(self, d_model, modes, forecast_ff=None, backcast_ff=None, fourier_weight=None, factor=4,
                 ff_weight_norm=False, n_ff_layers=2, layer_norm=False, use_fork=False, dropout=0.0, mode="full",
                 activation="identity", fft_norm="ortho", **kwargs):
        super().__init__()
        self.in_dim = self.out_dim = d_model
        self.n_modes, self.mode, self.use_fork, self.fft_norm = modes, mode, use_fork, fft_norm
        self.fourier_weight = fourier_weight if fourier_weight else _fourier_weights(d_model, modes, 1)
        self.backcast_ff = backcast_ff if backcast_ff else FeedForward(
            d_model, factor, n_layers=n_ff_layers, ff_weight_norm=ff_weight_norm, layer_norm=layer_norm, dropout=dropout)
        self.activation = activation
        self.act = act_registry[activation]

    def forward_fourier(self, x, with_skip=False):
        return ops.fspectral1d(x, self.fourier_weight[0], self.n_modes, self.mode, self.fft_norm, with_skip=with_skip)

    def forward(self, x, batch_dt=None, residual=None):
        if self.mode != "no-fourier":
            if residual is x and torch.is_grad_enabled() and x.requires_grad:
                # skip connection around this layer: its gradient is summed inside the spectral backward
                x, residual = self.forward_fourier(x, with_skip=True)
            else:
                x = self.forward_fourier(x)
        return self.backcast_ff(x, residual=residual, post_act=self.activation), None


class FSpectralConv2d(nn.Module):
    """Factorised spectral layer on channels-last [B,M,N,C]: independent 1-D
    spectral convolutions along y and x, summed in physical space ('ortho'
    norm, modes clamped to the available bins), then the FeedForward tail(s);
    returns (b, f)."""

    def __init__(self, d_model, modes, forecast_ff=None, backcast_ff=None, fourier_weight=None, factor=4,
                 ff_weight_norm=False, n_ff_layers=2, layer_norm=False, use_fork=False, dropout=0.0, mode="full"):
        super().__init__()
        self.in_dim = self.out_dim = d_model
        self.n_modes, self.mode, self.use_fork = modes, mode, use_fork
        self.fourier_weight = fourier_weight if fourier_weight else _fourier_weights(d_model, modes, 2)
        self.backcast_ff = backcast_ff if backcast_ff else FeedForward(
            d_model, factor, n_layers=n_ff_layers, ff_weight_norm=ff_weight_norm, layer_norm=layer_norm, dropout=dropout)
        self.forecast_ff = forecast_ff
        if use_fork and not self.forecast_ff:
            self.forecast_ff = FeedForward(
                d_model, factor, n_layers=n_ff_layers, ff_weight_norm=ff_weight_norm, layer_norm=layer_norm,
                dropout=dropout)

    def forward_fourier(self, x, with_skip=False):
        return ops.fspectral2d(x, self.fourier_weight[0], self.fourier_weight[1], self.n_modes, self.mode,
                               with_skip=with_skip)

    def forward(self, x, batch_dt=None, residual=None):
        if self.mode != "no-fourier":
            if residual is x and torch.is_grad_enabled() and x.requires_grad:
                x, residual = self.forward_fourier(x, with_skip=True)
            else:
                x = self.forward_fourier(x)
        b = self.backcast_ff(x, residual=residual)
        f = self.forecast_ff(x) if self.use_fork else None
        return b, f
