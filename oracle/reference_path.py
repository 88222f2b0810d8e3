"""CPU restatement of the reference's spectral-convolution hot path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Plain PyTorch-CPU,
functional style: every function takes tensors / a ``state_dict``-shaped
mapping that uses the reference's parameter names, so the same weights can be
fed to the reference (in the build container), to this oracle and to the HIP
product path.  All ``file:line`` citations are into ``/root/reference``.

Parity: PINNED by ``tests/golden/*.npz`` (generated from the imported
reference by ``tests/golden/make_golden.py``) and checked in
``tests/test_oracle_golden.py``.
"""
from __future__ import annotations

import math
from typing import Dict, Mapping, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# activations (models/spectral_convolution.py:104-106)
# --------------------------------------------------------------------------
def _act(name):
    if name == "gelu":
        return F.gelu
    if name == "relu":
        return F.relu
    if name == "identity":
        return lambda t: t
    raise KeyError(name)


# --------------------------------------------------------------------------
# SpectralConv1d.forward  (models/spectral_convolution.py:38-55)
# --------------------------------------------------------------------------
def spectral_conv1d(x: Tensor, weights1: Tensor) -> Tensor:
    """x [B,Cin,n] real, weights1 [Cin,Cout,K] complex -> [B,Cout,n].

    norm='backward'; modes above n//2+1 raise (quirk Q5); the C2R transform
    drops Im(DC)/Im(Nyquist) (quirk Q7)."""
    b, cin, n = x.shape
    cout, k = weights1.shape[1], weights1.shape[2]
    if k > n // 2 + 1:
        raise RuntimeError(f"modes1={k} exceeds n//2+1={n // 2 + 1}")
    x_ft = torch.fft.rfft(x)
    out_ft = torch.zeros(b, cout, n // 2 + 1, dtype=x_ft.dtype)
    out_ft[:, :, :k] = torch.einsum("bix,iox->box", x_ft[:, :, :k], weights1)
    return torch.fft.irfft(out_ft, n=n)


# --------------------------------------------------------------------------
# SpectralConv2d.forward  (models/spectral_convolution.py:79-98)
# --------------------------------------------------------------------------
def spectral_conv2d(x: Tensor, weights1: Tensor, weights2: Tensor) -> Tensor:
    """x [B,Cin,M,N]; weights [Cin,Cout,m1,m2] complex.  Rows [:m1] use
    weights1, rows [-m1:] use weights2 and overwrite on overlap (quirk Q6)."""
    b, cin, m, n = x.shape
    cout, m1, m2 = weights1.shape[1], weights1.shape[2], weights1.shape[3]
    if m2 > n // 2 + 1 or m1 > m:
        raise RuntimeError("modes exceed the available spectrum")
    x_ft = torch.fft.rfft2(x)
    out_ft = torch.zeros(b, cout, m, n // 2 + 1, dtype=x_ft.dtype)
    out_ft[:, :, :m1, :m2] = torch.einsum(
        "bixy,ioxy->boxy", x_ft[:, :, :m1, :m2], weights1)
    out_ft[:, :, m - m1:, :m2] = torch.einsum(
        "bixy,ioxy->boxy", x_ft[:, :, m - m1:, :m2], weights2)
    return torch.fft.irfft2(out_ft, s=(m, n))


# --------------------------------------------------------------------------
# FSpectralConv1d.forward_fourier  (models/spectral_convolution.py:158-204)
# --------------------------------------------------------------------------
def fspectral1d_fourier(x: Tensor, fourier_weight: Tensor, n_modes: int,
                        mode: str = "full", fft_norm: str = "ortho") -> Tensor:
    """x [B,n,C] channels-last; fourier_weight [C,C,K,2] real."""
    xt = x.transpose(1, 2)
    b, h, sx = xt.shape
    x_ft = torch.fft.rfft(xt, dim=-1, norm=fft_norm)
    out_ft = x_ft.new_zeros(b, h, sx // 2 + 1)
    keff = min(n_modes, sx // 2 + 1)
    if mode == "full":
        w = torch.view_as_complex(fourier_weight[:, :, :keff].contiguous())
        out_ft[:, :, :keff] = torch.einsum("bix,iox->box", x_ft[:, :, :keff], w)
    elif mode == "low-pass":
        out_ft[:, :, :keff] = x_ft[:, :, :keff]
    else:
        raise ValueError(f"Mode {mode} not recognized")
    out = torch.fft.irfft(out_ft, n=sx, dim=-1, norm=fft_norm)
    return out.transpose(1, 2)


# --------------------------------------------------------------------------
# FSpectralConv2d.forward_fourier  (models/spectral_convolution.py:256-318)
# --------------------------------------------------------------------------
def fspectral2d_fourier(x: Tensor, w_y: Tensor, w_x: Tensor, n_modes: int,
                        mode: str = "full") -> Tensor:
    """x [B,M,N,C] channels-last; w_y, w_x [C,C,K,2] real; 'ortho' norm.

    Unlike the 1-D layer an unknown ``mode`` does not raise here: the
    reference's if/elif has no else branch, so both spectra stay zero."""
    xt = x.permute(0, 3, 1, 2)
    b, i, m, n = xt.shape
    x_fty = torch.fft.rfft(xt, dim=-1, norm="ortho")
    out_ft = x_fty.new_zeros(b, i, m, n // 2 + 1)
    ky = min(n_modes, n // 2 + 1)
    if mode == "full":
        wy = torch.view_as_complex(w_y[:, :, :ky].contiguous())
        out_ft[:, :, :, :ky] = torch.einsum("bixy,ioy->boxy", x_fty[:, :, :, :ky], wy)
    elif mode == "low-pass":
        out_ft[:, :, :, :ky] = x_fty[:, :, :, :ky]
    xy = torch.fft.irfft(out_ft, n=n, dim=-1, norm="ortho")

    x_ftx = torch.fft.rfft(xt, dim=-2, norm="ortho")
    out_ft = x_ftx.new_zeros(b, i, m // 2 + 1, n)
    kx = min(n_modes, m // 2 + 1)
    if mode == "full":
        wx = torch.view_as_complex(w_x[:, :, :kx].contiguous())
        out_ft[:, :, :kx, :] = torch.einsum("bixy,iox->boxy", x_ftx[:, :, :kx, :], wx)
    elif mode == "low-pass":
        out_ft[:, :, :kx, :] = x_ftx[:, :, :kx, :]
    xx = torch.fft.irfft(out_ft, n=m, dim=-2, norm="ortho")
    return (xx + xy).permute(0, 2, 3, 1)


# --------------------------------------------------------------------------
# FeedForward  (models/custom_layer.py:49-68)
# --------------------------------------------------------------------------
def feedforward(x: Tensor, sd: Mapping[str, Tensor], prefix: str, n_layers: int,
                layer_norm: bool, dropout: float = 0.0, training: bool = False) -> Tensor:
    """Linear -> Dropout -> GELU (Identity on the last) -> LayerNorm (last, optional).
    ``ff_weight_norm`` is accepted by the reference and ignored (quirk Q3)."""
    for i in range(n_layers):
        x = F.linear(x, sd[f"{prefix}layers.{i}.0.weight"], sd[f"{prefix}layers.{i}.0.bias"])
        if dropout > 0.0 and training:
            x = F.dropout(x, dropout, True)
        if i < n_layers - 1:
            x = F.gelu(x)
        elif layer_norm:
            g = sd[f"{prefix}layers.{i}.3.weight"]
            x = F.layer_norm(x, (g.shape[0],), g, sd[f"{prefix}layers.{i}.3.bias"], 1e-5)
    return x


# --------------------------------------------------------------------------
# WNLinear  (models/custom_layer.py:70-108): old-style weight_norm, dim=0
# --------------------------------------------------------------------------
def wn_linear(x: Tensor, sd: Mapping[str, Tensor], prefix: str) -> Tensor:
    if f"{prefix}weight_g" in sd:
        g, v = sd[f"{prefix}weight_g"], sd[f"{prefix}weight_v"]
        w = v * (g / v.norm(2, dim=1, keepdim=True))
    else:
        w = sd[f"{prefix}weight"]
    return F.linear(x, w, sd.get(f"{prefix}bias"))


# --------------------------------------------------------------------------
# grids (quirk Q10): endpoint-inclusive numpy linspace cast to float32
# --------------------------------------------------------------------------
def _lin(lo: float, hi: float, n: int) -> Tensor:
    return torch.tensor(np.linspace(lo, hi, n), dtype=torch.float)


# --------------------------------------------------------------------------
# FNO1d / FNO2d forward  (models/fno.py:56-76, 130-150; fno_blocks.py)
# --------------------------------------------------------------------------
def fno_block(sd: Mapping[str, Tensor], x: Tensor, prefix: str = "", activation: str = "gelu") -> Tensor:
    """FNOBlock1d / FNOBlock2d (models/fno_blocks.py:25-33, 63-71): act(spectral_conv(x) + bypass 1x1 conv(x));
    the rank of x picks the 1-D or 2-D pair"""
    if x.dim() == 3:
        s = spectral_conv1d(x, sd[prefix + "spectral_conv.weights1"])
        c = F.conv1d(x, sd[prefix + "bypass_conv.weight"], sd[prefix + "bypass_conv.bias"])
    else:
        s = spectral_conv2d(x, sd[prefix + "spectral_conv.weights1"], sd[prefix + "spectral_conv.weights2"])
        c = F.conv2d(x, sd[prefix + "bypass_conv.weight"], sd[prefix + "bypass_conv.bias"])
    return _act(activation)(s + c)


def conv_mlp(sd: Mapping[str, Tensor], x: Tensor, prefix: str = "") -> Tensor:
    """MLP1d / MLP2d (models/fno_blocks.py:35-45, 73-83): 1x1 conv -> GELU -> 1x1 conv"""
    conv = F.conv1d if x.dim() == 3 else F.conv2d
    h = F.gelu(conv(x, sd[prefix + "mlp1.weight"], sd[prefix + "mlp1.bias"]))
    return conv(h, sd[prefix + "mlp2.weight"], sd[prefix + "mlp2.bias"])


def fno1d_forward(sd: Mapping[str, Tensor], x: Tensor, n_blocks: int = 4,
                  activation: str = "relu") -> Tensor:
    b, _, n = x.shape
    grid = _lin(0.0, 2 * np.pi, n).reshape(1, 1, n).repeat(b, 1, 1)
    h = torch.cat((x, grid), dim=1)
    h = F.conv1d(h, sd["lifting.weight"], sd["lifting.bias"])
    act = _act(activation)
    for i in range(n_blocks):
        h = fno_block(sd, h, f"fno_blocks.{i}.", activation)
    return conv_mlp(sd, h, "projection.")


def fno2d_forward(sd: Mapping[str, Tensor], x: Tensor, n_blocks: int = 4,
                  activation: str = "gelu") -> Tensor:
    b, _, m, n = x.shape
    gx = _lin(0.0, 1.0, m).reshape(1, 1, m, 1).repeat(b, 1, 1, n)
    gy = _lin(0.0, 1.0, n).reshape(1, 1, 1, n).repeat(b, 1, m, 1)
    h = torch.cat((x, gx, gy), dim=1)
    h = F.conv2d(h, sd["lifting.weight"], sd["lifting.bias"])
    for i in range(n_blocks):
        h = fno_block(sd, h, f"fno_blocks.{i}.", activation)
    return conv_mlp(sd, h, "projection.")


# --------------------------------------------------------------------------
# FFNO1D / FFNO2D forward  (models/ffno.py:96-125, 210-237)
# --------------------------------------------------------------------------
def ffno1d_forward(sd: Mapping[str, Tensor], x: Tensor, n_layers: int, n_modes: int,
                   n_ff_layers: int, layer_norm: bool, dropout: float = 0.0,
                   mode: str = "full", fft_norm: str = "ortho", activation: str = "identity",
                   grid=None, training: bool = False) -> Tensor:
    """``use_grid`` is not a parameter: the reference overwrites it with
    ``grid`` (quirk Q1), so the grid channel exists iff ``grid`` is truthy."""
    b, _, n = x.shape
    if grid:
        g = torch.as_tensor(grid, dtype=torch.float).reshape(1, 1, n).repeat(b, 1, 1)
        x = torch.cat((x, g), dim=1)
    h = wn_linear(x.permute(0, 2, 1), sd, "in_proj.")
    act = _act(activation)
    for i in range(n_layers):
        p = f"fourier_layers.{i}."
        t = h
        if mode != "no-fourier":
            t = fspectral1d_fourier(t, sd[p + "fourier_weight.0"], n_modes, mode, fft_norm)
        t = feedforward(t, sd, p + "backcast_ff.", n_ff_layers, layer_norm, dropout, training)
        h = h + act(t)
    return wn_linear(h, sd, "out_proj.").permute(0, 2, 1)


def ffno2d_forward(sd: Mapping[str, Tensor], x: Tensor, n_layers: int, n_modes: int,
                   n_ff_layers: int, layer_norm: bool, dropout: float = 0.0,
                   mode: str = "full", use_grid: bool = True, training: bool = False) -> Tensor:
    b, _, m, n = x.shape
    if use_grid:
        gx = _lin(0.0, 1.0, m).reshape(1, 1, m, 1).repeat(b, 1, 1, n).to(x.device)
        gy = _lin(0.0, 1.0, n).reshape(1, 1, 1, n).repeat(b, 1, m, 1).to(x.device)
        x = torch.cat((x, gx, gy), dim=1)
    h = wn_linear(x.permute(0, 2, 3, 1), sd, "in_proj.")
    for i in range(n_layers):
        p = f"fourier_layers.{i}."
        t = h
        if mode != "no-fourier":
            t = fspectral2d_fourier(t, sd[p + "fourier_weight.0"], sd[p + "fourier_weight.1"],
                                    n_modes, mode)
        t = feedforward(t, sd, p + "backcast_ff.", n_ff_layers, layer_norm, dropout, training)
        h = h + t
    return wn_linear(h, sd, "out_proj.").permute(0, 3, 1, 2)


# --------------------------------------------------------------------------
# RelativeL2Loss.forward  (utils/loss.py:31-59); eps on the denominator (Q16)
# --------------------------------------------------------------------------
def relative_l2(x: Tensor, y: Tensor, size_average: bool = True, reduction: bool = True) -> Tensor:
    nb = x.shape[0]
    diff = torch.norm(x.reshape(nb, -1) - y.reshape(nb, -1), 2, 1)
    ynorm = torch.norm(y.reshape(nb, -1), 2, 1)
    rel = diff / (ynorm + 1e-8)
    if reduction:
        return rel.mean() if size_average else rel.sum()
    return rel


# --------------------------------------------------------------------------
# spectral resize  (utils/res_utils.py:93-125 resize_1d, :29-50 resize)
# --------------------------------------------------------------------------
def resize_1d(x: Tensor, out_size: int) -> Tensor:
    n = x.shape[-1]
    f = torch.fft.rfft(x, norm="backward")
    fz = torch.zeros((*x.shape[:-1], out_size // 2 + 1), dtype=f.dtype)
    k = min(f.shape[-1], out_size // 2 + 1)
    fz[..., :k] = f[..., :k]
    return torch.fft.irfft(fz, n=out_size) * (out_size / n)


def resize_2d(x: Tensor, out_size) -> Tensor:
    m, n = x.shape[-2], x.shape[-1]
    mo, no = out_size
    f = torch.fft.rfft2(x, norm="backward")
    fz = torch.zeros((*x.shape[:-2], mo, no // 2 + 1), dtype=f.dtype)
    top1, bot1 = min((m + 1) // 2, (mo + 1) // 2), min(m // 2, mo // 2)
    k2 = min(f.shape[-1], no // 2 + 1)
    fz[..., :top1, :k2] = f[..., :top1, :k2]
    if bot1 > 0:
        fz[..., mo - bot1:, :k2] = f[..., m - bot1:, :k2]
    return torch.fft.irfft2(fz, s=(mo, no)) * (mo / m) * (no / n)


# --------------------------------------------------------------------------
# 1-D autoregressive rollout core  (utils/autoregressive_step.py:284-309)
# --------------------------------------------------------------------------
def rollout_1d(step_fn, state: Tensor, steps: int, mean: float, std: float) -> Tensor:
    """state [B,n] normalised; each step: model -> decode with the y-normaliser
    -> re-encode with the x-normaliser (SimpleNormalizer, same mean/std)."""
    outs = []
    for _ in range(steps):
        pred = step_fn(state.unsqueeze(1)).squeeze(1)
        outs.append(pred)
        state = ((pred * std + mean) - mean) / std
    return torch.stack(outs, dim=1)


# --------------------------------------------------------------------------
# training step with the shape of train/training.py:29-47 (CPU baseline leg)
# --------------------------------------------------------------------------
def make_params(sd: Mapping[str, Tensor]) -> Dict[str, Tensor]:
    return {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}


def train_step(forward, params: Dict[str, Tensor], opt: torch.optim.Optimizer,
               x: Tensor, y: Tensor) -> float:
    opt.zero_grad()
    pred = forward(params, x)
    loss = relative_l2(pred, y)
    loss.backward()
    opt.step()
    return float(loss.item())
