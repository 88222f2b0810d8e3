"""FeedForward / WNLinear / UnitGaussianNormalizer with the reference's
constructor signatures and state_dict layout (reference:
models/custom_layer.py:19-108), computing on the MI355X through librpde_hip.so.

The module tree mirrors the reference's container layout only so that
checkpoints interchange (``layers.{i}.0.weight`` for the Linear,
``layers.{last}.3.weight`` for the LayerNorm); ``forward`` never runs those
containers -- it hands their parameters to the fused HIP FeedForward.
"""
from __future__ import annotations

import copy

import torch
import torch.nn as nn

from rpde import ops


class UnitGaussianNormalizer(object):
    """Point-wise (x - mean) / (std + eps) normaliser (reference :19-47)."""

    def __init__(self, x, eps=0.00001):
        self.mean = torch.mean(x, 0)
        self.std = torch.std(x, 0)
        self.eps = eps

    def encode(self, x):
        return (x - self.mean) / (self.std + self.eps)

    def decode(self, x, device="cuda:0"):
        return x * (self.std + self.eps).to(device) + self.mean.to(device)

    def cuda(self):
        self.mean, self.std = self.mean.cuda(), self.std.cuda()

    def cpu(self):
        self.mean, self.std = self.mean.cpu(), self.std.cpu()


class FeedForward(nn.Module):
    """dim -> dim*factor -> ... -> dim pointwise MLP: per layer Linear, Dropout,
    GELU (Identity on the last), and LayerNorm after the last layer when
    ``layer_norm``.  ``ff_weight_norm`` is accepted and unused, as in the
    reference."""

    def __init__(self, dim, factor, n_layers=2, ff_weight_norm=False, layer_norm=False, dropout=0.0):
        super().__init__()
        self.dim, self.factor, self.n_layers = dim, factor, n_layers
        self.use_layer_norm, self.dropout = bool(layer_norm), float(dropout)
        self.layers = nn.ModuleList()
        for i in range(n_layers):
            last = i == n_layers - 1
            fan_in = dim if i == 0 else dim * factor
            fan_out = dim if last else dim * factor
            self.layers.append(nn.Sequential(
                nn.Linear(fan_in, fan_out),
                nn.Dropout(dropout),
                nn.Identity() if last else nn.GELU(),
                nn.LayerNorm(fan_out) if (layer_norm and last) else nn.Identity()))

    def forward(self, x, residual=None, post_act="identity"):
        """out = [residual +] post_act(FeedForward(x)); the extra arguments let
        the FFNO blocks fuse their skip connection into the last kernel."""
        lin = [blk[0] for blk in self.layers]
        ln = None
        if self.use_layer_norm:
            norm = self.layers[-1][3]
            ln = (norm.weight, norm.bias)
        p = self.dropout if self.training else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,), device="cpu").item()) if p > 0.0 else 0
        eps = self.layers[-1][3].eps if self.use_layer_norm else 1e-5
        return ops.feedforward(x, residual, [l.weight for l in lin], [l.bias for l in lin], ln, self.dim, self.factor,
                               p, seed, post_act, eps)


class WNLinear(nn.Linear):
    """nn.Linear with optional old-style weight normalisation
    (parameters ``weight_g`` [out,1], ``weight_v`` [out,in]; w = g * v / |v|_row),
    reference :70-108.  Deep copies are rebuilt from the state_dict, which is
    what the reference's ``_fix_weight_norm_deepcopy`` works around."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None, dtype=None, wnorm=False):
        super().__init__(in_features, out_features, bias=bias, device=device, dtype=dtype)
        self.wnorm = bool(wnorm)
        if self.wnorm:
            w = self.weight.detach()
            del self._parameters["weight"]
            self.register_parameter("weight_g", nn.Parameter(w.norm(2, dim=1, keepdim=True)))
            self.register_parameter("weight_v", nn.Parameter(w.clone()))

    def effective_weight(self) -> torch.Tensor:
        if not self.wnorm:
            return self.weight
        v = self.weight_v
        if v.is_cuda and v.dtype == torch.float32:
            return ops.weight_norm(v, self.weight_g)
        return v * (self.weight_g / v.norm(2, dim=1, keepdim=True))       # parameter bookkeeping on the host only

    def __getattr__(self, name):
        if name == "weight" and "weight_v" in self.__dict__.get("_parameters", {}):
            return self.effective_weight()
        return super().__getattr__(name)

    def forward(self, x):
        return ops.linear(x, self.effective_weight(), self.bias)

    def __deepcopy__(self, memo):
        ref = self.weight_v if self.wnorm else self.weight
        new = WNLinear(self.in_features, self.out_features, self.bias is not None, ref.device, ref.dtype, self.wnorm)
        new.load_state_dict(copy.deepcopy(self.state_dict(), memo))
        new.train(self.training)
        return new
