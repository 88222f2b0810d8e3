"""FNO blocks and projection MLPs (reference: models/fno_blocks.py:25-83) on the
HIP path.  Inside an FNO the blocks pass *pre-activations*: each consumer
applies the activation while it stages its operand, so act(x) is never written
to HBM; called stand-alone (``block(x)``) they behave exactly like the
reference and return activation(spectral_conv(x) + bypass_conv(x))."""
from __future__ import annotations

import torch.nn as nn
from torch.nn import functional as F

from models.spectral_convolution import SpectralConv1d, SpectralConv2d
from rpde import ops


def act_name(fn) -> str:
    """Map the reference's activation callables onto kernel ids."""
    if fn in (F.gelu,) or isinstance(fn, nn.GELU):
        return "gelu"
    if fn in (F.relu,) or isinstance(fn, nn.ReLU):
        return "relu"
    if fn is None or isinstance(fn, nn.Identity):
        return "identity"
    raise ValueError(f"activation {fn!r} has no HIP kernel (gelu, relu, identity are supported)")


class _FNOBlock(nn.Module):
    def pre_activation(self, x, act_in="identity"):
        """spectral_conv(act_in(x)) + bypass_conv(act_in(x)) (bias included)"""
        spec = self.spectral_conv(x, act_in)
        return ops.conv1x1(x, self.bypass_conv.weight, self.bypass_conv.bias, act_in, acc=spec, acc_owned=True)

    def activated(self, x):
        """evaluation only: activation(spectral_conv(x) + bypass_conv(x)) with the activation applied by the
        kernel that writes the block's output (stored once, already activated, so that the next block's truncated
        DFT streams it without evaluating the activation again)"""
        if isinstance(self.spectral_conv, SpectralConv2d):
            # the spectral branch's last transform, the bypass convolution and the activation in one pass
            out = ops.fnoblock2d_eval(x, self.spectral_conv.weights1, self.spectral_conv.weights2, self.bypass_conv.weight,
                                      self.bypass_conv.bias, act_name(self.activation))
            if out is not None:
                return out
        spec = self.spectral_conv(x, "identity")
        return ops.conv1x1_act_eval(x, self.bypass_conv.weight, self.bypass_conv.bias, spec, act_name(self.activation))

    def forward(self, x):
        import torch
        if not torch.is_grad_enabled():
            return self.activated(x)
        return ops.activation(self.pre_activation(x), act_name(self.activation))


class FNOBlock1d(_FNOBlock):
    def __init__(self, in_channels, out_channels, modes, activation=F.relu):
        super().__init__()
        self.spectral_conv = SpectralConv1d(in_channels, out_channels, modes)
        self.bypass_conv = nn.Conv1d(in_channels, out_channels, 1)
        self.activation = activation


class FNOBlock2d(_FNOBlock):
    def __init__(self, in_channels, out_channels, modes1, modes2, activation=F.gelu):
        super().__init__()
        self.spectral_conv = SpectralConv2d(in_channels, out_channels, modes1, modes2)
        self.bypass_conv = nn.Conv2d(in_channels, out_channels, 1)
        self.activation = activation


class _MLP(nn.Module):
    def forward(self, x, act_in="identity"):
        import torch
        if not torch.is_grad_enabled():
            # evaluation: one pass, the hidden tensor never reaches HBM (csrc/conv_mlp.hip)
            out = ops.conv_mlp_eval(x, self.mlp1.weight, self.mlp1.bias, self.mlp2.weight, self.mlp2.bias, act_in)
            if out is not None:
                return out
        h = ops.conv1x1(x, self.mlp1.weight, self.mlp1.bias, act_in)
        return ops.conv1x1(h, self.mlp2.weight, self.mlp2.bias, "gelu")


class MLP1d(_MLP):
    def __init__(self, in_channels, out_channels, mid_channels):
        super().__init__()
        self.mlp1 = nn.Conv1d(in_channels, mid_channels, 1)
        self.mlp2 = nn.Conv1d(mid_channels, out_channels, 1)


class MLP2d(_MLP):
    def __init__(self, in_channels, out_channels, mid_channels):
        super().__init__()
        self.mlp1 = nn.Conv2d(in_channels, mid_channels, 1)
        self.mlp2 = nn.Conv2d(mid_channels, out_channels, 1)


class LinearMLP1d(nn.Module):
    def __init__(self, in_features, out_features, mid_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, mid_features)
        self.fc2 = nn.Linear(mid_features, out_features)

    def forward(self, x):
        h = ops.linear(x, self.fc1.weight, self.fc1.bias)
        return ops.linear(ops.activation(h, "gelu"), self.fc2.weight, self.fc2.bias)
