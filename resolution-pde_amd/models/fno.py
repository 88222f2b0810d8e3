"""FNO1d / FNO2d (reference: models/fno.py:24-150): grid concat -> 1x1 lifting
-> n_blocks x act(spectral + bypass) -> 1x1 projection MLP, channels-first."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
from torch.nn import functional as F

from models.fno_blocks import FNOBlock1d, FNOBlock2d, LinearMLP1d, MLP1d, MLP2d, act_name  # noqa: F401
from rpde import ops


def _coord(c, device):
    if c is None:
        return None
    if not isinstance(c, torch.Tensor):
        c = torch.tensor(c, dtype=torch.float)
    return c.to(device=device, dtype=torch.float32).contiguous()


class _FNO(nn.Module):
    def _run(self, lifted):
        h = ops.conv1x1(lifted, self.lifting.weight, self.lifting.bias)
        if not torch.is_grad_enabled():      # evaluation / rollout: every block writes its activated output once
            for blk in self.fno_blocks:
                h = blk.activated(h)
            return self.projection(h, "identity")
        act_in = "identity"
        for blk in self.fno_blocks:          # h holds the pre-activation of the previous block
            h = blk.pre_activation(h, act_in)
            act_in = act_name(blk.activation)
        return self.projection(h, act_in)


class FNO1d(_FNO):
    def __init__(self, in_channels, out_channels, modes, width, grid=None, activation=F.relu, n_blocks=4):
        super().__init__()
        self.width, self.grid = width, grid
        self.lifting = nn.Conv1d(in_channels + 1, width, 1)
        self.fno_blocks = nn.ModuleList([FNOBlock1d(width, width, modes, activation) for _ in range(n_blocks)])
        self.projection = MLP1d(width, out_channels, width * 4)

    def get_grid(self, shape, device):
        """[B,n,1] coordinates: the user grid or linspace(0, 2*pi, n) (endpoint included)."""
        b, n = shape[0], shape[1]
        g = _coord(self.grid, device) if self.grid is not None else torch.tensor(
            np.linspace(0, 2 * np.pi, n), dtype=torch.float).to(device)
        return g.reshape(1, n, 1).repeat([b, 1, 1])

    def forward(self, x):
        gx = _coord(self.grid, x.device) if self.grid is not None else None
        return self._run(ops.concat_grid(x, 1, 0.0, 2 * np.pi, channels_last=False, gridx=gx))


class FNO2d(_FNO):
    def __init__(self, in_channels, out_channels, modes1, modes2, width, grid=None, activation=F.gelu, n_blocks=4):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.modes1, self.modes2, self.width, self.grid = modes1, modes2, width, grid
        self.lifting = nn.Conv2d(in_channels + 2, width, 1)
        self.fno_blocks = nn.ModuleList(
            [FNOBlock2d(width, width, modes1, modes2, activation) for _ in range(n_blocks)])
        self.projection = MLP2d(width, out_channels, width * 4)

    def get_grid(self, shape, device):
        """[B,M,N,2] coordinates (x then y), linspace(0,1) with the endpoint."""
        b, m, n = shape[0], shape[1], shape[2]
        if self.grid is not None:
            gx, gy = _coord(self.grid[0], device), _coord(self.grid[1], device)
        else:
            gx = torch.tensor(np.linspace(0, 1, m), dtype=torch.float).to(device)
            gy = torch.tensor(np.linspace(0, 1, n), dtype=torch.float).to(device)
        gx = gx.reshape(1, m, 1, 1).repeat([b, 1, n, 1])
        gy = gy.reshape(1, 1, n, 1).repeat([b, m, 1, 1])
        return torch.cat((gx, gy), dim=-1)

    def _grid_axes(self, m, n, device):
        """the two coordinate vectors as device arrays (cached per shape: evaluation loops call this every step)"""
        if self.grid is not None:
            return _coord(self.grid[0], device), _coord(self.grid[1], device)
        key = (m, n, str(device))
        hit = getattr(self, "_axes_cache", None)
        if hit is None or hit[0] != key:
            gx = torch.tensor(np.linspace(0, 1, m), dtype=torch.float).to(device)
            gy = torch.tensor(np.linspace(0, 1, n), dtype=torch.float).to(device)
            self._axes_cache = hit = (key, gx, gy)
        return hit[1], hit[2]

    def forward(self, x):
        if not torch.is_grad_enabled() and len(self.fno_blocks) > 0 and x.dim() == 4 and x.shape[1] == 1 and x.is_cuda:
            # evaluation / rollout: grid concat, lifting and the first block in one library call -- the lifted field
            # (width x the input) is formed inside the kernels and never crosses HBM
            blk = self.fno_blocks[0]
            ax, ay = self._grid_axes(x.shape[2], x.shape[3], x.device)
            h = ops.fno2d_lift_block_eval(x, ax, ay, self.lifting.weight, self.lifting.bias, blk.spectral_conv.weights1,
                                          blk.spectral_conv.weights2, blk.bypass_conv.weight, blk.bypass_conv.bias,
                                          act_name(blk.activation))
            if h is not None:
                for b in self.fno_blocks[1:-1]:
                    h = b.activated(h)
                if len(self.fno_blocks) > 1:
                    # the last block and the projection in one pass over its input (the block's output is never written)
                    last, pr = self.fno_blocks[-1], self.projection
                    out = ops.fnoblock2d_proj_eval(h, last.spectral_conv.weights1, last.spectral_conv.weights2,
                                                   last.bypass_conv.weight, last.bypass_conv.bias, act_name(last.activation),
                                                   pr.mlp1.weight, pr.mlp1.bias, pr.mlp2.weight, pr.mlp2.bias)
                    if out is not None:
                        return out
                    h = last.activated(h)
                return self.projection(h, "identity")
        gx = gy = None
        if self.grid is not None:
            gx, gy = _coord(self.grid[0], x.device), _coord(self.grid[1], x.device)
        return self._run(ops.concat_grid(x, 2, 0.0, 1.0, channels_last=False, gridx=gx, gridy=gy))
