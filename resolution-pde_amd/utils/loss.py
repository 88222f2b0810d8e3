"""RelativeL2Loss (reference: utils/loss.py:17-59): per-sample
|x - y|_2 / (|y|_2 + 1e-8), then mean / sum / none -- one fused HIP pass over
prediction and target (wavefront-shuffle reductions, deterministic)."""
from __future__ import annotations

import torch.nn as nn

from rpde import ops


class RelativeL2Loss(nn.Module):
    def __init__(self, size_average=True, reduction=True):
        super().__init__()
        self.size_average = size_average
        self.reduction = reduction

    def forward(self, x, y):
        return ops.relative_l2(x, y, self.size_average, self.reduction)
