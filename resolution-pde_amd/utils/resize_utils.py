"""All-resolution evaluation core (reference: utils/naive_utils.py:253-507,
utils/resize_utils.py:27-46): for every resolution in [min, ..., max] obtain the
test fields at that resolution (stride subsampling = 'naive_downsample', or the
spectral ``resize``), normalise, run the model, de-normalise, relative L2.
Plotting / CSV / wandb of the reference are out of scope; the numbers are
returned."""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import torch

from utils.loss import RelativeL2Loss
from utils.res_utils import resize, resize_1d


def get_lower_resolutions(base_resolution: int, min_resolution: int = 32) -> List[int]:
    """base / 2^k down to ``min_resolution``, ascending, ending with the base (e.g. 256 -> [32,64,128,256])"""
    out = [base_resolution]
    res = base_resolution // 2
    while res >= min_resolution:
        out.insert(0, res)
        res //= 2
    return out


def to_resolution(fields: torch.Tensor, target: int, how: str = "naive_downsample") -> torch.Tensor:
    """fields [B,C,n] or [B,C,M,N] at full resolution -> resolution ``target``"""
    nd = fields.dim() - 2
    full = fields.shape[-1]
    if target == full:
        return fields
    if how == "naive_downsample":
        r = full // target
        return (fields[..., ::r] if nd == 1 else fields[..., ::r, ::r]).contiguous()
    if how == "resize":
        return resize_1d(fields, target) if nd == 1 else resize(fields, (target, target))
    raise ValueError(f"unknown evaluation_type {how}")


@torch.no_grad()
def evaluate_all_resolutions(model, test_x: torch.Tensor, test_y: torch.Tensor, max_resolution: Optional[int] = None,
                             min_resolution: int = 32, how: str = "naive_downsample", batch_size: int = 16,
                             x_encode: Optional[Callable] = None, y_decode: Optional[Callable] = None,
                             device="cuda") -> Dict[int, float]:
    """{resolution: mean relative L2 over the test samples}; encode/decode are the x / y normalisers.  With
    torch.distributed initialised every rank evaluates samples rank, rank + world, ... and the sample-weighted sums
    are reduced (NaN for a resolution without samples)."""
    import torch.distributed as dist
    on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
    model.eval()
    loss_fn = RelativeL2Loss(size_average=True)
    full = test_x.shape[-1]
    mine_x, mine_y = test_x[rank::world], test_y[rank::world]
    resolutions = get_lower_resolutions(max_resolution or full, min_resolution)
    acc = torch.zeros(len(resolutions), 2, dtype=torch.float64, device=device)
    from rpde.ops import frozen_weights
    with frozen_weights():                      # one weight preparation per layer (and grid) for the whole sweep
        for k, res in enumerate(resolutions):
            for i in range(0, mine_x.shape[0], batch_size):
                x = to_resolution(mine_x[i:i + batch_size].to(device), res, how)
                y = to_resolution(mine_y[i:i + batch_size].to(device), res, how)
                pred = model(x_encode(x) if x_encode else x)
                if y_decode:
                    pred = y_decode(pred)
                acc[k, 0] += loss_fn(pred, y).double() * x.shape[0]
                acc[k, 1] += x.shape[0]
    if on:
        dist.all_reduce(acc)
    acc = acc.cpu()
    return {res: (float(acc[k, 0] / acc[k, 1]) if float(acc[k, 1]) > 0 else float("nan")) for k, res in enumerate(resolutions)}
