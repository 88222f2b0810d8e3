"""Two ranks sharing cuda:0 over gloo (the 1-GPU rehearsal of the one-process-per-GPU design): the real FFNO2D,
each rank on its half of the batch, gradients averaged through the flat bucket == single-process gradients on the
full batch (reference semantics: nn.DataParallel, main_2d.py:89-94,147-149)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import DROPIN, REPO

pytestmark = pytest.mark.gpu
CFG = dict(in_channels=1, out_channels=1, width=64, n_layers=2, n_modes=8, factor=4, ff_weight_norm=True, n_ff_layers=3,
           layer_norm=True, dropout=0.0)


def _worker(rank, world, port, path):
    for p in (REPO, DROPIN):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from models.ffno import FFNO2D
    from rpde.parallel import FlatGradBucket
    from utils.loss import RelativeL2Loss
    from utils.synthetic import advance, random_fields
    dev = "cuda:0"
    torch.manual_seed(0)
    model = FFNO2D(**CFG).to(dev).train()
    bucket = FlatGradBucket(model.parameters())
    x = random_fields(4, 64, 2, seed=7)
    y = advance(x, 2)
    xs, ys = x[rank * 2:(rank + 1) * 2].to(dev), y[rank * 2:(rank + 1) * 2].to(dev)
    bucket.zero()
    RelativeL2Loss()(model(xs), ys).backward()
    bucket.gather()
    flat_cpu = bucket.flat.cpu()                       # gloo reduces host tensors
    dist.all_reduce(flat_cpu)
    flat_cpu /= world
    if rank == 0:
        torch.save(flat_cpu, path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_reproduce_full_batch_gradients(gpu_device, tmp_path):
    from models.ffno import FFNO2D
    from rpde.launch import free_port
    from rpde.parallel import FlatGradBucket
    from utils.loss import RelativeL2Loss
    from utils.synthetic import advance, random_fields
    path = str(tmp_path / "flat.pt")
    mp.spawn(_worker, args=(2, free_port(), path), nprocs=2, join=True)
    got = torch.load(path, weights_only=True)
    torch.manual_seed(0)
    model = FFNO2D(**CFG).to(gpu_device).train()
    bucket = FlatGradBucket(model.parameters())
    x = random_fields(4, 64, 2, seed=7)
    y = advance(x, 2)
    bucket.zero()
    RelativeL2Loss()(model(x.to(gpu_device)), y.to(gpu_device)).backward()
    bucket.gather()
    ref = bucket.flat.cpu()
    rel = float((got - ref).norm() / ref.norm())
    assert rel < 1e-5, rel
