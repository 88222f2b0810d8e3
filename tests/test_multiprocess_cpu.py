"""CPU, world_size 2 over gloo: the N>1 path of the data-parallel design --
flat gradient bucket + one all-reduce == single-process full-batch gradients;
the sharded resolution-grouped loader gives every rank the same resolution at
the same step, disjoint samples, equal batch sizes."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import DROPIN, REPO


def _worker(rank, world, port, tmp):
    for p in (REPO, DROPIN):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rpde.parallel import FlatGradBucket
    from train.mres_training import ResolutionGroupedDataLoader
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 1))
    cw = torch.nn.Parameter(torch.randn(3, dtype=torch.cfloat))           # complex parameters (FNO weights)
    params = list(model.parameters()) + [cw]
    bucket = FlatGradBucket(params)
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 1, generator=g)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
    bucket.zero()
    loss = ((model(xs) - ys) ** 2).mean() + (cw.abs() ** 2).sum() * xs.mean()
    loss.backward()
    assert all(p.grad.data_ptr() >= bucket.flat.data_ptr() for p in params)   # grads are views into the bucket
    bucket.all_reduce_mean()
    # single-process reference on the full batch (mean of the two local means)
    ref_model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 1))
    ref_model.load_state_dict(model.state_dict())
    rcw = cw.detach().clone().requires_grad_(True)
    ref = 0.5 * sum(((ref_model(X[r * 4:(r + 1) * 4]) - Y[r * 4:(r + 1) * 4]) ** 2).mean()
                    + (rcw.abs() ** 2).sum() * X[r * 4:(r + 1) * 4].mean() for r in range(2))
    ref.backward()
    for p, q in zip(model.parameters(), ref_model.parameters()):
        assert torch.allclose(p.grad, q.grad, atol=1e-6), (p.grad, q.grad)
    assert torch.allclose(cw.grad, rcw.grad, atol=1e-6)

    # sharded multi-resolution loader
    data = [(torch.full((1, r), float(i)), torch.full((1, r), float(i))) for i, r in
            enumerate([64] * 9 + [128] * 5 + [256] * 8)]
    loader = ResolutionGroupedDataLoader(data, batch_size=2, shuffle=True, seed=7, rank=rank, world_size=world,
                                         verbose=False)
    steps = [(x.shape[-1], x[:, 0, 0].tolist()) for x, _ in loader]
    torch.save(steps, os.path.join(tmp, f"steps{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_and_sharded_loader_world2(tmp_path):
    port = 29600 + os.getpid() % 300
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    s0 = torch.load(os.path.join(tmp_path, "steps0.pt"))
    s1 = torch.load(os.path.join(tmp_path, "steps1.pt"))
    assert len(s0) == len(s1) == 9 // 4 + 5 // 4 + 8 // 4
    for (r0, i0), (r1, i1) in zip(s0, s1):
        assert r0 == r1 and len(i0) == len(i1) == 2 and not set(i0) & set(i1)


def test_grouped_loader_single_process_matches_reference_semantics():
    """quirk Q15: 5/3/6 samples at 64/128/256 with batch_size 4 -> batches 4,1 / 3 / 4,2 and len()==5"""
    sys.path.insert(0, DROPIN)
    from train.mres_training import ResolutionGroupedDataLoader, create_grouped_dataloaders
    data = [(torch.zeros(1, r), torch.zeros(1, r)) for r in [64] * 5 + [128] * 3 + [256] * 6]
    dl = ResolutionGroupedDataLoader(data, batch_size=4, shuffle=False, verbose=False)
    sizes = [(x.shape[-1], x.shape[0]) for x, _ in dl]
    assert len(dl) == 5 and sorted(sizes) == [(64, 1), (64, 4), (128, 3), (256, 2), (256, 4)]
    for x, y in ResolutionGroupedDataLoader(data, batch_size=4, shuffle=True, seed=3, verbose=False):
        assert x.shape == y.shape and x.dim() == 3
    tr, va, te = create_grouped_dataloaders(data, data, data, 4, seed=1)
    assert len(tr) == len(va) == len(te) == 5
