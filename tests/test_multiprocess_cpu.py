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
    bucket.all_reduce_mean()                                                  # gathers first, then reduces
    assert all(p.grad.data_ptr() >= bucket.flat.data_ptr() for p in params)   # grads are views into the bucket
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


# ---------------------------------------------------------------------------------------------------------------
# evaluation loaders keep every sample under world_size > 1; an empty split reads NaN, not 0.0
# ---------------------------------------------------------------------------------------------------------------
def _eval_worker(rank, world, port, tmp):
    for p in (REPO, DROPIN):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import train.training as T
    from train.mres_training import ResolutionGroupedDataLoader

    class CpuRelL2(torch.nn.Module):                       # the HIP loss has no CPU path; evaluate() is loss-agnostic here
        def __init__(self, size_average=True):
            super().__init__()

        def forward(self, a, b):
            a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
            return ((a - b).norm(dim=1) / (b.norm(dim=1) + 1e-8)).mean()
    T.RelativeL2Loss = CpuRelL2
    g = torch.Generator().manual_seed(0)
    data = [(torch.randn(1, 16, generator=g), torch.randn(1, 16, generator=g)) for _ in range(3)]   # n_test=3 < batch*world
    model = torch.nn.Identity()
    loader = ResolutionGroupedDataLoader(data, 2, shuffle=False, seed=0, rank=rank, world_size=world, verbose=False)
    assert loader.drop_last is False and sum(x.shape[0] for x, _ in loader) == (2 if rank == 0 else 1)
    got = T.evaluate(model, loader, normalization_type="simple", device="cpu")
    want = float(torch.stack([CpuRelL2()(x[None], y[None]) for x, y in data]).mean())
    assert abs(got - want) < 1e-6, (got, want)
    # training loaders still drop incomplete global batches (identical shapes on every rank) -> here: none at all
    tl = ResolutionGroupedDataLoader(data, 2, shuffle=True, seed=0, rank=rank, world_size=world, verbose=False)
    assert len(tl) == 0 and list(tl) == []
    # an empty evaluation reads NaN on every rank, never 0.0
    empty = ResolutionGroupedDataLoader([], 2, shuffle=False, seed=0, rank=rank, world_size=world, verbose=False)
    val = T.evaluate(model, empty, normalization_type="simple", device="cpu")
    assert val != val
    try:
        T.evaluate(model, loader, normalization_type="zscore", device="cpu")
        raise AssertionError("unknown normalisation must raise like the reference")
    except ValueError:
        pass
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_evaluation_counts_every_sample_once(tmp_path):
    from rpde.launch import free_port
    mp.spawn(_eval_worker, args=(2, free_port(), str(tmp_path)), nprocs=2, join=True)


def test_untouched_parameters_keep_grad_none_and_reference_modules_stay_importable():
    from rpde.parallel import FlatGradBucket
    a, b = torch.nn.Linear(3, 2), torch.nn.Linear(3, 2)           # b takes no part in the graph
    params = list(a.parameters()) + list(b.parameters())
    bucket = FlatGradBucket(params)
    opt = torch.optim.AdamW(params, lr=0.1, weight_decay=0.5)
    before = b.weight.detach().clone()
    for _ in range(2):
        bucket.zero()
        a(torch.ones(1, 3)).sum().backward()
        bucket.all_reduce_mean()
        bucket.detach_untouched()
        assert b.weight.grad is None and a.weight.grad is not None
        opt.step()
    assert torch.equal(b.weight, before)                          # no weight decay, no moments: as after zero_grad()
    bucket.zero()                                                 # every grad None: backward assigns, gather() collects
    assert all(p.grad is None for p in params)
    (a(torch.ones(1, 3)).sum() + b(torch.ones(1, 3)).sum()).backward()
    bucket.gather()
    assert b.weight.grad is not None and b.weight.grad.data_ptr() >= bucket.flat.data_ptr()
    assert torch.equal(bucket.flat[bucket.offsets[2]:bucket.offsets[2] + 6].view(2, 3), torch.ones(2, 3))
    # drop-in route of INTEGRATION.md A.2: with this tree ahead of the reference root the shared package names are
    # namespace packages, so the reference's own modules (utils/naive_utils.py ...) still resolve
    import importlib.util
    ref = "/root/reference"
    if os.path.isdir(ref):
        sys.path.append(ref)
        try:
            importlib.invalidate_caches()
            for pkg in ("utils", "models", "train"):
                sys.modules.pop(pkg, None) if not hasattr(sys.modules.get(pkg), "__file__") else None
            spec = importlib.util.find_spec("utils.naive_utils")
            assert spec is not None and spec.origin.startswith(ref)
            assert importlib.util.find_spec("utils.loss").origin.startswith(DROPIN)
        finally:
            sys.path.remove(ref)
