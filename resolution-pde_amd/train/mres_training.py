"""Resolution-grouped batching for true multi-resolution training
(reference: train/mres_training.py:11-166), with the two additions the
one-process-per-GPU design needs: a seed (the reference shuffles with the
unseeded global ``random`` module) and rank sharding.

Semantics kept from the reference: every sample is
materialised once in ``__init__``; samples are grouped by ``x.shape[-1]`` only;
each batch holds a single resolution; batch order is shuffled across
resolutions; ragged tail batches are kept (single process).  With
``world_size > 1`` a *global* batch of ``batch_size * world_size`` samples of
one resolution is cut into equal contiguous rank slices and incomplete global
batches are dropped, so every rank sees the same resolution at the same step
(identical kernel shapes and collective sizes) and the mean over the
concatenated batch equals the reference's DataParallel loss.
"""
from __future__ import annotations

import random
from collections import defaultdict
from typing import Dict, Iterator, List, Optional, Tuple

import torch
from torch.utils.data import Dataset, Sampler


def multires_collate_fn(batch):
    """lists instead of stacked tensors, for variable spatial sizes"""
    n = len(batch[0])
    if n not in (2, 3):
        raise ValueError(f"Unexpected batch item length: {n}")
    return tuple([item[i] for item in batch] for i in range(n))


class SimpleDataset(Dataset):
    def __init__(self, samples):
        self.samples = samples

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, idx):
        return self.samples[idx]


def _group_by_resolution(dataset) -> Dict[int, List[int]]:
    groups: Dict[int, List[int]] = defaultdict(list)
    for idx in range(len(dataset)):
        groups[int(dataset[idx][0].shape[-1])].append(idx)
    return groups


class ResolutionGroupedSampler(Sampler):
    """index sampler: same-resolution runs of ``batch_size`` indices"""

    def __init__(self, dataset, batch_size, shuffle=True, seed: Optional[int] = None):
        self.dataset, self.batch_size, self.shuffle = dataset, batch_size, shuffle
        self.rng = random.Random(seed) if seed is not None else random
        self.resolution_groups = _group_by_resolution(dataset)

    def __iter__(self):
        batches = []
        for _, indices in self.resolution_groups.items():
            indices = list(indices)
            if self.shuffle:
                self.rng.shuffle(indices)
            batches += [indices[i:i + self.batch_size] for i in range(0, len(indices), self.batch_size)]
        if self.shuffle:
            self.rng.shuffle(batches)
        for b in batches:
            yield from b

    def __len__(self):
        return len(self.dataset)


class ResolutionGroupedDataLoader:
    def __init__(self, dataset, batch_size, shuffle=True, num_workers=0, seed: Optional[int] = None,
                 rank: int = 0, world_size: int = 1, verbose: bool = True, drop_last: Optional[bool] = None,
                 pin_memory: bool = False):
        """drop_last (world_size > 1 only; default = ``shuffle``): True for training -- equal rank slices of whole
        global batches, so that every rank runs the same shapes in the same order beside the gradient all-reduce;
        False for validation / test -- every sample is used exactly once: rank r takes samples r, r+world, ... of
        each resolution and keeps its ragged (possibly empty) tail; the caller reduces a sample-weighted sum.
        pin_memory: batches are stacked into page-locked staging buffers (two per resolution and tensor, used in
        turn), so that ``batch.to(device, non_blocking=True)`` is an asynchronous copy; a batch is valid until the
        second-next batch of the same resolution is drawn"""
        self.dataset, self.batch_size, self.shuffle = dataset, int(batch_size), shuffle
        self.drop_last = bool(shuffle) if drop_last is None else bool(drop_last)
        self.seed, self.rank, self.world_size = seed, int(rank), int(world_size)
        self.epoch = 0
        self.pin_memory = bool(pin_memory) and torch.cuda.is_available()
        self._pinned: Dict[Tuple[int, int, int], torch.Tensor] = {}
        self._turn: Dict[int, int] = defaultdict(int)
        if not 0 <= self.rank < self.world_size:
            raise ValueError(f"rank {rank} outside world of {world_size}")
        self.resolution_groups: Dict[int, List[Tuple[torch.Tensor, torch.Tensor]]] = defaultdict(list)
        for idx in range(len(dataset)):
            x, y = dataset[idx]
            self.resolution_groups[int(x.shape[-1])].append((x, y))
        if verbose:
            print("Created resolution groups:")
            for res, samples in self.resolution_groups.items():
                print(f"  Resolution {res}: {len(samples)} samples")

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def _plan(self) -> List[Tuple[int, List[int]]]:
        """[(resolution, sample indices of THIS rank)] for one epoch"""
        rng = random.Random(self.seed + self.epoch) if self.seed is not None else random
        if self.world_size > 1 and self.seed is None:
            raise ValueError("sharded loading needs a seed: every rank must draw the same order")
        glob = self.batch_size * self.world_size
        plan: List[Tuple[int, List[int]]] = []
        for res, samples in self.resolution_groups.items():
            order = list(range(len(samples)))
            if self.shuffle:
                rng.shuffle(order)
            if self.world_size == 1:
                plan += [(res, order[i:i + self.batch_size]) for i in range(0, len(order), self.batch_size)]
            elif not self.drop_last:
                mine = order[self.rank::self.world_size]
                plan += [(res, mine[i:i + self.batch_size]) for i in range(0, len(mine), self.batch_size)]
            else:
                for i in range(0, len(order) - glob + 1, glob):
                    lo = i + self.rank * self.batch_size
                    plan.append((res, order[lo:lo + self.batch_size]))
        if self.shuffle:
            rng.shuffle(plan)
        return plan

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        for res, idxs in self._plan():
            samples = self.resolution_groups[res]
            if not self.pin_memory:
                yield (torch.stack([samples[i][0] for i in idxs]), torch.stack([samples[i][1] for i in idxs]))
                continue
            turn = self._turn[res]
            self._turn[res] = turn ^ 1
            out = []
            for which in (0, 1):
                parts = [samples[i][which] for i in idxs]
                key = (res, which, turn)
                buf = self._pinned.get(key)
                if buf is None or buf.shape[0] < len(parts) or buf.shape[1:] != parts[0].shape or buf.dtype != parts[0].dtype:
                    buf = torch.empty((self.batch_size,) + tuple(parts[0].shape), dtype=parts[0].dtype).pin_memory()
                    self._pinned[key] = buf
                # (sample by sample: torch.stack(..., out=) into a preallocated buffer took ~50 ms for 32 fields of 256^2
                #  against ~1 ms this way.  Root cause, found later: torch's intra-op pool is sized by the host's 256 logical
                #  CPUs inside a 16-core quota -- rpde.launch.limit_host_threads(), called by the entry points, fixes that
                #  for every host op; the small copies stay, they never wake the pool)
                view = buf[:len(parts)]
                for i, part in enumerate(parts):
                    view[i].copy_(part)
                out.append(view)
            yield tuple(out)
        self.epoch += 1

    def __len__(self):
        glob = self.batch_size * self.world_size
        if self.world_size == 1:
            return sum((len(s) + self.batch_size - 1) // self.batch_size for s in self.resolution_groups.values())
        if not self.drop_last:
            return sum((len(range(self.rank, len(s), self.world_size)) + self.batch_size - 1) // self.batch_size
                       for s in self.resolution_groups.values())
        return sum(len(s) // glob for s in self.resolution_groups.values())


def create_grouped_dataloaders(train_dataset, val_dataset, test_dataset, batch_size, seed: Optional[int] = None,
                               rank: int = 0, world_size: int = 1):
    train = ResolutionGroupedDataLoader(train_dataset, batch_size, shuffle=True, seed=seed, rank=rank,
                                        world_size=world_size)
    val = ResolutionGroupedDataLoader(val_dataset, batch_size, shuffle=False, seed=seed, rank=rank,
                                      world_size=world_size)
    test = ResolutionGroupedDataLoader(test_dataset, batch_size, shuffle=False, seed=seed, rank=rank,
                                       world_size=world_size)
    return train, val, test
