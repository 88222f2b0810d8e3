"""What the two 1-D true-multi-resolution loaders share (reference dataloaders/ks_naive_true_multires.py:173-533 and
burger_naive_true_multires.py:139-420 are the same class twice, differing in where a resolution's file lives, how the
array is found inside it, and which time steps make a pair):

  file leg        for every resolution named in ``data_mres_size`` with a non-zero target: read [N,T,X], stride samples
                  and time, split 0.8 / 0.1 / 0.1 in file order, and -- when 0 < target < split size -- draw
                  int(target * split_ratio[split]) trajectories WITHOUT replacement under the global numpy seed
                  ``random_seed + resolution + split_idx``; then pair.
  downsample leg  ``add_res``: the ``downsample_from_res`` file is split first (no reductions), drawn WITH replacement
                  under ``random_seed + target + split_idx + 10000``, strided over (draw, time), then either subsampled
                  by ``full // target`` or low-passed at ``target / full * lowpass_cutoff_ratio`` ON THE ORIGINAL GRID
                  (such samples share the base resolution's shape).
  rollout set     the test split's whole trajectories of the file leg (same draw, split index 2), un-normalised.
  normalisation   "simple": one mean / std over every value of the training pairs; "minmax": global extrema of inputs
                  and of targets.  The return arity follows ``normalization_type`` (6 or 8 values), also when
                  ``data_normalizer`` is off.

Deliberate difference: the reference's "minmax" statistics go through a DataLoader with batch 512 and ``torch.cat``,
which raises on a training set with more than one grid size; here the extrema are taken item by item.

HDF5 is read through dataloaders/_store.py (h5py when importable; an ``.npz`` archive with the same member names next
to the expected file otherwise).  The reference reads these data sets through h5py only and holds no fixture of
them: nothing here can be pinned against it -- "parity unpinned"; tests/test_data_layer_cpu.py checks the semantics
above on synthetic archives."""
from __future__ import annotations

import os
from typing import Callable, Dict, Iterable, List, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

from dataloaders.ns_naive_markov import NormalizedDataset, SimpleNormalizer
from utils.low_pass_filter import lowpass_filter_1d

SPLIT_INDEX = {"train": 0, "valid": 1, "val": 1, "test": 2}


def with_npz_fallback(path: str) -> Optional[str]:
    """the file itself, or the .npz archive of the same stem (the build image has no h5py), or None"""
    if os.path.exists(path):
        return path
    alt = os.path.splitext(path)[0] + ".npz"
    return alt if os.path.exists(alt) else None


def split_rows(data: np.ndarray, split: str, ratio) -> np.ndarray:
    n = data.shape[0]
    train_end = int(n * ratio[0])
    val_end = train_end + int(n * ratio[1])
    if split == "train":
        return data[:train_end]
    if split in ("val", "valid"):
        return data[train_end:val_end]
    if split == "test":
        return data[val_end:]
    raise ValueError(f"Invalid split: {split}")


def draw_for_split(part: np.ndarray, target: int, split_idx: int, ratio, seed: int) -> Optional[np.ndarray]:
    """the file leg's subsample; None: nothing allocated to this split"""
    if 0 < target < part.shape[0]:
        take = int(target * ratio[split_idx])
        if take <= 0:
            return None
        np.random.seed(seed)                                   # the reference seeds the global stream
        return part[np.random.choice(part.shape[0], take, replace=False)]
    return part


class TrajectoryList(Dataset):
    """whole trajectories [T,X] with their provenance (reference: *TrajectoryDatasetFromExtracted)"""

    def __init__(self, trajectories, trajectory_info):
        self.trajectories, self.trajectory_info = trajectories, trajectory_info

    def __len__(self):
        return len(self.trajectories)

    def __getitem__(self, idx):
        return self.trajectories[idx]

    def get_trajectory_info(self, idx):
        return self.trajectory_info[idx]

    def get_all_info(self):
        return self.trajectory_info


def extract_test_trajectories(locate: Callable[[int], Optional[str]], read: Callable[[str], np.ndarray], data_mres_size,
                              split_ratio, reduced_batch, reduced_resolution_t, random_seed) -> Tuple[list, list]:
    trajectories, info = [], []
    for resolution, target in data_mres_size.items():
        if target == 0:
            continue
        path = locate(int(resolution))
        if path is None:
            print(f"Warning: no file for resolution {resolution}. Skipping.")
            continue
        u = read(path)[::reduced_batch, ::reduced_resolution_t, :]
        test = draw_for_split(split_rows(u, "test", split_ratio), int(target), 2, split_ratio, random_seed + int(resolution) + 2)
        if test is None:
            print(f"  No test samples allocated for resolution {resolution}")
            continue
        for i in range(test.shape[0]):
            trajectories.append(torch.tensor(test[i], dtype=torch.float))
            info.append({"resolution": resolution, "original_index": i, "source": f"res_{resolution}_file"})
    return trajectories, info


class TrueMultiRes1dMarkovDataset(Dataset):
    """items (x, y), each [1, X]; ``pair(u) -> (inputs, targets)`` cuts [n,T,X] along time"""

    def __init__(self, locate: Callable[[int], Optional[str]], locate_base: Callable[[int], Optional[str]],
                 read: Callable[[str], np.ndarray], pair: Callable[[np.ndarray], Tuple[np.ndarray, np.ndarray]],
                 default_mres: Dict[int, int], default_add: Dict[int, int], reduced_batch=1, reduced_resolution_t=1,
                 data_mres_size=None, add_res=None, add_res_samples=None, downsample_from_res=None,
                 use_low_pass_filter=False, lowpass_cutoff_ratio=1.0, split_ratio=None, random_seed=42, split="train"):
        self.random_seed, self.split = random_seed, split
        self.use_low_pass_filter, self.lowpass_cutoff_ratio = use_low_pass_filter, lowpass_cutoff_ratio
        self._read, self._pair, self._locate_base = read, pair, locate_base
        split_ratio = [0.8, 0.1, 0.1] if split_ratio is None else split_ratio
        data_mres_size = dict(default_mres) if data_mres_size is None else {int(k): int(v) for k, v in dict(data_mres_size).items()}
        if downsample_from_res is None and data_mres_size:       # highest resolution that has samples, else highest named
            have = [r for r, n in data_mres_size.items() if n > 0]
            downsample_from_res = max(have) if have else max(data_mres_size)
        self.downsample_from_res = downsample_from_res
        add_res_samples = dict(default_add) if add_res_samples is None else {int(k): int(v) for k, v in dict(add_res_samples).items()}

        self.x: List[torch.Tensor] = []
        self.y: List[torch.Tensor] = []
        self.resolution_info: List[str] = []
        split_idx = SPLIT_INDEX.get(split, 0)
        for resolution, target in data_mres_size.items():
            if target == 0:
                continue
            path = locate(resolution)
            if path is None:
                print(f"Warning: no file for resolution {resolution}. Skipping.")
                continue
            u = read(path)[::reduced_batch, ::reduced_resolution_t, :]
            part = draw_for_split(split_rows(u, split, split_ratio), target, split_idx, split_ratio,
                                  random_seed + resolution + split_idx)
            if part is None:
                print(f"  No samples allocated for {split} split at resolution {resolution}")
                continue
            self._append(part, f"{resolution}_file")
        if add_res is not None and add_res_samples is not None:
            if self.downsample_from_res is not None:
                self._add_downsampled(self.downsample_from_res, add_res, add_res_samples, split_ratio, reduced_batch,
                                      reduced_resolution_t)
            else:
                print("Warning: No resolution specified for downsampling and no available resolutions found.")
        assert len(self.x) == len(self.y), "Invalid input output pairs"

    def _append(self, u: np.ndarray, tag: str) -> None:
        a, b = self._pair(u)
        xs = torch.tensor(a, dtype=torch.float).reshape(-1, 1, a.shape[-1])
        ys = torch.tensor(b, dtype=torch.float).reshape(-1, 1, b.shape[-1])
        self.x.extend(xs.unbind(0))
        self.y.extend(ys.unbind(0))
        self.resolution_info.extend([tag] * xs.shape[0])

    def _add_downsampled(self, base_resolution, add_res: Iterable[int], add_res_samples: Dict[int, int], split_ratio,
                         reduced_batch, reduced_resolution_t) -> None:
        path = self._locate_base(int(base_resolution))
        if path is None:
            print(f"Warning: Base file for resolution {base_resolution} does not exist. Cannot create downsampled data.")
            return
        part = split_rows(self._read(path), self.split, split_ratio)
        full = part.shape[2]
        split_idx = SPLIT_INDEX.get(self.split, 0)
        for target in add_res:
            target = int(target)
            if target >= full:
                print(f"  Warning: Target resolution {target} >= original {full}. Skipping.")
                continue
            take = int(add_res_samples.get(target, 100) * split_ratio[split_idx])
            if take == 0:
                print(f"  No downsampled samples allocated for {self.split} split at resolution {target}")
                continue
            np.random.seed(self.random_seed + target + split_idx + 10000)
            drawn = part[np.random.choice(part.shape[0], take, replace=True)]
            drawn = drawn[::reduced_batch, ::reduced_resolution_t, :]
            if self.use_low_pass_filter:
                cutoff = (target / full) * self.lowpass_cutoff_ratio
                low = lowpass_filter_1d(torch.from_numpy(np.ascontiguousarray(drawn)).float(), cutoff_ratio=cutoff).numpy()
            else:
                low = drawn[:, :, ::full // target]
            self._append(low, f"{target}_downsampled_{'lowpass' if self.use_low_pass_filter else 'naive'}")

    def __len__(self):
        return len(self.x)

    def __getitem__(self, idx):
        return self.x[idx], self.y[idx]

    def get_resolution_info(self):
        return self.resolution_info


class MinMaxNormalizedDataset(Dataset):
    def __init__(self, dataset, min_data, max_data, min_model, max_model):
        self.dataset = dataset
        self.min_data, self.max_data, self.min_model, self.max_model = min_data, max_data, min_model, max_model

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, idx):
        x, y = self.dataset[idx]
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x).float()
        if isinstance(y, np.ndarray):
            y = torch.from_numpy(y).float()
        return (x - self.min_data) / (self.max_data - self.min_data), (y - self.min_model) / (self.max_model - self.min_model)


def normalise_and_pack(train, val, test, rollout, data_normalizer: bool, normalization_type: str):
    """the tail both factories share: statistics from the training pairs, wrapped splits, return arity by type"""
    x_normalizer = y_normalizer = None
    min_data = max_data = min_model = max_model = None
    if data_normalizer:
        if normalization_type == "simple":
            xs = torch.cat([x.reshape(-1) for x, _ in train])
            ys = torch.cat([y.reshape(-1) for _, y in train])
            x_normalizer, y_normalizer = SimpleNormalizer(xs.mean(), xs.std()), SimpleNormalizer(ys.mean(), ys.std())
            train, val, test = (NormalizedDataset(d, x_normalizer, y_normalizer) for d in (train, val, test))
        elif normalization_type == "minmax":
            min_data = min(float(x.min()) for x, _ in train)
            max_data = max(float(x.max()) for x, _ in train)
            min_model = min(float(y.min()) for _, y in train)
            max_model = max(float(y.max()) for _, y in train)
            train, val, test = (MinMaxNormalizedDataset(d, min_data, max_data, min_model, max_model) for d in (train, val, test))
        else:
            raise ValueError(f"Invalid normalization_type: {normalization_type}. Must be 'simple' or 'minmax'")
    if normalization_type == "simple":
        return train, val, test, rollout, x_normalizer, y_normalizer
    return train, val, test, rollout, min_data, max_data, min_model, max_model
