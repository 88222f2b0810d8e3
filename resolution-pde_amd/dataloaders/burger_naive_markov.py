"""Burgers (PDEBench 1-D) trajectories -> Markov pairs, rollout trajectories, normalisers.  Public names, arguments,
defaults and return values of the reference's dataloaders/burger_naive_markov.py
(BurgersTrajectoryDatasetFromExtracted :13-31, extract_burgers_test_trajectories_for_rollout_single :34-119,
H5pyMarkovDataset :124-201, burger_markov_dataset :204-453).

File members: ``tensor`` [N,T,X] and ``x-coordinate`` [X] (PDEBench).  Pairs are (u[t], u[t+1]) for t = 1 .. T-2,
each [1,X]; the 0.8 / 0.1 / 0.1 split is a seeded random split of the PAIRS (torch.Generator 42), while the rollout
set is the last 10 % of the TRAJECTORIES in file order.

The return arity follows ``normalization_type`` as in the reference, also when ``data_normalizer`` is off:
  "simple" -> train, val, test, rollout, x_normalizer, y_normalizer
  "minmax" -> train, val, test, rollout, min_data, max_data, min_model, max_model   (the default)

Formats: ``.h5`` / ``.hdf5`` through h5py as the reference, ``.npz`` with the same member names in addition
(dataloaders/_store.py).  The reference has no HDF5-free leg for Burgers: "parity unpinned"
(tests/test_data_layer_cpu.py checks the semantics above on synthetic archives)."""
from __future__ import annotations

import os

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset, random_split

from dataloaders._store import Store
from dataloaders.ks_naive_markov import _reduce_1d
from dataloaders.ns_naive_markov import NormalizedDataset, SimpleNormalizer


class BurgersTrajectoryDatasetFromExtracted(Dataset):
    """whole test trajectories [T,X] for rollout evaluation"""

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


def _read_tensor(saved_folder, filename, with_grid=False):
    path = os.path.join(os.path.abspath(saved_folder), filename)
    with Store(path) as f:
        u = np.array(f["tensor"], dtype=np.float32)
        grid = np.array(f["x-coordinate"], dtype=np.float32) if with_grid else None
    return u, grid


def extract_burgers_test_trajectories_for_rollout_single(filename, saved_folder, reduced_batch=1, reduced_resolution=1,
                                                          reduced_resolution_t=1, use_low_pass_filter=False,
                                                          lowpass_cutoff_ratio=1.0, num_samples_max=-1, split_ratio=None,
                                                          random_seed=42):
    """the trajectories behind the last (1 - split_ratio[0] - split_ratio[1]) share of the file, reduced like the
    training data, before any pairing"""
    split_ratio = [0.8, 0.1, 0.1] if split_ratio is None else split_ratio
    u, _ = _read_tensor(saved_folder, filename)
    u = _reduce_1d(u, reduced_batch, reduced_resolution, reduced_resolution_t, use_low_pass_filter, lowpass_cutoff_ratio,
                   num_samples_max)
    n = u.shape[0]
    test = u[int(n * split_ratio[0]) + int(n * split_ratio[1]):]
    trajectories = [torch.tensor(test[i], dtype=torch.float) for i in range(test.shape[0])]
    info = [{"original_index": i, "source": "single_resolution_file"} for i in range(test.shape[0])]
    return trajectories, info


class H5pyMarkovDataset(Dataset):
    """x = u[:, 1:-1], y = u[:, 2:] flattened over (sample, time) to [(N*(T-2)), 1, X] -- numpy arrays, as the
    reference keeps them (the collate function / the normalising wrapper makes tensors); ``grid`` [X,1]"""

    def __init__(self, filename, saved_folder, reduced_batch=1, reduced_resolution=1, reduced_resolution_t=1,
                 use_low_pass_filter=False, lowpass_cutoff_ratio=1.0, num_samples_max=-1, **kwargs):
        self.use_low_pass_filter, self.lowpass_cutoff_ratio = use_low_pass_filter, lowpass_cutoff_ratio
        u, grid = _read_tensor(saved_folder, filename, with_grid=True)
        self.data = _reduce_1d(u, reduced_batch, reduced_resolution, reduced_resolution_t, use_low_pass_filter,
                               lowpass_cutoff_ratio, num_samples_max)
        if reduced_resolution > 1 and not use_low_pass_filter:
            grid = grid[::reduced_resolution]
        self.grid = torch.tensor(grid, dtype=torch.float).unsqueeze(-1)
        m = self.data.shape[-1]
        self.x = self.data[:, 1:-1, :].reshape(-1, 1, m)
        self.y = self.data[:, 2:, :].reshape(-1, 1, m)
        assert len(self.x) == len(self.y), "Invalid input output pairs"

    def __len__(self):
        return len(self.x)

    def __getitem__(self, idx):
        return self.x[idx], self.y[idx]


class MinMaxNormalizedDataset(Dataset):
    """x -> (x - min_data) / (max_data - min_data), y likewise with the model-side range"""

    def __init__(self, dataset, min_data, max_data, min_model, max_model):
        self.dataset = dataset
        self.min_data, self.max_data, self.min_model, self.max_model = min_data, max_data, min_model, max_model

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, idx):
        x, y = self.dataset[idx][:2]
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x).float()
        if isinstance(y, np.ndarray):
            y = torch.from_numpy(y).float()
        return (x - self.min_data) / (self.max_data - self.min_data), (y - self.min_model) / (self.max_model - self.min_model)


def burger_markov_dataset(filename, saved_folder, data_normalizer=True, normalization_type="minmax",
                          use_low_pass_filter=False, lowpass_cutoff_ratio=1.0, **kwargs):
    full = H5pyMarkovDataset(filename, saved_folder, use_low_pass_filter=use_low_pass_filter,
                             lowpass_cutoff_ratio=lowpass_cutoff_ratio, **kwargs)
    n = len(full)
    n_train, n_val = int(0.8 * n), int(0.1 * n)
    train, val, test = random_split(full, [n_train, n_val, n - n_train - n_val], generator=torch.Generator().manual_seed(42))
    keep = ("reduced_batch", "reduced_resolution", "reduced_resolution_t", "num_samples_max")
    trajs, info = extract_burgers_test_trajectories_for_rollout_single(
        filename=filename, saved_folder=saved_folder, use_low_pass_filter=use_low_pass_filter,
        lowpass_cutoff_ratio=lowpass_cutoff_ratio, split_ratio=[0.8, 0.1, 0.1], random_seed=42,
        **{k: v for k, v in kwargs.items() if k in keep})
    rollout = BurgersTrajectoryDatasetFromExtracted(trajs, info)
    x_normalizer = y_normalizer = min_data = max_data = min_model = max_model = None
    if data_normalizer:
        if normalization_type == "simple":
            xs = torch.cat([torch.as_tensor(x).reshape(-1) for x, _ in train])
            ys = torch.cat([torch.as_tensor(y).reshape(-1) for _, y in train])
            x_normalizer, y_normalizer = SimpleNormalizer(xs.mean(), xs.std()), SimpleNormalizer(ys.mean(), ys.std())
            train, val, test = (NormalizedDataset(d, x_normalizer, y_normalizer) for d in (train, val, test))
        elif normalization_type == "minmax":
            lo_x = lo_y = float("inf")
            hi_x = hi_y = float("-inf")
            for xb, yb in DataLoader(train, batch_size=512, shuffle=False):
                lo_x, hi_x = min(lo_x, float(xb.min())), max(hi_x, float(xb.max()))
                lo_y, hi_y = min(lo_y, float(yb.min())), max(hi_y, float(yb.max()))
            min_data, max_data, min_model, max_model = lo_x, hi_x, lo_y, hi_y
            train, val, test = (MinMaxNormalizedDataset(d, min_data, max_data, min_model, max_model) for d in (train, val, test))
        else:
            raise ValueError(f"Invalid normalization_type: {normalization_type}. Must be 'simple' or 'minmax'")
    if normalization_type == "simple":
        return train, val, test, rollout, x_normalizer, y_normalizer
    return train, val, test, rollout, min_data, max_data, min_model, max_model
