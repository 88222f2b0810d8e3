"""Navier-Stokes vorticity trajectories -> single-step (Markov) training pairs, normalisers, rollout
trajectories.  Same public names, arguments and return values as the reference's
dataloaders/ns_naive_markov.py (NSMarkovDataset :166-323, ns_markov_dataset :325-511,
extract_ns_test_trajectories_for_rollout_single :33-146, NSTrajectoryDatasetFromExtracted :12-31).

File formats: ``.mat`` (scipy, key ``u`` [N,H,W,T]) as the reference; ``.h5`` (key ``u``, [N,T,H,W] or
[N,H,W,T] told apart by the reference's heuristic) when h5py is importable; additionally ``.npz`` / ``.npy``
with the ``.h5`` conventions, so the pipeline can be exercised on hosts without h5py.  Pinned against the
reference through its ``.mat`` branch (tests/golden/data_layer.npz); the ``.h5`` reader itself is unpinned
here (no h5py in the build image)."""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset, random_split

from models.custom_layer import UnitGaussianNormalizer
from utils.low_pass_filter import lowpass_filter_2d


def _to_time_major(raw: np.ndarray) -> np.ndarray:
    """[N,T,H,W] <- either layout; the reference's test (ns_naive_markov.py:308-314): the last axis is time when
    it is short (< 100) and shorter than both spatial axes"""
    if raw.ndim == 4 and raw.shape[-1] < 100 and raw.shape[-1] < min(raw.shape[1], raw.shape[2]):
        return np.transpose(raw, (0, 3, 1, 2))
    return raw


def _read_u(file_path: str) -> np.ndarray:
    """vorticity array as [N,T,H,W] float32"""
    if not os.path.exists(file_path):
        raise FileNotFoundError(f"File not found: {file_path}")
    ext = os.path.splitext(file_path)[1].lower()
    if ext == ".mat":
        from scipy.io import loadmat
        blob = loadmat(file_path)
        if "u" not in blob:
            raise KeyError(f"'u' key not found in {file_path}. Available keys: {[k for k in blob if not k.startswith('__')]}")
        raw = np.array(blob["u"], dtype=np.float32)
        if raw.ndim != 4:
            raise ValueError(f"Expected 4D array, got {raw.shape}")
        return np.transpose(raw, (0, 3, 1, 2))          # .mat files are always [N,H,W,T]
    if ext == ".h5":
        try:
            import h5py
        except ImportError as e:                        # no silent fallback: say what is missing
            raise ImportError(f"{file_path}: reading .h5 needs h5py, which is not installed here; "
                              "convert the 'u' dataset to .npz/.npy or install h5py") from e
        with h5py.File(file_path, "r") as f:
            if "u" not in f:
                raise KeyError(f"'u' key not found in {file_path}. Available keys: {list(f.keys())}")
            raw = np.array(f["u"], dtype=np.float32)
    elif ext == ".npz":
        with np.load(file_path) as blob:
            if "u" not in blob:
                raise KeyError(f"'u' key not found in {file_path}. Available keys: {list(blob.keys())}")
            raw = np.asarray(blob["u"], dtype=np.float32)
    elif ext == ".npy":
        raw = np.asarray(np.load(file_path), dtype=np.float32)
    else:
        raise ValueError(f"Unsupported file extension: {ext}. Supported extensions: .mat, .h5, .npz, .npy")
    if raw.ndim != 4:
        raise ValueError(f"Expected 4D array, got {raw.shape}")
    return _to_time_major(raw)


def _reduce(data: np.ndarray, reduced_batch: int, reduced_resolution: int, reduced_resolution_t: int,
            use_low_pass_filter: bool, lowpass_cutoff_ratio: float, num_samples_max: int) -> np.ndarray:
    """[N,T,H,W]: stride over samples and time, then space (naive stride, or the reference's low-pass variant
    which filters at 1/reduced_resolution of the band and KEEPS the grid), then cap the sample count"""
    data = data[::reduced_batch, ::reduced_resolution_t]
    if reduced_resolution > 1:
        if use_low_pass_filter:
            cutoff = (1.0 / reduced_resolution) * lowpass_cutoff_ratio
            data = lowpass_filter_2d(torch.from_numpy(np.ascontiguousarray(data)).float(), cutoff_ratio=cutoff).numpy()
        else:
            data = data[:, :, ::reduced_resolution, ::reduced_resolution]
    if num_samples_max > 0:
        data = data[:min(num_samples_max, data.shape[0])]
    return data


class NSTrajectoryDatasetFromExtracted(Dataset):
    """full test trajectories [T,H,W] for rollout evaluation"""

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


def extract_ns_test_trajectories_for_rollout_single(filename, saved_folder, reduced_batch=1, reduced_resolution=1,
                                                     reduced_resolution_t=1, use_low_pass_filter=False,
                                                     lowpass_cutoff_ratio=1.0, num_samples_max=-1, split_ratio=None,
                                                     random_seed=42):
    """the LAST (1 - split_ratio[0] - split_ratio[1]) share of the samples, in file order (the reference takes a
    contiguous tail here, not the random split of the Markov pairs), as whole trajectories"""
    split_ratio = [0.8, 0.1, 0.1] if split_ratio is None else split_ratio
    data = _reduce(_read_u(os.path.join(saved_folder, filename)), reduced_batch, reduced_resolution,
                   reduced_resolution_t, use_low_pass_filter, lowpass_cutoff_ratio, num_samples_max)
    total = data.shape[0]
    val_end = int(total * split_ratio[0]) + int(total * split_ratio[1])
    test = data[val_end:]
    trajectories = [torch.tensor(test[i], dtype=torch.float) for i in range(test.shape[0])]
    info = [{"original_index": i, "source": "single_resolution_file"} for i in range(test.shape[0])]
    return trajectories, info


class NSMarkovDataset(Dataset):
    """x = u[:, 1:-1], y = u[:, 2:] flattened over (sample, time) to [(N*(T-2)), 1, H, W]"""

    def __init__(self, filename, saved_folder, reduced_batch=1, reduced_resolution=1, reduced_resolution_t=1,
                 use_low_pass_filter=False, lowpass_cutoff_ratio=1.0, num_samples_max=-1, **kwargs):
        self.use_low_pass_filter, self.lowpass_cutoff_ratio = use_low_pass_filter, lowpass_cutoff_ratio
        data = _reduce(_read_u(os.path.join(saved_folder, filename)), reduced_batch, reduced_resolution,
                       reduced_resolution_t, use_low_pass_filter, lowpass_cutoff_ratio, num_samples_max)
        self.data = data[..., None]                                            # [N,T,H,W,1] as the reference keeps it
        u = torch.from_numpy(np.ascontiguousarray(data)).float()
        s = u.shape[-2:]
        self.x = u[:, 1:-1].reshape(-1, 1, *s).contiguous()
        self.y = u[:, 2:].reshape(-1, 1, *s).contiguous()
        assert len(self.x) == len(self.y), "Invalid input output pairs"

    def __len__(self):
        return len(self.x)

    def __getitem__(self, idx):
        return self.x[idx], self.y[idx]


class SimpleNormalizer:
    """one global mean / std (reference: the closure class at ns_naive_markov.py:423-440)"""

    def __init__(self, mean, std, eps=1e-8):
        self.mean, self.std, self.eps = float(mean), float(std), eps

    def encode(self, x):
        return (x - self.mean) / (self.std + self.eps)

    def decode(self, x, device="cuda"):
        return x * (self.std + self.eps) + self.mean

    def cuda(self):
        return self

    def cpu(self):
        return self


class NormalizedDataset(Dataset):
    def __init__(self, dataset, x_normalizer, y_normalizer):
        self.dataset, self.x_normalizer, self.y_normalizer = dataset, x_normalizer, y_normalizer

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, idx):
        x, y = self.dataset[idx]
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x).float()
        if isinstance(y, np.ndarray):
            y = torch.from_numpy(y).float()
        return self.x_normalizer.encode(x), self.y_normalizer.encode(y)


def _gather(dataset) -> Tuple[torch.Tensor, torch.Tensor]:
    xs, ys = zip(*(b for b in DataLoader(dataset, batch_size=512, shuffle=False)))
    return torch.cat(xs, dim=0), torch.cat(ys, dim=0)


def ns_markov_dataset(filename, saved_folder, use_low_pass_filter=False, lowpass_cutoff_ratio=1.0, data_normalizer=True,
                      normalization_type="unit_gaussian", **kwargs):
    """-> train, val, test, x_normalizer, y_normalizer: 0.8 / 0.1 / 0.1 random split of the Markov pairs
    (torch.Generator seed 42), statistics from the training split only ("simple": one scalar mean / std;
    "unit_gaussian": per grid point)"""
    full = NSMarkovDataset(filename, saved_folder, use_low_pass_filter=use_low_pass_filter,
                           lowpass_cutoff_ratio=lowpass_cutoff_ratio, **kwargs)
    n = len(full)
    n_train, n_val = int(0.8 * n), int(0.1 * n)
    train, val, test = random_split(full, [n_train, n_val, n - n_train - n_val], generator=torch.Generator().manual_seed(42))
    x_normalizer = y_normalizer = None
    if data_normalizer:
        x_all, y_all = _gather(train)
        if normalization_type == "simple":
            x_normalizer = SimpleNormalizer(x_all.mean(), x_all.std())
            y_normalizer = SimpleNormalizer(y_all.mean(), y_all.std())
        elif normalization_type == "unit_gaussian":
            x_normalizer, y_normalizer = UnitGaussianNormalizer(x_all), UnitGaussianNormalizer(y_all)
        else:
            raise ValueError(f"Invalid normalization_type: {normalization_type}. Must be 'simple' or 'unit_gaussian'")
        train, val, test = (NormalizedDataset(d, x_normalizer, y_normalizer) for d in (train, val, test))
    return train, val, test, x_normalizer, y_normalizer


def ns_rollout_test_dataset(filename, saved_folder, **kwargs) -> NSTrajectoryDatasetFromExtracted:
    """the rollout set the reference builds inside ns_markov_dataset (:377-395) but does not return"""
    keep = ("reduced_batch", "reduced_resolution", "reduced_resolution_t", "num_samples_max", "use_low_pass_filter",
            "lowpass_cutoff_ratio")
    trajs, info = extract_ns_test_trajectories_for_rollout_single(filename, saved_folder,
                                                                  **{k: v for k, v in kwargs.items() if k in keep})
    return NSTrajectoryDatasetFromExtracted(trajs, info)
