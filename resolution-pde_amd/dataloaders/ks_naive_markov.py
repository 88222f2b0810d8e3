"""Kuramoto-Sivashinsky trajectories -> Markov pairs, rollout trajectories, normalisers.  Public names, arguments,
defaults and return values of the reference's dataloaders/ks_naive_markov.py (KSTrajectoryDatasetFromFile :11-131,
KSMarkovDataset :134-306, ks_markov_dataset :309-444).

One file per split (train / valid / test, told from the FILE NAME); inside, a group named after the split (or the
only group) holding ``pde_<nt>-<nx>`` [N,T,X] and optionally ``t``, ``x``, ``dx``, ``dt`` -- the layout of the
LPSDA generator the reference trains on, which ks_naive_true_multires.py:297-307 reads the same way.
Pairs here are (u[t], u[t+1]) for EVERY t = 0 .. T-2 (the NS and Burgers loaders drop the first step; KS does not).

Formats: ``.h5`` through h5py as the reference, ``.npz`` with ``/``-separated member names in addition
(dataloaders/_store.py).  The reference has no HDF5-free leg for KS, so nothing here is pinned against it:
"parity unpinned" (tests/test_data_layer_cpu.py checks the semantics above on synthetic archives)."""
from __future__ import annotations

import os

import numpy as np
import torch
from torch.utils.data import Dataset

from dataloaders._store import Store
from dataloaders.ns_naive_markov import NormalizedDataset, SimpleNormalizer
from utils.low_pass_filter import lowpass_filter_1d


def _resolve(saved_folder: str, filename: str) -> str:
    """<saved_folder>/<filename>, or the .npz archive of the same stem when that file is absent (the default validation /
    test names end in .h5; a tree of .npz archives -- the build image has no h5py -- then works with the defaults)"""
    path = os.path.join(os.path.abspath(saved_folder), filename)
    if not os.path.exists(path):
        alt = os.path.splitext(path)[0] + ".npz"
        if os.path.exists(alt):
            return alt
    return path


def _split_of(filename: str, say) -> str:
    low = filename.lower()
    for s in ("train", "valid", "test"):
        if s in low:
            return s
    say(f"Warning: Could not determine split from filename {filename}, assuming 'train'")
    return "train"


def _open_group(f, split: str):
    if split in f:
        return f[split]
    keys = list(f.keys())
    if len(keys) == 1:
        return f[keys[0]]
    raise ValueError(f"Could not find split '{split}' in file. Available keys: {keys}")


def _pde_key(group) -> str:
    keys = list(group.keys())
    for k in keys:
        if "pde" in k.lower() and "-" in k:
            return k
    raise ValueError(f"Could not find PDE data key in {keys}")


def _reduce_1d(u: np.ndarray, reduced_batch, reduced_resolution, reduced_resolution_t, use_low_pass_filter,
               lowpass_cutoff_ratio, num_samples_max) -> np.ndarray:
    """[N,T,X]: stride samples and time; space by stride, or low-pass at 1/reduced_resolution of the band on the
    SAME grid; then cap the sample count (reference :253-283)"""
    u = u[::reduced_batch, ::reduced_resolution_t, :]
    if reduced_resolution > 1:
        if use_low_pass_filter:
            cutoff = (1.0 / reduced_resolution) * lowpass_cutoff_ratio
            u = lowpass_filter_1d(torch.from_numpy(np.ascontiguousarray(u)).float(), cutoff_ratio=cutoff).numpy()
        else:
            u = u[:, :, ::reduced_resolution]
    n = min(num_samples_max, u.shape[0]) if num_samples_max > 0 else u.shape[0]
    return u[:n]


class KSTrajectoryDatasetFromFile(Dataset):
    """whole trajectories [T,X] of one file, for rollout evaluation (not normalised)"""

    def __init__(self, filename, saved_folder, reduced_batch=1, reduced_resolution=1, reduced_resolution_t=1,
                 use_low_pass_filter=False, lowpass_cutoff_ratio=1.0, num_samples_max=-1, **kwargs):
        path = _resolve(saved_folder, filename)
        self.split = _split_of(filename, print)
        with Store(path) as f:
            group = _open_group(f, self.split)
            u = np.array(group[_pde_key(group)], dtype=np.float32)
        u = _reduce_1d(u, reduced_batch, reduced_resolution, reduced_resolution_t, use_low_pass_filter,
                       lowpass_cutoff_ratio, num_samples_max)
        self.trajectories = [torch.tensor(u[i], dtype=torch.float) for i in range(u.shape[0])]
        self.trajectory_info = [{"original_index": i, "source": f"{self.split}_file", "filename": os.path.basename(path)}
                                for i in range(u.shape[0])]

    def __len__(self):
        return len(self.trajectories)

    def __getitem__(self, idx):
        return self.trajectories[idx]

    def get_trajectory_info(self, idx):
        return self.trajectory_info[idx]

    def get_all_info(self):
        return self.trajectory_info


class KSMarkovDataset(Dataset):
    """x = u[:, :-1], y = u[:, 1:] flattened over (sample, time) to [(N*(T-1)), 1, X]; keeps ``time``, ``x_coords``,
    ``grid`` [X,1], ``dx``, ``dt`` when the file has them"""

    def __init__(self, filename, saved_folder, reduced_batch=1, reduced_resolution=1, reduced_resolution_t=1,
                 use_low_pass_filter=False, lowpass_cutoff_ratio=1.0, num_samples_max=-1, **kwargs):
        self.use_low_pass_filter, self.lowpass_cutoff_ratio = use_low_pass_filter, lowpass_cutoff_ratio
        path = _resolve(saved_folder, filename)
        self.split = _split_of(filename, print)
        with Store(path) as f:
            group = _open_group(f, self.split)
            keys = list(group.keys())
            u = np.array(group[_pde_key(group)], dtype=np.float32)
            opt = lambda k: np.array(group[k], dtype=np.float32) if k in keys else None      # noqa: E731
            self.time, xc, self.dx, self.dt = opt("t"), opt("x"), opt("dx"), opt("dt")
        self.x_coords = xc[0] if xc is not None and xc.ndim == 2 else xc       # one grid for the whole file
        self.data = _reduce_1d(u, reduced_batch, reduced_resolution, reduced_resolution_t, use_low_pass_filter,
                               lowpass_cutoff_ratio, num_samples_max)
        if self.time is not None:
            self.time = (self.time[:self.data.shape[0], ::reduced_resolution_t] if self.time.ndim == 2
                         else self.time[::reduced_resolution_t])
        if self.x_coords is not None:
            if not use_low_pass_filter and reduced_resolution > 1:                 # the filtered leg keeps every point
                self.x_coords = self.x_coords[::reduced_resolution]
            self.grid = torch.tensor(self.x_coords, dtype=torch.float).unsqueeze(-1)
        t = torch.tensor(self.data, dtype=torch.float)
        self.x = t[:, :-1].reshape(-1, 1, t.shape[-1])
        self.y = t[:, 1:].reshape(-1, 1, t.shape[-1])
        assert len(self.x) == len(self.y), "Invalid input output pairs"

    def __len__(self):
        return len(self.x)

    def __getitem__(self, idx):
        return self.x[idx], self.y[idx]


def ks_markov_dataset(filename, saved_folder, data_normalizer=True, use_low_pass_filter=False, lowpass_cutoff_ratio=1.0,
                      val_filename="KS_valid.h5", test_filename="KS_test.h5", **kwargs):
    """-> train, val, test, rollout_test, x_normalizer, y_normalizer.  The three Markov sets come from three files;
    the rollout set is the test file's whole trajectories, left un-normalised; statistics are one scalar mean /
    std over the training pairs ("simple" is the only normalisation this loader has)"""
    mk = lambda name: KSMarkovDataset(name, saved_folder, use_low_pass_filter=use_low_pass_filter,     # noqa: E731
                                      lowpass_cutoff_ratio=lowpass_cutoff_ratio, **kwargs)
    train, val, test = mk(filename), mk(val_filename), mk(test_filename)
    rollout = KSTrajectoryDatasetFromFile(filename=test_filename, saved_folder=saved_folder,
                                          use_low_pass_filter=use_low_pass_filter,
                                          lowpass_cutoff_ratio=lowpass_cutoff_ratio, **kwargs)
    x_normalizer = y_normalizer = None
    if data_normalizer:
        x_normalizer = SimpleNormalizer(train.x.reshape(-1).mean(), train.x.reshape(-1).std())
        y_normalizer = SimpleNormalizer(train.y.reshape(-1).mean(), train.y.reshape(-1).std())
        train, val, test = (NormalizedDataset(d, x_normalizer, y_normalizer) for d in (train, val, test))
    return train, val, test, rollout, x_normalizer, y_normalizer
