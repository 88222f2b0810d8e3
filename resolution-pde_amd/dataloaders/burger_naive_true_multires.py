"""Burgers / PDEBench Markov pairs at several resolutions in ONE dataset (reference
dataloaders/burger_naive_true_multires.py: BurgersTrajectoryDatasetFromExtracted :11-29,
extract_burgers_test_trajectories_for_rollout :32-136, H5pyTrueMultiResMarkovDataset :139-420,
burger_true_multires_markov_dataset :423-690; Hydra target of conf/dataset/burger/burger_naive_true_mres*.yaml).  Same
names, arguments, defaults, sampling seeds and return values.

A resolution's trajectories are the member ``tensor`` [N,T,X] of the first file matching ``filename_pattern`` in
    <saved_folder>/burgers_<resolution>_<viscosity>/
(the downsample leg always looks for ``1D_Burgers_Sols_Nu*.hdf5`` there, whatever ``filename_pattern`` says, as the
reference does).  Pairs are (u[t], u[t+1]) for t = 1 .. T-2: the first step is dropped, as in burger_naive_markov.py.
Sampling, the downsample leg, the rollout set and normalisation: dataloaders/_true_multires_1d.py.  Default
normalisation is "minmax" (8 return values).

"parity unpinned": the reference reads these files through h5py only (absent from the build image) and ships no
fixture of them; ``.npz`` archives with the member ``tensor`` are accepted where no ``.hdf5`` file matches (matches are
taken in name order; the reference takes ``glob``'s first, which is unordered)."""
from __future__ import annotations

import glob
import os

import numpy as np

from dataloaders._store import Store
from dataloaders._true_multires_1d import (TrajectoryList, TrueMultiRes1dMarkovDataset, extract_test_trajectories,
                                           normalise_and_pack)

_BASE_PATTERN = "1D_Burgers_Sols_Nu*.hdf5"


class BurgersTrajectoryDatasetFromExtracted(TrajectoryList):
    """whole test trajectories [T,X] for rollout evaluation"""


def _burgers_path(saved_folder, resolution, viscosity, pattern):
    folder = os.path.join(saved_folder, f"burgers_{resolution}_{viscosity}")
    hits = sorted(glob.glob(os.path.join(folder, pattern)))
    if not hits:
        hits = sorted(glob.glob(os.path.join(folder, os.path.splitext(pattern)[0] + ".npz")))
    return hits[0] if hits else None


def _burgers_read(path: str) -> np.ndarray:
    with Store(path) as f:
        return np.array(f["tensor"], dtype=np.float32)


def _burgers_pairs(u: np.ndarray):
    return u[:, 1:-1, :], u[:, 2:, :]


def extract_burgers_test_trajectories_for_rollout(saved_folder, viscosity=0.001, filename_pattern="1D_Burgers_Sols_Nu*.hdf5",
                                                  data_mres_size=None, split_ratio=None, reduced_batch=1,
                                                  reduced_resolution_t=1, random_seed=42):
    """-> (trajectories, trajectory_info): the test split's whole trajectories, before any pairing"""
    split_ratio = [0.8, 0.1, 0.1] if split_ratio is None else split_ratio
    data_mres_size = {1024: 0, 512: 0, 256: 0, 128: 0} if data_mres_size is None else data_mres_size
    locate = lambda r: _burgers_path(saved_folder, r, viscosity, filename_pattern)      # noqa: E731
    return extract_test_trajectories(locate, _burgers_read, data_mres_size, split_ratio, reduced_batch, reduced_resolution_t,
                                     random_seed)


class H5pyTrueMultiResMarkovDataset(TrueMultiRes1dMarkovDataset):
    def __init__(self, saved_folder, viscosity=0.001, filename_pattern="1D_Burgers_Sols_Nu*.hdf5", reduced_batch=1,
                 reduced_resolution_t=1, data_mres_size=None, add_res=None, add_res_samples=None, downsample_from_res=None,
                 use_low_pass_filter=False, lowpass_cutoff_ratio=1.0, split_ratio=None, random_seed=42, split="train",
                 **kwargs):
        self.viscosity = viscosity
        locate = lambda r: _burgers_path(saved_folder, r, viscosity, filename_pattern)   # noqa: E731
        locate_base = lambda r: _burgers_path(saved_folder, r, viscosity, _BASE_PATTERN)  # noqa: E731
        super().__init__(locate, locate_base, _burgers_read, _burgers_pairs, {1024: 0, 512: 0, 256: 0, 128: 0},
                         {64: 0, 32: 0}, reduced_batch=reduced_batch, reduced_resolution_t=reduced_resolution_t,
                         data_mres_size=data_mres_size, add_res=add_res, add_res_samples=add_res_samples,
                         downsample_from_res=downsample_from_res, use_low_pass_filter=use_low_pass_filter,
                         lowpass_cutoff_ratio=lowpass_cutoff_ratio, split_ratio=split_ratio, random_seed=random_seed,
                         split=split)


def burger_true_multires_markov_dataset(saved_folder, viscosity=0.001, filename_pattern="1D_Burgers_Sols_Nu*.hdf5",
                                        data_mres_size=None, add_res=None, add_res_samples=None, downsample_from_res=None,
                                        use_low_pass_filter=False, lowpass_cutoff_ratio=1.0, data_normalizer=True,
                                        normalization_type="minmax", random_seed=42, **kwargs):
    """"simple" -> train, val, test, rollout_test, x_normalizer, y_normalizer;
    "minmax" (default) -> train, val, test, rollout_test, min_data, max_data, min_model, max_model"""
    data_mres_size = {1024: 200, 512: 100, 256: 1000, 128: 100} if data_mres_size is None else data_mres_size
    add_res_samples = {64: 150, 32: 100} if add_res_samples is None else add_res_samples
    split_ratio = [0.8, 0.1, 0.1]
    common = dict(saved_folder=saved_folder, viscosity=viscosity, filename_pattern=filename_pattern,
                  data_mres_size=data_mres_size, add_res=add_res, add_res_samples=add_res_samples,
                  downsample_from_res=downsample_from_res, use_low_pass_filter=use_low_pass_filter,
                  lowpass_cutoff_ratio=lowpass_cutoff_ratio, split_ratio=split_ratio, random_seed=random_seed)
    train, val, test = (H5pyTrueMultiResMarkovDataset(split=s, **common, **kwargs) for s in ("train", "val", "test"))
    trajectories, info = extract_burgers_test_trajectories_for_rollout(
        saved_folder=saved_folder, viscosity=viscosity, filename_pattern=filename_pattern, data_mres_size=data_mres_size,
        split_ratio=split_ratio, reduced_batch=1, reduced_resolution_t=1, random_seed=random_seed)
    rollout = BurgersTrajectoryDatasetFromExtracted(trajectories, info)
    return normalise_and_pack(train, val, test, rollout, data_normalizer, normalization_type)
