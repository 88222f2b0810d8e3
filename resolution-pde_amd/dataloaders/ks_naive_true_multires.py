"""Kuramoto-Sivashinsky Markov pairs at several resolutions in ONE dataset (reference
dataloaders/ks_naive_true_multires.py: KSTrajectoryDatasetFromExtracted :11-29,
extract_ks_test_trajectories_for_rollout :32-170, KSTrueMultiResMarkovDataset :173-533,
ks_true_multires_markov_dataset :535-826; Hydra target of conf/dataset/ks/ks_naive_true_mres*.yaml).  Same names,
arguments, defaults, sampling seeds and return values.

A resolution's trajectories live in
    <saved_folder>/res_<resolution>/visc_<viscosity>_L<L>_lmax<lmax>_et<et>_nte<nte>_nt<nt>/KS_train_<train_s>.h5
inside the group ``train`` (or the only group, or the first whose name is data / pde / train or contains "pde") as
``pde_<nt>-<nx>`` [N,T,X] -- the LPSDA generator's layout, the one ks_naive_markov.py reads.  Pairs are (u[t], u[t+1])
for EVERY t = 0 .. T-2.  Sampling, the downsample leg, the rollout set and normalisation: dataloaders/_true_multires_1d.py.

"parity unpinned": the reference reads these files through h5py only (absent from the build image) and ships no
fixture of them; an ``.npz`` archive with the same member names is accepted in place of the ``.h5`` file."""
from __future__ import annotations

import os

import numpy as np

from dataloaders._store import Store
from dataloaders._true_multires_1d import (TrajectoryList, TrueMultiRes1dMarkovDataset, extract_test_trajectories,
                                           normalise_and_pack, with_npz_fallback)


class KSTrajectoryDatasetFromExtracted(TrajectoryList):
    """whole test trajectories [T,X] for rollout evaluation"""


def _ks_path(saved_folder, resolution, viscosity, L, lmax, et, nte, nt, train_s):
    sub = f"visc_{viscosity}_L{L}_lmax{lmax}_et{et}_nte{nte}_nt{nt}"
    return with_npz_fallback(os.path.join(saved_folder, f"res_{resolution}", sub, f"KS_train_{train_s}.h5"))


def _ks_read(path: str) -> np.ndarray:
    with Store(path) as f:
        if "train" in f:
            group = f["train"]
        else:
            keys = list(f.keys())
            group = f[keys[0]] if len(keys) == 1 else None
            if group is None:
                for k in keys:
                    if k.lower() in ("data", "pde", "train") or "pde" in k.lower():
                        group = f[k]
                        break
            if group is None:
                raise ValueError(f"Could not find data group in file. Available keys: {keys}")
        names = list(group.keys())
        for k in names:
            if "pde" in k.lower() and "-" in k:
                return np.array(group[k], dtype=np.float32)
        raise ValueError(f"Could not find PDE data key in {names}")


def _ks_pairs(u: np.ndarray):
    return u[:, :-1, :], u[:, 1:, :]


def extract_ks_test_trajectories_for_rollout(saved_folder, viscosity=0.05, L=64.0, lmax=8, et=5.0, nte=51, nt=51,
                                             train_s=2048, data_mres_size=None, split_ratio=None, reduced_batch=1,
                                             reduced_resolution_t=1, random_seed=42):
    """-> (trajectories, trajectory_info): the test split's whole trajectories, before any pairing"""
    split_ratio = [0.8, 0.1, 0.1] if split_ratio is None else split_ratio
    data_mres_size = {800: 0, 512: 0, 400: 0} if data_mres_size is None else data_mres_size
    locate = lambda r: _ks_path(saved_folder, r, viscosity, L, lmax, et, nte, nt, train_s)      # noqa: E731
    return extract_test_trajectories(locate, _ks_read, data_mres_size, split_ratio, reduced_batch, reduced_resolution_t,
                                     random_seed)


class KSTrueMultiResMarkovDataset(TrueMultiRes1dMarkovDataset):
    def __init__(self, saved_folder, viscosity=0.05, L=64.0, lmax=8, et=5.0, nte=51, nt=51, train_s=2048, reduced_batch=1,
                 reduced_resolution_t=1, data_mres_size=None, add_res=None, add_res_samples=None, downsample_from_res=None,
                 use_low_pass_filter=False, lowpass_cutoff_ratio=1.0, split_ratio=None, random_seed=42, split="train",
                 **kwargs):
        self.viscosity, self.L, self.lmax, self.et, self.nte, self.nt, self.train_s = viscosity, L, lmax, et, nte, nt, train_s
        locate = lambda r: _ks_path(saved_folder, r, viscosity, L, lmax, et, nte, nt, train_s)  # noqa: E731
        super().__init__(locate, locate, _ks_read, _ks_pairs, {800: 0, 512: 0, 400: 0}, {200: 0, 128: 0},
                         reduced_batch=reduced_batch, reduced_resolution_t=reduced_resolution_t,
                         data_mres_size=data_mres_size, add_res=add_res, add_res_samples=add_res_samples,
                         downsample_from_res=downsample_from_res, use_low_pass_filter=use_low_pass_filter,
                         lowpass_cutoff_ratio=lowpass_cutoff_ratio, split_ratio=split_ratio, random_seed=random_seed,
                         split=split)


def ks_true_multires_markov_dataset(saved_folder, viscosity=0.05, L=64.0, lmax=8, et=5.0, nte=51, nt=51, train_s=2048,
                                    data_mres_size=None, add_res=None, add_res_samples=None, downsample_from_res=None,
                                    use_low_pass_filter=False, lowpass_cutoff_ratio=1.0, data_normalizer=True,
                                    normalization_type="simple", random_seed=42, **kwargs):
    """"simple" -> train, val, test, rollout_test, x_normalizer, y_normalizer;
    "minmax" -> train, val, test, rollout_test, min_data, max_data, min_model, max_model"""
    data_mres_size = {800: 200, 512: 100, 400: 1000} if data_mres_size is None else data_mres_size
    add_res_samples = {200: 150, 128: 100} if add_res_samples is None else add_res_samples
    split_ratio = [0.8, 0.1, 0.1]
    common = dict(saved_folder=saved_folder, viscosity=viscosity, L=L, lmax=lmax, et=et, nte=nte, nt=nt, train_s=train_s,
                  data_mres_size=data_mres_size, add_res=add_res, add_res_samples=add_res_samples,
                  downsample_from_res=downsample_from_res, use_low_pass_filter=use_low_pass_filter,
                  lowpass_cutoff_ratio=lowpass_cutoff_ratio, split_ratio=split_ratio, random_seed=random_seed)
    train, val, test = (KSTrueMultiResMarkovDataset(split=s, **common, **kwargs) for s in ("train", "val", "test"))
    trajectories, info = extract_ks_test_trajectories_for_rollout(
        saved_folder=saved_folder, viscosity=viscosity, L=L, lmax=lmax, et=et, nte=nte, nt=nt, train_s=train_s,
        data_mres_size=data_mres_size, split_ratio=split_ratio, reduced_batch=1, reduced_resolution_t=1,
        random_seed=random_seed)
    rollout = KSTrajectoryDatasetFromExtracted(trajectories, info)
    return normalise_and_pack(train, val, test, rollout, data_normalizer, normalization_type)
