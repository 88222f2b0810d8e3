"""Navier-Stokes Markov pairs at several resolutions in ONE dataset: the dataset module of the north-star run
(reference main_2d.py:72 instantiates dataloaders.ns_naive_true_multires.ns_true_multires_markov_dataset, :396;
the per-split class is NSVTrueMultiResMarkovDataset, :10-394).  Same names, arguments, defaults, sampling seeds
and return values.

Two sources of samples per split:
  * files ``ns_{resolution}_{viscosity}{ext}`` named by ``data_mres_size`` ({resolution: target sample count};
    0 skips the resolution) -- the whole split when the target is 0 < . < split size is false, else a seeded
    draw without replacement of ``int(target * split_ratio[split])`` trajectories;
  * ``add_res``: resolutions derived from the ``downsample_from_res`` file, a seeded draw WITH replacement, then
    either a stride subsample (true lower-resolution grids) or, with ``use_low_pass_filter``, a spectral low-pass
    at ``target/original * lowpass_cutoff_ratio`` that KEEPS the original grid (such samples join the
    base resolution's group in ResolutionGroupedDataLoader).
Items are (x, y) = (u[t], u[t+1]) for t = 1 .. T-2, each [1, H, W].

Formats: ``.mat`` (key ``u`` [N,H,W,T]) and ``.h5`` (key ``u``, either axis order, the reference's heuristic) as
the reference; ``.npz`` / ``.npy`` with the ``.h5`` conventions in addition.  Pinned against the imported reference
through the ``.mat`` leg (tests/golden/data_layer_mres.npz); the ``.h5`` leg is parity-unpinned (no h5py in the
build image).

Deliberate difference: ``normalization_type="unit_gaussian"`` works here when every sample has one shape; the
reference's live code never imports UnitGaussianNormalizer in this module and raises NameError."""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional

import numpy as np
import torch
from torch.utils.data import Dataset

from dataloaders.ns_naive_markov import NormalizedDataset, SimpleNormalizer, _gather, _to_time_major
from models.custom_layer import UnitGaussianNormalizer
from utils.low_pass_filter import lowpass_filter_2d

_SPLIT_INDEX = {"train": 0, "valid": 1, "val": 1, "test": 2}
_EXTENSIONS = (".mat", ".h5", ".npz", ".npy")


def _load_u(path: str, ext: str, say) -> Optional[np.ndarray]:
    """[N,T,H,W] float32, or None (with a printed reason) when the file is unusable -- the reference skips such
    a resolution instead of failing (ns_naive_true_multires.py:214-250)"""
    try:
        if ext == ".mat":
            from scipy.io import loadmat
            blob = loadmat(path)
            raw = np.array(blob["u"], dtype=np.float32) if "u" in blob else None
        elif ext == ".h5":
            import h5py                      # ImportError is reported like any other unreadable file, below
            with h5py.File(path, "r") as f:
                raw = np.array(f["u"], dtype=np.float32) if "u" in f else None
        elif ext == ".npz":
            with np.load(path) as z:
                raw = np.asarray(z["u"], dtype=np.float32) if "u" in z.files else None
        else:
            raw = np.asarray(np.load(path), dtype=np.float32)
    except Exception as e:      # noqa: BLE001 -- same net as the reference's loaders
        print(f"  Error loading {path}: {e}")
        return None
    if raw is None:
        print(f"  Warning: 'u' key not found in {path}. Skipping.")
        return None
    if raw.ndim != 4:
        print(f"  Warning: Expected 4D array, got {raw.shape}. Skipping {path}")
        return None
    say(f"  raw data {raw.shape} from {path}")
    return np.transpose(raw, (0, 3, 1, 2)) if ext == ".mat" else _to_time_major(raw)


def _split(data: np.ndarray, split: str, ratio) -> np.ndarray:
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


class NSVTrueMultiResMarkovDataset(Dataset):
    def __init__(self, saved_folder, viscosity="1e-3", file_extension=".h5", reduced_batch=1, reduced_resolution=1,
                 reduced_resolution_t=1, data_mres_size=None, add_res=None, add_res_samples=None,
                 downsample_from_res=None, use_low_pass_filter=False, lowpass_cutoff_ratio=1.0, split_ratio=None,
                 random_seed=42, split="train", verbose=False, **kwargs):
        self.random_seed, self.split, self.viscosity = random_seed, split, viscosity
        self.file_extension = file_extension.lower()
        self.use_low_pass_filter, self.lowpass_cutoff_ratio = use_low_pass_filter, lowpass_cutoff_ratio
        self._say = print if verbose else (lambda *a, **k: None)
        if self.file_extension not in _EXTENSIONS:
            raise ValueError(f"Unsupported file extension: {self.file_extension}. Supported: {', '.join(_EXTENSIONS)}")
        split_ratio = [0.8, 0.1, 0.1] if split_ratio is None else split_ratio
        data_mres_size = {512: 0, 64: 0, 32: 0} if data_mres_size is None else {int(k): int(v) for k, v in dict(data_mres_size).items()}
        if downsample_from_res is None and data_mres_size:      # highest resolution that has samples, else highest named
            have = [r for r, n in data_mres_size.items() if n > 0]
            downsample_from_res = max(have) if have else max(data_mres_size)
        self.downsample_from_res = downsample_from_res
        add_res_samples = {64: 0, 32: 0} if add_res_samples is None else {int(k): int(v) for k, v in dict(add_res_samples).items()}

        self.x: List[torch.Tensor] = []
        self.y: List[torch.Tensor] = []
        self.resolution_info: List[str] = []
        split_idx = _SPLIT_INDEX.get(split, 0)

        for resolution, target in data_mres_size.items():
            if target == 0:
                continue
            path = os.path.join(saved_folder, f"ns_{resolution}_{viscosity}{self.file_extension}")
            if not os.path.exists(path):
                print(f"Warning: File {path} does not exist. Skipping resolution {resolution}")
                continue
            u = _load_u(path, self.file_extension, self._say)
            if u is None:
                continue
            u = u[::reduced_batch, ::reduced_resolution_t, ::reduced_resolution, ::reduced_resolution]
            part = _split(u, split, split_ratio)
            if 0 < target < part.shape[0]:
                take = int(target * split_ratio[split_idx])
                if take <= 0:
                    self._say(f"  no samples allocated for {split} at resolution {resolution}")
                    continue
                np.random.seed(random_seed + resolution + split_idx)          # the reference seeds the global stream
                part = part[np.random.choice(part.shape[0], take, replace=False)]
            self._append_pairs(part, f"{resolution}_file")

        if add_res is not None and add_res_samples is not None:
            if self.downsample_from_res is not None:
                self._add_downsampled_data(saved_folder, self.downsample_from_res, add_res, add_res_samples, split_ratio,
                                           reduced_batch, reduced_resolution, reduced_resolution_t)
            else:
                print("Warning: No resolution specified for downsampling and no available resolutions found.")
        assert len(self.x) == len(self.y), "Invalid input output pairs"
        self._say(f"split {split}: {len(self.x)} pairs, resolutions {sorted(set(self.resolution_info))}")

    def _append_pairs(self, u: np.ndarray, tag: str) -> None:
        """u [n,T,H,W] -> (u[:,1:-1], u[:,2:]) flattened over (n, t), one [1,H,W] tensor per item"""
        t = torch.tensor(u, dtype=torch.float)
        s = t.shape[-2:]
        xs = t[:, 1:-1].reshape(-1, 1, *s)
        ys = t[:, 2:].reshape(-1, 1, *s)
        self.x.extend(xs.unbind(0))
        self.y.extend(ys.unbind(0))
        self.resolution_info.extend([tag] * xs.shape[0])

    def _add_downsampled_data(self, saved_folder, base_resolution, add_res: Iterable[int], add_res_samples: Dict[int, int],
                              split_ratio, reduced_batch, reduced_resolution, reduced_resolution_t) -> None:
        """reference :263-383.  The base file is split first (no reductions), drawn WITH replacement under seed
        random_seed + target + split_idx + 10000, strided over (draw, time) only -- reduced_resolution is not
        applied on this leg -- and then brought to the target resolution"""
        path = os.path.join(saved_folder, f"ns_{base_resolution}_{self.viscosity}{self.file_extension}")
        if not os.path.exists(path):
            print(f"Warning: Base file {path} does not exist. Cannot create downsampled data.")
            return
        u = _load_u(path, self.file_extension, self._say)
        if u is None:
            print(f"Warning: Could not load base file {path}. Cannot create downsampled data.")
            return
        part = _split(u, self.split, split_ratio)
        full = part.shape[2]
        split_idx = _SPLIT_INDEX.get(self.split, 0)
        for target in add_res:
            target = int(target)
            if target >= full:
                print(f"  Warning: Target resolution {target} >= original {full}. Skipping.")
                continue
            take = int(add_res_samples.get(target, 100) * split_ratio[split_idx])
            if take == 0:
                self._say(f"  no downsampled samples allocated for {self.split} at resolution {target}")
                continue
            np.random.seed(self.random_seed + target + split_idx + 10000)
            drawn = part[np.random.choice(part.shape[0], take, replace=True)]
            drawn = drawn[::reduced_batch, ::reduced_resolution_t]
            if self.use_low_pass_filter:
                cutoff = (target / full) * self.lowpass_cutoff_ratio
                low = lowpass_filter_2d(torch.from_numpy(np.ascontiguousarray(drawn)).float()[:, :, None], cutoff_ratio=cutoff)
                low = low[:, :, 0].numpy()                                   # same grid as the base file
            else:
                f = full // target
                low = drawn[:, :, ::f, ::f]
            self._append_pairs(low, f"{target}_downsampled_{'lowpass' if self.use_low_pass_filter else 'naive'}")

    def __len__(self):
        return len(self.x)

    def __getitem__(self, idx):
        return self.x[idx], self.y[idx]

    def get_resolution_info(self):
        return self.resolution_info


def _flat_stats(dataset):
    """mean / std over every value of the split, as one float32 vector in item order (the reduction the reference
    runs on its python list of values, :513-526)"""
    xs = torch.cat([x.reshape(-1) for x, _ in dataset])
    ys = torch.cat([y.reshape(-1) for _, y in dataset])
    return xs.mean(), xs.std(), ys.mean(), ys.std()


def ns_true_multires_markov_dataset(saved_folder, viscosity="1e-3", file_extension=".mat", data_mres_size=None,
                                    add_res=None, add_res_samples=None, downsample_from_res=None,
                                    use_low_pass_filter=False, lowpass_cutoff_ratio=1.0, data_normalizer=True,
                                    normalization_type="simple", random_seed=42, **kwargs):
    """-> train, val, test, x_normalizer, y_normalizer.  Trajectories are split 0.8 / 0.1 / 0.1 in file order BEFORE
    pairing (no trajectory crosses splits); statistics come from the training split only"""
    data_mres_size = {512: 16, 64: 16, 32: 16} if data_mres_size is None else data_mres_size
    add_res_samples = {16: 12, 8: 8} if add_res_samples is None else add_res_samples
    common = dict(saved_folder=saved_folder, viscosity=viscosity, file_extension=file_extension,
                  data_mres_size=data_mres_size, add_res=add_res, add_res_samples=add_res_samples,
                  downsample_from_res=downsample_from_res, use_low_pass_filter=use_low_pass_filter,
                  lowpass_cutoff_ratio=lowpass_cutoff_ratio, split_ratio=[0.8, 0.1, 0.1], random_seed=random_seed)
    train, val, test = (NSVTrueMultiResMarkovDataset(split=s, **common, **kwargs) for s in ("train", "val", "test"))
    x_normalizer = y_normalizer = None
    if data_normalizer:
        if normalization_type == "simple":
            xm, xs, ym, ys = _flat_stats(train)
            x_normalizer, y_normalizer = SimpleNormalizer(xm, xs), SimpleNormalizer(ym, ys)
        elif normalization_type == "unit_gaussian":
            x_all, y_all = _gather(train)                # one shape for every sample, or torch.cat refuses
            x_normalizer, y_normalizer = UnitGaussianNormalizer(x_all), UnitGaussianNormalizer(y_all)
        else:
            raise ValueError(f"Invalid normalization_type: {normalization_type}. Must be 'simple' or 'unit_gaussian'")
        train, val, test = (NormalizedDataset(d, x_normalizer, y_normalizer) for d in (train, val, test))
    return train, val, test, x_normalizer, y_normalizer
