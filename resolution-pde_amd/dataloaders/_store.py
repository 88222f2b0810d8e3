"""One read-only view over the on-disk containers the loaders accept, shaped like the part of ``h5py.File`` the
reference's readers use (``key in f``, ``f[key]``, ``list(f.keys())``, nested groups).

  .h5 / .hdf5   h5py, when importable (the reference's only format for KS and Burgers/PDEBench); without it the
                open fails with an ImportError naming the alternative -- never a silent fallback
  .npz          numpy archive; a ``/`` in a key makes a group (``train/pde_140-256``), so one archive mirrors one
                HDF5 file
  .mat          scipy.io.loadmat (MATLAB v5/v7 files); top-level variables only

The .npz leg is what the build image can exercise (no h5py there); see DESIGN.md "data layer"."""
from __future__ import annotations

import os
from typing import Dict, Iterator, List

import numpy as np


class _Group:
    """a flat {"a/b/c": array} mapping seen as nested groups"""

    def __init__(self, flat: Dict[str, np.ndarray], prefix: str = ""):
        self._flat, self._prefix = flat, prefix

    def keys(self) -> List[str]:
        n = len(self._prefix)
        out: List[str] = []
        for k in self._flat:
            if k.startswith(self._prefix):
                head = k[n:].split("/", 1)[0]
                if head not in out:
                    out.append(head)
        return sorted(out)                      # h5py lists members in name order

    def __iter__(self) -> Iterator[str]:
        return iter(self.keys())

    def __contains__(self, key: str) -> bool:
        return key in self.keys()

    def __getitem__(self, key: str):
        full = self._prefix + key
        if full in self._flat:
            return self._flat[full]
        if any(k.startswith(full + "/") for k in self._flat):
            return _Group(self._flat, full + "/")
        raise KeyError(key)


class Store:
    """``with Store(path) as f:`` -- f behaves like the root group"""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise FileNotFoundError(f"File not found: {path}")
        self.path, self._h5, self._root = path, None, None
        ext = os.path.splitext(path)[1].lower()
        if ext in (".h5", ".hdf5"):
            try:
                import h5py
            except ImportError as e:
                raise ImportError(f"{path}: reading HDF5 needs h5py, which is not installed here; convert the file to "
                                  "an .npz archive with the same member names ('/' separates groups)") from e
            self._h5 = h5py.File(path, "r")
            self._root = self._h5
        elif ext == ".npz":
            with np.load(path) as z:                                  # allow_pickle stays False
                self._root = _Group({k: z[k] for k in z.files})
        elif ext == ".mat":
            from scipy.io import loadmat
            self._root = _Group({k: v for k, v in loadmat(path).items() if not k.startswith("__")})
        else:
            raise ValueError(f"Unsupported file extension: {ext}. Supported: .h5, .hdf5, .npz, .mat")

    def __enter__(self):
        return self._root

    def __exit__(self, *exc):
        if self._h5 is not None:
            self._h5.close()
        return False
