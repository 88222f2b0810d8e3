#!/usr/bin/env python3
"""Golden vectors for the data layer, from the IMPORTED reference (build container only).

    python tests/golden/make_golden_data.py          # data_layer.npz   (ns_naive_markov + low-pass filters)
    python tests/golden/make_golden_data.py mres     # data_layer_mres.npz (ns_naive_true_multires)

The reference's NS loader is run on a small synthetic ``.mat`` file written here with scipy (``u`` [N,H,W,T],
numpy-seeded) -- its ``.mat`` branch needs only scipy, so nothing is stubbed beyond the unused ``import h5py``
(SURVEY 8c); the ``.h5`` branch cannot run here (no h5py) and stays unpinned.  Stored: split sizes, first / last
items of every split, normaliser statistics, rollout trajectories, and the low-pass filters' outputs on seeded
inputs -- data only, nothing of the reference's text."""
from __future__ import annotations

import contextlib
import io
import os
import sys
import tempfile
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

NS_CASES = {
    "ns_default": dict(),
    "ns_reduced_simple": dict(reduced_batch=2, reduced_resolution=2, reduced_resolution_t=2, normalization_type="simple"),
    "ns_lowpass_capped": dict(reduced_resolution=2, use_low_pass_filter=True, lowpass_cutoff_ratio=1.0, num_samples_max=8),
    "ns_raw": dict(data_normalizer=False, reduced_resolution_t=3),
}


# ns_true_multires_markov_dataset on two files ns_32_1e-3.mat / ns_16_1e-3.mat of 20 trajectories each
MRES_FILES = {32: dict(seed=21, n=20, h=32, w=32, t=7), 16: dict(seed=22, n=20, h=16, w=16, t=7)}
MRES_CASES = {
    "mres_all": dict(data_mres_size={32: 20, 16: 20}),
    "mres_sampled": dict(data_mres_size={32: 10, 16: 5}, random_seed=7),
    "mres_add_naive": dict(data_mres_size={32: 20, 16: 0}, add_res=[16, 8], add_res_samples={16: 10, 8: 5}),
    "mres_add_lowpass_q14": dict(data_mres_size={32: 12}, add_res=[16, 8, 64], add_res_samples={16: 10, 8: 5},
                                 use_low_pass_filter=True, lowpass_cutoff_ratio=1.0, reduced_resolution_t=2),
    "mres_from16_raw": dict(data_mres_size={32: 0, 16: 20}, add_res=[8], add_res_samples={8: 20}, downsample_from_res=16,
                            data_normalizer=False, reduced_batch=2),
}


def probe_indices(n):
    return sorted({0, n // 3, n // 2, n - 1}) if n else []


def synthetic_u(seed=11, n=12, h=16, w=16, t=9):
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((n, h, w, 1)).astype(np.float32)
    drift = rng.standard_normal((n, h, w, t)).astype(np.float32) * 0.3
    return (base + np.cumsum(drift, axis=-1)).astype(np.float32)          # [N,H,W,T] as the .mat files store it


def main():
    sys.dont_write_bytecode = True
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))
    sys.path.insert(0, REF)
    warnings.filterwarnings("ignore")
    import torch
    from scipy.io import savemat
    from dataloaders.ns_naive_markov import extract_ns_test_trajectories_for_rollout_single, ns_markov_dataset
    from utils.low_pass_filter import lowpass_filter_1d, lowpass_filter_2d
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        savemat(os.path.join(tmp, "ns_16_synth.mat"), {"u": synthetic_u()})
        for name, kw in NS_CASES.items():
            with contextlib.redirect_stdout(io.StringIO()):
                train, val, test, xn, yn = ns_markov_dataset("ns_16_synth.mat", tmp, **kw)
                rk = {k: v for k, v in kw.items() if k in ("reduced_batch", "reduced_resolution", "reduced_resolution_t",
                                                           "use_low_pass_filter", "lowpass_cutoff_ratio", "num_samples_max")}
                trajs, info = extract_ns_test_trajectories_for_rollout_single("ns_16_synth.mat", tmp, **rk)
            out[f"{name}/sizes"] = np.array([len(train), len(val), len(test), len(trajs)])
            for split, ds in (("train", train), ("val", val), ("test", test)):
                for tag, idx in (("first", 0), ("last", len(ds) - 1)):
                    x, y = ds[idx]
                    out[f"{name}/{split}_{tag}_x"] = np.asarray(x, dtype=np.float32)
                    out[f"{name}/{split}_{tag}_y"] = np.asarray(y, dtype=np.float32)
            if xn is not None:
                for tag, nrm in (("x", xn), ("y", yn)):
                    out[f"{name}/{tag}_mean"] = np.asarray(nrm.mean, dtype=np.float32)
                    out[f"{name}/{tag}_std"] = np.asarray(nrm.std, dtype=np.float32)
                probe = torch.from_numpy(synthetic_u(seed=5, n=1)[0, :, :, :1].transpose(2, 0, 1).copy())
                if probe.shape[-1] == np.asarray(train[0][0]).shape[-1]:
                    out[f"{name}/decode_of_encode"] = np.asarray(yn.decode(xn.encode(probe), device="cpu"), dtype=np.float32)
            out[f"{name}/traj0"] = np.asarray(trajs[0], dtype=np.float32)
            out[f"{name}/traj_last"] = np.asarray(trajs[-1], dtype=np.float32)
    rng = np.random.default_rng(3)
    f1 = torch.from_numpy(rng.standard_normal((2, 3, 2, 32)).astype(np.float32))
    f1b = torch.from_numpy(rng.standard_normal((2, 3, 20)).astype(np.float32))
    f2 = torch.from_numpy(rng.standard_normal((2, 2, 1, 16, 16)).astype(np.float32))
    f2b = torch.from_numpy(rng.standard_normal((1, 3, 12, 12)).astype(np.float32))
    out["lp/in1"], out["lp/in1b"], out["lp/in2"], out["lp/in2b"] = f1.numpy(), f1b.numpy(), f2.numpy(), f2b.numpy()
    for c in (0.25, 0.5, 1.0):
        out[f"lp/1d_{c}"] = lowpass_filter_1d(f1.clone(), cutoff_ratio=c).numpy()
        out[f"lp/1db_{c}"] = lowpass_filter_1d(f1b.clone(), cutoff_ratio=c).numpy()
        out[f"lp/2d_{c}"] = lowpass_filter_2d(f2.clone(), cutoff_ratio=c).numpy()
        out[f"lp/2db_{c}"] = lowpass_filter_2d(f2b.clone(), cutoff_ratio=c).numpy()
    path = os.path.join(HERE, "data_layer.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


def main_mres():
    sys.dont_write_bytecode = True
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))
    sys.path.insert(0, REF)
    warnings.filterwarnings("ignore")
    from scipy.io import savemat
    from dataloaders.ns_naive_true_multires import ns_true_multires_markov_dataset
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for res, kw in MRES_FILES.items():
            savemat(os.path.join(tmp, f"ns_{res}_1e-3.mat"), {"u": synthetic_u(**kw)})
        for name, kw in MRES_CASES.items():
            with contextlib.redirect_stdout(io.StringIO()):
                train, val, test, xn, yn = ns_true_multires_markov_dataset(tmp, viscosity="1e-3", file_extension=".mat", **kw)
            out[f"{name}/sizes"] = np.array([len(train), len(val), len(test)])
            for split, ds in (("train", train), ("val", val), ("test", test)):
                raw = ds.dataset if hasattr(ds, "dataset") else ds
                out[f"{name}/{split}_info"] = np.array(raw.get_resolution_info(), dtype="U40")
                for idx in probe_indices(len(ds)):
                    x, y = ds[idx]
                    out[f"{name}/{split}_{idx}_x"] = np.asarray(x, dtype=np.float32)
                    out[f"{name}/{split}_{idx}_y"] = np.asarray(y, dtype=np.float32)
            if xn is not None:
                out[f"{name}/stats"] = np.array([xn.mean, xn.std, yn.mean, yn.std], dtype=np.float64)
            out[f"{name}/np_random_after"] = np.array(np.random.get_state()[1][:4], dtype=np.uint32)
    path = os.path.join(HERE, "data_layer_mres.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main_mres() if sys.argv[1:] == ["mres"] else main()
